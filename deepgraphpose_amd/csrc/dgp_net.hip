// Host engine + C-ABI of libdgp_hip.so (see include/dgp_hip.h).
//
// The engine walks the TF-slim resnet_v1 plan (same table as deepgraphpose_amd/arch.py):
//   preprocess -> conv1 7x7/2 (+BN+ReLU) -> maxpool 3x3/2 SAME -> bottleneck units
//   (shortcut | conv1 | conv2 | conv3+residual+ReLU) -> transposed-conv heads,
// every convolution through the one implicit-GEMM kernel.  It owns only the repacked
// weights; activations live in the caller's workspace.
#include "dgp_engine.h"

using namespace dgp;

static thread_local std::string g_err;
namespace dgp {
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

int coutp_for(int cout) {
    if (cout <= 32) return 32;
    if (cout <= 64) return 64;
    return round_up(cout, 128);
}
// padded width of a head's pointwise panel (columns = (tap, phase, joint)): the H2 cell kernels start at 64-column tiles, so one or two
// joints (16 / 32 columns) pad to 64.  ONE rule for dgp_net_load_weights and dgp_trainer_sync_weights: they share l.d_w_pw / l.d_wh3_pw.
int head_pw_coutp(int cpw) { const int c = coutp_for(cpw); return c < 64 ? 64 : c; }
int nk_for(int kh, int kw, int cin) { return (kh * kw * (cin / 4) + 7) / 8; }

void tf_same(int n, int k, int s, int d, int* out, int* pad_before) {
    const int keff = (k - 1) * d + 1;
    *out = (n + s - 1) / s;
    int total = (*out - 1) * s + keff - n;
    if (total < 0) total = 0;
    *pad_before = total / 2;
}

// conv2d_same / SAME padding-before for a k x k conv with given stride / rate on extent n
int pad_before_for(int n, int k, int stride, int rate, bool conv2d_same_explicit) {
    const int keff = (k - 1) * rate + 1;
    if (conv2d_same_explicit && stride > 1) return (keff - 1) / 2;     // slim conv2d_same
    int out, pb;
    tf_same(n, k, stride, rate, &out, &pb);
    return pb;
}
// The 16-bit tier (dgp_net::tier == 1; DGP_CONV_MODE=f16 makes it the default of every net of the process): H1 cells -- 2-byte
// activations -- from the pool output to the block4 features, high-only weight cells, one MFMA per product.  Not inside the 1e-3 px gate;
// bench.py reports what it measures beside the parity tier.
static bool tier16_env() {
    static const bool on = getenv("DGP_CONV_MODE") && !strcmp(getenv("DGP_CONV_MODE"), "f16");
    return on;
}

static int pow2_exp_for_max(float mx) {        // e with max * 2^e in [2^14, 2^15) -- the weight scale of the fp16 split (pow2_scale_for)
    if (!(mx > 0.f) || !std::isfinite(mx)) return 0;
    int ex; (void)frexpf(mx, &ex);             // mx = f 2^ex, f in [0.5, 1)
    return 14 - (ex - 1);
}

void free_chain_plan(ChainPlan& pl) {
    for (void* q : {pl.d_frags, pl.d_frags_h1, (void*)pl.d_sc1, (void*)pl.d_bi1}) if (q) (void)hipFree(q);
    pl = ChainPlan();
}

int build_chain_plan(ChainPlan& pl, int C, int C1, int CIN2, int res, const float* w3cat, const float* sc3, const float* bi3,
                     const float* w1, const float* sc1, const float* bi1, const float* w2, const float* sc2, const float* bi2) {
    free_chain_plan(pl);
    if (!chain_supported(C, C1, CIN2, res)) return DGP_OK;       // (ok stays false: the engine runs the two convs layer by layer)
    const bool unit = w2 && unit_supported(C, C1, CIN2, res);
    const int C4 = 4 * C, KS1 = (C + CIN2) / 32, NJP = C4 / 32, NCB = C1 / 16;
    const int pairs = KS1 * 2 + NCB;
    const size_t chunk_f = (size_t)pairs * 512 + 256;             // floats of the fp32 staging image per chunk
    const int KS2 = C / 32, NCB2 = C / 16, pairs2 = KS2 * NCB2;
    const size_t chunk2_f = (size_t)pairs2 * 512 + 256;
    std::vector<float> src((size_t)NJP * chunk_f, 0.f), src2(unit ? 9 * chunk2_f : 0, 0.f);
    float m3 = 0.f, m1 = 0.f, m2 = 0.f;
    for (size_t i = 0; i < (size_t)(C + CIN2) * C4; ++i) m3 = std::max(m3, fabsf(w3cat[i]));
    for (size_t i = 0; i < (size_t)C4 * C1; ++i) m1 = std::max(m1, fabsf(w1[i]));
    // column of MFMA row i of 16-channel block cb: the permutation that gives a lane 8 consecutive channels per block pair
    auto col_of = [](int cb, int i) { return 32 * (cb >> 1) + 8 * (i >> 2) + 4 * (cb & 1) + (i & 3); };
    for (int jp = 0; jp < NJP; ++jp) {
        float* ch = src.data() + (size_t)jp * chunk_f;
        for (int s = 0; s < KS1; ++s)
            for (int b = 0; b < 2; ++b) {
                float* f = ch + (size_t)(s * 2 + b) * 512;
                for (int ln = 0; ln < 64; ++ln) {
                    const int g = ln >> 4, i = ln & 15, col = 32 * jp + col_of(b, i);
                    for (int j = 0; j < 8; ++j) f[ln * 8 + j] = w3cat[(size_t)(32 * s + 8 * g + j) * C4 + col];
                }
            }
        for (int cb = 0; cb < NCB; ++cb) {
            float* f = ch + (size_t)(KS1 * 2 + cb) * 512;
            for (int ln = 0; ln < 64; ++ln) {
                const int g = ln >> 4, i = ln & 15, col = col_of(cb, i);
                for (int j = 0; j < 8; ++j) f[ln * 8 + j] = w1[(size_t)(32 * jp + 8 * g + j) * C1 + col];
            }
        }
        float* af = ch + (size_t)pairs * 512;
        for (int c = 0; c < 32; ++c) { af[c] = sc3 ? sc3[32 * jp + c] : 1.f; af[32 + c] = bi3 ? bi3[32 * jp + c] : 0.f; }
    }
    if (unit) {
        for (size_t i = 0; i < (size_t)9 * C * C; ++i) m2 = std::max(m2, fabsf(w2[i]));
        for (int t = 0; t < 9; ++t) {
            float* ch = src2.data() + (size_t)t * chunk2_f;
            for (int ks = 0; ks < KS2; ++ks)
                for (int cb = 0; cb < NCB2; ++cb) {
                    float* f = ch + (size_t)(ks * NCB2 + cb) * 512;
                    for (int ln = 0; ln < 64; ++ln) {
                        const int g = ln >> 4, i = ln & 15, col = col_of(cb, i);
                        for (int j = 0; j < 8; ++j) f[ln * 8 + j] = w2[((size_t)t * C + 32 * ks + 8 * g + j) * C + col];
                    }
                }
            float* af = ch + (size_t)pairs2 * 512;
            for (int c = 0; c < C; ++c) { af[c] = sc2 ? sc2[c] : 1.f; af[C + c] = bi2 ? bi2[c] : 0.f; }
        }
    }
    pl.C = C; pl.C1 = C1; pl.CIN2 = CIN2; pl.res = res; pl.unit = unit;
    pl.w3_exp = pow2_exp_for_max(m3); pl.w1_exp = pow2_exp_for_max(m1); pl.w2_exp = pow2_exp_for_max(m2);
    const int nf = chain_frags_per_chunk(C, C1, CIN2), nf2 = 2 * pairs2 + 1;
    const size_t head = unit ? (size_t)9 * nf2 * 1024 : 0;
    pl.frag_bytes = (unsigned)(head + (size_t)NJP * nf * 1024);
    pl.head_bytes = (unsigned)head;
    float *d_src = nullptr, *d_src2 = nullptr;
    HIP_TRY(hipMalloc(&d_src, src.size() * sizeof(float)));
    hipError_t e = hipMemcpy(d_src, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&pl.d_frags, pl.frag_bytes);
    if (e == hipSuccess && unit) e = hipMalloc(&d_src2, src2.size() * sizeof(float));
    if (e == hipSuccess && unit) e = hipMemcpy(d_src2, src2.data(), src2.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && unit) e = launch_chain_pack(d_src2, 9, pairs2, 0, ldexpf(1.f, pl.w2_exp), 1.f, pl.d_frags, nullptr);
    if (e == hipSuccess) e = launch_chain_pack(d_src, NJP, KS1 * 2, NCB, ldexpf(1.f, pl.w3_exp), ldexpf(1.f, pl.w1_exp), (char*)pl.d_frags + head, nullptr);
    // the 16-bit tier's chunks: the high fragments alone, same order
    const int nf_h1 = chain_frags_per_chunk(C, C1, CIN2, 1), nf2_h1 = pairs2 + 1;
    const size_t head_h1 = unit ? (size_t)9 * nf2_h1 * 1024 : 0;
    pl.frag_bytes_h1 = (unsigned)(head_h1 + (size_t)NJP * nf_h1 * 1024);
    pl.head_bytes_h1 = (unsigned)head_h1;
    if (e == hipSuccess) e = hipMalloc(&pl.d_frags_h1, pl.frag_bytes_h1);
    if (e == hipSuccess && unit) e = launch_chain_pack(d_src2, 9, pairs2, 0, ldexpf(1.f, pl.w2_exp), 1.f, pl.d_frags_h1, nullptr, 1);
    if (e == hipSuccess) e = launch_chain_pack(d_src, NJP, KS1 * 2, NCB, ldexpf(1.f, pl.w3_exp), ldexpf(1.f, pl.w1_exp), (char*)pl.d_frags_h1 + head_h1, nullptr, 1);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(d_src);
    if (d_src2) (void)hipFree(d_src2);
    if (e == hipSuccess) e = hipMalloc(&pl.d_sc1, C1 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&pl.d_bi1, C1 * sizeof(float));
    std::vector<float> one(C1, 1.f), zero(C1, 0.f);
    if (e == hipSuccess) e = hipMemcpy(pl.d_sc1, sc1 ? sc1 : one.data(), C1 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pl.d_bi1, bi1 ? bi1 : zero.data(), C1 * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { free_chain_plan(pl); return fail(DGP_ERR_HIP, std::string("chain plan: ") + hipGetErrorString(e)); }
    pl.ok = true;
    return DGP_OK;
}
}  // namespace dgp

namespace {

void pack_panels(const float* hwio, int KH, int KW, int Cin, int Cout, float* packed, int coutp = 0) {
    const int cin4 = Cin / 4, CoutP = coutp ? coutp : coutp_for(Cout), nk = nk_for(KH, KW, Cin);
    const size_t total = (size_t)nk * 8 * CoutP * 4;
    memset(packed, 0, total * sizeof(float));
    const int nchunks = KH * KW * cin4;
    for (int q = 0; q < nchunks; ++q) {
        const int tap = q / cin4, ch = (q % cin4) * 4;
        for (int e = 0; e < 4; ++e) {
            const float* src = hwio + ((size_t)tap * Cin + ch + e) * Cout;
            float* dst = packed + (size_t)q * CoutP * 4 + e;
            for (int co = 0; co < Cout; ++co) dst[(size_t)co * 4] = src[co];
        }
    }
}

}  // namespace

static int add_layer(dgp_net* net, const std::string& scope, int cin, int cout, int k, int stride, int rate,
                     bool bn, bool relu) {
    ConvLayer l;
    l.scope = scope; l.Cin = cin; l.Cout = cout; l.KH = l.KW = k; l.stride = stride; l.rate = rate;
    l.CoutP = coutp_for(cout); l.nk = nk_for(k, k, cin); l.ntaps = k * k; l.has_bn = bn; l.relu = relu;
    net->layers.push_back(l);
    return (int)net->layers.size() - 1;
}

extern "C" {

int dgp_version(void) { return DGP_ABI_VERSION; }

// Host-side CRC-32C (Castagnoli), slicing-by-8; checksum of TF tensor-bundle entries (tf_checkpoint.py).
uint32_t dgp_crc32c(const void* data, size_t n, uint32_t crc) {
    static uint32_t tab[8][256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            tab[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) tab[t][i] = (tab[t - 1][i] >> 8) ^ tab[0][tab[t - 1][i] & 0xFF];
        init = true;
    }
    const uint8_t* p = (const uint8_t*)data;
    crc = ~crc;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= crc;
        crc = tab[7][lo & 0xFF] ^ tab[6][(lo >> 8) & 0xFF] ^ tab[5][(lo >> 16) & 0xFF] ^ tab[4][lo >> 24] ^
              tab[3][hi & 0xFF] ^ tab[2][(hi >> 8) & 0xFF] ^ tab[1][(hi >> 16) & 0xFF] ^ tab[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) crc = tab[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
const char* dgp_last_error(void) { return g_err.c_str(); }
int dgp_tuning_build(void) {
#ifdef DGP_TUNING
    return 1;
#else
    return 0;
#endif
}

int dgp_net_create(const dgp_net_desc* d, dgp_net** out) {
    if (!d || !out) return fail(DGP_ERR_INVALID, "dgp_net_create: null argument");
    int nunits[4];
    if (d->depth == 50) { int v[4] = {3, 4, 6, 3}; memcpy(nunits, v, sizeof v); }
    else if (d->depth == 101) { int v[4] = {3, 4, 23, 3}; memcpy(nunits, v, sizeof v); }
    else if (d->depth == 152) { int v[4] = {3, 8, 36, 3}; memcpy(nunits, v, sizeof v); }
    else return fail(DGP_ERR_INVALID, "dgp_net_create: depth must be 50, 101 or 152");
    if (d->num_joints < 1 || d->in_h < 32 || d->in_w < 32 || d->max_batch < 1)
        return fail(DGP_ERR_INVALID, "dgp_net_create: bad num_joints / frame size / max_batch");
    dgp_net* net = new dgp_net();
    net->desc = *d;
    if (net->desc.bn_eps <= 0.f) net->desc.bn_eps = 1e-5f;
    HIP_TRY(hipGetDevice(&net->device));
    const std::string name = "resnet_v1_" + std::to_string(d->depth);
    // root block: conv2d_same(64, 7, stride 2) on the channel-padded (3->4) centred frame
    net->conv1 = add_layer(net, name + "/conv1", 4, 64, 7, 2, 1, true, true);
    // slim stack_blocks_dense with output_stride 16 (target 4 after the root block)
    const int base[4] = {64, 128, 256, 512};
    const int bstride[4] = {2, 2, 2, 1};
    int cur = 1, rate = 1, depth_in = 64;
    for (int b = 0; b < 4; ++b)
        for (int u = 1; u <= nunits[b]; ++u) {
            Unit un;
            const int s = (u == nunits[b]) ? bstride[b] : 1;
            if (cur == 4) { un.stride = 1; un.rate = rate; rate *= s; }
            else { un.stride = s; un.rate = 1; cur *= s; }
            un.depth_in = depth_in; un.depth = base[b] * 4; un.depth_bn = base[b];
            const std::string sc = name + "/block" + std::to_string(b + 1) + "/unit_" + std::to_string(u) +
                                   "/bottleneck_v1";
            if (un.depth_in != un.depth)
                un.sc = add_layer(net, sc + "/shortcut", un.depth_in, un.depth, 1, un.stride, 1, true, false);
            un.c1 = add_layer(net, sc + "/conv1", un.depth_in, un.depth_bn, 1, 1, 1, true, true);
            un.c2 = add_layer(net, sc + "/conv2", un.depth_bn, un.depth_bn, 3, un.stride, un.rate, true, true);
            un.c3 = add_layer(net, sc + "/conv3", un.depth_bn, un.depth, 1, 1, 1, true, true /*after add*/);
            net->units.push_back(un);
            depth_in = un.depth;
        }
    // heads: 3x3/stride-2 SAME transposed conv == 2x2 conv over (i-1..i, j-1..j) producing the
    // 4 output phases as 4*nj channels, scattered by the epilogue.
    {
        ConvLayer l;
        l.scope = "pose/part_pred/block4"; l.Cin = 2048; l.Cout = 4 * d->num_joints; l.KH = l.KW = 2;
        l.CoutP = coutp_for(l.Cout); l.nk = nk_for(2, 2, 2048); l.ntaps = 4; l.has_bn = false; l.relu = false;
        net->layers.push_back(l);
        net->head_part = (int)net->layers.size() - 1;
        if (d->with_locref) {
            l.scope = "pose/locref_pred/block4"; l.Cout = 8 * d->num_joints; l.CoutP = coutp_for(l.Cout);
            net->layers.push_back(l);
            net->head_locref = (int)net->layers.size() - 1;
        }
    }
    int pb;
    net->h1 = (d->in_h + 1) / 2; net->w1 = (d->in_w + 1) / 2;     // conv2d_same stride 2
    tf_same(net->h1, 3, 2, 1, &net->hp, &pb); tf_same(net->w1, 3, 2, 1, &net->wp, &pb);
    net->fh = (((net->hp + 1) / 2) + 1) / 2; net->fw = (((net->wp + 1) / 2) + 1) / 2;
    net->tier = tier16_env() ? 1 : 0;
    *out = net;
    return DGP_OK;
}

void dgp_net_destroy(dgp_net* net) { delete net; }

int dgp_net_set_tier(dgp_net* net, int32_t tier) {
    if (!net) return fail(DGP_ERR_INVALID, "dgp_net_set_tier: null net");
    if (tier != 0 && tier != 1) return fail(DGP_ERR_INVALID, "dgp_net_set_tier: tier must be 0 (parity) or 1 (16-bit)");
    if (tier != net->tier) { net->tier = tier; net->h2_calibrated = false; }      // the next forward calibrates the scales on this tier's tensors
    return DGP_OK;
}
int dgp_net_get_tier(const dgp_net* net) { return net ? net->tier : 0; }

int dgp_net_set_input_size(dgp_net* net, int32_t in_h, int32_t in_w) {
    if (!net) return fail(DGP_ERR_INVALID, "dgp_net_set_input_size: null net");
    if (in_h < 32 || in_w < 32) return fail(DGP_ERR_INVALID, "dgp_net_set_input_size: frames must be at least 32 x 32");
    net->desc.in_h = in_h; net->desc.in_w = in_w;
    int pb;
    net->h1 = (in_h + 1) / 2; net->w1 = (in_w + 1) / 2;
    tf_same(net->h1, 3, 2, 1, &net->hp, &pb); tf_same(net->w1, 3, 2, 1, &net->wp, &pb);
    net->fh = (((net->hp + 1) / 2) + 1) / 2; net->fw = (((net->wp + 1) / 2) + 1) / 2;
    return DGP_OK;
}

static const dgp_tensor_view* find_t(const std::map<std::string, const dgp_tensor_view*>& m,
                                     const std::string& k) {
    auto it = m.find(k);
    return it == m.end() ? nullptr : it->second;
}

int dgp_net_load_weights(dgp_net* net, const dgp_tensor_view* tensors, int32_t n) {
    if (!net || !tensors) return fail(DGP_ERR_INVALID, "dgp_net_load_weights: null argument");
    std::map<std::string, const dgp_tensor_view*> m;
    for (int i = 0; i < n; ++i)
        if (tensors[i].name) m[tensors[i].name] = &tensors[i];
    HIP_TRY(hipSetDevice(net->device));
    const int nj = net->desc.num_joints;
    std::map<int, std::vector<float>> keep_w, keep_scale, keep_bias;      // 1x1 convs that get fused (conv3 + shortcut)
    std::map<int, bool> wanted;
    for (const Unit& u : net->units) { wanted[u.c1] = true; wanted[u.c2] = true; wanted[u.c3] = true; if (u.sc >= 0) wanted[u.sc] = true; }
    for (size_t li = 0; li < net->layers.size(); ++li) {
        ConvLayer& l = net->layers[li];
        const bool is_head = ((int)li == net->head_part || (int)li == net->head_locref);
        const dgp_tensor_view* w = find_t(m, l.scope + "/weights");
        if (!w) return fail(DGP_ERR_MISSING, "missing tensor " + l.scope + "/weights");
        std::vector<float> hwio;
        std::vector<float> scale(l.Cout, 1.f), bias(l.Cout, 0.f);
        if ((int)li == net->conv1) {
            if (w->ndim != 4 || w->shape[0] != 7 || w->shape[1] != 7 || w->shape[2] != 3 || w->shape[3] != 64)
                return fail(DGP_ERR_INVALID, "bad shape for " + l.scope + "/weights (want [7,7,3,64])");
            hwio.assign((size_t)49 * 4 * 64, 0.f);
            for (int tp = 0; tp < 49; ++tp)
                for (int ci = 0; ci < 3; ++ci)
                    memcpy(&hwio[((size_t)tp * 4 + ci) * 64], w->data + ((size_t)tp * 3 + ci) * 64, 64 * sizeof(float));
        } else if (is_head) {
            const int njt = l.Cout / 4;      // nj or 2*nj
            if (w->ndim != 4 || w->shape[0] != 3 || w->shape[1] != 3 || w->shape[2] != njt || w->shape[3] != 2048)
                return fail(DGP_ERR_INVALID, "bad shape for " + l.scope + "/weights (want [3,3,Cout,2048])");
            // W'[kh'][kw'][ci][(a,b),c] = w[a+2-2kh'][b+2-2kw'][c][ci] when both tap ids <= 2
            hwio.assign((size_t)4 * 2048 * l.Cout, 0.f);
            for (int khp = 0; khp < 2; ++khp)
                for (int kwp = 0; kwp < 2; ++kwp)
                    for (int a = 0; a < 2; ++a)
                        for (int b = 0; b < 2; ++b) {
                            const int ka = a + 2 - 2 * khp, kb = b + 2 - 2 * kwp;
                            if (ka > 2 || kb > 2) continue;
                            for (int c = 0; c < njt; ++c) {
                                const float* src = w->data + (((size_t)ka * 3 + kb) * njt + c) * 2048;
                                const int co = (a * 2 + b) * njt + c;
                                for (int ci = 0; ci < 2048; ++ci)
                                    hwio[(((size_t)khp * 2 + kwp) * 2048 + ci) * l.Cout + co] = src[ci];
                            }
                        }
            const dgp_tensor_view* bb = find_t(m, l.scope + "/biases");
            if (!bb) return fail(DGP_ERR_MISSING, "missing tensor " + l.scope + "/biases");
            for (int ph = 0; ph < 4; ++ph)
                for (int c = 0; c < njt; ++c) bias[ph * njt + c] = bb->data[c];
            (void)nj;
        } else {
            if (w->ndim != 4 || w->shape[0] != l.KH || w->shape[1] != l.KW || w->shape[2] != l.Cin ||
                w->shape[3] != l.Cout)
                return fail(DGP_ERR_INVALID, "bad shape for " + l.scope + "/weights");
            hwio.assign(w->data, w->data + (size_t)l.KH * l.KW * l.Cin * l.Cout);
        }
        if (l.has_bn) {
            const dgp_tensor_view* g = find_t(m, l.scope + "/BatchNorm/gamma");
            const dgp_tensor_view* be = find_t(m, l.scope + "/BatchNorm/beta");
            const dgp_tensor_view* mu = find_t(m, l.scope + "/BatchNorm/moving_mean");
            const dgp_tensor_view* var = find_t(m, l.scope + "/BatchNorm/moving_variance");
            if (!g || !be || !mu || !var) return fail(DGP_ERR_MISSING, "missing BatchNorm tensors under " + l.scope);
            for (int c = 0; c < l.Cout; ++c) {
                // slim.batch_norm(is_training=False): y = (x-mean)*gamma*rsqrt(var+eps)+beta
                const float inv = g->data[c] / sqrtf(var->data[c] + net->desc.bn_eps);
                scale[c] = inv;
                bias[c] = be->data[c] - mu->data[c] * inv;
            }
        }
        if (wanted.count((int)li)) { keep_w[(int)li] = hwio; keep_scale[(int)li] = scale; keep_bias[(int)li] = bias; }
        const size_t nfl = (size_t)l.nk * 8 * l.CoutP * 4;
        std::vector<float> packed(nfl);
        pack_panels(hwio.data(), l.KH, l.KW, l.Cin, l.Cout, packed.data());
        if (!l.d_w) HIP_TRY(hipMalloc(&l.d_w, nfl * sizeof(float)));
        if (!l.d_scale) HIP_TRY(hipMalloc(&l.d_scale, l.Cout * sizeof(float)));
        if (!l.d_bias) HIP_TRY(hipMalloc(&l.d_bias, l.Cout * sizeof(float)));
        HIP_TRY(hipMemcpy(l.d_w, packed.data(), nfl * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(l.d_scale, scale.data(), l.Cout * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(l.d_bias, bias.data(), l.Cout * sizeof(float), hipMemcpyHostToDevice));
        if (is_head) {      // the same weights as a pointwise panel: column (tap, phase, joint) of row ci
            const int cpw = 4 * l.Cout;
            std::vector<float> wpw((size_t)2048 * cpw);
            for (int tap = 0; tap < 4; ++tap)
                for (int ci = 0; ci < 2048; ++ci)
                    memcpy(&wpw[(size_t)ci * cpw + (size_t)tap * l.Cout], &hwio[((size_t)tap * 2048 + ci) * l.Cout], l.Cout * sizeof(float));
            l.coutp_pw = head_pw_coutp(cpw);
            const size_t npw = (size_t)nk_for(1, 1, 2048) * 8 * l.coutp_pw * 4;
            std::vector<float> ppw(npw);
            pack_panels(wpw.data(), 1, 1, 2048, cpw, ppw.data(), l.coutp_pw);
            if (!l.d_w_pw) HIP_TRY(hipMalloc(&l.d_w_pw, npw * sizeof(float)));
            if (!l.d_wh3_pw) HIP_TRY(hipMalloc(&l.d_wh3_pw, npw * sizeof(float)));
            HIP_TRY(hipMemcpy(l.d_w_pw, ppw.data(), npw * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    // operand ranges for the fp16-split kernels
    const size_t nl = net->layers.size();
    const size_t rb = (size_t)ABSMAX_SLOTS * sizeof(float);
    if (!net->d_wmax) HIP_TRY(hipMalloc(&net->d_wmax, 2 * nl * rb));
    if (!net->d_amax) HIP_TRY(hipMalloc(&net->d_amax, nl * rb));
    if (!net->d_inmax) HIP_TRY(hipMalloc(&net->d_inmax, rb));
    HIP_TRY(hipMemset(net->d_wmax, 0, 2 * nl * rb));
    HIP_TRY(hipMemset(net->d_amax, 0, nl * rb));
    {
        std::vector<float> in_rng(ABSMAX_SLOTS, 0.f);
        float bound = 1.f;                           // max |u8 - mean_pixel| over the three channels
        for (int c = 0; c < 3; ++c)
            bound = std::max(bound, std::max(fabsf(net->desc.mean_pixel[c]), fabsf(255.f - net->desc.mean_pixel[c])));
        in_rng[0] = bound;
        HIP_TRY(hipMemcpy(net->d_inmax, in_rng.data(), rb, hipMemcpyHostToDevice));
    }
    for (size_t li = 0; li < nl; ++li) {
        ConvLayer& l = net->layers[li];
        hipError_t e = launch_absmax(l.d_w, (long long)l.nk * 8 * l.CoutP * 4, net->d_wmax + li * ABSMAX_SLOTS, nullptr);
        if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("weight range: ") + hipGetErrorString(e));
        if ((int)li == net->conv1) {
            // stem as a row walk for the split kernels: K-step kh = the 8 pixels x 4 channels that start at the window's
            // left edge (one contiguous 128-byte read per output pixel and kernel row); pixel 7 has zero weights
            std::vector<float> rows((size_t)7 * 8 * l.CoutP * 4, 0.f), host((size_t)l.nk * 8 * l.CoutP * 4);
            HIP_TRY(hipMemcpy(host.data(), l.d_w, host.size() * sizeof(float), hipMemcpyDeviceToHost));
            // Channel slot 3 (a zero in every input the layer kernels see) carries, for the fused root kernel, the part of the mean
            // subtraction that is not an integer: x - mean_c = (x - round(mean_c)) + d_c with d_c = round(mean_c) - mean_c, and
            // sum_c d_c w[tap][c][co] is a weight of a fourth input channel that is 1 on every pixel inside the frame.  The kernel then
            // multiplies exact small integers (one fp16 plane, two MFMAs per product instead of three).
            float dmean[3];
            for (int c = 0; c < 3; ++c) dmean[c] = roundf(net->desc.mean_pixel[c]) - net->desc.mean_pixel[c];
            for (int kh = 0; kh < 7; ++kh)
                for (int kw = 0; kw < 7; ++kw) {      // generic panel row = tap (Cin = 4: one row of 4 k-values per tap)
                    float* dst = &rows[((size_t)(kh * 8 + kw)) * l.CoutP * 4];
                    memcpy(dst, &host[((size_t)(kh * 7 + kw)) * l.CoutP * 4], (size_t)l.CoutP * 4 * sizeof(float));
                    for (int co = 0; co < l.CoutP; ++co)
                        dst[co * 4 + 3] = dmean[0] * dst[co * 4] + dmean[1] * dst[co * 4 + 1] + dmean[2] * dst[co * 4 + 2];
                }
            if (!l.d_w_rows) HIP_TRY(hipMalloc(&l.d_w_rows, rows.size() * sizeof(float)));
            HIP_TRY(hipMemcpy(l.d_w_rows, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice));
            if (!l.d_wh3) HIP_TRY(hipMalloc(&l.d_wh3, rows.size() * sizeof(float)));
            e = launch_pack_h3(l.d_w_rows, 7, l.CoutP, net->d_wmax + li * ABSMAX_SLOTS, l.d_wh3, nullptr);
            if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("stem weight cells: ") + hipGetErrorString(e));
        } else if (l.d_w_pw) {                                   // heads: range + cells of the pointwise panel (slot nl + li)
            const int nkpw = nk_for(1, 1, 2048);
            float* wm = net->d_wmax + (nl + li) * ABSMAX_SLOTS;
            e = launch_absmax(l.d_w_pw, (long long)nkpw * 8 * l.coutp_pw * 4, wm, nullptr);
            if (e == hipSuccess) e = launch_pack_h3(l.d_w_pw, nkpw, l.coutp_pw, wm, l.d_wh3_pw, nullptr);
            if (e == hipSuccess && !l.d_wh1_pw) e = hipMalloc(&l.d_wh1_pw, (size_t)nkpw * 8 * l.coutp_pw * 8);
            if (e == hipSuccess) e = launch_pack_h1(l.d_w_pw, nkpw, l.coutp_pw, wm, l.d_wh1_pw, nullptr);
            if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("head pointwise panel: ") + hipGetErrorString(e));
        } else if (l.Cin >= 32 && l.CoutP % 64 == 0) {          // layers the fp16-split kernels can take: pre-split cells
            const size_t bytes = (size_t)l.nk * 8 * l.CoutP * 16;
            if (!l.d_wh3) HIP_TRY(hipMalloc(&l.d_wh3, bytes));
            e = launch_pack_h3(l.d_w, l.nk, l.CoutP, net->d_wmax + li * ABSMAX_SLOTS, l.d_wh3, nullptr);
            if (e == hipSuccess && (l.Cin % 64) == 0 && (l.nk % 2) == 0) {      // 16-bit tier: a K-step is 64 channels
                if (!l.d_wh1) e = hipMalloc(&l.d_wh1, bytes / 2);
                if (e == hipSuccess) e = launch_pack_h1(l.d_w, l.nk, l.CoutP, net->d_wmax + li * ABSMAX_SLOTS, l.d_wh1, nullptr);
            }
            if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("weight cells: ") + hipGetErrorString(e));
        }
    }
    // conv3 + shortcut conv of a block's first unit as one GEMM over the concatenated channels (BN scales folded into
    // the weights, biases added): saves writing and re-reading the shortcut tensor
    for (const Unit& u : net->units) {
        if (u.sc < 0) continue;
        ConvLayer& l3 = net->layers[u.c3];
        const ConvLayer& ls = net->layers[u.sc];
        if (l3.KH != 1 || ls.KH != 1 || ls.stride != 1 || l3.Cin % 32 || ls.Cin % 32 || l3.Cout != ls.Cout) continue;
        const int c1 = l3.Cin, c2 = ls.Cin, co = l3.Cout;
        std::vector<float> w((size_t)(c1 + c2) * co), b(co);
        const std::vector<float>&w3 = keep_w[u.c3], &wsc = keep_w[u.sc], &s3 = keep_scale[u.c3], &ssc = keep_scale[u.sc];
        for (int k = 0; k < c1; ++k) for (int o = 0; o < co; ++o) w[(size_t)k * co + o] = w3[(size_t)k * co + o] * s3[o];
        for (int k = 0; k < c2; ++k) for (int o = 0; o < co; ++o) w[(size_t)(c1 + k) * co + o] = wsc[(size_t)k * co + o] * ssc[o];
        for (int o = 0; o < co; ++o) b[o] = keep_bias[u.c3][o] + keep_bias[u.sc][o];
        l3.nk_fused = (c1 + c2) / 32; l3.cin2 = c2;
        const size_t nfl = (size_t)l3.nk_fused * 8 * l3.CoutP * 4;
        std::vector<float> packed(nfl);
        pack_panels(w.data(), 1, 1, c1 + c2, co, packed.data());
        if (!l3.d_w_fused) HIP_TRY(hipMalloc(&l3.d_w_fused, nfl * sizeof(float)));
        if (!l3.d_bias_fused) HIP_TRY(hipMalloc(&l3.d_bias_fused, co * sizeof(float)));
        if (!l3.d_wh3_fused) HIP_TRY(hipMalloc(&l3.d_wh3_fused, nfl * sizeof(float)));
        HIP_TRY(hipMemcpy(l3.d_w_fused, packed.data(), nfl * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(l3.d_bias_fused, b.data(), co * sizeof(float), hipMemcpyHostToDevice));
        float* wm = net->d_wmax + (nl + (size_t)u.c3) * ABSMAX_SLOTS;
        hipError_t e = launch_absmax(l3.d_w_fused, (long long)nfl, wm, nullptr);
        if (e == hipSuccess) e = launch_pack_h3(l3.d_w_fused, l3.nk_fused, l3.CoutP, wm, l3.d_wh3_fused, nullptr);
        if (e == hipSuccess && (c1 % 64) == 0 && (c2 % 64) == 0) {
            if (!l3.d_wh1_fused) e = hipMalloc(&l3.d_wh1_fused, nfl * 2);
            if (e == hipSuccess) e = launch_pack_h1(l3.d_w_fused, l3.nk_fused, l3.CoutP, wm, l3.d_wh1_fused, nullptr);
        }
        if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("fused shortcut panel: ") + hipGetErrorString(e));
    }
    // conv3 of unit k (+ shortcut) and conv1 of unit k + 1 as one launch (dgp_chain.hip) wherever a kernel instance exists
    for (auto& c : net->chains) free_chain_plan(c);
    net->chains.assign(net->units.size(), ChainPlan());
    for (size_t ui = 0; ui + 1 < net->units.size(); ++ui) {
        const Unit &u = net->units[ui], &un = net->units[ui + 1];
        const ConvLayer &l3 = net->layers[u.c3], &l1 = net->layers[un.c1];
        if (l3.KH != 1 || l1.KH != 1 || l1.stride != 1 || l3.Cout != 4 * l3.Cin || l1.Cin != l3.Cout) continue;
        const int C = l3.Cin, C1 = l1.Cout;
        // block3 (C = 256): the kernel instance exists and is tested, but with 65-KiB weight chunks per 80 pixels it is paced by the
        // weight stream (0.18 ms per pair against 0.17 layer by layer on the batch-32 shapes): off unless DGP_CHAIN_WIDE=1
        static const bool wide_env = (dgp_tune("DGP_CHAIN_WIDE", 0) != 0);
        if (C > 128 && !wide_env) continue;
        const ConvLayer& l2 = net->layers[u.c2];
        const bool c2_plain = l2.KH == 3 && l2.stride == 1 && l2.rate == 1 && l2.Cin == C && l2.Cout == C;      // conv2 the unit kernel can take
        const float *w2 = c2_plain ? keep_w[u.c2].data() : nullptr, *s2 = c2_plain ? keep_scale[u.c2].data() : nullptr,
                    *b2 = c2_plain ? keep_bias[u.c2].data() : nullptr;
        int rc2;
        if (u.sc >= 0) {
            const ConvLayer& ls = net->layers[u.sc];
            if (ls.stride != 1 || u.stride != 1 || !l3.d_w_fused) continue;
            const int c2 = ls.Cin, co = l3.Cout;
            std::vector<float> w((size_t)(C + c2) * co), b(co);
            const std::vector<float>&w3 = keep_w[u.c3], &wsc = keep_w[u.sc], &s3 = keep_scale[u.c3], &ssc = keep_scale[u.sc];
            for (int k = 0; k < C; ++k) for (int o = 0; o < co; ++o) w[(size_t)k * co + o] = w3[(size_t)k * co + o] * s3[o];
            for (int k = 0; k < c2; ++k) for (int o = 0; o < co; ++o) w[(size_t)(C + k) * co + o] = wsc[(size_t)k * co + o] * ssc[o];
            for (int o = 0; o < co; ++o) b[o] = keep_bias[u.c3][o] + keep_bias[u.sc][o];
            rc2 = build_chain_plan(net->chains[ui], C, C1, c2, 0, w.data(), nullptr, b.data(), keep_w[un.c1].data(),
                                   keep_scale[un.c1].data(), keep_bias[un.c1].data(), w2, s2, b2);
        } else {
            rc2 = build_chain_plan(net->chains[ui], C, C1, 0, u.stride == 1 ? 1 : 2, keep_w[u.c3].data(), keep_scale[u.c3].data(),
                                   keep_bias[u.c3].data(), keep_w[un.c1].data(), keep_scale[un.c1].data(), keep_bias[un.c1].data(), w2, s2, b2);
        }
        if (rc2) return rc2;
    }
    HIP_TRY(hipDeviceSynchronize());
    net->wmax_valid = true;
    net->loaded = true;
    net->h2_calibrated = false;                 // new weights, new ranges
    net->h2_head = 4;
    net->act_exp.assign(net->layers.size(), 0);
    net->unit_fuse_ok.assign(net->units.size(), 1);
    if (!net->d_exps) HIP_TRY(hipMalloc(&net->d_exps, net->layers.size() * sizeof(int)));
    if (!net->d_flag) { HIP_TRY(hipMalloc(&net->d_flag, 2 * sizeof(int))); HIP_TRY(hipMemset(net->d_flag, 0, 2 * sizeof(int))); }
    return DGP_OK;
}

}  // extern "C"

namespace {

constexpr int HEAD_KSPLIT_MAX = 8;

struct Plan {
    size_t off_p0, off_c1, off_x0, off_x1, off_sc, off_r1, off_r2, off_scmap, off_locref, off_slabs, off_tail, total;
};

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

Plan make_plan(const dgp_net* net, int B) {
    const dgp_net_desc& d = net->desc;
    size_t p0 = (size_t)B * d.in_h * d.in_w * 4;
    size_t c1 = (size_t)B * net->h1 * net->w1 * 64;
    size_t x = (size_t)B * net->hp * net->wp * 64, sc = 0, r1 = 0, r2 = 0;
    int h = net->hp, w = net->wp;
    for (const Unit& u : net->units) {
        const int ho = (h + u.stride - 1) / u.stride, wo = (w + u.stride - 1) / u.stride;
        r1 = std::max(r1, (size_t)B * h * w * u.depth_bn);
        r2 = std::max(r2, (size_t)B * ho * wo * u.depth_bn);
        x = std::max(x, (size_t)B * ho * wo * u.depth);
        if (u.sc >= 0) sc = std::max(sc, (size_t)B * ho * wo * u.depth);
        h = ho; w = wo;
    }
    r1 = r2 = std::max(r1, r2);       // the unit kernel ping-pongs conv1's output between the two regions
    Plan p{};
    size_t o = 0;
    auto take = [&](size_t nfloats) { size_t r = o; o += align256(nfloats * sizeof(float)); return r; };
    p.off_p0 = take(p0); p.off_c1 = take(c1); p.off_x0 = take(x); p.off_x1 = take(x); p.off_sc = take(sc);
    p.off_r1 = take(r1); p.off_r2 = take(r2);
    p.off_scmap = take((size_t)B * 4 * net->fh * net->fw * d.num_joints);
    p.off_locref = take((size_t)B * 4 * net->fh * net->fw * 2 * d.num_joints);
    p.off_slabs = take((size_t)HEAD_KSPLIT_MAX * B * 4 * net->fh * net->fw * 2 * d.num_joints);
    p.off_tail = take(TAIL_SLAB_FLOATS);
    p.total = o;
    return p;
}

struct ProfScope {
    dgp_net* net; hipStream_t s; int li;
    ProfScope(dgp_net* n, hipStream_t st, const std::string& name, double flops) : net(n), s(st), li(-1) {
        if (!net->prof_on || net->prof_used >= net->prof_slots) return;
        li = net->prof_cursor++;
        if (li >= net->prof_launches) { li = -1; return; }
        if (net->prof_used == 0) { net->prof_names[li] = name; net->prof_flops[li] = flops; }
        (void)hipEventRecord(net->prof_ev[((size_t)net->prof_used * net->prof_launches + li) * 2], s);
    }
    ~ProfScope() {
        if (li >= 0) (void)hipEventRecord(net->prof_ev[((size_t)net->prof_used * net->prof_launches + li) * 2 + 1], s);
    }
};

double conv_flops_of(const ConvLayer& l, int M, bool is_head) {
    // algorithmic: the transposed-conv heads count their true 9 taps per input pixel
    if (is_head) return 2.0 * M * 9.0 * l.Cin * (l.Cout / 4);
    const int cin = (l.KH == 7) ? 3 : l.Cin;
    return 2.0 * M * (double)l.KH * l.KW * cin * l.Cout;
}

// H2 formats / scale exponents of a launch's tensors (all zero: fp32 everywhere)
struct H2Spec { int in_fmt = 0, in_exp = 0, out_fmt = 0, out_exp = 0, res_fmt = 0, res_exp = 0; };

void apply_h2(ConvArgs& a, const H2Spec& h) {
    a.in_fmt = h.in_fmt; a.out_fmt = h.out_fmt; a.res_fmt = h.res_fmt;
    a.in_scale = ldexpf(1.f, h.in_exp); a.out_scale = ldexpf(1.f, h.out_exp); a.res_inv_scale = ldexpf(1.f, -h.res_exp);
}

int run_conv(dgp_net* net, const ConvLayer& l, const float* in, int N, int H, int W, int pad_t, int pad_l, int Ho, int Wo,
             const float* res, int res_s, int res_H, int res_W, bool relu, int out_mode, int dc_nj, float* out,
             hipStream_t s, float* slabs = nullptr, const float* in_absmax = nullptr, const H2Spec& h2 = H2Spec()) {
    ConvArgs a{};
    const int li = (int)(&l - net->layers.data());
    if (!net->wmax_valid) in_absmax = nullptr;
    apply_h2(a, h2);
    const bool ranged = in_absmax || h2.in_fmt;
    a.in_absmax = in_absmax; a.w_absmax = ranged ? net->wmax(li) : nullptr;
    a.out_absmax = out_mode == 0 ? net->amax(li) : nullptr;
    a.slab = net->tail_slab; a.slab_bytes = net->tail_slab_bytes;
    a.in = in; a.wpk = l.d_w; a.scale = l.has_bn ? l.d_scale : nullptr; a.bias = l.d_bias; a.res = res; a.out = out;
    a.N = N; a.H = H; a.W = W; a.Cin = l.Cin; a.log2cin4 = ilog2(l.Cin / 4);
    a.Ho = Ho; a.Wo = Wo; a.Cout = l.Cout; a.CoutP = l.CoutP;
    a.KH = l.KH; a.KW = l.KW; a.stride = l.stride; a.dil = l.rate; a.pad_t = pad_t; a.pad_l = pad_l;
    a.ntaps = l.ntaps; a.nk = l.nk; a.M = N * Ho * Wo;
    a.res_s = res ? res_s : 0; a.res_H = res_H; a.res_W = res_W;
    a.relu = relu ? 1 : 0; a.out_mode = out_mode; a.dc_nj = dc_nj;
    {
        const double lim = 4294967000.0;
        const double inb = (double)N * H * W * l.Cin * 4, outb = (double)a.M * l.Cout * 4;
        const double resb = res ? (double)N * res_H * res_W * l.Cout * 4 : 0.0;
        if (inb > lim || outb > lim || resb > lim)
            return fail(DGP_ERR_INVALID, "activation tensor exceeds the 4 GiB buffer-descriptor range; lower the batch");
        if (out_mode == 0 && (l.Cout & 3)) return fail(DGP_ERR_INVALID, "conv: Cout must be a multiple of 4");
        a.in_bytes = (unsigned)inb; a.out_bytes = (unsigned)outb; a.res_bytes = (unsigned)resb;
        a.w_bytes = (unsigned)((size_t)l.nk * 8 * l.CoutP * 16);
    }
    static const bool stem_rows = (dgp_tune("DGP_STEM_ROWS", 1) != 0);        // A/B switch
    static const bool f32_mode = getenv("DGP_CONV_MODE") && !strcmp(getenv("DGP_CONV_MODE"), "f32");
    if (li == net->conv1 && l.d_w_rows && net->wmax_valid && stem_rows && !f32_mode && l.CoutP % 64 == 0) {
        a.stem = 1; a.tap_rows = 8; a.KH = 7; a.KW = 1; a.ntaps = 7; a.nk = 7; a.wpk = l.d_w_rows;
        a.w_bytes = (unsigned)((size_t)7 * 8 * l.CoutP * 16);
    }
    static const bool use_cells = (dgp_tune("DGP_PRESPLIT_WEIGHTS", 1) != 0);   // A/B switch
    if (use_cells && ranged && l.d_wh3 && (li != net->conv1 || a.stem)) { a.wh3 = l.d_wh3; a.wh3_bytes = a.w_bytes; }
    if (h2.in_fmt == 2) { a.wh3 = l.d_wh1; a.wh3_bytes = a.w_bytes; }       // (null: launch_conv refuses)
    const int tile_cfg = pick_tile(a.M, a.CoutP, a.nk * BK, ranged && a.w_absmax);
    ProfScope ps(net, s, "conv:" + l.scope + "|" + conv_kernel_name(a, tile_cfg), conv_flops_of(l, a.M, out_mode == 1));
    const long long out_n = (long long)N * 4 * Ho * Wo * dc_nj;
    if (out_mode == 1 && slabs && (out_n & 3) == 0) {
        // the heads have a tiny N (4*nj channels) and a huge K (4 x 2048): split K so the grid fills the chip;
        // slabs are summed in a fixed order (deterministic, unlike float atomics)
        const int mtiles = (a.M + 127) / 128;
        int ks = 1;
        while (ks < HEAD_KSPLIT_MAX && mtiles * ks < 1024 && l.nk % (ks * 2) == 0) ks *= 2;
        if (ks > 1) {
            a.ksplit = ks; a.split_stride = out_n; a.out = slabs;
            hipError_t e = launch_conv(a, tile_cfg, s);
            if (e == hipSuccess) e = launch_reduce_slabs(slabs, out_n, out_n, ks, out, s);
            if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("head conv (") + l.scope + "): " + hipGetErrorString(e));
            return DGP_OK;
        }
    }
    hipError_t e = launch_conv(a, tile_cfg, s);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("conv launch (") + l.scope + "): " + hipGetErrorString(e));
    return DGP_OK;
}

// A head (3x3 / stride-2 SAME transposed conv, run as a 2x2 conv with 4 output phases) as a pointwise GEMM + gather:
// T[pixel][(tap, phase, joint)] = x[pixel][:] . W'[tap][:][(phase, joint)], then
// out[2 ho + a][2 wo + b][c] = bias[c] + sum over taps (kh', kw') of T[ho - 1 + kh'][wo - 1 + kw'][(kh', kw'), (a, b), c]
// (taps outside the map contribute nothing; fixed summation order).  T lives in the (free) R1 region of the workspace.
int run_head_pointwise(dgp_net* net, const ConvLayer& l, const float* feat, int B, int h, int w, int njt, float* out, hipStream_t s,
                       float* T, const float* feat_absmax, const H2Spec& h2 = H2Spec()) {
    const int li = (int)(&l - net->layers.data());
    const int nl = (int)net->layers.size();
    ConvArgs a{};
    apply_h2(a, h2);
    a.in = feat; a.wpk = l.d_w_pw; a.wh3 = h2.in_fmt == 2 ? l.d_wh1_pw : l.d_wh3_pw; a.out = T;
    a.in_absmax = feat_absmax; a.w_absmax = net->d_wmax + (size_t)(nl + li) * ABSMAX_SLOTS;
    a.slab = net->tail_slab; a.slab_bytes = net->tail_slab_bytes;
    a.N = B; a.H = h; a.W = w; a.Cin = 2048; a.log2cin4 = ilog2(2048 / 4);
    a.Ho = h; a.Wo = w; a.Cout = l.coutp_pw; a.CoutP = l.coutp_pw;
    a.KH = 1; a.KW = 1; a.stride = 1; a.dil = 1; a.ntaps = 1; a.nk = nk_for(1, 1, 2048); a.M = B * h * w;
    a.in_bytes = (unsigned)((size_t)a.M * 2048 * 4); a.out_bytes = (unsigned)((size_t)a.M * l.coutp_pw * 4);
    a.w_bytes = (unsigned)((size_t)a.nk * 8 * l.coutp_pw * 16); a.wh3_bytes = a.w_bytes;
    const int tile_cfg = pick_tile(a.M, a.CoutP, a.nk * BK, true);
    hipError_t e;
    {
        ProfScope ps(net, s, "conv:" + l.scope + "(pointwise+gather)|" + conv_kernel_name(a, tile_cfg), conv_flops_of(l, a.M, true));
        e = launch_conv(a, tile_cfg, s);
        if (e == hipSuccess) e = launch_head_gather(T, l.d_bias, B, h, w, njt, l.coutp_pw, out, s);
    }
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("head (") + l.scope + "): " + hipGetErrorString(e));
    return DGP_OK;
}

// conv3 and the shortcut conv of a unit as one K-concatenated 1x1 conv: out = relu([r2 | x] Wf + bf)
int run_conv_fused_shortcut(dgp_net* net, const Unit& u, const float* r2, const float* x, int N, int H, int W, float* out,
                            hipStream_t s, const float* r2_absmax, const float* x_absmax, const H2Spec& h2 = H2Spec()) {
    const ConvLayer& l = net->layers[u.c3];
    const ConvLayer& ls = net->layers[u.sc];
    const int nl = (int)net->layers.size();
    if (!net->wmax_valid) { r2_absmax = nullptr; x_absmax = nullptr; }
    ConvArgs a{};
    apply_h2(a, h2);
    a.in = r2; a.in2 = x; a.cin_split = l.Cin; a.Cin = l.Cin + l.cin2; a.log2cin4 = 0;
    a.wpk = l.d_w_fused; a.scale = nullptr; a.bias = l.d_bias_fused; a.res = nullptr; a.out = out;
    a.N = N; a.H = H; a.W = W; a.Ho = H; a.Wo = W; a.Cout = l.Cout; a.CoutP = l.CoutP;
    a.KH = 1; a.KW = 1; a.stride = 1; a.dil = 1; a.pad_t = 0; a.pad_l = 0; a.ntaps = 1; a.nk = l.nk_fused; a.M = N * H * W;
    a.relu = 1; a.out_mode = 0;
    const double lim = 4294967000.0;
    const double inb = (double)a.M * l.Cin * 4, in2b = (double)a.M * l.cin2 * 4, outb = (double)a.M * l.Cout * 4;
    if (inb > lim || in2b > lim || outb > lim)
        return fail(DGP_ERR_INVALID, "activation tensor exceeds the 4 GiB buffer-descriptor range; lower the batch");
    a.in_bytes = (unsigned)inb; a.in2_bytes = (unsigned)in2b; a.out_bytes = (unsigned)outb;
    a.w_bytes = (unsigned)((size_t)l.nk_fused * 8 * l.CoutP * 16);
    if ((r2_absmax && x_absmax) || h2.in_fmt) {
        a.in_absmax = r2_absmax; a.in2_absmax = x_absmax; a.w_absmax = net->wmax(nl + u.c3);
        a.wh3 = h2.in_fmt == 2 ? l.d_wh1_fused : l.d_wh3_fused; a.wh3_bytes = a.w_bytes;
    }
    a.out_absmax = net->amax(u.c3);
    a.slab = net->tail_slab; a.slab_bytes = net->tail_slab_bytes;
    const int tile_cfg = pick_tile(a.M, a.CoutP, a.nk * BK, (a.in_absmax || a.in_fmt) && a.w_absmax);
    ProfScope ps(net, s, "conv:" + l.scope + "+shortcut|" + conv_kernel_name(a, tile_cfg),
                 conv_flops_of(l, a.M, false) + conv_flops_of(ls, a.M, false));
    hipError_t e = launch_conv(a, tile_cfg, s);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("fused conv3+shortcut (") + l.scope + "): " + hipGetErrorString(e));
    return DGP_OK;
}

// conv3 of unit ui (+ shortcut, ReLU) and conv1 of unit ui + 1 as one launch (dgp_chain.hip).  r2 [M][C]; x: the unit's input (identity
// shortcut [M][4C], stride-2 unit [N, H, W, 4C], or the K-concatenated source of the shortcut conv [M][CIN2]); all tensors H2
// r1in != null: the unit kernel -- conv2 of unit ui runs in the same launch, reading R1 (r1in) with its halo; r2 is not used
int run_chain(dgp_net* net, int ui, const float* r2, const float* x, int N, int Ho, int Wo, int H, int W, float* xout, float* r1out,
              hipStream_t s, int x_exp, const float* r1in = nullptr) {
    const ChainPlan& cp = net->chains[ui];
    const Unit &u = net->units[ui], &un = net->units[ui + 1];
    const ConvLayer &l3 = net->layers[u.c3], &l1 = net->layers[un.c1];
    ChainArgs a{};
    a.h1 = net->tier ? 1 : 0;                     // the 16-bit tier: H1 tensors, the same weight chunks (high fragments only)
    const double EB = a.h1 ? 2.0 : 4.0;           // bytes per channel
    a.r2 = r2; a.src2 = x; a.xout = xout; a.r1out = r1out; a.sc1 = cp.d_sc1; a.bi1 = cp.d_bi1;
    a.wfrag = a.h1 ? (const char*)cp.d_frags_h1 + (r1in ? 0 : cp.head_bytes_h1) : (const char*)cp.d_frags + (r1in ? 0 : cp.head_bytes);
    a.M = N * Ho * Wo; a.HoWo = Ho * Wo; a.Wo = Wo; a.res_H = H; a.res_W = W;
    if (r1in) {                                   // unit kernel: conv2 (stride 1: Ho x Wo = H x W) in front
        const ConvLayer& l2 = net->layers[u.c2];
        a.r1in = r1in; a.H = H; a.W = W;
        a.post0 = ldexpf(1.f, -(net->act_exp[u.c1] + cp.w2_exp));
        a.r2_scale = ldexpf(1.f, net->act_exp[u.c2]);
        a.r2_absmax = net->amax(u.c2);
        a.r1in_bytes = (unsigned)((double)N * H * W * l2.Cin * EB);
    }
    a.post1 = ldexpf(1.f, -(net->act_exp[u.c2] + cp.w3_exp));
    a.post2 = ldexpf(1.f, -(net->act_exp[u.c3] + cp.w1_exp));
    a.res_inv_scale = ldexpf(1.f, -x_exp);
    a.xout_scale = ldexpf(1.f, net->act_exp[u.c3]);
    a.r1_scale = ldexpf(1.f, net->act_exp[un.c1]);
    a.xout_absmax = net->amax(u.c3); a.r1_absmax = net->amax(un.c1);
    const double lim = 4294967000.0;
    const double r2b = (double)a.M * cp.C * EB, xob = (double)a.M * cp.C * 4 * EB, r1b = (double)a.M * cp.C1 * EB;
    const double s2b = cp.res == 0 ? (double)a.M * cp.CIN2 * EB : (double)N * H * W * cp.C * 4 * EB;
    if (r2b > lim || xob > lim || r1b > lim || s2b > lim)
        return fail(DGP_ERR_INVALID, "activation tensor exceeds the 4 GiB buffer-descriptor range; lower the batch");
    a.r2_bytes = (unsigned)r2b; a.xout_bytes = (unsigned)xob; a.r1_bytes = (unsigned)r1b; a.src2_bytes = (unsigned)s2b;
    a.w_bytes = a.h1 ? cp.frag_bytes_h1 - (r1in ? 0 : cp.head_bytes_h1) : cp.frag_bytes - (r1in ? 0 : cp.head_bytes);
    double flops = conv_flops_of(l3, a.M, false) + conv_flops_of(l1, a.M, false);
    if (u.sc >= 0) flops += conv_flops_of(net->layers[u.sc], a.M, false);
    if (r1in) flops += conv_flops_of(net->layers[u.c2], a.M, false);
    std::string kname = chain_kernel_name(cp.C, cp.C1, cp.CIN2, cp.res);
    if (r1in) kname = "unit" + kname.substr(5);
    if (a.h1) kname = "h1_" + kname;
    ProfScope ps(net, s, "conv:" + (r1in ? net->layers[u.c2].scope + "+" : std::string()) + l3.scope + (u.sc >= 0 ? "+shortcut" : "") + "+" + l1.scope + "|" + kname, flops);
    hipError_t e = r1in ? launch_unit(a, N, cp.C, cp.C1, cp.CIN2, cp.res, s) : launch_chain(a, cp.C, cp.C1, cp.CIN2, cp.res, s);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("chain launch (") + l3.scope + "): " + hipGetErrorString(e));
    return DGP_OK;
}

}  // namespace

extern "C" {

int dgp_net_workspace_bytes(const dgp_net* net, int32_t batch, size_t* out_bytes) {
    if (!net || !out_bytes || batch < 1 || batch > net->desc.max_batch)
        return fail(DGP_ERR_INVALID, "dgp_net_workspace_bytes: bad argument (batch > max_batch?)");
    *out_bytes = make_plan(net, batch).total;
    return DGP_OK;
}

int dgp_net_output_dims(const dgp_net* net, int32_t* out_h, int32_t* out_w, int32_t* feat_h, int32_t* feat_w) {
    if (!net) return fail(DGP_ERR_INVALID, "dgp_net_output_dims: null net");
    if (out_h) *out_h = 2 * net->fh;
    if (out_w) *out_w = 2 * net->fw;
    if (feat_h) *feat_h = net->fh;
    if (feat_w) *feat_w = net->fw;
    return DGP_OK;
}

int dgp_net_stats(const dgp_net* net, int32_t batch, int32_t* n_launches, double* conv_flops) {
    if (!net) return fail(DGP_ERR_INVALID, "dgp_net_stats: null net");
    double macs = (double)net->h1 * net->w1 * 49 * 3 * 64;
    int launches = 3;   // preprocess, conv1, pool
    int h = net->hp, w = net->wp;
    for (const Unit& u : net->units) {
        const int ho = (h + u.stride - 1) / u.stride, wo = (w + u.stride - 1) / u.stride;
        if (u.sc >= 0) { macs += (double)ho * wo * u.depth_in * u.depth; ++launches; }
        macs += (double)h * w * u.depth_in * u.depth_bn;
        macs += (double)ho * wo * 9 * u.depth_bn * u.depth_bn;
        macs += (double)ho * wo * u.depth_bn * u.depth;
        launches += 3;
        h = ho; w = wo;
    }
    const int heads = net->desc.num_joints * (net->desc.with_locref ? 3 : 1);
    macs += (double)h * w * 9 * 2048 * heads;
    launches += net->desc.with_locref ? 2 : 1;
    if (n_launches) *n_launches = launches;
    if (conv_flops) *conv_flops = 2.0 * macs * batch;
    return DGP_OK;
}

int dgp_forward(dgp_net* net, const uint8_t* frames, int32_t batch, void* workspace, size_t workspace_bytes,
                float* scmap, float* locref, float* features, void* stream) {
    if (!net || !frames || !workspace) return fail(DGP_ERR_INVALID, "dgp_forward: null argument");
    if (!net->loaded) return fail(DGP_ERR_STATE, "dgp_forward: weights not loaded");
    if (batch < 1 || batch > net->desc.max_batch) return fail(DGP_ERR_INVALID, "dgp_forward: batch out of range");
    const Plan pl = make_plan(net, batch);
    if (workspace_bytes < pl.total) return fail(DGP_ERR_INVALID, "dgp_forward: workspace too small");
    if (locref && net->head_locref < 0) return fail(DGP_ERR_INVALID, "dgp_forward: net built without locref head");
    if (net->owner_sync) { const int rco = net->owner_sync(net->owner, stream); if (rco) return rco; }
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* P0 = (float*)(ws + pl.off_p0);
    float* C1 = (float*)(ws + pl.off_c1);
    float* X[2] = {(float*)(ws + pl.off_x0), (float*)(ws + pl.off_x1)};
    float* SC = (float*)(ws + pl.off_sc);
    float* R1 = (float*)(ws + pl.off_r1);
    float* R2 = (float*)(ws + pl.off_r2);
    const dgp_net_desc& d = net->desc;
    const int B = batch;
    int rc;

    net->prof_cursor = 0;
    net->tail_slab = (float*)(ws + pl.off_tail); net->tail_slab_bytes = (unsigned)(TAIL_SLAB_FLOATS * sizeof(float));
    hipError_t e;
    if (net->d_amax) HIP_TRY(hipMemsetAsync(net->d_amax, 0, net->layers.size() * ABSMAX_SLOTS * sizeof(float), s));   // ranges are per forward
    // ---- activation format of this forward.  H2 (default): every tensor from the pool output to the block4 features lives in HBM as
    // fp16 high / low cells with a calibrated per-tensor scale, so the conv kernels' K loops are ds_read + MFMA only (DGP_H2=0: fp32
    // activations, split in the consumers' K loops -- also what the other DGP_CONV_MODEs and a trainer-owned net use).
    static const bool h2_env = (dgp_env("DGP_H2", 1) != 0);
    static const bool f16_mode = !getenv("DGP_CONV_MODE") || !strcmp(getenv("DGP_CONV_MODE"), "f16x3") || tier16_env();
    const int tier = net->tier, FMT = tier ? 2 : 1;      // activation cells of this forward: H2 (parity tier) or H1 (16-bit tier)
    static const bool head_pw = (dgp_tune("DGP_HEAD_PW", 1) != 0);      // A/B switch
    static const bool fuse_env = (dgp_env("DGP_FUSE_SHORTCUT", 1) != 0);      // A/B switch
    static const bool f32_mode = getenv("DGP_CONV_MODE") && !strcmp(getenv("DGP_CONV_MODE"), "f32");
    static const bool cells_env = (dgp_tune("DGP_PRESPLIT_WEIGHTS", 1) != 0);
    const bool h2 = h2_env && cells_env && f16_mode && head_pw && net->wmax_valid && net->d_exps && net->layers[net->head_part].d_wh3_pw &&
                    (net->head_locref < 0 || net->layers[net->head_locref].d_wh3_pw) && net->act_exp.size() == net->layers.size();
    if (tier) {            // the 16-bit tier has no fallback: say what is missing instead of silently running another tier
        bool ok = h2 && net->layers[net->head_part].d_wh1_pw && (net->head_locref < 0 || net->layers[net->head_locref].d_wh1_pw);
        for (const Unit& u : net->units) {
            ok = ok && net->layers[u.c1].d_wh1 && net->layers[u.c2].d_wh1 && net->layers[u.c3].d_wh1 && (u.sc < 0 || net->layers[u.sc].d_wh1);
            if (u.sc >= 0 && net->layers[u.c3].d_w_fused && !net->layers[u.c3].d_wh1_fused) ok = false;
        }
        if (!ok) return fail(DGP_ERR_STATE, "dgp_forward: the 16-bit tier needs the H2 engine's switches at their defaults (DGP_H2, DGP_PRESPLIT_WEIGHTS, "
                                            "DGP_HEAD_PW, DGP_CONV_MODE unset or f16) and weights loaded by dgp_net_load_weights");
    }
    const bool calib = h2 && !net->h2_calibrated;
    const int H2_HEAD = net->h2_head;             // bits of headroom between a calibrated maximum and the fp16 limit
    auto exp_for = [H2_HEAD](float mx) {          // scale exponent that puts mx into [2^(14 - H2_HEAD), 2^(15 - H2_HEAD))
        if (!(mx > 0.f) || !std::isfinite(mx)) return 0;
        int ex; (void)frexpf(mx, &ex);            // mx = f 2^ex, f in [0.5, 1)
        return (14 - H2_HEAD) - (ex - 1);
    };
    auto read_range = [&](int li, float* mx) -> int {
        float host[ABSMAX_SLOTS];
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(host, net->amax(li), sizeof host, hipMemcpyDeviceToHost));
        float m = 0.f;
        for (float v : host) m = (v > m || v != v) ? v : m;
        *mx = m;
        return DGP_OK;
    };
    // calibration: launch, read the layer's range, fix the exponent, launch again if it changed (first forward only: hidden syncs)
    auto layer = [&](int li, auto&& launch, int forced_exp = dgp_net::H2_NONE, bool* forced_ok = nullptr) -> int {
        int r = launch();
        if (r || !calib) return r;
        float mx = 0.f;
        if ((r = read_range(li, &mx))) return r;
        int e = exp_for(mx);
        if (forced_exp != dgp_net::H2_NONE) {     // must share another tensor's scale (K-concatenated second source)
            const bool ok = forced_exp <= e + H2_HEAD - 1 && forced_exp >= e - 6;
            if (forced_ok) *forced_ok = ok;
            if (ok) e = forced_exp;
        }
        if (e != net->act_exp[li]) {
            net->act_exp[li] = e;
            HIP_TRY(hipMemsetAsync(net->amax(li), 0, ABSMAX_SLOTS * sizeof(float), s));
            r = launch();
        }
        return r;
    };
    // ---- root block.  H2 engine: ONE kernel from the uint8 frame to the pool output's cells (stem_pool_fused_kernel);
    // otherwise preprocess -> conv1 -> max-pool as three launches
    static const bool stem_fused_env = (dgp_tune("DGP_STEM_FUSED", 1) != 0);
    const ConvLayer& lstem = net->layers[net->conv1];
    const bool stem_fused = h2 && (stem_fused_env || tier) && lstem.d_wh3 && lstem.d_w_rows && lstem.CoutP == 64 && lstem.d_scale && lstem.d_bias;
    if (tier && !stem_fused) return fail(DGP_ERR_STATE, "dgp_forward: the 16-bit tier needs the fused root block");
    if (stem_fused) {
        rc = layer(net->conv1, [&] {
            ProfScope ps(net, s, "conv:" + lstem.scope + "+pool|stem_pool_fused", conv_flops_of(lstem, B * net->h1 * net->w1, false));
            hipError_t e2 = launch_stem_pool_fused(frames, B, d.in_h, d.in_w, lstem.d_wh3, net->wmax(net->conv1), lstem.d_scale, lstem.d_bias,
                                                   d.mean_pixel[0], d.mean_pixel[1], d.mean_pixel[2],
                                                   ldexpf(1.f, net->act_exp[net->conv1]), X[0], net->amax(net->conv1), s, tier ? 1 : 0);
            return e2 == hipSuccess ? (int)DGP_OK : fail(DGP_ERR_HIP, std::string("stem + pool: ") + hipGetErrorString(e2));
        });
        if (rc) return rc;
    } else {
        {
            ProfScope ps(net, s, "preprocess_u8", 0.0);
            e = launch_preprocess(frames, (long long)B * d.in_h * d.in_w, d.mean_pixel[0], d.mean_pixel[1],
                                  d.mean_pixel[2], P0, s);
        }
        if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("preprocess: ") + hipGetErrorString(e));
        // conv1: conv2d_same(7, stride 2): explicit pad 3 before
        rc = run_conv(net, net->layers[net->conv1], P0, B, d.in_h, d.in_w, 3, 3, net->h1, net->w1, nullptr, 0, 0, 0, true, 0,
                      0, C1, s, nullptr, net->d_inmax);
        if (rc) return rc;
        if (calib) {
            float mx = 0.f;
            if ((rc = read_range(net->conv1, &mx))) return rc;
            net->act_exp[net->conv1] = exp_for(mx);   // (max-pooling cannot raise the maximum of conv1's output)
        }
        {
            ProfScope ps(net, s, "maxpool3x3s2", 0.0);
            e = launch_maxpool(C1, B, net->h1, net->w1, 64, X[0], s, h2 ? ldexpf(1.f, net->act_exp[net->conv1]) : 0.f);
        }
        if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("maxpool: ") + hipGetErrorString(e));

    }
    const int pool_exp = h2 ? net->act_exp[net->conv1] : 0;
    // conv3(k) + conv1(k + 1) as one launch (DGP_CHAIN=0: layer by layer).  Calibration runs layer by layer (it needs every tensor's
    // range before the next layer runs) and is followed by a second, chained pass, so results never depend on which pass produced them
    static const bool chain_env = (dgp_env("DGP_CHAIN", 1) != 0);
    static const bool chain_h1_env = (dgp_env("DGP_CHAIN_H1", 1) != 0);      // the chain / unit kernels on H1 tensors (the 16-bit tier)
    const bool chain_tier = chain_env && (!tier || chain_h1_env);
    const bool chain_on = h2 && !calib && chain_tier && net->chains.size() == net->units.size();
    static const bool unit_env = (dgp_tune("DGP_UNIT", 1) != 0);      // conv2 inside the chain launch (block1)
    bool r1_ready = false;                            // R1 of this unit came out of the previous unit's chain launch
    float *Ra = R1, *Rb = R2;
    int cur = 0, h = net->hp, w = net->wp;
    const float* x_rng = net->amax(net->conv1);      // max-pooling cannot raise the maximum of conv1's output
    int x_exp = pool_exp;
    int ui = 0;
    for (const Unit& u : net->units) {
        const int ho = (h + u.stride - 1) / u.stride, wo = (w + u.stride - 1) / u.stride;
        const float* xin = X[cur];
        float* xout = X[cur ^ 1];
        const float* res = xin;
        int res_s = u.stride, res_H = h, res_W = w;
        int res_exp = x_exp;
        const bool can_fuse = fuse_env && !f32_mode && net->wmax_valid && u.sc >= 0 && net->layers[u.c3].d_w_fused && u.stride == 1 && ho == h && wo == w &&
                              net->layers[u.c3].CoutP % 64 == 0;
        bool fuse = can_fuse && (!h2 || calib || net->unit_fuse_ok[ui]);
        const int pb_h = pad_before_for(h, 3, u.stride, u.rate, true);
        const int pb_w = pad_before_for(w, 3, u.stride, u.rate, true);
        if (!h2) {
            if (u.sc >= 0 && !fuse) {
                // slim.conv2d(1x1, stride, SAME): pad 0, samples x[::s, ::s]
                rc = run_conv(net, net->layers[u.sc], xin, B, h, w, 0, 0, ho, wo, nullptr, 0, 0, 0, false, 0, 0, SC, s, nullptr, x_rng);
                if (rc) return rc;
                res = SC; res_s = 1; res_H = ho; res_W = wo;
            }
            rc = run_conv(net, net->layers[u.c1], xin, B, h, w, 0, 0, h, w, nullptr, 0, 0, 0, true, 0, 0, R1, s, nullptr, x_rng);
            if (rc) return rc;
            rc = run_conv(net, net->layers[u.c2], R1, B, h, w, pb_h, pb_w, ho, wo, nullptr, 0, 0, 0, true, 0, 0, R2, s, nullptr,
                          net->amax(u.c1));
            if (rc) return rc;
            if (fuse)
                rc = run_conv_fused_shortcut(net, u, R2, xin, B, h, w, xout, s, net->amax(u.c2), x_rng);
            else
                rc = run_conv(net, net->layers[u.c3], R2, B, ho, wo, 0, 0, ho, wo, res, res_s, res_H, res_W, true, 0, 0, xout, s,
                              nullptr, net->amax(u.c2));
            if (rc) return rc;
        } else {
            auto spec = [&](int in_exp, int out_li, int r_fmt = 0, int r_exp = 0) {
                H2Spec q; q.in_fmt = FMT; q.in_exp = in_exp; q.out_fmt = FMT; q.out_exp = net->act_exp[out_li]; q.res_fmt = r_fmt ? FMT : 0; q.res_exp = r_exp;
                return q;
            };
            // Ra: this unit's conv1 output; Rb: its conv2 output (the chain writes the NEXT unit's conv1 output back into Ra; the unit
            // kernel, which still reads Ra's halos while it writes, into Rb -- the two regions then trade places)
            if (!r1_ready) {
                rc = layer(u.c1, [&] { return run_conv(net, net->layers[u.c1], xin, B, h, w, 0, 0, h, w, nullptr, 0, 0, 0, true, 0, 0, Ra, s,
                                                       nullptr, nullptr, spec(x_exp, u.c1)); });
                if (rc) return rc;
            }
            r1_ready = false;
            const bool chain_ok = chain_on && (size_t)ui + 1 < net->units.size() && net->chains[ui].ok;
            // the K-concatenated shortcut needs R2 on X's scale: known from calibration (unit_fuse_ok) before conv2 runs
            const bool unit_k = chain_ok && unit_env && net->chains[ui].unit && (net->chains[ui].res != 0 || fuse);
            bool share_ok = true;
            if (!unit_k) {
                rc = layer(u.c2, [&] { return run_conv(net, net->layers[u.c2], Ra, B, h, w, pb_h, pb_w, ho, wo, nullptr, 0, 0, 0, true, 0, 0, Rb, s,
                                                       nullptr, nullptr, spec(net->act_exp[u.c1], u.c2)); },
                           (fuse && calib) ? x_exp : dgp_net::H2_NONE, &share_ok);
                if (rc) return rc;
            }
            if (calib && can_fuse) { net->unit_fuse_ok[ui] = share_ok ? 1 : 0; fuse = share_ok; }
            const bool chain = chain_ok && (net->chains[ui].res != 0 || fuse);
            if (u.sc >= 0 && !fuse) {
                rc = layer(u.sc, [&] { return run_conv(net, net->layers[u.sc], xin, B, h, w, 0, 0, ho, wo, nullptr, 0, 0, 0, false, 0, 0, SC, s,
                                                       nullptr, nullptr, spec(x_exp, u.sc)); });
                if (rc) return rc;
                res = SC; res_s = 1; res_H = ho; res_W = wo; res_exp = net->act_exp[u.sc];
            }
            if (unit_k) {
                rc = run_chain(net, ui, nullptr, xin, B, ho, wo, h, w, xout, Rb, s, x_exp, Ra);
                std::swap(Ra, Rb);
                r1_ready = true;
            } else if (chain) {
                rc = run_chain(net, ui, Rb, xin, B, ho, wo, h, w, xout, Ra, s, x_exp);
                r1_ready = true;
            } else if (fuse)
                rc = layer(u.c3, [&] { return run_conv_fused_shortcut(net, u, Rb, xin, B, h, w, xout, s, nullptr, nullptr,
                                                                      spec(x_exp, u.c3)); });      // R2 shares X's scale (calibration)
            else
                rc = layer(u.c3, [&] { return run_conv(net, net->layers[u.c3], Rb, B, ho, wo, 0, 0, ho, wo, res, res_s, res_H, res_W, true, 0,
                                                       0, xout, s, nullptr, nullptr, spec(net->act_exp[u.c2], u.c3, 1, res_exp)); });
            if (rc) return rc;
            x_exp = net->act_exp[u.c3];
        }
        x_rng = net->amax(u.c3);
        cur ^= 1; h = ho; w = wo;
        ++ui;
    }
    const float* feat = X[cur];
    if (features) {
        if (h2) {
            e = tier ? launch_h1_to_f32(feat, (long long)B * h * w * 2048 / 8, ldexpf(1.f, -x_exp), features, s)
                     : launch_h2_to_f32(feat, (long long)B * h * w * 2048 / 8, ldexpf(1.f, -x_exp), features, s);
            if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("features: ") + hipGetErrorString(e));
        } else
            HIP_TRY(hipMemcpyAsync(features, feat, (size_t)B * h * w * 2048 * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    float* sm = scmap ? scmap : (float*)(ws + pl.off_scmap);
    float* slabs = (float*)(ws + pl.off_slabs);
    H2Spec hs;
    if (h2) { hs.in_fmt = FMT; hs.in_exp = x_exp; }
    const bool pw = h2 || (head_pw && f16_mode && net->wmax_valid && x_rng && net->layers[net->head_part].d_wh3_pw);
    if (pw) rc = run_head_pointwise(net, net->layers[net->head_part], feat, B, h, w, d.num_joints, sm, s, R1, x_rng, hs);
    else rc = run_conv(net, net->layers[net->head_part], feat, B, h, w, 1, 1, h, w, nullptr, 0, 0, 0, false, 1, d.num_joints,
                       sm, s, slabs);
    if (rc) return rc;
    if (locref) {
        if (pw && net->layers[net->head_locref].d_wh3_pw)
            rc = run_head_pointwise(net, net->layers[net->head_locref], feat, B, h, w, 2 * d.num_joints, locref, s, R1, x_rng, hs);
        else rc = run_conv(net, net->layers[net->head_locref], feat, B, h, w, 1, 1, h, w, nullptr, 0, 0, 0, false, 1,
                           2 * d.num_joints, locref, s, slabs);
        if (rc) return rc;
    }
    if (h2) {
        if (calib) {                      // freeze the scales: the device copy feeds the per-forward range check
            std::vector<int> ex(net->layers.size(), dgp_net::H2_NONE);
            ex[net->conv1] = net->act_exp[net->conv1];
            int k = 0;
            for (const Unit& u : net->units) {
                ex[u.c1] = net->act_exp[u.c1]; ex[u.c2] = net->act_exp[u.c2]; ex[u.c3] = net->act_exp[u.c3];
                if (u.sc >= 0) ex[u.sc] = net->act_exp[u.sc];       // (an unlaunched shortcut conv tracks 0: never flagged)
                ++k;
            }
            HIP_TRY(hipStreamSynchronize(s));
            HIP_TRY(hipMemcpy(net->d_exps, ex.data(), ex.size() * sizeof(int), hipMemcpyHostToDevice));
            net->h2_calibrated = true;
            ++net->h2_calibrations;
        }
        e = launch_h2_range_check(net->d_amax, net->d_exps, (int)net->layers.size(), net->d_flag, s);
        if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("range check: ") + hipGetErrorString(e));
    }
    if (calib && chain_tier) return dgp_forward(net, frames, batch, workspace, workspace_bytes, scmap, locref, features, stream);   // the chained pass
    if (net->prof_on && net->prof_used < net->prof_slots && !net->prof_in_infer) ++net->prof_used;
    return DGP_OK;
}

int dgp_soft_argmax(const float* scmap, int32_t B, int32_t H, int32_t W, int32_t C, float gamma, int32_t gauss_len,
                    float* mu, float* conf, int32_t* idx, float* pmap, void* stream) {
    if (!scmap || !mu || !conf || !idx) return fail(DGP_ERR_INVALID, "dgp_soft_argmax: null argument");
    if (B < 0 || H < 1 || W < 1 || C < 1 || gauss_len < 0 || gauss_len > 7)
        return fail(DGP_ERR_INVALID, "dgp_soft_argmax: bad shape / gauss_len (0..7)");
    if (B == 0) return DGP_OK;
    hipError_t e = launch_soft_argmax(scmap, B, H, W, C, gamma, gauss_len, mu, conf, idx, pmap, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("soft_argmax: ") + hipGetErrorString(e));
    return DGP_OK;
}

int dgp_pmap_threshold(float* pmap, int32_t B, int32_t H, int32_t W, int32_t C, float th, float* mu, void* stream) {
    if (!pmap || !mu) return fail(DGP_ERR_INVALID, "dgp_pmap_threshold: null argument");
    if (B < 0 || H < 1 || W < 1 || C < 1 || C > 65535 || B > 65535) return fail(DGP_ERR_INVALID, "dgp_pmap_threshold: bad shape");
    if (!(th >= 0.f)) return fail(DGP_ERR_INVALID, "dgp_pmap_threshold: th must be >= 0");
    if (B == 0) return DGP_OK;
    hipError_t e = launch_pmap_threshold(pmap, B, H, W, C, th, mu, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("pmap_threshold: ") + hipGetErrorString(e));
    return DGP_OK;
}

int dgp_hard_argmax(const float* scmap, const float* locref, int32_t B, int32_t H, int32_t W, int32_t C,
                    int32_t* idx, float* prob, float* offs, void* stream) {
    if (!scmap || !idx || !prob || !offs) return fail(DGP_ERR_INVALID, "dgp_hard_argmax: null argument");
    if (B < 0 || H < 1 || W < 1 || C < 1) return fail(DGP_ERR_INVALID, "dgp_hard_argmax: bad shape");
    if (B == 0) return DGP_OK;
    hipError_t e = launch_hard_argmax(scmap, locref, B, H, W, C, idx, prob, offs, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("hard_argmax: ") + hipGetErrorString(e));
    return DGP_OK;
}

static int infer_impl(dgp_net* net, const uint8_t* frames, int32_t batch, void* workspace, size_t workspace_bytes, float gamma,
                      int32_t gauss_len, float* mu, float* conf, int32_t* idx, float* scmap_out, int record_stride, void* stream) {
    if (!net) return fail(DGP_ERR_INVALID, "dgp_infer: null net");
    if (!mu || !conf || !idx) return fail(DGP_ERR_INVALID, "dgp_infer: null output");
    if (gauss_len < 0 || gauss_len > 7) return fail(DGP_ERR_INVALID, "dgp_infer: gauss_len (0..7)");
    // (the soft-argmax keeps one joint's map in LDS up to 38 400 cells -- frames up to ~1920 x 1280 -- and streams larger ones: launch_soft_argmax)
    net->prof_in_infer = true;
    int rc = dgp_forward(net, frames, batch, workspace, workspace_bytes, scmap_out, nullptr, nullptr, stream);
    net->prof_in_infer = false;
    if (rc) return rc;
    const float* sm = scmap_out ? scmap_out : (const float*)((char*)workspace + make_plan(net, batch).off_scmap);
    if (batch > 0) {
        ProfScope ps(net, (hipStream_t)stream, "soft_argmax", 0.0);
        hipError_t e = launch_soft_argmax(sm, batch, 2 * net->fh, 2 * net->fw, net->desc.num_joints, gamma, gauss_len, mu, conf, idx,
                                          nullptr, (hipStream_t)stream, record_stride);
        if (e != hipSuccess) rc = fail(DGP_ERR_HIP, std::string("soft_argmax: ") + hipGetErrorString(e));
    }
    if (net->prof_on && net->prof_used < net->prof_slots) ++net->prof_used;
    return rc;
}

int dgp_infer(dgp_net* net, const uint8_t* frames, int32_t batch, void* workspace, size_t workspace_bytes, float gamma,
              int32_t gauss_len, float* mu, float* conf, int32_t* idx, float* scmap_out, void* stream) {
    return infer_impl(net, frames, batch, workspace, workspace_bytes, gamma, gauss_len, mu, conf, idx, scmap_out, 0, stream);
}

int dgp_infer_packed(dgp_net* net, const uint8_t* frames, int32_t batch, void* workspace, size_t workspace_bytes, float gamma,
                     int32_t gauss_len, float* traj, float* scmap_out, void* stream) {
    if (!traj) return fail(DGP_ERR_INVALID, "dgp_infer_packed: null trajectory");
    return infer_impl(net, frames, batch, workspace, workspace_bytes, gamma, gauss_len, traj, traj + 2,
                      reinterpret_cast<int32_t*>(traj) + 3, scmap_out, 5, stream);
}

int dgp_net_profile_begin(dgp_net* net, int32_t max_steps) {
    if (!net || max_steps < 1) return fail(DGP_ERR_INVALID, "dgp_net_profile_begin: bad argument");
    int nl = 0;
    dgp_net_stats(net, 1, &nl, nullptr);
    nl += 1;   // soft-argmax
    net->prof_free();
    net->prof_launches = nl; net->prof_slots = max_steps; net->prof_used = 0; net->prof_cursor = 0;
    net->prof_names.assign(nl, ""); net->prof_flops.assign(nl, 0.0);
    net->prof_ev.resize((size_t)max_steps * nl * 2);
    for (auto& e : net->prof_ev) HIP_TRY(hipEventCreate(&e));
    net->prof_on = true;
    return DGP_OK;
}

int dgp_net_profile_end(dgp_net* net, int32_t* n_steps, int32_t* n_launches) {
    if (!net) return fail(DGP_ERR_INVALID, "dgp_net_profile_end: null net");
    net->prof_on = false;
    // launches actually recorded per step (fused layers make it smaller than the slots reserved by dgp_net_stats)
    int recorded = 0;
    while (recorded < net->prof_launches && !net->prof_names[recorded].empty()) ++recorded;
    net->prof_launches_used = recorded;
    if (n_steps) *n_steps = net->prof_used;
    if (n_launches) *n_launches = recorded;
    return DGP_OK;
}

int dgp_net_profile_launch(dgp_net* net, int32_t launch, char* name, int32_t name_cap, double* flops, double* avg_ms) {
    if (!net || launch < 0 || launch >= net->prof_launches_used) return fail(DGP_ERR_INVALID, "dgp_net_profile_launch: bad index");
    if (name && name_cap > 0) { strncpy(name, net->prof_names[launch].c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (flops) *flops = net->prof_flops[launch];
    double tot = 0; int cnt = 0;
    for (int st = 0; st < net->prof_used; ++st) {
        hipEvent_t a = net->prof_ev[((size_t)st * net->prof_launches + launch) * 2];
        hipEvent_t b = net->prof_ev[((size_t)st * net->prof_launches + launch) * 2 + 1];
        float ms = 0.f;
        if (hipEventSynchronize(b) != hipSuccess) continue;
        if (hipEventElapsedTime(&ms, a, b) != hipSuccess) continue;
        tot += ms; ++cnt;
    }
    if (avg_ms) *avg_ms = cnt ? tot / cnt : 0.0;
    return DGP_OK;
}

static size_t loss_scratch_layout(const dgp_loss_desc* d, size_t off[8]) {
    const size_t nm = (size_t)d->nt * d->nj;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) / 256 * 256; return r; };
    off[0] = take(nm * 4);        // conf
    off[1] = take(nm * 2 * 4);    // idx
    off[2] = take(nm * 2 * 4);    // t_all
    off[3] = take(nm * 2 * 4);    // dLdt
    off[4] = take(nm * 4);        // kind
    off[5] = take(nm * 4);        // stats
    off[6] = take(8 * 4);         // norm
    off[7] = take(nm * 5 * 4);    // temporal weights [nm] + their derivatives by the four marker coordinates [nm][4]
    return o;
}

int dgp_loss_scratch_bytes(const dgp_loss_desc* d, size_t* out_bytes) {
    if (!d || !out_bytes || d->nt < 1 || d->nj < 1) return fail(DGP_ERR_INVALID, "dgp_loss_scratch_bytes: bad argument");
    size_t off[8];
    *out_bytes = loss_scratch_layout(d, off);
    return DGP_OK;
}

int dgp_loss_fwd_bwd(const dgp_loss_desc* d, const float* pred, const float* locref_pred, const float* targets,
                     const float* locref_map, const float* locref_mask, const int32_t* visible_marker,
                     const int32_t* hidden_marker, const int32_t* visible_in_targets, const float* S0, const float* ws,
                     const float* ws_max, const float* vector_field, const float* wt_batch, float* dpred, float* dlocref,
                     float* mu, float* losses, void* scratch, size_t scratch_bytes, void* stream) {
    if (!d || !pred || !locref_pred || !dpred || !dlocref || !mu || !losses || !scratch)
        return fail(DGP_ERR_INVALID, "dgp_loss_fwd_bwd: null argument");
    if (d->nt < 1 || d->H < 1 || d->W < 1 || d->nj < 1 || d->nl < 0 || d->n_visible < 0 || d->n_hidden < 0)
        return fail(DGP_ERR_INVALID, "dgp_loss_fwd_bwd: bad shape");
    if (d->gauss_len < 1 || d->gauss_len > 7)       // 2 gauss_len + 1 taps in a 16-float LDS array; gauss_len 0 would give 0/0 taps
        return fail(DGP_ERR_INVALID, "dgp_loss_fwd_bwd: gauss_len must be 1..7");
    if (!(d->gm2 == 0 || d->gm2 == 1 || d->gm2 == 2) || !(d->gm3 == 0 || d->gm3 == 3) || (d->gm3 == 3 && d->gm2 == 0))
        return fail(DGP_ERR_INVALID, "dgp_loss_fwd_bwd: Not implemented (gm2 in {0,1,2}, gm3 in {0,3}, gm3=3 needs gm2>0)");
    if ((d->n_visible > 0 && (!visible_marker || !visible_in_targets || !targets || !locref_map || !locref_mask)) ||
        (d->n_hidden > 0 && !hidden_marker) || (d->nl > 0 && (!S0 || !ws || !ws_max)))
        return fail(DGP_ERR_INVALID, "dgp_loss_fwd_bwd: missing marker / target / skeleton arrays");
    size_t off[8];
    if (scratch_bytes < loss_scratch_layout(d, off)) return fail(DGP_ERR_INVALID, "dgp_loss_fwd_bwd: scratch too small");
    char* sc = (char*)scratch;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = launch_soft_argmax(pred, d->nt, d->H, d->W, d->nj, d->gamma, d->gauss_len, mu, (float*)(sc + off[0]),
                                      (int*)(sc + off[1]), nullptr, s);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("loss soft_argmax: ") + hipGetErrorString(e));
    LossArgs a{};
    a.pred = pred; a.locref_pred = locref_pred; a.mu = mu; a.targets = targets; a.locref_map = locref_map;
    a.locref_mask = locref_mask; a.visible_marker = visible_marker; a.hidden_marker = hidden_marker;
    a.visible_in_targets = visible_in_targets; a.S0 = S0; a.ws = ws; a.ws_max = ws_max; a.dpred = dpred; a.dlocref = dlocref;
    a.losses = losses; a.t_all = (float*)(sc + off[2]); a.dLdt = (float*)(sc + off[3]); a.kind = (int*)(sc + off[4]);
    a.stats = (float*)(sc + off[5]); a.norm = (float*)(sc + off[6]);
    a.nt = d->nt; a.H = d->H; a.W = d->W; a.nj = d->nj; a.nl = d->nl; a.n_v = d->n_visible; a.n_h = d->n_hidden;
    a.gm2 = d->gm2; a.gm3 = d->gm3; a.gauss_len = d->gauss_len; a.huber = d->huber;
    a.gamma = d->gamma; a.lengthscale = d->lengthscale; a.stride = d->stride; a.locref_weight = d->locref_loss_weight;
    const double n_vis_tot = d->n_visible_frames_total, n_hid_tot = d->n_frames_total - d->n_visible_frames_total;
    const double n_h = d->n_hidden, n_v_eff = d->n_visible > 0 ? d->n_visible : d->n_hidden;     // fitdgp.py:983-984
    a.hidden_scale = (n_h > 0 && n_hid_tot > 0) ? (float)(n_vis_tot / n_hid_tot * n_h / n_v_eff * d->wn_hidden / d->wn_visible) : 0.f;
    a.use_wt = (d->use_wt && d->nt > 1) ? 1 : 0;
    if (a.use_wt && (!vector_field || !wt_batch || d->Hin < 1 || d->Win < 1))
        return fail(DGP_ERR_INVALID, "dgp_loss_fwd_bwd: use_wt needs vector_field [nt-1,Hin,Win] and wt_batch [nt-1]");
    a.vector_field = vector_field; a.wt_batch = wt_batch; a.wt_w = (float*)(sc + off[7]); a.Hin = d->Hin; a.Win = d->Win;
    a.wt_max = d->wt_max;
    a.temporal_scale = n_v_eff > 0 ? (float)(n_vis_tot / n_v_eff / (n_vis_tot + n_hid_tot) / d->wn_visible) : 0.f;
    a.clique_scale = n_v_eff > 0 ? (float)(1.0 / ((double)d->H * d->W) * n_vis_tot / n_v_eff / (n_vis_tot + n_hid_tot) / d->wn_visible) : 0.f;
    e = launch_loss(a, s);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("loss kernels: ") + hipGetErrorString(e));
    return DGP_OK;
}

int dgp_dlc_loss_fwd_bwd(const float* pred, const float* locref_pred, const float* part_targets,
                         const float* part_weights, const float* locref_targets, const float* locref_mask, int32_t nt,
                         int32_t H, int32_t W, int32_t nj, float locref_loss_weight, int32_t huber, float* dpred,
                         float* dlocref, float* losses, void* scratch, size_t scratch_bytes, void* stream) {
    if (!pred || !part_targets || !dpred || !losses || !scratch)
        return fail(DGP_ERR_INVALID, "dgp_dlc_loss_fwd_bwd: null argument");
    if (locref_pred && (!locref_targets || !locref_mask || !dlocref))
        return fail(DGP_ERR_INVALID, "dgp_dlc_loss_fwd_bwd: locref targets / mask / gradient missing");
    if (nt < 1 || H < 1 || W < 1 || nj < 1) return fail(DGP_ERR_INVALID, "dgp_dlc_loss_fwd_bwd: bad shape");
    if (scratch_bytes < 4 * sizeof(double) || ((uintptr_t)scratch & 7))
        return fail(DGP_ERR_INVALID, "dgp_dlc_loss_fwd_bwd: scratch must be >= 32 bytes, 8-byte aligned");
    DlcLossArgs a{};
    a.pred = pred; a.part_targets = part_targets; a.part_weights = part_weights;
    a.locref_pred = locref_pred; a.locref_targets = locref_targets; a.locref_mask = locref_mask;
    a.dpred = dpred; a.dlocref = dlocref; a.losses = losses; a.acc = (double*)scratch;
    a.n_part = (long long)nt * H * W * nj; a.n_loc = locref_pred ? 2 * a.n_part : 0;
    a.locref_loss_weight = locref_loss_weight; a.huber = huber;
    hipError_t e = launch_dlc_loss(a, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("dlc loss kernels: ") + hipGetErrorString(e));
    return DGP_OK;
}

size_t dgp_packed_weight_floats(int32_t KH, int32_t KW, int32_t Cin, int32_t Cout) {
    if (KH < 1 || KW < 1 || Cin < 4 || (Cin & 3) || Cout < 1) return 0;
    return (size_t)nk_for(KH, KW, Cin) * 8 * coutp_for(Cout) * 4;
}

int dgp_pack_conv_weights(const float* hwio, int32_t KH, int32_t KW, int32_t Cin, int32_t Cout, float* packed) {
    if (!hwio || !packed || dgp_packed_weight_floats(KH, KW, Cin, Cout) == 0)
        return fail(DGP_ERR_INVALID, "dgp_pack_conv_weights: bad argument (Cin must be a multiple of 4)");
    if ((Cin / 4) & (Cin / 4 - 1)) return fail(DGP_ERR_INVALID, "dgp_pack_conv_weights: Cin/4 must be a power of two");
    pack_panels(hwio, KH, KW, Cin, Cout, packed);
    return DGP_OK;
}

int dgp_tensor_absmax(const float* x, size_t n, float* absmax_dev, void* stream) {
    if (!x || !absmax_dev) return fail(DGP_ERR_INVALID, "dgp_tensor_absmax: null argument");
    hipError_t e = launch_absmax(x, (long long)n, absmax_dev, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("absmax: ") + hipGetErrorString(e));
    return DGP_OK;
}

int dgp_conv2d(const dgp_conv_desc* d, const float* x, const float* packed_w, const float* scale, const float* bias,
               const float* residual, float* y, void* stream) {
    return dgp_conv2d_ranged(d, x, packed_w, scale, bias, residual, y, nullptr, nullptr, nullptr, stream);
}

int dgp_conv2d_ranged(const dgp_conv_desc* d, const float* x, const float* packed_w, const float* scale, const float* bias,
                      const float* residual, float* y, const float* x_absmax, const float* w_absmax, float* y_absmax,
                      void* stream) {
    if (!d || !x || !packed_w || !y) return fail(DGP_ERR_INVALID, "dgp_conv2d: null argument");
    if (d->Cin < 4 || (d->Cin & 3) || ((d->Cin / 4) & (d->Cin / 4 - 1)))
        return fail(DGP_ERR_INVALID, "dgp_conv2d: Cin must be 4 * 2^k");
    if (d->res_stride > 0 && !residual) return fail(DGP_ERR_INVALID, "dgp_conv2d: residual missing");
    ConvArgs a{};
    a.in = x; a.wpk = packed_w; a.scale = scale; a.bias = bias; a.res = d->res_stride > 0 ? residual : nullptr;
    a.out = y; a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.log2cin4 = ilog2(d->Cin / 4);
    a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.CoutP = coutp_for(d->Cout);
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.dil = d->rate; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
    a.ntaps = d->KH * d->KW; a.nk = nk_for(d->KH, d->KW, d->Cin); a.M = d->N * d->Ho * d->Wo;
    a.res_s = d->res_stride; a.res_H = d->res_H; a.res_W = d->res_W; a.relu = d->relu; a.out_mode = 0; a.dc_nj = 0;
    if (d->Cout & 3) return fail(DGP_ERR_INVALID, "dgp_conv2d: Cout must be a multiple of 4");
    {
        const double lim = 4294967000.0;
        const double inb = (double)d->N * d->H * d->W * d->Cin * 4, outb = (double)a.M * d->Cout * 4;
        const double resb = a.res ? (double)d->N * d->res_H * d->res_W * d->Cout * 4 : 0.0;
        if (inb > lim || outb > lim || resb > lim) return fail(DGP_ERR_INVALID, "dgp_conv2d: tensor exceeds 4 GiB");
        a.in_bytes = (unsigned)inb; a.out_bytes = (unsigned)outb; a.res_bytes = (unsigned)resb;
        a.w_bytes = (unsigned)((size_t)a.nk * 8 * a.CoutP * 16);
    }
    a.in_absmax = x_absmax; a.w_absmax = w_absmax; a.out_absmax = y_absmax;
    // DGP_CONV2D_CELLS=1 (tests, tuning): split the panel into fp16 cells per call (grow-only scratch, stream-ordered) so that a single
    // layer runs on the engine's compute-side-split / LDS-DMA kernels; the network packs its cells once at load instead
    static const bool cells_env = (dgp_env("DGP_CONV2D_CELLS", 0) != 0);
    if (cells_env && x_absmax && w_absmax && d->Cin >= 32) {
        static void* cells = nullptr;
        static size_t cells_bytes = 0;
        if (cells_bytes < a.w_bytes) {
            if (cells) { (void)hipDeviceSynchronize(); (void)hipFree(cells); }
            if (hipMalloc(&cells, a.w_bytes) != hipSuccess) return fail(DGP_ERR_HIP, "dgp_conv2d: cell scratch");
            cells_bytes = a.w_bytes;
        }
        hipError_t pe = launch_pack_h3(packed_w, a.nk, a.CoutP, w_absmax, cells, (hipStream_t)stream);
        if (pe != hipSuccess) return fail(DGP_ERR_HIP, std::string("dgp_conv2d: pack cells: ") + hipGetErrorString(pe));
        a.wh3 = cells; a.wh3_bytes = a.w_bytes;
    }
    hipError_t e = launch_conv(a, pick_tile(a.M, a.CoutP, a.nk * BK, x_absmax && w_absmax), (hipStream_t)stream);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("dgp_conv2d: ") + hipGetErrorString(e));
    return DGP_OK;
}

/* ---- H2 activation format at the boundary (tests, PoseNet.extract_features): converters, a single conv layer on H2 tensors, and
 * the engine's range status.  See ConvArgs::in_fmt (csrc/dgp_internal.h) and EXPERIMENTS.md section 3. */
int dgp_f32_to_h2(const float* x, size_t n_floats, int32_t scale_exp, void* out, void* stream) {
    if (!x || !out || (n_floats & 7)) return fail(DGP_ERR_INVALID, "dgp_f32_to_h2: null argument / length not a multiple of 8");
    hipError_t e = launch_f32_to_h2(x, (long long)(n_floats / 8), ldexpf(1.f, scale_exp), out, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("dgp_f32_to_h2: ") + hipGetErrorString(e));
    return DGP_OK;
}

int dgp_h2_to_f32(const void* x, size_t n_floats, int32_t scale_exp, float* out, void* stream) {
    if (!x || !out || (n_floats & 7)) return fail(DGP_ERR_INVALID, "dgp_h2_to_f32: null argument / length not a multiple of 8");
    hipError_t e = launch_h2_to_f32(x, (long long)(n_floats / 8), ldexpf(1.f, -scale_exp), out, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("dgp_h2_to_f32: ") + hipGetErrorString(e));
    return DGP_OK;
}

int dgp_f32_to_h1(const float* x, size_t n_floats, int32_t scale_exp, void* out, void* stream) {
    if (!x || !out || (n_floats & 7)) return fail(DGP_ERR_INVALID, "dgp_f32_to_h1: null argument / length not a multiple of 8");
    hipError_t e = launch_f32_to_h1(x, (long long)(n_floats / 8), ldexpf(1.f, scale_exp), out, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("dgp_f32_to_h1: ") + hipGetErrorString(e));
    return DGP_OK;
}

int dgp_h1_to_f32(const void* x, size_t n_floats, int32_t scale_exp, float* out, void* stream) {
    if (!x || !out || (n_floats & 7)) return fail(DGP_ERR_INVALID, "dgp_h1_to_f32: null argument / length not a multiple of 8");
    hipError_t e = launch_h1_to_f32(x, (long long)(n_floats / 8), ldexpf(1.f, -scale_exp), out, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("dgp_h1_to_f32: ") + hipGetErrorString(e));
    return DGP_OK;
}

static int conv2d_cells(int fmt, const dgp_conv_desc* d, const void* x_h2, int32_t x_exp, const float* packed_w, const float* w_absmax,
                        const float* scale, const float* bias, const void* residual, int32_t res_is_h2, int32_t res_exp, void* y,
                        int32_t y_is_h2, int32_t y_exp, float* y_absmax, void* cells_scratch, void* stream);

int dgp_conv2d_h2(const dgp_conv_desc* d, const void* x_h2, int32_t x_exp, const float* packed_w, const float* w_absmax,
                  const float* scale, const float* bias, const void* residual, int32_t res_is_h2, int32_t res_exp, void* y,
                  int32_t y_is_h2, int32_t y_exp, float* y_absmax, void* cells_scratch, void* stream) {
    return conv2d_cells(1, d, x_h2, x_exp, packed_w, w_absmax, scale, bias, residual, res_is_h2, res_exp, y, y_is_h2, y_exp, y_absmax,
                        cells_scratch, stream);
}

int dgp_conv2d_h1(const dgp_conv_desc* d, const void* x_h1, int32_t x_exp, const float* packed_w, const float* w_absmax,
                  const float* scale, const float* bias, const void* residual_h1, int32_t res_exp, void* y,
                  int32_t y_is_h1, int32_t y_exp, float* y_absmax, void* cells_scratch, void* stream) {
    if (d && (d->Cin & 63)) return fail(DGP_ERR_INVALID, "dgp_conv2d_h1: Cin must be a multiple of 64 (a K-step is 64 channels)");
    return conv2d_cells(2, d, x_h1, x_exp, packed_w, w_absmax, scale, bias, residual_h1, 1, res_exp, y, y_is_h1, y_exp, y_absmax,
                        cells_scratch, stream);
}

static int conv2d_cells(int fmt, const dgp_conv_desc* d, const void* x_h2, int32_t x_exp, const float* packed_w, const float* w_absmax,
                        const float* scale, const float* bias, const void* residual, int32_t res_is_h2, int32_t res_exp, void* y,
                        int32_t y_is_h2, int32_t y_exp, float* y_absmax, void* cells_scratch, void* stream) {
    if (!d || !x_h2 || !packed_w || !w_absmax || !y || !cells_scratch) return fail(DGP_ERR_INVALID, "dgp_conv2d_h2: null argument");
    if (d->Cin < 32 || (d->Cin & 7) || ((d->Cin / 4) & (d->Cin / 4 - 1)) || (d->Cout & 7))
        return fail(DGP_ERR_INVALID, "dgp_conv2d_h2: Cin must be 4 * 2^k >= 32, Cout a multiple of 8");
    if (d->res_stride > 0 && !residual) return fail(DGP_ERR_INVALID, "dgp_conv2d_h2: residual missing");
    if (d->res_stride > 0 && res_is_h2 && !y_is_h2)
        return fail(DGP_ERR_INVALID, "dgp_conv2d_h2: an H2 residual needs an H2 output (the fp32-output epilogue adds fp32 residuals only)");
    ConvArgs a{};
    a.in = (const float*)x_h2; a.wpk = packed_w; a.scale = scale; a.bias = bias; a.res = d->res_stride > 0 ? (const float*)residual : nullptr;
    a.out = (float*)y; a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.log2cin4 = ilog2(d->Cin / 4);
    a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.CoutP = coutp_for(d->Cout);
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.dil = d->rate; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
    a.ntaps = d->KH * d->KW; a.nk = nk_for(d->KH, d->KW, d->Cin); a.M = d->N * d->Ho * d->Wo;
    a.res_s = d->res_stride; a.res_H = d->res_H; a.res_W = d->res_W; a.relu = d->relu;
    const double lim = 4294967000.0;
    const double inb = (double)d->N * d->H * d->W * d->Cin * 4, outb = (double)a.M * d->Cout * 4;
    const double resb = a.res ? (double)d->N * d->res_H * d->res_W * d->Cout * 4 : 0.0;
    if (inb > lim || outb > lim || resb > lim) return fail(DGP_ERR_INVALID, "dgp_conv2d_h2: tensor exceeds 4 GiB");
    a.in_bytes = (unsigned)inb; a.out_bytes = (unsigned)outb; a.res_bytes = (unsigned)resb;
    a.w_bytes = (unsigned)((size_t)a.nk * 8 * a.CoutP * 16);
    H2Spec h; h.in_fmt = fmt; h.in_exp = x_exp; h.out_fmt = y_is_h2 ? fmt : 0; h.out_exp = y_exp; h.res_fmt = (a.res && res_is_h2) ? fmt : 0; h.res_exp = res_exp;
    apply_h2(a, h);
    a.w_absmax = w_absmax; a.out_absmax = y_absmax;
    hipError_t pe = fmt == 2 ? launch_pack_h1(packed_w, a.nk, a.CoutP, w_absmax, cells_scratch, (hipStream_t)stream)
                             : launch_pack_h3(packed_w, a.nk, a.CoutP, w_absmax, cells_scratch, (hipStream_t)stream);
    if (pe != hipSuccess) return fail(DGP_ERR_HIP, std::string("dgp_conv2d_h2: pack cells: ") + hipGetErrorString(pe));
    a.wh3 = cells_scratch; a.wh3_bytes = a.w_bytes;
    hipError_t e = launch_conv(a, pick_tile(a.M, a.CoutP, a.nk * BK, true), (hipStream_t)stream);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("dgp_conv2d_h2: ") + hipGetErrorString(e));
    return DGP_OK;
}

/* conv3 (+ shortcut, ReLU) of a bottleneck unit and conv1 of the next unit as ONE launch on H2 tensors (the engine's chain kernel,
 * csrc/dgp_chain.hip) -- layer-level entry for tests: weights and BN affines are HOST arrays, packed per call. */
static int chain_layer(bool h1, int32_t N, int32_t Ho, int32_t Wo, int32_t C, int32_t C1, int32_t CIN2, int32_t res_mode, int32_t res_H, int32_t res_W,
                 const void* r2_h2, int32_t r2_exp, const void* src2_h2, int32_t src2_exp,
                 const float* w3cat, const float* scale3, const float* bias3, const float* w1, const float* scale1, const float* bias1,
                 void* xout_h2, int32_t xout_exp, void* r1_h2, int32_t r1_exp, float* xout_absmax, float* r1_absmax, void* stream) {
    const size_t EB = h1 ? 2 : 4;                  // bytes per channel: H1 cells / H2 cell pairs
    if (!r2_h2 || !src2_h2 || !w3cat || !w1 || !xout_h2 || !r1_h2) return fail(DGP_ERR_INVALID, h1 ? "dgp_chain_h1: null argument" : "dgp_chain_h2: null argument");
    if (!chain_supported(C, C1, CIN2, res_mode))
        return fail(DGP_ERR_INVALID, "dgp_chain_h2 / _h1: no kernel instance for this (C, C1, CIN2, res_mode)");
    if (res_mode == 0 && src2_exp != r2_exp) return fail(DGP_ERR_INVALID, "dgp_chain_h2 / _h1: the K-concatenated source must share R2's scale");
    ChainPlan cp;
    int rc = build_chain_plan(cp, C, C1, CIN2, res_mode, w3cat, scale3, bias3, w1, scale1, bias1);
    if (rc) return rc;
    ChainArgs a{};
    a.h1 = h1 ? 1 : 0;
    a.r2 = r2_h2; a.src2 = src2_h2; a.xout = xout_h2; a.r1out = r1_h2; a.wfrag = h1 ? cp.d_frags_h1 : cp.d_frags; a.sc1 = cp.d_sc1; a.bi1 = cp.d_bi1;
    a.M = N * Ho * Wo; a.HoWo = Ho * Wo; a.Wo = Wo; a.res_H = res_H; a.res_W = res_W;
    a.post1 = ldexpf(1.f, -(r2_exp + cp.w3_exp)); a.post2 = ldexpf(1.f, -(xout_exp + cp.w1_exp));
    a.res_inv_scale = ldexpf(1.f, -src2_exp); a.xout_scale = ldexpf(1.f, xout_exp); a.r1_scale = ldexpf(1.f, r1_exp);
    a.xout_absmax = xout_absmax; a.r1_absmax = r1_absmax;
    a.r2_bytes = (unsigned)((size_t)a.M * C * EB); a.xout_bytes = (unsigned)((size_t)a.M * C * 4 * EB); a.r1_bytes = (unsigned)((size_t)a.M * C1 * EB);
    a.src2_bytes = (unsigned)(res_mode == 0 ? (size_t)a.M * CIN2 * EB : (size_t)N * res_H * res_W * C * 4 * EB);
    a.w_bytes = h1 ? cp.frag_bytes_h1 : cp.frag_bytes;
    hipError_t e = launch_chain(a, C, C1, CIN2, res_mode, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);       // (the fragments are freed below)
    free_chain_plan(cp);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string(h1 ? "dgp_chain_h1: " : "dgp_chain_h2: ") + hipGetErrorString(e));
    return DGP_OK;
}

/* The unit kernel at layer level (tests): conv2 (3x3, stride 1, SAME) + BN + ReLU of a bottleneck unit, its conv3 + shortcut + ReLU and
 * conv1 of the next unit in one launch; R2 never leaves the registers. */
static int unit_layer(bool h1, int32_t N, int32_t H, int32_t W, int32_t C, int32_t C1, int32_t CIN2, int32_t res_mode,
                const void* r1_h2, int32_t r1_exp, const void* src2_h2, int32_t src2_exp,
                const float* w2, const float* scale2, const float* bias2, int32_t r2_exp,
                const float* w3cat, const float* scale3, const float* bias3, const float* w1, const float* scale1, const float* bias1,
                void* xout_h2, int32_t xout_exp, void* r1out_h2, int32_t r1out_exp, float* r2_absmax, float* xout_absmax, float* r1_absmax,
                void* stream) {
    const size_t EB = h1 ? 2 : 4;
    if (!r1_h2 || !src2_h2 || !w2 || !w3cat || !w1 || !xout_h2 || !r1out_h2) return fail(DGP_ERR_INVALID, h1 ? "dgp_unit_h1: null argument" : "dgp_unit_h2: null argument");
    if (!unit_supported(C, C1, CIN2, res_mode))
        return fail(DGP_ERR_INVALID, "dgp_unit_h2 / _h1: no kernel instance for this (C, C1, CIN2, res_mode)");
    if (res_mode == 0 && src2_exp != r2_exp) return fail(DGP_ERR_INVALID, "dgp_unit_h2 / _h1: the K-concatenated source must share R2's scale");
    if (r1_h2 == r1out_h2) return fail(DGP_ERR_INVALID, "dgp_unit_h2 / _h1: r1out must not alias r1 (halo reads)");
    ChainPlan cp;
    int rc = build_chain_plan(cp, C, C1, CIN2, res_mode, w3cat, scale3, bias3, w1, scale1, bias1, w2, scale2, bias2);
    if (rc) return rc;
    ChainArgs a{};
    a.h1 = h1 ? 1 : 0;
    a.r1in = r1_h2; a.src2 = src2_h2; a.xout = xout_h2; a.r1out = r1out_h2; a.wfrag = h1 ? cp.d_frags_h1 : cp.d_frags; a.sc1 = cp.d_sc1; a.bi1 = cp.d_bi1;
    a.M = N * H * W; a.HoWo = H * W; a.Wo = W; a.res_H = H; a.res_W = W; a.H = H; a.W = W;
    a.post0 = ldexpf(1.f, -(r1_exp + cp.w2_exp)); a.r2_scale = ldexpf(1.f, r2_exp);
    a.post1 = ldexpf(1.f, -(r2_exp + cp.w3_exp)); a.post2 = ldexpf(1.f, -(xout_exp + cp.w1_exp));
    a.res_inv_scale = ldexpf(1.f, -src2_exp); a.xout_scale = ldexpf(1.f, xout_exp); a.r1_scale = ldexpf(1.f, r1out_exp);
    a.r2_absmax = r2_absmax; a.xout_absmax = xout_absmax; a.r1_absmax = r1_absmax;
    a.r1in_bytes = (unsigned)((size_t)a.M * C * EB); a.xout_bytes = (unsigned)((size_t)a.M * C * 4 * EB); a.r1_bytes = (unsigned)((size_t)a.M * C1 * EB);
    a.src2_bytes = (unsigned)(res_mode == 0 ? (size_t)a.M * CIN2 * 4 : (size_t)a.M * C * 4 * EB);
    a.w_bytes = h1 ? cp.frag_bytes_h1 : cp.frag_bytes;
    hipError_t e = launch_unit(a, N, C, C1, CIN2, res_mode, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    free_chain_plan(cp);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string(h1 ? "dgp_unit_h1: " : "dgp_unit_h2: ") + hipGetErrorString(e));
    return DGP_OK;
}
int dgp_chain_h2(int32_t N, int32_t Ho, int32_t Wo, int32_t C, int32_t C1, int32_t CIN2, int32_t res_mode, int32_t res_H, int32_t res_W,
                 const void* r2_h2, int32_t r2_exp, const void* src2_h2, int32_t src2_exp,
                 const float* w3cat, const float* scale3, const float* bias3, const float* w1, const float* scale1, const float* bias1,
                 void* xout_h2, int32_t xout_exp, void* r1_h2, int32_t r1_exp, float* xout_absmax, float* r1_absmax, void* stream) {
    return chain_layer(false, N, Ho, Wo, C, C1, CIN2, res_mode, res_H, res_W, r2_h2, r2_exp, src2_h2, src2_exp, w3cat, scale3, bias3, w1, scale1, bias1,
                       xout_h2, xout_exp, r1_h2, r1_exp, xout_absmax, r1_absmax, stream);
}
/* the same launches on H1 tensors (the 16-bit tier: 2 bytes per channel, high weight fragments only, one MFMA per product) */
int dgp_chain_h1(int32_t N, int32_t Ho, int32_t Wo, int32_t C, int32_t C1, int32_t CIN2, int32_t res_mode, int32_t res_H, int32_t res_W,
                 const void* r2_h1, int32_t r2_exp, const void* src2_h1, int32_t src2_exp,
                 const float* w3cat, const float* scale3, const float* bias3, const float* w1, const float* scale1, const float* bias1,
                 void* xout_h1, int32_t xout_exp, void* r1_h1, int32_t r1_exp, float* xout_absmax, float* r1_absmax, void* stream) {
    return chain_layer(true, N, Ho, Wo, C, C1, CIN2, res_mode, res_H, res_W, r2_h1, r2_exp, src2_h1, src2_exp, w3cat, scale3, bias3, w1, scale1, bias1,
                       xout_h1, xout_exp, r1_h1, r1_exp, xout_absmax, r1_absmax, stream);
}
int dgp_unit_h2(int32_t N, int32_t H, int32_t W, int32_t C, int32_t C1, int32_t CIN2, int32_t res_mode,
                const void* r1_h2, int32_t r1_exp, const void* src2_h2, int32_t src2_exp,
                const float* w2, const float* scale2, const float* bias2, int32_t r2_exp,
                const float* w3cat, const float* scale3, const float* bias3, const float* w1, const float* scale1, const float* bias1,
                void* xout_h2, int32_t xout_exp, void* r1out_h2, int32_t r1out_exp, float* r2_absmax, float* xout_absmax, float* r1_absmax,
                void* stream) {
    return unit_layer(false, N, H, W, C, C1, CIN2, res_mode, r1_h2, r1_exp, src2_h2, src2_exp, w2, scale2, bias2, r2_exp, w3cat, scale3, bias3, w1, scale1, bias1,
                      xout_h2, xout_exp, r1out_h2, r1out_exp, r2_absmax, xout_absmax, r1_absmax, stream);
}
int dgp_unit_h1(int32_t N, int32_t H, int32_t W, int32_t C, int32_t C1, int32_t CIN2, int32_t res_mode,
                const void* r1_h1, int32_t r1_exp, const void* src2_h1, int32_t src2_exp,
                const float* w2, const float* scale2, const float* bias2, int32_t r2_exp,
                const float* w3cat, const float* scale3, const float* bias3, const float* w1, const float* scale1, const float* bias1,
                void* xout_h1, int32_t xout_exp, void* r1out_h1, int32_t r1out_exp, float* r2_absmax, float* xout_absmax, float* r1_absmax,
                void* stream) {
    return unit_layer(true, N, H, W, C, C1, CIN2, res_mode, r1_h1, r1_exp, src2_h1, src2_exp, w2, scale2, bias2, r2_exp, w3cat, scale3, bias3, w1, scale1, bias1,
                      xout_h1, xout_exp, r1out_h1, r1out_exp, r2_absmax, xout_absmax, r1_absmax, stream);
}

int dgp_net_range_status(dgp_net* net, int32_t* overflow, int32_t* calibrations, void* stream) {
    if (!net) return fail(DGP_ERR_INVALID, "dgp_net_range_status: null net");
    int flag[2] = {0, 0};
    if (net->d_flag) {
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        HIP_TRY(hipMemcpy(flag, net->d_flag, sizeof flag, hipMemcpyDeviceToHost));
        if (flag[0]) {
            HIP_TRY(hipMemset(net->d_flag, 0, sizeof flag));
            net->h2_calibrated = false;         // the next forward re-calibrates on its batch, with more headroom: a quiet first
            net->h2_head = std::min(net->h2_head + 3, 12);      // batch must not under-size the scales again (costs no accuracy up to ~10 bits)
        }
    }
    if (overflow) *overflow = flag[0] ? 1 : 0;
    if (calibrations) *calibrations = net->h2_calibrations;
    return DGP_OK;
}

int dgp_net_recalibrate(dgp_net* net) {
    if (!net) return fail(DGP_ERR_INVALID, "dgp_net_recalibrate: null net");
    net->h2_calibrated = false;
    return DGP_OK;
}

int dgp_net_widen(dgp_net* net) {
    if (!net) return fail(DGP_ERR_INVALID, "dgp_net_widen: null net");
    net->h2_calibrated = false;
    net->h2_head = std::min(net->h2_head + 3, 12);
    return DGP_OK;
}

int dgp_net_copy_scales(dgp_net* dst, const dgp_net* src, void* stream) {
    if (!dst || !src) return fail(DGP_ERR_INVALID, "dgp_net_copy_scales: null net");
    if (!src->h2_calibrated) return fail(DGP_ERR_STATE, "dgp_net_copy_scales: the source engine is not calibrated");
    if (!dst->loaded || !dst->d_exps || dst->layers.size() != src->layers.size() || dst->units.size() != src->units.size() || dst->tier != src->tier ||
        dst->desc.in_h != src->desc.in_h || dst->desc.in_w != src->desc.in_w)
        return fail(DGP_ERR_INVALID, "dgp_net_copy_scales: the engines differ (layers, tier or frame size)");
    dst->act_exp = src->act_exp;
    dst->unit_fuse_ok = src->unit_fuse_ok;
    dst->h2_head = src->h2_head;
    std::vector<int> ex(dst->layers.size(), dgp_net::H2_NONE);          // (what the calibration pass uploads for the per-forward range check)
    ex[dst->conv1] = dst->act_exp[dst->conv1];
    for (const Unit& u : dst->units) {
        ex[u.c1] = dst->act_exp[u.c1]; ex[u.c2] = dst->act_exp[u.c2]; ex[u.c3] = dst->act_exp[u.c3];
        if (u.sc >= 0) ex[u.sc] = dst->act_exp[u.sc];
    }
    hipError_t e;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));                  // (a forward of `dst` still in flight reads the old exponents)
    HIP_TRY(hipMemcpy(dst->d_exps, ex.data(), ex.size() * sizeof(int), hipMemcpyHostToDevice));
    dst->h2_calibrated = true;
    ++dst->h2_calibrations;
    return DGP_OK;
}

int dgp_net_reset_scales(dgp_net* net) {
    if (!net) return fail(DGP_ERR_INVALID, "dgp_net_reset_scales: null net");
    net->h2_calibrated = false;
    net->h2_head = 4;                   // (dgp_net_load_weights' value)
    return DGP_OK;
}

int dgp_maxpool_3x3s2_same(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, float* y, void* stream) {
    if (!x || !y || (C & 3)) return fail(DGP_ERR_INVALID, "dgp_maxpool_3x3s2_same: bad argument (C % 4)");
    hipError_t e = launch_maxpool(x, N, H, W, C, y, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("maxpool: ") + hipGetErrorString(e));
    return DGP_OK;
}

int dgp_preprocess_u8(const uint8_t* frames, int64_t n_pixels, const float mean[3], float* out_nhwc4, void* stream) {
    if (!frames || !mean || !out_nhwc4) return fail(DGP_ERR_INVALID, "dgp_preprocess_u8: null argument");
    hipError_t e = launch_preprocess(frames, n_pixels, mean[0], mean[1], mean[2], out_nhwc4, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("preprocess: ") + hipGetErrorString(e));
    return DGP_OK;
}

int dgp_motion_energy(const uint8_t* frames, int64_t frame_bytes, int32_t n_frames, const uint8_t* prev_frame, uint64_t* sums,
                      void* stream) {
    if (!frames || !sums || frame_bytes <= 0 || n_frames <= 0) return fail(DGP_ERR_INVALID, "dgp_motion_energy: bad argument");
    hipError_t e = launch_motion_energy(frames, frame_bytes, n_frames, prev_frame, (unsigned long long*)sums, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("motion energy: ") + hipGetErrorString(e));
    return DGP_OK;
}

}  // extern "C"
