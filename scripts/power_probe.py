#!/usr/bin/env python3
"""Probe: board power and shader clock (sysfs: hwmon power1_average / power1_cap, pp_dpm_sclk) while the engine runs -- the whole chip
(two plain streams), half of it (one CU-masked stream: CUs 0-127), and one layer type at a time is not needed: the two numbers say whether
the bench workload runs at the power cap.  python scripts/power_probe.py [parity|f16]"""
import ctypes, glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepgraphpose_amd import engine, synthetic
tier = sys.argv[1] if len(sys.argv) > 1 else "parity"
H, W, NJ, B = 480, 640, 4, 32
torch.zeros(1, device="cuda")
pr = torch.cuda.get_device_properties(0)
want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
cards = [os.path.dirname(q) for q in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk") if want in os.path.realpath(os.path.dirname(q)).lower()]
card = cards[0] if cards else None
hw = glob.glob(os.path.join(card, "hwmon", "hwmon*")) if card else []
def rd(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None
def sample():
    out = {}
    if hw:
        for k in ("power1_average", "power1_input", "power1_cap"):
            v = rd(os.path.join(hw[0], k))
            if v and v.isdigit():
                out[k] = int(v) / 1e6
    s = rd(os.path.join(card, "pp_dpm_sclk")) if card else None
    if s:
        for line in s.splitlines():
            if line.strip().endswith("*"):
                out["sclk"] = int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
    return out
print("idle:", sample(), flush=True)
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(lo, hi):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if lo <= 32 * w + b < hi) for w in range(8)])
    st = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words) == 0
    return torch.cuda.ExternalStream(st.value)
wts = synthetic.make_weights(50, NJ, False, seed=0, head_std=0.05)
frames = torch.from_numpy(synthetic.make_frames(B, H, W, NJ, seed=100)).cuda()
nets = [engine.DGPNet(50, NJ, H, W, max_batch=B, tier=tier) for _ in range(2)]
outs = [torch.zeros((B, NJ, 5), device="cuda") for _ in range(2)]
for n in nets:
    n.load_weights(wts)
def load(streams, seconds):
    acc, stop = [], [False]
    def watch():
        while not stop[0]:
            time.sleep(0.1); acc.append(sample())
    th = threading.Thread(target=watch); th.start()
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(16):
            with torch.cuda.stream(streams[k % len(streams)]):
                nets[k % len(streams)].infer_packed(frames, outs[k % len(streams)])
            k += 1
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop[0] = True; th.join()
    acc = acc[len(acc) // 3:]                      # (after the clocks have settled)
    avg = {key: round(sum(a[key] for a in acc if key in a) / max(1, sum(1 for a in acc if key in a)), 1) for key in ("power1_average", "power1_input", "power1_cap", "sclk")}
    return k * B / dt, avg
for name, st in (("whole chip, two streams", [torch.cuda.Stream(), torch.cuda.Stream()]), ("whole chip, one stream", [torch.cuda.Stream()]),
                 ("CUs 0-127, one stream", [masked_stream(0, 128)])):
    fps, avg = load(st, 6.0)
    print("tier %s  %-26s %6.0f frames/s  %s" % (tier, name, fps, avg), flush=True)
