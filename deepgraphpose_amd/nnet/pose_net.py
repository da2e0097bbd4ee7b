"""Drop-in for deeplabcut/pose_estimation_tensorflow/nnet/pose_net.py on the MI355X engine (north_star: "replaces
pose_estimation_tensorflow/nnet/pose_net.py").

Same names and argument meaning as the reference (PET/nnet/pose_net.py:14-100, DGP/models/fitdgp_util.py:18-74):
  PoseNet(cfg).extract_features(inputs) -> (net, end_points)        :36-54
  PoseNet(cfg).prediction_layers(features, end_points) -> dict       :56-78
  PoseNet(cfg).get_net(inputs) / .test(inputs)                       :80-90
  prediction_layer(cfg, input, name, num_outputs)                    :18-26
  dgp_prediction_layer(weight_dlc, bias_dlc, dlc_cfg, inputs, name, num_outputs, init_flag, nc, train_flag, ...)  fitdgp_util.py:18
The reference functions build TF graph nodes whose variables a Saver fills later; here the variables are a weights dict
(TF variable names -> arrays, what the snapshot holds) bound with `PoseNet.restore(weights)` / the `weights=` arguments, and
the calls run eagerly: `inputs` are uint8 / float frames [B,H,W,3] (numpy or device tensor), results are device tensors.
Every contraction is a HIP kernel: the backbone through DGPNet (dgp_forward), a head applied to caller-supplied features
through the implicit-GEMM conv kernel as the 2 x 2 convolution over the four output phases of the 3 x 3 / stride-2 SAME
transposed convolution (EXPERIMENTS.md section 3), the phase interleave being a view + copy.
"""
from __future__ import annotations

import re
from typing import Dict, Optional

import numpy as np

net_funcs = {"resnet_50": 50, "resnet_101": 101}          # DGP hard-codes these two (DGP/models/eval.py:272-276)


def _to_device_u8(inputs, device):
    import torch
    if isinstance(inputs, np.ndarray):
        a = inputs
        if a.dtype != np.uint8:
            a = np.clip(np.rint(a), 0, 255).astype(np.uint8)       # the reference feeds img_as_ubyte frames cast to fp32
        return torch.from_numpy(np.ascontiguousarray(a)).to(device)
    t = inputs
    if t.dtype != torch.uint8:
        t = t.round().clamp(0, 255).to(torch.uint8)
    return t.to(device).contiguous()


def _deconv_as_phase_conv(w: np.ndarray) -> np.ndarray:
    """conv2d_transpose weights [3,3,Cout,Cin] (stride 2, SAME) -> HWIO [2,2,Cin,4*Cout] of the equivalent 2 x 2 convolution over
    x[i-1..i, j-1..j] whose output channels are (phase a, phase b, c):  y[2i+a, 2j+b, c] = sum x[i-1+kh', j-1+kw', :] . W'[kh', kw', :, (a,b,c)]
    with W'[kh', kw', ci, (a,b,c)] = w[a+2-2kh', b+2-2kw', c, ci] where that tap exists (index map y[o] = sum_{2i'+k=o} x[i'] w[k])."""
    kh, kw, cout, cin = w.shape
    assert (kh, kw) == (3, 3)
    out = np.zeros((2, 2, cin, 2, 2, cout), dtype=np.float32)
    for khp in range(2):
        for kwp in range(2):
            for a in range(2):
                for b in range(2):
                    th, tw = a + 2 - 2 * khp, b + 2 - 2 * kwp
                    if 0 <= th <= 2 and 0 <= tw <= 2:
                        out[khp, kwp, :, a, b, :] = w[th, tw].T
    return out.reshape(2, 2, cin, 4 * cout)


def _apply_head(features, w: np.ndarray, b: Optional[np.ndarray], stride: int = 2):
    """features [B,h,w,Cin] device fp32 -> [B,2h,2w,Cout] through the HIP conv kernel."""
    import torch
    from .. import engine
    if stride != 2 or tuple(w.shape[:2]) != (3, 3):
        raise NotImplementedError("heads are 3x3 / stride-2 transposed convolutions (pose_net.py:22-25, deconvolutionstride 2)")
    B, h, wd, cin = features.shape
    cout = w.shape[2]
    if w.shape[3] != cin:
        raise ValueError("head weights expect %d input channels, features have %d" % (w.shape[3], cin))
    wp = _deconv_as_phase_conv(np.asarray(w, dtype=np.float32))
    bias4 = None if b is None else np.tile(np.asarray(b, dtype=np.float32).reshape(-1), 4)
    y = engine.conv2d(features.contiguous(), wp, stride=1, rate=1, pad_t=1, pad_l=1, out_hw=(h, wd), bias=bias4)
    y = y.view(B, h, wd, 2, 2, cout).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * h, 2 * wd, cout)
    return y.contiguous()


def prediction_layer(cfg, input, name, num_outputs, weights: Optional[Dict[str, np.ndarray]] = None):
    """pose_net.py:18-26: slim.conv2d_transpose(input, num_outputs, [3,3], stride=cfg.deconvolutionstride, scope='block4')
    under variable scope pose/<name>.  weights: the snapshot's variables ('pose/<name>/block4/{weights,biases}')."""
    if weights is None:
        weights = getattr(cfg, "__dict__", {}).get("_dgp_weights")
    if weights is None:
        raise ValueError("prediction_layer needs the variables: pass weights= or call PoseNet(cfg).restore(weights) first")
    scope = name if name.startswith("pose/") else "pose/" + name
    w, b = weights[scope + "/block4/weights"], weights.get(scope + "/block4/biases")
    if w.shape[2] != num_outputs:
        raise ValueError("%s holds %d outputs, %d requested" % (scope, w.shape[2], num_outputs))
    return _apply_head(input, w, b, int(cfg.get("deconvolutionstride", 2)))


def dgp_prediction_layer(weight_dlc, bias_dlc, dlc_cfg, inputs, name, num_outputs, init_flag, nc, train_flag, stride=2,
                         kernel_size=[3, 3], scope="block4"):
    """fitdgp_util.py:18-74: the same transposed convolution initialised from (weight_dlc [3,3,nj,>=nc], bias_dlc) when init_flag
    is set.  train_flag only marks the variables trainable in the reference; the trainer owns that here (all heads train)."""
    if not init_flag:
        raise ValueError("dgp_prediction_layer without init_flag has no variables to read: pass the snapshot's weights "
                         "(init_flag=True) -- the reference relies on a later Saver.restore")
    if list(kernel_size) != [3, 3]:
        raise NotImplementedError("kernel_size [3, 3] only")
    w = np.asarray(weight_dlc, dtype=np.float32)[:, :, :, :nc]
    if w.shape[2] != num_outputs:
        raise ValueError("weight_dlc holds %d outputs, %d requested" % (w.shape[2], num_outputs))
    return _apply_head(inputs, w, None if bias_dlc is None else np.asarray(bias_dlc, dtype=np.float32).reshape(-1), stride)


class PoseNet:
    def __init__(self, cfg, weights: Optional[Dict[str, np.ndarray]] = None, device: int = 0, max_batch: int = 32):
        self.cfg = cfg
        if "output_stride" not in self.cfg.keys():
            self.cfg.output_stride = 16
        if "deconvolutionstride" not in self.cfg.keys():
            self.cfg.deconvolutionstride = 2
        if self.cfg.net_type not in net_funcs:
            raise KeyError(self.cfg.net_type)
        if int(self.cfg.output_stride) != 16 or int(self.cfg.deconvolutionstride) != 2:
            raise NotImplementedError("output_stride 16 / deconvolutionstride 2 (the only DGP configuration)")
        self.device, self.max_batch = device, max_batch
        self._net = None
        self.weights = None
        if weights is not None:
            self.restore(weights)

    def restore(self, weights: Dict[str, np.ndarray]):
        """What Saver.restore does for the reference graph: bind the snapshot's variables."""
        depth = net_funcs[self.cfg.net_type]
        if ("resnet_v1_%d/conv1/weights" % depth) not in weights:
            raise KeyError("snapshot holds no resnet_v1_%d variables" % depth)
        self.weights = weights
        self.cfg.__dict__["_dgp_weights"] = weights           # (instance attribute, not a config key: prediction_layer(cfg, ...) finds it)
        self._net = None
        return self

    def _engine(self, h, w):
        from .. import engine
        if self.weights is None:
            raise ValueError("PoseNet has no variables yet: call restore(weights)")
        nj = int(self.cfg.num_joints)
        if self._net is None:
            self._net = engine.DGPNet(net_funcs[self.cfg.net_type], nj, h, w, max_batch=self.max_batch,
                                      with_locref=bool(self.cfg.get("location_refinement", False)) and
                                      "pose/locref_pred/block4/weights" in self.weights,
                                      device=self.device, mean_pixel=tuple(self.cfg.get("mean_pixel", (123.68, 116.779, 103.939))))
            self._net.load_weights(self.weights)
        elif (self._net.in_h, self._net.in_w) != (h, w):
            self._net.set_input_size(h, w)
        return self._net

    def extract_features(self, inputs):
        """pose_net.py:36-54: (inputs - mean_pixel) -> resnet_v1_<depth>(global_pool=False, output_stride=16, is_training=False).
        -> (net [B,h,w,2048] device fp32, end_points {'resnet_v1_<d>/block4': net})."""
        import torch
        fr = _to_device_u8(inputs, torch.device("cuda", self.device))
        net = self._engine(fr.shape[1], fr.shape[2])
        feats = []
        for s in range(0, fr.shape[0], self.max_batch):
            _, f = net.forward(fr[s:s + self.max_batch].contiguous(), want_features=True)
            feats.append(f)
        out = feats[0] if len(feats) == 1 else torch.cat(feats, 0)
        num_layers = re.findall("resnet_([0-9]*)", self.cfg.net_type)[0]
        return out, {"resnet_v1_{}/block4".format(num_layers): out}

    def prediction_layers(self, features, end_points, reuse=None):
        cfg = self.cfg
        out = {"part_pred": prediction_layer(cfg, features, "part_pred", cfg.num_joints, self.weights)}
        if cfg.get("location_refinement", False):
            out["locref"] = prediction_layer(cfg, features, "locref_pred", cfg.num_joints * 2, self.weights)
        if cfg.get("intermediate_supervision", False):
            raise NotImplementedError("intermediate_supervision is off in every DGP configuration (pose_cfg.yaml) and is not built")
        return out

    def get_net(self, inputs):
        net, end_points = self.extract_features(inputs)
        return self.prediction_layers(net, end_points)

    def test(self, inputs):
        """pose_net.py:83-90: {'part_prob': sigmoid(part_pred)[, 'locref']}."""
        import torch
        heads = self.get_net(inputs)
        prob = torch.sigmoid(heads["part_pred"])
        if self.cfg.get("location_refinement", False):
            return {"part_prob": prob, "locref": heads["locref"]}
        return {"part_prob": prob}
