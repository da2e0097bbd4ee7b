"""deepgraphpose_amd -- MI355X-native hot path for Deep Graph Pose.

ResNet-50/101 (output stride 16) backbone, part-detection / locref transposed-conv
heads, DGP 2-D soft-argmax + likelihood, as hand-written gfx950 HIP kernels behind
the C-ABI declared in include/dgp_hip.h.  See DESIGN.md.
"""
__version__ = "0.1.0"
