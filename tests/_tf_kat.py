"""Known answers TensorFlow's own unit tests assert for conv2d (tensorflow/python/kernel_tests/conv_ops_test.py, class Conv2DTest), restated
as data: the test's name, input shape (NHWC), filter shape (HWIO), stride, padding and the expected output (row-major NHWC).  As in
`_SetupValuesForDevice`, both tensors are filled with 1, 2, 3, ... in row-major order.  Used by tests/test_oracle_cpu.py (the oracle) and
tests/test_parity_gpu.py (the product's conv kernel through the C-ABI)."""
import numpy as np

TF_CONV2D_KNOWN_ANSWERS = [
    ("testConv2D1x1Filter", (1, 2, 3, 3), (1, 1, 3, 3), 1, "VALID",
     [30.0, 36.0, 42.0, 66.0, 81.0, 96.0, 102.0, 126.0, 150.0, 138.0, 171.0, 204.0, 174.0, 216.0, 258.0, 210.0, 261.0, 312.0]),
    ("testConv2D2x2Filter", (1, 2, 3, 3), (2, 2, 3, 3), 1, "VALID", [2271.0, 2367.0, 2463.0, 2901.0, 3033.0, 3165.0]),
    ("testConv2D2x2FilterStride2", (1, 2, 3, 3), (2, 2, 3, 3), 2, "VALID", [2271.0, 2367.0, 2463.0]),
    ("testConv2D2x2FilterStride2Same", (1, 2, 3, 3), (2, 2, 3, 3), 2, "SAME", [2271.0, 2367.0, 2463.0, 1230.0, 1305.0, 1380.0]),
    ("testConv2D1x2Filter", (1, 2, 3, 3), (1, 2, 3, 3), 1, "VALID",
     [231.0, 252.0, 273.0, 384.0, 423.0, 462.0, 690.0, 765.0, 840.0, 843.0, 936.0, 1029.0]),
    ("testConv2DKernelSmallerThanStrideValid (3x3)", (1, 3, 3, 1), (1, 1, 1, 1), 2, "VALID", [1, 3, 7, 9]),
    ("testConv2DKernelSmallerThanStrideValid (7x7)", (1, 7, 7, 1), (2, 2, 1, 1), 3, "VALID", [65, 95, 275, 305]),
    ("testConv2DKernelSmallerThanStrideSame (3x3)", (1, 3, 3, 1), (1, 1, 1, 1), 2, "SAME", [1, 3, 7, 9]),
    ("testConv2DKernelSmallerThanStrideSame (4x4)", (1, 4, 4, 1), (1, 1, 1, 1), 2, "SAME", [1, 3, 9, 11]),
    ("testConv2DKernelSizeMatchesInputSize", (1, 2, 2, 1), (2, 2, 1, 2), 1, "VALID", [50.0, 60.0]),
]


def tf_test_values(shape):
    """_SetupValuesForDevice: 1, 2, 3, ... in row-major order"""
    return np.arange(1, int(np.prod(shape)) + 1, dtype=np.float32).reshape(shape)


def tf_out_and_pads(size, k, stride, padding):
    """TF's output size and (before, after) padding of one spatial axis (common_shape_fns / the SAME rule: the extra pixel goes AFTER)"""
    if padding == "VALID":
        return (size - k) // stride + 1, 0, 0
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return out, total // 2, total - total // 2


def brute_force(in_shape, f_shape, stride, padding):
    """the definition, in float64 loops: what the published vectors are re-derived from before anything is held to them"""
    x, w = tf_test_values(in_shape).astype(np.float64), tf_test_values(f_shape).astype(np.float64)
    n, h, wd, _ = in_shape
    kh, kw, _, co = f_shape
    oh, pt, _ = tf_out_and_pads(h, kh, stride, padding)
    ow, pl, _ = tf_out_and_pads(wd, kw, stride, padding)
    y = np.zeros((n, oh, ow, co))
    for i in range(oh):
        for j in range(ow):
            for a in range(kh):
                for b in range(kw):
                    r, c = i * stride + a - pt, j * stride + b - pl
                    if 0 <= r < h and 0 <= c < wd:
                        y[:, i, j] += x[:, r, c] @ w[a, b]
    return y
