"""TEST INFRASTRUCTURE ONLY: the oracle's own restatement of slim's ResNet-v1 unit plan.

Independent of deepgraphpose_amd.arch (the product's table): a wrong stride / rate in the product plan must show up as a
parity failure, so the oracle cannot borrow it.  Follows the structure of TF-1.15 slim itself (third-party, not vendored;
call sites PET/nnet/pose_net.py:14-16,46-52):

  * resnet_v1.py `resnet_v1_block(scope, base_depth, num_units, stride)`: num_units - 1 units of
    {depth: 4 base, depth_bottleneck: base, stride: 1} followed by ONE unit with the block's stride (stride on the LAST unit);
  * resnet_v1.py `resnet_v1_50`: block1 (64, 3, 2), block2 (128, 4, 2), block3 (256, 6, 2), block4 (512, 3, 1);
    `resnet_v1_101`: block3 has 23 units;
  * resnet_utils.py `stack_blocks_dense(net, blocks, output_stride)`: walks the units with `current_stride` (starts at 1;
    resnet_v1 passes output_stride / 4 because the root conv + pool already stride by 4) and `rate` (starts at 1):
        if current_stride == output_stride:  unit runs with stride 1 and the current atrous rate; rate *= unit stride
        else:                                 unit runs with its stride and rate 1;               current_stride *= unit stride
  * resnet_v1.py `bottleneck`: shortcut = subsample (1x1 max-pool, stride s) when depth_in == depth, else a 1x1 conv of stride s
    without activation.
"""
from collections import namedtuple

Unit = namedtuple("Unit", "scope depth_in depth depth_bottleneck stride rate has_shortcut_conv")

_BLOCK_ARGS = {
    50: (("block1", 64, 3, 2), ("block2", 128, 4, 2), ("block3", 256, 6, 2), ("block4", 512, 3, 1)),
    101: (("block1", 64, 3, 2), ("block2", 128, 4, 2), ("block3", 256, 23, 2), ("block4", 512, 3, 1)),
}


def _resnet_v1_block(scope, base_depth, num_units, stride):
    args = [dict(depth=base_depth * 4, depth_bottleneck=base_depth, stride=1)] * (num_units - 1)
    args = args + [dict(depth=base_depth * 4, depth_bottleneck=base_depth, stride=stride)]
    return scope, args


def units(depth=50, output_stride=16):
    if depth not in _BLOCK_ARGS:
        raise ValueError("resnet_v1_%r is not one of DGP's backbones (DGP/models/eval.py:272-276)" % (depth,))
    if output_stride % 4:
        raise ValueError("The output_stride needs to be a multiple of 4.")
    target = output_stride // 4
    name = "resnet_v1_%d" % depth
    current_stride, rate = 1, 1
    depth_in = 64                                   # root: conv1 7x7/2 -> 64 channels, pool 3x3/2
    out = []
    for blk in _BLOCK_ARGS[depth]:
        scope, args = _resnet_v1_block(*blk)
        for i, unit in enumerate(args):
            if current_stride > target:
                raise ValueError("The target output_stride cannot be reached.")
            if current_stride == target:
                stride_used, rate_used = 1, rate
                rate *= unit.get("stride", 1)
            else:
                stride_used, rate_used = unit["stride"], 1
                current_stride *= unit.get("stride", 1)
            out.append(Unit("%s/%s/unit_%d/bottleneck_v1" % (name, scope, i + 1), depth_in, unit["depth"],
                            unit["depth_bottleneck"], stride_used, rate_used, depth_in != unit["depth"]))
            depth_in = unit["depth"]
    if current_stride != target:
        raise ValueError("The target output_stride cannot be reached.")
    return out
