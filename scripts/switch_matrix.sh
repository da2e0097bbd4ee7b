#!/bin/bash
# Every shipped native switch (README "Run-time switches") at its non-default values, one at a time and in pairs, against the whole-network parity tests of
# both tiers and the trainer's gradient tests: the product must stay inside its gates under ANY of them.  Usage (GPU box): bash scripts/switch_matrix.sh [pairs]
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/switch_matrix; mkdir -p $OUT; rc=0
NET="tests/test_parity_gpu.py -k network_small_matches_oracle or infer_640x480_r50_matches_oracle or demo_frame_size"
H1="tests/test_h1_gpu.py"
TRAIN="tests/test_train_gpu.py -k gradients or tier"
run() {   # name, env assignments...
  name=$1; shift
  for suite in NET H1 TRAIN; do
    case $suite in NET) args=(tests/test_parity_gpu.py -k "network_small_matches_oracle or infer_640x480_r50_matches_oracle or demo_frame_size");;
                   H1) args=(tests/test_h1_gpu.py);; TRAIN) args=(tests/test_train_gpu.py -k "gradients or tier");; esac
    if [ "$name" = "DGP_H2=0" ] && [ $suite = H1 ]; then      # (the 16-bit tier lives on the cell engine: dgp_forward refuses it under DGP_H2=0 with DGP_ERR_STATE, by design)
      line=$(env "$@" timeout 600 python -m pytest tests/test_h1_gpu.py -x -q -m gpu -k network_stays_within 2>&1 | grep -c "the 16-bit tier needs the H2 engine")
      echo "$name | $suite | refused loudly ($line message lines), as documented"; continue
    fi
    line=$(env "$@" timeout 1200 python -m pytest "${args[@]}" -x -q -m gpu 2>&1 | tail -1)
    echo "$name | $suite | $line"
    case "$line" in *failed*|*error*) rc=1;; esac
  done
}
run "defaults" DGP_NOP=1
for s in DGP_H2=0 DGP_CHAIN=0 DGP_CHAIN_H1=0 DGP_HALO=0 DGP_HALO=2 DGP_FUSE_SHORTCUT=0 DGP_TAIL_SPLIT=0 DGP_TAIL_SPLIT=2 DGP_W64=2 DGP_TRAIN_HEADS_H1=0 \
         DGP_SOFTARGMAX_STREAM=1 DGP_LOSS_STREAM=1 DGP_CONV2D_CELLS=1; do run "$s" $s; done
if [ "${1:-}" = pairs ]; then
  run "CHAIN=0 HALO=0" DGP_CHAIN=0 DGP_HALO=0
  run "CHAIN=0 FUSE_SHORTCUT=0" DGP_CHAIN=0 DGP_FUSE_SHORTCUT=0
  run "W64=2 HALO=0" DGP_W64=2 DGP_HALO=0
  run "W64=2 CHAIN=0 TAIL_SPLIT=2" DGP_W64=2 DGP_CHAIN=0 DGP_TAIL_SPLIT=2
  run "CHAIN_H1=0 TRAIN_HEADS_H1=0" DGP_CHAIN_H1=0 DGP_TRAIN_HEADS_H1=0
  run "HALO=2 TAIL_SPLIT=2 SOFTARGMAX_STREAM=1" DGP_HALO=2 DGP_TAIL_SPLIT=2 DGP_SOFTARGMAX_STREAM=1
fi
echo "switch matrix: $([ $rc -eq 0 ] && echo clean || echo FAILURES)"
exit $rc
