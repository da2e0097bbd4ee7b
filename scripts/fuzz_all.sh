#!/bin/bash
# Every fuzzer of the repo, one after the other (each compares the HIP path with the oracle / float64 on random configurations and exits non-zero on
# a failure).  Usage (on a GPU box): bash scripts/fuzz_all.sh [seed] [scale]   -- scale multiplies the default case counts (default 1: ~9 minutes, most of it the CPU oracle of fuzz_net)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; SEED=${1:-0}; K=${2:-1}; OUT=gpurun_out/fuzz_all_$SEED; mkdir -p "$OUT"; rc=0
run() { name=$1; shift; timeout 3000 python "scripts/$name.py" "$@" > "$OUT/$name.log" 2>&1; r=$?; [ $r -ne 0 ] && rc=1
        echo "$name: rc $r, $(grep -c '^ok' "$OUT/$name.log") ok lines, $(tail -1 "$OUT/$name.log")"; grep -v '^ok\|^rej \|amdgpu' "$OUT/$name.log" | head -5 | cut -c1-220; }
run fuzz_net $((40 * K)) $SEED                 # whole networks (size, bodyparts, batch, locref, depth) vs the oracle
mv "$OUT/fuzz_net.log" "$OUT/fuzz_net_parity.log"; run fuzz_net $((20 * K)) $SEED --f16       # ... the 16-bit tier inside its band
run fuzz_estimate_pose $((12 * K)) $SEED       # the A0 entry point: batch sizes, chunk rounds, ragged tails
run fuzz_resize $((4 * K)) $SEED                # one engine through sequences of frame sizes, batch sizes and brightness
run fuzz_pipeline $((6 * K)) $SEED              # DGPPipeline submit patterns vs one engine, bit for bit
run fuzz_readout $((150 * K)) $SEED            # soft-argmax, likelihood window, th branch, hard arg-max
run fuzz_conv_h2 $((200 * K)) $SEED            # every loader of the cell kernels
run fuzz_halo $((100 * K)) $SEED               # the halo walk and its fall-backs
run fuzz_fused $((120 * K)) $SEED              # chain / unit kernels
mv "$OUT/fuzz_fused.log" "$OUT/fuzz_fused_h2.log"; run fuzz_fused $((120 * K)) $SEED --h1        # ... their instances on H1 tensors (the 16-bit tier)
run fuzz_conv_f32 $((200 * K)) $SEED           # fp32-activation tiles, max-pool, H2 converters
run fuzz_backward_layers $((100 * K)) $SEED    # weight-gradient tiles, data gradient
run fuzz_train $((14 * K)) $SEED               # whole DGP training steps vs fp64 autograd
mv "$OUT/fuzz_train.log" "$OUT/fuzz_train_single.log"; run fuzz_train $((12 * K)) $SEED --sequence     # ... one trainer through four different steps in a row
run fuzz_dlc                                   # DLC (step-0) training steps
mv "$OUT/fuzz_dlc.log" "$OUT/fuzz_dlc_single.log"; run fuzz_dlc --sequence     # ... one trainer through five frame sizes in a row
run fuzz_motion $((150 * K)) $SEED             # motion-energy scan, whole and chunked
echo "fuzz_all: $([ $rc -eq 0 ] && echo clean || echo FAILURES)"
exit $rc
