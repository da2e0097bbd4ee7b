#!/usr/bin/env python3
"""Where the gradient exchange of a data-parallel step starts relative to the backward pass (scripts/timeline_dp.sh): from rank 0's
rocprofv3 kernel + memory-copy trace, per step (delimited by the optimiser's kernel): when gradient group 0 is final (end of the first
bn_param_grads_all launch of the step), when the backward pass ends (end of the last one), and when the first copy of the exchange starts
(gloo stages a group through the host: device-to-host copies / copyBuffer kernels; with RCCL it would be the first all-reduce kernel)."""
import csv, glob, sys
d = sys.argv[1]
kern = [r for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
cop = [r for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
K = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in kern)
mom = [s for s, e, n in K if "momentum_kernel" in n]
fin = [(s, e) for s, e, n in K if "bn_param_grads_all" in n]
xfer = sorted([int(r["Start_Timestamp"]) for r in cop if "DEVICE_TO_HOST" in r["Direction"]] + [s for s, e, n in K if "copyBuffer" in n])
dgrad = [(s, e) for s, e, n in K if "conv_igemm_split_ls" in n or "wgrad" in n]
print("step | group 0 final | backward ends | first copy of the exchange | exchange starts before the end of the backward pass by | conv / wgrad kernels still to run then")
for i in range(max(1, len(mom) - 6), len(mom)):
    lo, hi = mom[i - 1], mom[i]
    f = [x for x in fin if lo < x[0] < hi]
    if len(f) < 2:
        continue
    g0, end = f[0][1], f[-1][1]
    xs = [t for t in xfer if g0 <= t < hi]
    if not xs:
        continue
    left = sum(1 for s, e in dgrad if xs[0] < s < end)
    print("%4d | %10.3f ms | %10.3f ms | %10.3f ms | %+8.3f ms | %d" % (i, (g0 - lo) / 1e6, (end - lo) / 1e6, (xs[0] - lo) / 1e6, (end - xs[0]) / 1e6, left))
