"""Joins the per-dispatch PMC values of scripts/traffic_per_layer.sh with the engine's launch table (same order in every step)."""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgraphpose_amd import arch
out, tier = sys.argv[1], sys.argv[2]
eb = 2.0 if tier == "f16" else 4.0
CONV = ("conv_igemm", "chain_kernel", "unit_kernel", "stem_pool_fused", "tail_fixup", "reduce_slabs")
def last_step(sub, counter):
    f = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    ends = [i for i, r in enumerate(rows) if "soft_argmax" in r["Kernel_Name"]]
    seg = rows[ends[-2] + 1:ends[-1] + 1]
    return [(r["Kernel_Name"].replace("void ", "").replace("dgp::", "").split("(")[0], float(r["Counter_Value"]) * 1024.0) for r in seg]
fe, wr = last_step("f", "FETCH_SIZE"), last_step("w", "WRITE_SIZE")
assert [k for k, _ in fe] == [k for k, _ in wr], "dispatch sequences of the two passes differ"
names = [r["name"] for r in csv.DictReader(open(os.path.join(out, "table.tsv")), delimiter="\t")]
us = [float(r["us"]) for r in csv.DictReader(open(os.path.join(out, "table.tsv")), delimiter="\t")]
convs = [i for i, (k, _) in enumerate(fe) if any(c in k for c in CONV)]
tabc = [i for i, n in enumerate(names) if n.startswith("conv:")]
print("%-58s %9s %9s %9s %9s %6s %8s" % ("launch", "fetch MB", "write MB", "total MB", "alg MB", "ratio", "us"))
tf = tw = ta = 0.0
j = 0
for ti in tabc:
    n = names[ti]
    if j >= len(convs): break
    i = convs[j]; j += 1
    f, w = fe[i][1] * 2.0, wr[i][1]
    # tail fix-up / slab reduce launches that follow a conv belong to it
    while j < len(convs) and any(c in fe[convs[j]][0] for c in ("tail_fixup", "reduce_slabs")):
        f += fe[convs[j]][1] * 2.0; w += wr[convs[j]][1]; j += 1
    alg = arch.launch_algorithmic_bytes(n, 480, 640, 50, 32, eb)
    tf += f; tw += w; ta += alg
    short = n.split("|")[0].replace("conv:resnet_v1_50/", "").replace("bottleneck_v1/", "")
    print("%-58s %9.1f %9.1f %9.1f %9.1f %6.2f %8.1f" % (short[:58], f / 1e6, w / 1e6, (f + w) / 1e6, alg / 1e6, (f + w) / alg if alg else float("nan"), us[ti]))
print("%-58s %9.1f %9.1f %9.1f %9.1f %6.2f" % ("sum", tf / 1e6, tw / 1e6, (tf + tw) / 1e6, ta / 1e6, (tf + tw) / max(ta, 1)))
