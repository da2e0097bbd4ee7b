"""Fused-kernel rows (unit_* / chain_*) of the layer tables written by scripts/ab_loaders.sh.  Usage: python scripts/fused_rows.py <dir> <n> ..."""
import csv, sys
d = sys.argv[1]
for n in sys.argv[2:]:
    rows = list(csv.reader(open("%s/lt_%s.tsv" % (d, n)), delimiter="\t"))[1:]
    fused = [(r[1].split("|")[-1], float(r[3])) for r in rows if r[1].split("|")[-1].startswith(("unit_", "chain_"))]
    print(n, "  ".join("%s %.4f" % (k.replace("chain_", "ch_").replace("unit_", "u_"), t) for k, t in fused), " sum %.4f" % sum(t for _, t in fused))
