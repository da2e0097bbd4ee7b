#!/usr/bin/env python3
"""README.md's table of numbers, generated from the committed bench lines (profiles/r<N>_bench_line*.json: one JSON line of `python
bench.py` each, one box per file) -- never typed by hand.  `readme_numbers.py` rewrites the block between the two markers in
README.md; `--check` exits 1 when the block is out of date (tests/test_host_new_cpu.py)."""
import glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- numbers:begin (scripts/readme_numbers.py) -->", "<!-- numbers:end -->"


def lines_of_latest_round():
    files = glob.glob(os.path.join(ROOT, "profiles", "r*_bench_line*.json"))
    rounds = sorted({int(re.match(r"r(\d+)_", os.path.basename(f)).group(1)) for f in files})
    if not rounds:
        return 0, []
    r = rounds[-1]
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r%d_bench_line*.json" % r))):
        with open(f) as fh:
            txt = fh.read().strip()
        out.append((os.path.relpath(f, ROOT), json.loads(txt.splitlines()[-1])))
    return r, out


def span(vals, fmt="%.0f"):
    vals = [v for v in vals if v is not None]
    if not vals:
        return "n/a"
    lo, hi = min(vals), max(vals)
    return fmt % lo if fmt % lo == fmt % hi else (fmt % lo) + " – " + (fmt % hi)


def get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def table():
    r, lines = lines_of_latest_round()
    if not lines:
        return "(no profiles/r*_bench_line*.json)\n"
    L = [d for _, d in lines]
    src = ", ".join("`%s`" % f for f, _ in lines)
    rows = [
        ("inference, parity tier (`value`: ResNet-50, 640×480, 4 keypoints, batch 32, two batches in flight)",
         "**%s frames/s** (%s ms per step)" % (span([get(d, "value") for d in L]), span([get(d, "ms_per_step") for d in L], "%.2f"))),
        ("roofline of its dominant kernel (`conv_igemm_split_ls` 128×128, 3 MFMAs per product: peak 833 TFLOP/s)",
         "%s TFLOP/s = frac **%s**; PMC traffic ÷ algorithmic bytes %s" % (
             span([get(d, "roofline", "achieved") for d in L]), span([get(d, "roofline", "frac") for d in L], "%.3f"),
             span([(get(d, "roofline", "traffic") or 0) / get(d, "roofline", "algorithmic_bytes_per_launch")
                   if get(d, "roofline", "traffic") and get(d, "roofline", "algorithmic_bytes_per_launch") else None for d in L], "%.2f"))),
        ("parity vs the CPU oracle inside the timed run", "max %s px, indices bit-exact: %s" % (
            span([get(d, "accuracy_vs_oracle", "px_max") for d in L], "%.1e"),
            "yes" if all(get(d, "accuracy_vs_oracle", "idx_bit_exact") for d in L) else "NO")),
        ("16-bit tier (`tier_f16`: H1 cells, one MFMA per product; a reported tier outside the 1e-3 px gate)",
         "**%s frames/s** two batches in flight, %s on one stream; frac %s of 2500 TFLOP/s; max %s px, RMSE %s px, index agreement %s over %s frames" % (
             span([get(d, "tier_f16", "frames_per_s") for d in L]), span([get(d, "tier_f16", "one_stream", "frames_per_s") for d in L]),
             span([get(d, "tier_f16", "roofline", "frac") for d in L], "%.3f"),
             span([get(d, "tier_f16", "accuracy_vs_oracle", "px_max") for d in L], "%.3f"),
             span([get(d, "tier_f16", "accuracy_vs_oracle", "px_rmse") for d in L], "%.3f"),
             span([get(d, "tier_f16", "accuracy_vs_oracle", "idx_agreement_rate") for d in L], "%.4f"),
             span([get(d, "tier_f16", "accuracy_vs_oracle", "frames") for d in L], "%d"))),
        ("training step (configs[3]: 11 frames 640×480, gm2 = 1, gm3 = 3), parity tier", "%s ms (frac %s of 833 TFLOP/s at 3 × forward FLOPs)" % (
            span([get(d, "train_step", "ms_per_step") for d in L], "%.2f"), span([get(d, "train_step", "frac") for d in L], "%.3f"))),
        ("training step, 16-bit tier (`train_step_f16`)", "**%s ms** (%s frames/s)" % (
            span([get(d, "train_step_f16", "ms_per_step") for d in L], "%.2f"), span([get(d, "train_step_f16", "frames_per_s") for d in L]))),
        ("strict fp32 MFMA (`strict_f32`)", "%s frames/s" % span([get(d, "strict_f32", "frames_per_s") for d in L])),
        ("ResNet-101, 1280×720, 20 keypoints, batch 16 (`r101_1280x720`)", "%s frames/s; 16-bit tier (`r101_1280x720_f16`) %s" % (
            span([get(d, "r101_1280x720", "frames_per_s") for d in L]), span([get(d, "r101_1280x720_f16", "frames_per_s") for d in L]))),
        ("host pipeline (`host_pipeline` / `host_pipeline_f16`: `estimate_pose` on host frames -- staging threads → pinned ring → two copy streams → two engines; "
         "PCIe inclusive, never `value`)",
         "parity tier **%s frames/s**, 16-bit tier **%s frames/s** end to end on %s frames (second call on a snapshot: the session is kept)" % (
             span([get(d, "host_pipeline", "frames_per_s") for d in L]), span([get(d, "host_pipeline_f16", "frames_per_s") for d in L]),
             span([get(d, "host_pipeline", "frames") for d in L], "%d"))),
        ("whole step against the roof (`roofline.frac_whole_step`: stem, heads, soft-argmax, epilogues, launch gaps included)",
         "parity tier %s of 833 TFLOP/s, 16-bit tier %s of 2500" % (span([get(d, "roofline", "frac_whole_step") for d in L], "%.3f"),
                                                                 span([get(d, "tier_f16", "roofline", "frac_whole_step") for d in L], "%.3f"))),
        ("CPU baseline (`cpu_baseline`: the oracle on the box's host cores, bounded sample)", "%s frames/s on %s cores (`kind: %s`)" % (
            span([get(d, "cpu_baseline", "value") for d in L], "%.1f"), span([get(d, "cpu_baseline", "cores") for d in L], "%d"),
            get(L[0], "cpu_baseline", "kind"))),
    ]
    out = ["Round %d, one MI355X, from %s (a range = box to box):\n" % (r, src), "| | |", "|---|---|"]
    out += ["| %s | %s |" % (a, b) for a, b in rows]
    return "\n".join(out) + "\n"


def main():
    path = os.path.join(ROOT, "README.md")
    with open(path) as f:
        txt = f.read()
    if BEGIN not in txt or END not in txt:
        sys.exit("README.md has no numbers block (%s ... %s)" % (BEGIN, END))
    head, rest = txt.split(BEGIN, 1)
    _, tail = rest.split(END, 1)
    new = head + BEGIN + "\n" + table() + END + tail
    if "--check" in sys.argv:
        if new != txt:
            sys.exit("README.md's numbers block is out of date: run python scripts/readme_numbers.py")
        return
    with open(path, "w") as f:
        f.write(new)


if __name__ == "__main__":
    main()
