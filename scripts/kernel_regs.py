"""Register / spill / occupancy table of one HIP source's kernels (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python scripts/kernel_regs.py deepgraphpose_amd/csrc/dgp_kernels.hip [name-regex] [extra hipcc flags ...]"""
import re, subprocess, sys
src = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "."; extra = sys.argv[3:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + extra
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for l in err.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", l)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
names = subprocess.run(["/usr/bin/c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = n.replace("dgp::", "").split("(")[0]
    if re.search(pat, n):
        print("%-110s vgpr %4s agpr %3s spill %3s occ %s scratch %s" % (n[:110], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("Occupancy [waves/SIMD]"), r.get("ScratchSize [bytes/lane]")))
