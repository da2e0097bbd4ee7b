"""Conv launches of a rocprofv3 kernel trace grouped by (kernel instance, workgroups): launches per step, median duration, time per step."""
import csv, glob, re, collections, sys
root, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 7.0
f = sorted(glob.glob(root + '/**/*kernel_trace.csv', recursive=True))[-1]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    m = re.search(r'split_ls<([^>]*)>', n)
    if not m: continue
    key = (m.group(1).replace(' ', '').replace('true', 'T').replace('false', 'F'), int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))
    agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print('%-44s wgs %5d  n/step %5.1f  median %7.1f us  per step %.3f ms' % (k[0], k[1], len(v) / steps, v2[len(v2) // 2], sum(v) / steps / 1e3))
    tot += sum(v) / steps / 1e3
print('sum %.3f ms per step' % tot)
