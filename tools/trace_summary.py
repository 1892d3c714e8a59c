"""rocprofv3 --kernel-trace CSV -> per (kernel, grid) averages: `--stats` lumps every launch of a kernel together, but the
kernel table launches one kernel at several sizes.  usage: trace_summary.py <..._kernel_trace.csv> > summary.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
    key = (name, int(r.get("Grid_Size_X", r.get("Grid_Size", 0))), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0))))
    acc.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid_threads", "workgroup", "calls", "avg_us", "min_us", "max_us"])
for (name, grid, wg), v in acc.items():
    v2 = sorted(v)[len(v) // 10: len(v) - len(v) // 10] if len(v) >= 20 else v       # trimmed: first launches ramp the clocks
    w.writerow([name, grid, wg, len(v), round(sum(v2) / len(v2), 3), round(min(v), 3), round(max(v), 3)])
