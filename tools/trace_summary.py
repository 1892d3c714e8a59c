"""rocprofv3 --kernel-trace CSV -> per (kernel, grid) rows: `--stats` lumps every launch of a kernel together, but the
kernel table launches one kernel at several sizes -- and it averages per-dispatch durations, which would say nothing about
bandwidth if launches of several streams overlapped (bench.py's headline keeps 4 launches in flight on 4 HIP streams).  They do
NOT overlap in a trace: with --kernel-trace attached the dispatches of all streams run one at a time (in_flight 1.00-1.01 over
bench.py's 4-stream region), so the profiler's average is the one-launch-in-flight duration -- bench.py's roofline.avg_launch_us --
and the 4-stream rate can only be seen by HIP events and the wall clock.  Besides the trimmed average duration each row carries
    sum_us       the sum of the dispatches' durations
    union_us     the time during which AT LEAST ONE dispatch of the row was running (union of the [start, end] intervals)
    in_flight    sum_us / union_us: dispatches in flight on average while the kernel runs
so that  calls x bytes_per_launch / union_us  is the bandwidth the chip achieved while that kernel ran, from the profiler's
timestamps alone (= in_flight x bytes_per_launch / the mean duration).
usage: trace_summary.py <..._kernel_trace.csv> > summary.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
    key = (name, int(r.get("Grid_Size_X", r.get("Grid_Size", 0))), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0))))
    acc.setdefault(key, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid_threads", "workgroup", "calls", "avg_us", "min_us", "max_us", "sum_us", "union_us", "in_flight"])
for (name, grid, wg), iv in acc.items():
    v = [(e - s) * 1e-3 for s, e in iv]
    v2 = sorted(v)[len(v) // 10: len(v) - len(v) // 10] if len(v) >= 20 else v       # trimmed: first launches ramp the clocks
    union, cur_s, cur_e = 0, None, None
    for s, e in sorted(iv):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        union += cur_e - cur_s
    total = sum(v)
    w.writerow([name, grid, wg, len(v), round(sum(v2) / len(v2), 3), round(min(v), 3), round(max(v), 3), round(total, 1),
                round(union * 1e-3, 1), round(total / (union * 1e-3), 3) if union else ""])
