"""CPU tests of the small tools whose output the documents quote."""
import csv
import io
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_trace_summary_unions_overlapping_dispatches(tmp_path):
    """tools/trace_summary.py: per (kernel, grid) rows with the sum of the durations, the UNION of the [start, end] intervals
    and their ratio (`in_flight`) -- two kernels overlapping by half are 1.33 in flight, dispatches back to back exactly 1"""
    p = tmp_path / "x_kernel_trace.csv"
    rows = [("k<1>(int*)", 256, 64, 1000, 3000), ("k<1>(int*)", 256, 64, 2000, 4000),          # overlap: union 3000, sum 4000
            ("k<1>(int*)", 512, 64, 10000, 11000), ("k<1>(int*)", 512, 64, 11000, 12000),      # back to back: union = sum
            ("void (anonymous namespace)::other(float)", 64, 64, 0, 500)]
    with open(p, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel_Name", "Grid_Size_X", "Workgroup_Size_X", "Start_Timestamp", "End_Timestamp"])
        w.writerows(rows)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_summary.py"), str(p)], capture_output=True, text=True, check=True).stdout
    got = {(r["kernel"], int(r["grid_threads"])): r for r in csv.DictReader(io.StringIO(out))}
    a, b, c = got[("k<1>", 256)], got[("k<1>", 512)], got[("other", 64)]
    assert (float(a["sum_us"]), float(a["union_us"]), float(a["in_flight"])) == (4.0, 3.0, 1.333)
    assert (float(b["sum_us"]), float(b["union_us"]), float(b["in_flight"])) == (2.0, 2.0, 1.0)
    assert int(a["calls"]) == 2 and float(a["avg_us"]) == 2.0 and float(c["avg_us"]) == 0.5
