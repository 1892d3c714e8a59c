"""tools/pmc_summary.py attributes hardware counters to LAUNCHES by launch order (the launcher's manifest), never by kernel name
or grid: in round 5 two instantiations of one kernel template ran with the same grid and the byte counts of the N = 256 and the
N = 1024 aggregation were attached the wrong way round (4.13x and 0.275x in a committed table).  CPU only: canned CSV rows in
rocprofv3's counter_collection format."""
import csv
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

FIELDS = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
L, D = 83, 256
B1024 = 1024 * (L + 1) * 4 * D + L * 4 * D
B256 = 256 * (L + 1) * 4 * D + L * 4 * D
MANIFEST = [{"name": "aggregate1024", "kernel": "aggregate_", "signers": 1024, "bytes": B1024, "launches": 6},
            {"name": "aggregate256", "kernel": "aggregate_", "signers": 256, "bytes": B256, "launches": 6}]


def write_csv(path, launches):
    """launches: [(kernel name, grid, bytes read, bytes written)] in dispatch order; FETCH_SIZE is reported in KB and at HALF
    the bytes on gfx950 (the summary doubles it), one row per XCD"""
    with open(path, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=FIELDS)
        w.writeheader()
        for did, (name, grid, rd, wr) in enumerate(launches, 1):
            for xcd in range(8):
                for counter, val in (("FETCH_SIZE", rd / 2 / 1024 / 8), ("WRITE_SIZE", wr / 1024 / 8)):
                    w.writerow({"Dispatch_Id": did, "Kernel_Name": f"void (anonymous namespace)::{name}(int const*, int*)", "Grid_Size": grid,
                                "Counter_Name": counter, "Counter_Value": val, "Start_Timestamp": 1000 * did, "End_Timestamp": 1000 * did + 500})


def test_two_instantiations_with_one_grid_get_their_own_launches(tmp_path):
    import pmc_summary
    # the N = 1024 launches come first and run the <..., 3> instantiation; N = 256 runs <..., 4>; both with grid 131072
    rows = [("aggregate_onepass<8, FzNoRag, false, 3>", 131072, B1024 * 1.03, 83 * 1024)] * 6 + \
           [("aggregate_onepass<8, FzNoRag, false, 4>", 131072, B256 * 1.08, 83 * 1024)] * 6
    p = tmp_path / "x_counter_collection.csv"
    write_csv(p, rows)
    t = pmc_summary.scheme_table([str(p)], MANIFEST)
    a3 = t["aggregate_onepass<8, FzNoRag, false, 3> grid=131072"]
    a4 = t["aggregate_onepass<8, FzNoRag, false, 4> grid=131072"]
    assert a3["signers"] == 1024 and a3["launch"] == "aggregate1024" and a3["algorithmic_bytes_per_launch"] == B1024
    assert a4["signers"] == 256 and a4["launch"] == "aggregate256" and a4["algorithmic_bytes_per_launch"] == B256
    assert 1.02 < a3["traffic_over_algorithmic"] < 1.05 and 1.07 < a4["traffic_over_algorithmic"] < 1.10
    assert pmc_summary.unexplained(t) == {}
    # the same kernels in the OTHER launch order: the attribution follows the launches, not the names
    write_csv(p, rows[6:] + rows[:6])
    t = pmc_summary.scheme_table([str(p)], MANIFEST)
    assert t["aggregate_onepass<8, FzNoRag, false, 4> grid=131072"]["signers"] == 1024       # (and its ratio is now far off:)
    assert set(pmc_summary.unexplained(t)) == {"aggregate_onepass<8, FzNoRag, false, 4> grid=131072", "aggregate_onepass<8, FzNoRag, false, 3> grid=131072"}


def test_one_kernel_at_two_grids_is_two_entries_and_a_surplus_kernel_is_an_error(tmp_path):
    import pmc_summary
    rows = [("aggregate_direct<8, FzNoRag>", 65536, B1024, 83 * 1024)] * 5 + [("aggregate_direct<8, FzNoRag>", 16384, B256, 83 * 1024)] * 5
    p = tmp_path / "y_counter_collection.csv"
    write_csv(p, rows)
    t = pmc_summary.scheme_table([str(p)], MANIFEST)
    assert t["aggregate_direct<8, FzNoRag> grid=65536"]["signers"] == 1024 and t["aggregate_direct<8, FzNoRag> grid=16384"]["signers"] == 256
    write_csv(p, rows + [("aggregate_onepass<8, FzNoRag, true, 3>", 118784, 1, 1)] * 3)
    with pytest.raises(SystemExit) as e:
        pmc_summary.scheme_table([str(p)], MANIFEST)
    assert "manifest lists 2 launches" in str(e.value)


def test_a_ratio_outside_the_band_needs_an_explanation(tmp_path, monkeypatch):
    import pmc_summary
    rows = [("aggregate_onepass<8, FzNoRag, false, 3>", 131072, B1024 * 1.7, 83 * 1024)] * 6
    p = tmp_path / "z_counter_collection.csv"
    write_csv(p, rows)
    t = pmc_summary.scheme_table([str(p)], MANIFEST[:1])
    assert list(pmc_summary.unexplained(t).values()) == [pytest.approx(1.7, abs=0.01)]
    monkeypatch.setitem(pmc_summary.EXPLAINED, "aggregate1024", "a reason")
    t = pmc_summary.scheme_table([str(p)], MANIFEST[:1])
    assert pmc_summary.unexplained(t) == {} and next(iter(t.values()))["explained"] == "a reason"


def test_show_roofline_reads_a_committed_bench_line():
    """tools/show_roofline.py raised KeyError on every line from round 4 on: it is run here on the newest committed line"""
    import glob
    import subprocess
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default_line.json")))
    assert lines
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "show_roofline.py"), lines[-1]], capture_output=True, text=True)
    assert r.returncode == 0 and "roofline:" in r.stdout and "value" in r.stdout, r.stderr
