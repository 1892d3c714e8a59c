"""Summarise the rocprofv3 --pmc passes written by tools/collect_profiles.sh into <tag>_pmc_ntt.json.
usage: pmc_summary.py gpurun_out/<tag> <tag>
Per kernel and launch: HBM-side traffic = FETCH_SIZE (KB, doubled: MI355X_MICROARCH.md's gfx950 correction for
16 B/lane coalesced reads) + WRITE_SIZE (KB); busy cycles per XCD (GRBM_GUI_ACTIVE / 8); an fp64-VALU issue
estimate = SQ_INSTS_VALU x 4 cycles / (cycles per XCD x 128 SIMDs per XCD).  Counters are summed over the
dimension rows rocprofv3 emits per dispatch and averaged over the dispatches of the kernel (the first two
dispatches of each kernel are dropped as warm-up when there are more than four)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

# scheme kernels (tools/probes/prof_scheme.py, secpar 256: l = 83, d = 256): algorithmic bytes per launch from SURVEY.md 8d
L_, D_ = 83, 256


# ---- scheme kernels ----------------------------------------------------------------------------------------------
# Which launch a kernel instantiation served is NOT read off its name or grid (two instantiations of aggregate_onepass ran with
# the same grid in round 5 and the N = 256 / N = 1024 byte counts were attached the wrong way round: 4.13x and 0.275x).  The
# launcher writes what it launched, in order (tools/probes/prof_scheme.py -> pmcs_manifest.json); kernels are matched to manifest
# entries by the order of their FIRST dispatch.  Without a manifest the same order is assumed (prof_scheme.py's own).
DEFAULT_MANIFEST = [
    {"name": "keygen", "kernel": "keygen_fused", "bytes": 1024 * (4 * L_ + 2) * 4 * D_},
    {"name": "sign", "kernel": "sign_kernel", "bytes": 1024 * (3 * L_ + 1) * 4 * D_},
    {"name": "aggregate1024", "kernel": "aggregate_", "signers": 1024, "bytes": 1024 * (L_ + 1) * 4 * D_ + L_ * 4 * D_},
    {"name": "aggregate256", "kernel": "aggregate_", "signers": 256, "bytes": 256 * (L_ + 1) * 4 * D_ + L_ * 4 * D_},
    {"name": "sign+aggregate 4x256", "kernel": "aggregate_", "signers": 1024, "bytes": 1024 * (3 * L_ + 4) * 4 * D_,
     "what": "sign + aggregate + target sums in one pass, 4 x 256 signers: (3l + 4) rows moved per signature"},
    {"name": "matvec", "kernel": "matvec_", "bytes": 2048 * (L_ + 1) * 4 * D_},
    {"name": "pw_mul", "kernel": "pw_kernel", "bytes": 1024 * L_ * D_ * 12},
    {"name": "verify64", "kernel": "verify_fused", "bytes": 64 * (L_ + 2) * 4 * D_ + L_ * 4 * D_},
    {"name": "verify1024", "kernel": "verify_fused", "bytes": 1024 * (L_ + 2) * 4 * D_ + L_ * 4 * D_},
    {"name": "verify8192", "kernel": "verify_fused", "bytes": 8192 * (L_ + 2) * 4 * D_ + L_ * 4 * D_},
    {"name": "polymul 2^13", "kernel": "polymul_fused", "products": 1 << 13, "bytes": (1 << 13) * 12 * D_},
    {"name": "polymul 2^17", "kernel": "polymul16", "products": 1 << 17, "bytes": (1 << 17) * 12 * D_},
]
# ratios outside [0.9, 1.5] must be explained here (prefix of the manifest name -> why), or the script fails
EXPLAINED = {
}


def scheme_table(csv_paths, manifest=None):
    """{kernel: counters + traffic / algorithmic bytes} from rocprofv3 counter_collection CSVs of ONE prof_scheme.py run per counter
    set (the dispatch ids of every pass follow the same launch order)"""
    manifest = manifest or DEFAULT_MANIFEST
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    dur, first = defaultdict(dict), {}
    for path in csv_paths:
        with open(path) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
                if r.get("Grid_Size"):
                    k += f" grid={r['Grid_Size']}"              # one kernel serves launches of several sizes: one entry per size
                did = int(r["Dispatch_Id"])
                per[k][r["Counter_Name"]][did] += float(r["Counter_Value"])
                dur[k][did] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
                first[k] = min(first.get(k, did), did)
    # manifest entries in launch order <-> kernels in first-dispatch order, within each kernel family
    assigned = {}
    fams = []
    for m in manifest:
        if m["kernel"] not in fams:
            fams.append(m["kernel"])
    for fam in fams:
        entries = [m for m in manifest if m["kernel"] == fam]
        kernels = sorted((k for k in per if k.startswith(fam)), key=lambda k: first[k])
        if len(kernels) > len(entries):
            raise SystemExit(f"pmc_summary: {len(kernels)} kernels start with {fam!r} ({kernels}) but the manifest lists {len(entries)} launches of it")
        for k, m in zip(kernels, entries):
            assigned[k] = m
    table = {}
    for k, counters in per.items():
        if k.startswith(("fill_synthetic", "ntt_")):
            continue
        e = {}
        for c, by_dispatch in counters.items():
            ids = sorted(by_dispatch)
            ids = ids[2:] if len(ids) > 4 else ids
            e[c] = sum(by_dispatch[i] for i in ids) / len(ids)
            e.setdefault("dispatches_averaged", len(ids))
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["read_bytes_corrected"] = e["FETCH_SIZE"] * 1024 * 2
            e["write_bytes"] = e["WRITE_SIZE"] * 1024
            e["traffic_bytes_per_launch"] = e["read_bytes_corrected"] + e["write_bytes"]
            m = assigned.get(k)
            if m:
                e["launch"] = m["name"]
                for key in ("signers", "products", "what"):
                    if key in m:
                        e[key] = m[key]
                e["algorithmic_bytes_per_launch"] = m["bytes"]
                e["traffic_over_algorithmic"] = e["traffic_bytes_per_launch"] / m["bytes"]
                note = next((v for n, v in EXPLAINED.items() if m["name"].startswith(n)), None)
                if note:
                    e["explained"] = note
        if "GRBM_GUI_ACTIVE" in e:
            e["cycles_per_launch_per_xcd"] = e["GRBM_GUI_ACTIVE"] / 8
            if "SQ_INSTS_VALU" in e:
                e["valu_issue_utilisation_est"] = e["SQ_INSTS_VALU"] * 4 / (e["cycles_per_launch_per_xcd"] * 128 * 8)
        ds = sorted(dur[k])
        ds = ds[2:] if len(ds) > 4 else ds
        e["serialised_duration_us_under_pmc"] = sum(dur[k][i] for i in ds) / len(ds)
        table[k] = e
    return table


def unexplained(table):
    """kernels whose measured traffic is not within [0.9, 1.5] of the algorithmic bytes and carry no explanation"""
    return {k: round(e["traffic_over_algorithmic"], 3) for k, e in table.items()
            if "traffic_over_algorithmic" in e and not 0.9 <= e["traffic_over_algorithmic"] <= 1.5 and "explained" not in e}


def main():
    root, tag = sys.argv[1], sys.argv[2]
    GROUPS = {"pmcb": ("B=4096 (bench launch)", 4096), "pmc20": ("B=2^20", 1 << 20)}
    # what bench.py itself says its dominant launch is (the PMC passes run bench.py --headline-only --no-graph): kernel name and
    # transforms per launch -- the 16-per-lane multi-job kernel's grid is the resident grid whatever the launch holds, so its rows
    # cannot be read off the grid as the radix-4 kernels' can
    BENCH = {}
    try:
        with open(os.path.join(root, "pmcb_bench_full.json")) as fh:
            _r = json.load(fh)["roofline"]
        BENCH = {"kernel": _r["kernel"].split("(")[0], "rows": int(_r["units_per_launch"])}
    except Exception:
        pass
    out = {"source": "rocprofv3 --pmc <one set per pass> --output-format csv (tools/collect_profiles.sh, tools/pmc_summary.py); "
                     "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane coalesced reads on gfx950; KB -> bytes x1024",
           "kernels": {}}
    for sub, (label, rows) in GROUPS.items():
        per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))       # kernel -> counter -> dispatch -> value
        dur = defaultdict(dict)
        grid = {}
        for path in glob.glob(os.path.join(root, sub, "*", "*", "*_counter_collection.csv")):
            with open(path) as fh:
                for r in csv.DictReader(fh):
                    k = r["Kernel_Name"]
                    if "ntt_" not in k:
                        continue
                    short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
                    grid[short] = int(r.get("Grid_Size", 0) or 0)
                    per[short][r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
                    dur[short][int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        for k, counters in per.items():
            e = {}
            is_bench = sub == "pmcb" and BENCH and k.replace(" ", "") == BENCH["kernel"].replace(" ", "")
            for c, by_dispatch in counters.items():
                ids = sorted(by_dispatch)
                if len(ids) > 4:
                    ids = ids[2:]
                if is_bench and c in ("FETCH_SIZE", "WRITE_SIZE") and len(ids) > 4:
                    # FULL launches only: a run of pipelined steps opens with a forward-only launch and closes with an inverse-only
                    # one (half the jobs, half the bytes, the same kernel) -- dropped by their own counter value
                    med = sorted(by_dispatch[i] for i in ids)[len(ids) // 2]
                    ids = [i for i in ids if by_dispatch[i] >= 0.75 * med]
                e[c] = sum(by_dispatch[i] for i in ids) / len(ids)
                e.setdefault("dispatches_averaged", len(ids))
            if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
                e["read_bytes_corrected"] = e["FETCH_SIZE"] * 1024 * 2
                e["write_bytes"] = e["WRITE_SIZE"] * 1024
                e["traffic_bytes_per_launch"] = e["read_bytes_corrected"] + e["write_bytes"]
                # rows of the launch: the radix-4 kernels (ntt_fwd4 / ntt_inv4 / ntt_jobs4 <LOGD, FAST, NR, WAVES>) run one wave-task of
                # NR rows (degree 256) per wave, so rows = grid threads / 64 x NR -- the two-job launch of bench.py's pipelined step
                # (4096 forward + 4096 inverse rows) is ntt_jobs4<8, true, 2, 2> with 8192 rows; other kernels: the group's batch
                m4 = re.search(r"ntt_(?:fwd4|inv4|jobs4)<(\d+), (?:true|false), (\d+), (\d+)[,>]", k)
                rows_k = (grid.get(k, 0) // 64) * int(m4.group(2)) if m4 and grid.get(k) else rows
                if is_bench:
                    rows_k = BENCH["rows"]
                elif sub == "pmcb" and k.startswith("ntt_jobs16"):
                    # a pipelined run's first (forward jobs only) and last (inverse jobs only) launch: half the jobs through the same
                    # resident grid -- another table size, another instantiation; read off its own written bytes (4 B per coefficient)
                    rows_k = int(round(e["write_bytes"] / 1024 / 4096)) * 4096          # jobs of 4096 rows
                    e["what"] = "boundary launch of a pipelined run (half the jobs); rows from the bytes written"
                e["rows_per_launch"] = rows_k
                e["algorithmic_bytes_per_launch"] = rows_k * 2048
                e["traffic_over_algorithmic"] = e["traffic_bytes_per_launch"] / e["algorithmic_bytes_per_launch"]
            if "GRBM_GUI_ACTIVE" in e:
                e["cycles_per_launch_per_xcd"] = e["GRBM_GUI_ACTIVE"] / 8
                if "SQ_INSTS_VALU" in e:
                    e["valu_issue_utilisation_est"] = e["SQ_INSTS_VALU"] * 4 / (e["cycles_per_launch_per_xcd"] * 128 * 8)
            ds = sorted(dur[k])
            ds = ds[2:] if len(ds) > 4 else ds
            e["serialised_duration_us_under_pmc"] = sum(dur[k][i] for i in ds) / len(ds)
            if "cycles_per_launch_per_xcd" in e and e["serialised_duration_us_under_pmc"] > 100:     # the counter window of a
                # few-microsecond dispatch is longer than the kernel, so the ratio only means something for long kernels
                e["effective_clock_ghz_under_pmc"] = e["cycles_per_launch_per_xcd"] / e["serialised_duration_us_under_pmc"] * 1e-3
            out["kernels"][f"{k} {label}" if "rows_per_launch" not in e or e["rows_per_launch"] == rows else f"{k} B={e['rows_per_launch']} (bench launch)"] = e
    with open(os.path.join(root, f"{tag}_pmc_ntt.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))

    bad = unexplained(out["kernels"])
    manifest = None
    mpath = os.path.join(root, "pmcs_manifest.json")
    if os.path.exists(mpath):
        with open(mpath) as fh:
            manifest = json.load(fh)
    sch = {"source": out["source"] + "; launches of tools/probes/prof_scheme.py (cold operand sets), matched to kernels by launch order",
           "kernels": scheme_table(glob.glob(os.path.join(root, "pmcs", "*", "*", "*_counter_collection.csv")), manifest)}
    if sch["kernels"]:
        with open(os.path.join(root, f"{tag}_pmc_scheme.json"), "w") as fh:
            json.dump(sch, fh, indent=1)
    bad.update(unexplained(sch["kernels"]))
    if bad:
        raise SystemExit(f"pmc_summary: traffic / algorithmic bytes outside [0.9, 1.5] without an explanation: {bad} -- a wrong byte count "
                         "or a wrong attribution; fix it or add the reason to EXPLAINED")


if __name__ == "__main__":
    main()
