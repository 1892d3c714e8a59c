"""per-kernel averages of a rocprofv3 --pmc counter_collection.csv: python tools/probes/pmc_stalls.py <csv> [...]"""
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:] + " grid=" + r.get("Grid_Size", "?")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    print(k)
    wc = m.get("SQ_WAVE_CYCLES", 0)
    for c, v in sorted(m.items()):
        extra = f"  ({v / wc * 100:5.1f} % of wave cycles)" if wc and c.startswith(("SQ_WAIT", "SQ_ACTIVE_INST")) else ""
        print(f"    {c:28s} {v:16.0f}{extra}")
