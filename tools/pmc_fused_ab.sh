#!/bin/bash
# vector-instruction counts of the fused keygen / verify kernels for the A/B knobs, one box: FZ_FUSED_ROWS x FZ_NO_IMAD
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_pmc_fused
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "rows1_imad:FZ_FUSED_ROWS=1" "rows2_imad:FZ_FUSED_ROWS=2" "rows1_fp64:FZ_FUSED_ROWS=1 FZ_NO_IMAD=1" "rows1_imad_cent:FZ_FUSED_ROWS=1 FZ_VERIFY_CENT=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  export $envs
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/$name -- python3 $R/tools/prof_scheme.py 4 > $OUT/$name.log 2>&1 || echo "$name failed"
  unset FZ_FUSED_ROWS FZ_NO_IMAD FZ_VERIFY_CENT
  python3 - "$OUT/$name" "$name" <<'PY'
import csv, glob, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "keygen_fused" in k or "verify_fused" in k:
            acc[k.split("(")[0]+" grid="+r.get("Grid_Size","?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()):
    print(sys.argv[2], k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
done
