#!/bin/bash
# rocprofv3 view of the headline launch for three workgroup shapes, one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_prof_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in 8 4 1; do
  FZ_NTT_WAVES=$w timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/w$w -- python3 $R/bench.py --no-cpu-baseline --no-sweep --no-sign-verify --no-two-stream --no-kernel-table --no-end-to-end > $OUT/w$w.json 2> $OUT/w$w.err || echo "w=$w failed"
  cp $OUT/w$w/*/*_kernel_stats.csv $OUT/w${w}_kernel_stats.csv 2>/dev/null
  python3 $R/tools/trace_summary.py $OUT/w$w/*/*_kernel_trace.csv > $OUT/w${w}_by_grid.csv 2>/dev/null
  rm -rf $OUT/w$w
done
head -6 $OUT/w8_by_grid.csv $OUT/w4_by_grid.csv $OUT/w1_by_grid.csv
