#!/bin/bash
# Regenerates every measurement committed under profiles/ in ONE gpurun call (one box: numbers from different
# boxes differ by several percent).  usage (from the repo root):
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r02'      then copy gpurun_out/<tag>/<tag>_* to profiles/
# Stops at the first step that fails or times out: no further GPU work is started after a failed one.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
step() { echo "[collect] $*" >&2; "$@" || { echo "[collect] FAILED ($?): $*" >&2; exit 1; }; }
lscpu | grep -E "Model name|Socket|Thread|Core" > $OUT/${TAG}_host_cpu.txt
step timeout -k 10 200 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
step timeout -k 10 120 ./tools/microbench/build/launch_floor > $OUT/${TAG}_launch_floor.txt 2>&1
step timeout -k 10 300 python tools/kernel_table.py > $OUT/${TAG}_kernel_table.txt 2>&1
step timeout -k 10 300 python tools/agg_tune.py --twopass > $OUT/${TAG}_aggregate_shapes.txt 2>&1
step timeout -k 10 200 python tools/ntt_ab.py > $OUT/${TAG}_ntt_small_batches.txt 2>&1
step timeout -k 10 200 python tools/challenge_bench.py > $OUT/${TAG}_challenge_pipeline.txt 2>&1
step timeout -k 10 500 python bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
# per-kernel durations from the profiler: the bench's transform launches, the cold kernel table, the challenge pipeline
step timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --no-cpu-baseline --no-sweep --no-sign-verify --no-two-stream --no-kernel-table > $OUT/prof.log 2>&1
cp $OUT/prof/*/*_kernel_stats.csv $OUT/${TAG}_bench_rocprofv3_kernel_stats.csv 2>/dev/null
step timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/profk -- python3 $R/tools/kernel_table.py > $OUT/profk.log 2>&1
cp $OUT/profk/*/*_kernel_stats.csv $OUT/${TAG}_kernel_table_rocprofv3_kernel_stats.csv 2>/dev/null
step timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/profc -- python3 $R/tools/challenge_bench.py > $OUT/profc.log 2>&1
cp $OUT/profc/*/*_kernel_stats.csv $OUT/${TAG}_challenge_rocprofv3_kernel_stats.csv 2>/dev/null
# PMC passes: one counter set per pass, nothing else traced
for set in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU"; do
  n=$(echo $set | tr " " "_" | cut -c1-30)
  step timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/pmcb/$n -- python3 $R/bench.py --no-cpu-baseline --no-sweep --no-sign-verify --no-two-stream --no-kernel-table --no-graph --steps 50 --prewarm-ms 20 > $OUT/pmcb_$n.log 2>&1
  step timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc20/$n -- python3 $R/tools/prof_ntt.py 20 30 > $OUT/pmc20_$n.log 2>&1
  step timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/pmcs/$n -- python3 $R/tools/prof_scheme.py 12 > $OUT/pmcs_$n.log 2>&1
done
cd $R
step python3 tools/pmc_summary.py $OUT $TAG > /dev/null
echo collected into $OUT
