#!/bin/bash
# Regenerates every measurement committed under profiles/ in ONE gpurun call (one box: numbers from different
# boxes differ by several percent).  usage (from the repo root):
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r01'      then copy gpurun_out/<tag>/<tag>_* to profiles/
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
step timeout -k 10 120 ./tools/microbench/build/clock_probe > $OUT/${TAG}_clock_probe.txt 2>&1
step timeout -k 10 300 python tools/quick_bench.py > $OUT/${TAG}_ntt_batch_sweep.txt 2>&1
step timeout -k 10 200 python tools/quick_bench_scheme.py > $OUT/${TAG}_scheme_cores.txt 2>&1
step timeout -k 10 400 python bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
step timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --no-cpu-baseline --no-sweep --no-sign-verify --no-two-stream > $OUT/prof.log 2>&1
cp $OUT/prof/*/*_kernel_stats.csv $OUT/${TAG}_bench_rocprofv3_kernel_stats.csv 2>/dev/null
for set in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU"; do
  n=$(echo $set | tr " " "_" | cut -c1-30)
  step timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/pmcb/$n -- python3 $R/bench.py --no-cpu-baseline --no-sweep --no-sign-verify --no-two-stream --no-graph --steps 50 --prewarm-ms 20 > $OUT/pmcb_$n.log 2>&1
  step timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc20/$n -- python3 $R/tools/prof_ntt.py 20 30 > $OUT/pmc20_$n.log 2>&1
done
cd $R
step python3 tools/pmc_summary.py $OUT $TAG > /dev/null
echo collected into $OUT
