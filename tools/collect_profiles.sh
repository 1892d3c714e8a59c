#!/bin/bash
# Regenerates every measurement committed under profiles/ in ONE gpurun call (one box: numbers from different
# boxes differ by several percent).  usage (from the repo root):
#   gpurun --timeout 900 -- 'bash tools/collect_profiles.sh r01'      then copy gpurun_out/<tag>/* to profiles/
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 120 ./tools/microbench/build/launch_floor > $OUT/${TAG}_launch_floor.txt 2>&1
timeout -k 10 120 ./tools/microbench/build/clock_probe > $OUT/${TAG}_clock_probe.txt 2>&1
timeout -k 10 300 python tools/quick_bench.py > $OUT/${TAG}_ntt_batch_sweep.txt 2>&1
timeout -k 10 200 python tools/quick_bench_scheme.py > $OUT/${TAG}_scheme_cores.txt 2>&1
timeout -k 10 400 python bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --no-cpu-baseline --no-sweep --no-sign-verify --no-two-stream > $OUT/prof.log 2>&1
cp $OUT/prof/*/*_kernel_stats.csv $OUT/${TAG}_bench_rocprofv3_kernel_stats.csv 2>/dev/null
for set in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU"; do
  n=$(echo $set | tr " " "_" | cut -c1-30)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/pmcb/$n -- python3 $R/bench.py --no-cpu-baseline --no-sweep --no-sign-verify --no-two-stream --steps 50 > $OUT/pmcb_$n.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc20/$n -- python3 $R/tools/prof_ntt.py 20 3 > $OUT/pmc20_$n.log 2>&1
done
echo collected into $OUT
ls $OUT
