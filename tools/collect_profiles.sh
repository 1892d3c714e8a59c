#!/bin/bash
# Regenerates every measurement committed under profiles/ in ONE gpurun call (one box: numbers from different
# boxes differ by several percent).  usage (from the repo root):
#   gpurun --timeout 1150 -- 'bash tools/collect_profiles.sh r05'      then copy gpurun_out/<tag>/<tag>_* to profiles/
# rocprofv3 over bench.py, twice (`--headline-only`: the timed regions, the device-timestamp passes and the per-dispatch passes,
# nothing else).  (1) `--streams 1`: ONE launch in flight throughout -- the configuration roofline.frac is defined on; the
# kernel_stats average of the dominant kernel (ntt_jobs16: 134 217 728 B per launch / AverageNs / 8 TB/s) is the figure the
# line's roofline.frac / avg_launch_us must agree with.  (2) the default two streams: round 5's launches are long enough that
# the profiler no longer serialises them completely (by_grid's in_flight column > 1), so this run's per-kernel average MIXES
# overlapped launches (each slower) with the one-stream region's -- it corroborates the overlap, it is not the per-launch figure.
# Stops at the first step that fails or times out: no further GPU work is started after a failed one.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
step() { echo "[collect] $*" >&2; "$@" || { echo "[collect] FAILED ($?): $*" >&2; exit 1; }; }
lscpu | grep -E "Model name|Socket|Thread|Core" > $OUT/${TAG}_host_cpu.txt
# the stand-alone microbenchmarks travel prebuilt (tools/microbench/build/, git-ignored); build whatever is missing
mkdir -p tools/microbench/build
for mb in launch_floor ntt_variants ntt_structures access_pattern mixed_ceiling shape_ceiling crosslane_latency keccak_wave salu_chain; do
  [ -x tools/microbench/build/$mb ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/microbench/$mb.hip -o tools/microbench/build/$mb
done
[ -x tools/microbench/build/keccak_host_clang ] || /opt/rocm/lib/llvm/bin/clang++ -O3 -std=c++17 tools/microbench/keccak_host.cpp -o tools/microbench/build/keccak_host_clang
[ -x tools/microbench/build/x64_throughput ] || g++ -O2 -mbmi -mbmi2 tools/microbench/x64_throughput.cpp -o tools/microbench/build/x64_throughput
step timeout -k 10 200 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
step timeout -k 10 120 ./tools/microbench/build/launch_floor > $OUT/${TAG}_launch_floor.txt 2>&1
step timeout -k 10 300 python tools/kernel_table.py > $OUT/${TAG}_kernel_table.txt 2>&1
step timeout -k 10 300 python tools/kernel_table.py --secpar 128 > $OUT/${TAG}_kernel_table_secpar128.txt 2>&1
step timeout -k 10 400 python tools/probes/agg_direct_ab.py > $OUT/${TAG}_aggregate_direct_ab.txt 2>&1
step timeout -k 10 300 python tools/probes/agg_direct_ab.py --small >> $OUT/${TAG}_aggregate_direct_ab.txt 2>&1
step timeout -k 10 200 python tools/probes/ntt_ab.py > $OUT/${TAG}_ntt_small_batches.txt 2>&1
# the challenge pipeline: as the library chooses its form by batch size, then each form forced (3 = a wave per signer, 1 = lane pairs)
for f in 0 3 1; do FZ_SHAKE_FORM=$f timeout -k 10 200 python tools/probes/challenge_bench.py 2>&1 | grep -v amdgpu >> $OUT/${TAG}_challenge_pipeline.txt || exit 1; done
# round 6: the wave-wide Keccak state against hashlib with its chain's time, the cross-lane primitives it is made of, where the
# time of sign_batch(1024) goes, the host Keccak forms with the instruction latencies / throughputs behind them, and what a
# stand-alone kernel with verify_fused's / polymul_fused's / the small aggregations' shape sustains
step timeout -k 10 200 python tools/probes/keccak_wave_check.py 256 1024 2048 4096 > $OUT/${TAG}_keccak_wave_form.txt 2>&1
step timeout -k 10 100 ./tools/microbench/build/crosslane_latency > $OUT/${TAG}_crosslane_latency.txt 2>&1
# the seeding step of the device sampler as a dependent chain on the scalar and on the vector unit (csrc/fz_sample.hip: mt_seed_state)
step timeout -k 10 100 ./tools/microbench/build/salu_chain > $OUT/${TAG}_sampler_seed_chain.txt 2>&1
step timeout -k 10 200 python tools/probes/sign_latency.py > $OUT/${TAG}_sign_latency.txt 2>&1
step timeout -k 10 200 taskset -c 4 ./tools/microbench/build/keccak_host_clang > $OUT/${TAG}_keccak_host_forms.txt 2>&1
step timeout -k 10 100 taskset -c 4 ./tools/microbench/build/x64_throughput >> $OUT/${TAG}_keccak_host_forms.txt 2>&1
step timeout -k 10 200 ./tools/microbench/build/shape_ceiling 20 > $OUT/${TAG}_shape_ceilings.txt 2>&1
# the coefficient-domain product in its two fused forms over batch sizes (kPolymul16MinRows256 in csrc/fz_ntt.hip)
step timeout -k 10 250 python tools/probes/polymul_crossover.py > $OUT/${TAG}_polymul_crossover.txt 2>&1
step timeout -k 10 200 python tools/probes/keygen_probe.py > $OUT/${TAG}_keygen_end_to_end.txt 2>&1
step timeout -k 10 200 python tools/probes/agg_probe.py > $OUT/${TAG}_aggregate_end_to_end.txt 2>&1
step timeout -k 10 200 python tools/probes/copy_bw.py > $OUT/${TAG}_copy_ceiling.txt 2>&1
step timeout -k 10 200 python tools/probes/dispatch_dist.py > $OUT/${TAG}_dispatch_distribution.txt 2>&1
step timeout -k 10 300 python tools/probes/numa_placement.py > $OUT/${TAG}_numa_placement.txt 2>&1
for st in none 0 1; do FZ_NO_PIN=1 timeout -k 10 100 python tools/probes/numa_switch.py $st >> $OUT/${TAG}_numa_placement.txt 2>&1; done
step timeout -k 10 200 python tools/probes/keccak_bench.py > $OUT/${TAG}_keccak_variants_gpu_host.txt 2>&1
step timeout -k 10 300 ./tools/microbench/build/ntt_variants 300 > $OUT/${TAG}_ntt_variants_current_kernels.txt 2>&1
step timeout -k 10 120 ./tools/microbench/build/ntt_structures 200 > $OUT/${TAG}_ntt_structures.txt 2>&1
step timeout -k 10 120 ./tools/microbench/build/access_pattern 100 > $OUT/${TAG}_access_pattern.txt 2>&1
# what a streaming kernel with the transforms' memory schedule sustains while the fp64 pipes are busy: the transforms' ceiling
step timeout -k 10 120 ./tools/microbench/build/mixed_ceiling 18 20 > $OUT/${TAG}_mixed_ceiling_2p18.txt 2>&1
step timeout -k 10 120 ./tools/microbench/build/mixed_ceiling 16 100 > $OUT/${TAG}_mixed_ceiling_2p16.txt 2>&1
# the transform schedules' crossover: the same probe with each kernel forced, and as the library chooses
for k in 0 4 16; do echo "FZ_NTT_KERNEL=$k" >> $OUT/${TAG}_ntt_crossover.txt; FZ_NTT_KERNEL=$k timeout -k 10 200 python tools/probes/ntt_ab.py 2>&1 | grep -v amdgpu >> $OUT/${TAG}_ntt_crossover.txt; done
step timeout -k 10 100 python tools/probes/keygen_sign_pair.py > $OUT/${TAG}_keygen_sign_pair.txt 2>&1
step timeout -k 10 200 python tools/probes/clock_under_load.py > $OUT/${TAG}_shader_clock_under_load.txt 2>&1
step timeout -k 10 200 python tools/probes/clock_under_load.py --secpar 128 >> $OUT/${TAG}_shader_clock_under_load.txt 2>&1
step timeout -k 10 600 bash tools/probes/matvec_ab.sh > $OUT/${TAG}_matvec_ab_raw.txt 2>&1
step timeout -k 10 300 python tools/probes/queue_probe.py > $OUT/${TAG}_queue_probe.txt 2>&1
step timeout -k 10 300 python tools/probes/sharded_modes.py > $OUT/${TAG}_sharded_alpha_modes_raw.txt 2>&1
step timeout -k 10 300 python tools/benchmarks.py 256 128 > $OUT/${TAG}_api_benchmarks.json 2> $OUT/api_benchmarks.err
step timeout -k 10 200 python tools/probes/object_api_profile.py 256 16 > $OUT/${TAG}_object_api_profile.txt 2>&1
step timeout -k 10 600 bash tools/probes/exchange_overlap.sh > $OUT/${TAG}_exchange_overlap.txt 2>&1
step timeout -k 10 400 python tools/probes/hw_queue_probe.py > $OUT/${TAG}_hw_queue_oversubscription.txt 2>&1
step timeout -k 10 300 python tools/probes/stream_sweep.py > $OUT/${TAG}_multi_stream_sweep.txt 2>&1
step timeout -k 10 300 python tools/probes/queue_aggregates.py > $OUT/${TAG}_queue_aggregates.txt 2>&1
# (tools/rccl_exit_matrix.py, 15 minutes, is run by hand when the RCCL binding changes: profiles/r05_rccl_exit_matrix.txt stands)
step timeout -k 10 600 python bench.py --full --full-out $OUT/${TAG}_bench_full.json > $OUT/${TAG}_bench_n1.json 2> $OUT/bench.err
python3 tools/stamp_table.py $OUT/${TAG}_bench_full.json > $OUT/${TAG}_device_timestamps.txt
cd /tmp && export TMPDIR=/tmp
# per-kernel durations from the profiler: the bench's transform launches, the cold kernel table, the challenge pipeline
step timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --headline-only --streams 1 --full-out $OUT/prof_bench_full.json > $OUT/prof.log 2>&1
cp $OUT/prof/*/*_kernel_stats.csv $OUT/${TAG}_bench_rocprofv3_kernel_stats.csv 2>/dev/null
# the line bench.py printed IN THAT PROFILED RUN: its roofline.avg_launch_us and the kernel_stats average above are the same
# launches measured two ways
grep -a '^{"metric"' $OUT/prof.log | tail -1 > $OUT/${TAG}_bench_under_rocprofv3_line.json
step timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof2 -- python3 $R/bench.py --headline-only --full-out $OUT/prof2_bench_full.json > $OUT/prof2.log 2>&1
cp $OUT/prof2/*/*_kernel_stats.csv $OUT/${TAG}_bench_2streams_rocprofv3_kernel_stats.csv 2>/dev/null
grep -a '^{"metric"' $OUT/prof2.log | tail -1 > $OUT/${TAG}_bench_2streams_under_rocprofv3_line.json
# ... and the whole default run (the scheme legs' kernels as the bench runs them)
step timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/profd -- python3 $R/bench.py --no-cpu-baseline --full-out /dev/null > $OUT/profd.log 2>&1
cp $OUT/profd/*/*_kernel_stats.csv $OUT/${TAG}_bench_default_rocprofv3_kernel_stats.csv 2>/dev/null
step timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/profk -- python3 $R/tools/kernel_table.py > $OUT/profk.log 2>&1
cp $OUT/profk/*/*_kernel_stats.csv $OUT/${TAG}_kernel_table_rocprofv3_kernel_stats.csv 2>/dev/null
step timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/profc -- python3 $R/tools/probes/challenge_bench.py > $OUT/profc.log 2>&1
cp $OUT/profc/*/*_kernel_stats.csv $OUT/${TAG}_challenge_rocprofv3_kernel_stats.csv 2>/dev/null
# PMC passes: one counter set per pass, nothing else traced
for set in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU"; do
  n=$(echo $set | tr " " "_" | cut -c1-30)
  step timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/pmcb/$n -- python3 $R/bench.py --headline-only --streams 1 --no-stamps --no-graph --steps 50 --prewarm-ms 20 --full-out $OUT/pmcb_bench_full.json > $OUT/pmcb_$n.log 2>&1
  step timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc20/$n -- python3 $R/tools/probes/prof_ntt.py 20 30 > $OUT/pmc20_$n.log 2>&1
  step timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/pmcs/$n -- python3 $R/tools/probes/prof_scheme.py 12 $OUT/pmcs_manifest.json > $OUT/pmcs_$n.log 2>&1
done
# where the waves' cycles go (issue, stalls, LDS): two SQ counter sets over the cold scheme kernels
step timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/sq1 -- python3 $R/tools/probes/prof_scheme.py 6 > $OUT/sq1.log 2>&1
step timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/sq2 -- python3 $R/tools/probes/prof_scheme.py 6 > $OUT/sq2.log 2>&1
# ... and over the 16-per-lane transforms at 2^20 rows
step timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/sqa -- python3 $R/tools/probes/prof_ntt.py 20 6 > $OUT/sqa.log 2>&1
step timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/sqb -- python3 $R/tools/probes/prof_ntt.py 20 6 > $OUT/sqb.log 2>&1
cd $R
python3 tools/probes/pmc_stalls.py $OUT/sq1/*/*counter_collection.csv $OUT/sq2/*/*counter_collection.csv > $OUT/${TAG}_wave_cycles.txt
python3 tools/probes/pmc_stalls.py $OUT/sqa/*/*counter_collection.csv $OUT/sqb/*/*counter_collection.csv > $OUT/${TAG}_ntt_wave_cycles.txt
python3 tools/trace_summary.py $OUT/prof/*/*_kernel_trace.csv > $OUT/${TAG}_bench_rocprofv3_by_grid.csv 2>/dev/null
python3 tools/trace_summary.py $OUT/prof2/*/*_kernel_trace.csv > $OUT/${TAG}_bench_2streams_rocprofv3_by_grid.csv 2>/dev/null
python3 tools/trace_summary.py $OUT/profd/*/*_kernel_trace.csv > $OUT/${TAG}_bench_default_rocprofv3_by_grid.csv 2>/dev/null
python3 tools/trace_summary.py $OUT/profk/*/*_kernel_trace.csv > $OUT/${TAG}_kernel_table_rocprofv3_by_grid.csv 2>/dev/null
python3 tools/trace_summary.py $OUT/profc/*/*_kernel_trace.csv > $OUT/${TAG}_challenge_rocprofv3_by_grid.csv 2>/dev/null
step python3 tools/pmc_summary.py $OUT $TAG > /dev/null

# raw traces and counter dumps are large and have been summarised above: they do not travel back
find $OUT -name "*_kernel_trace.csv" -delete 2>/dev/null
find $OUT -name "*counter_collection.csv" -delete 2>/dev/null
echo collected into $OUT
