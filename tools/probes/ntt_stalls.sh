set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r05e
mkdir -p $OUT
rocprofv3 -L > $OUT/counters.txt 2>&1 || rocprofv3 --list-avail > $OUT/counters.txt 2>&1 || true
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/sqa -- python3 tools/probes/prof_ntt.py 20 6 > $OUT/sqa.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/sqb -- python3 tools/probes/prof_ntt.py 20 6 > $OUT/sqb.log 2>&1
python3 tools/probes/pmc_stalls.py $OUT/sqa/*/*counter_collection.csv $OUT/sqb/*/*counter_collection.csv > $OUT/ntt_stalls.txt
rm -rf $OUT/sqa $OUT/sqb
tail -60 $OUT/ntt_stalls.txt
