#!/bin/bash
# How much of an exchange step's latency does sign_verify's second stream hide?  ONE GPU: a one-wave kernel of known duration
# (fz_diag_delay) stands in for the multi-GPU all-reduce -- on the exchange stream (the default) and on the compute stream
# (--no-exchange-overlap, round 3's form); one verification launch per step in both.  Output: profiles/r04_exchange_overlap.txt
echo "stand-in us | exchange on a second stream: us/step, M sig/s | on the compute stream: us/step, M sig/s"
for us in 0 10 20 40 60 80 120; do
  a=$(python bench.py --no-cpu-baseline --exchange-standin-us $us --verify-per-step --full-out /dev/null 2>/dev/null | grep -a '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read())['sign_verify']; print(f\"{d['ms_per_step']*1e3:7.1f} {d['value']/1e6:6.2f}\")")
  b=$(python bench.py --no-cpu-baseline --exchange-standin-us $us --verify-per-step --no-exchange-overlap --full-out /dev/null 2>/dev/null | grep -a '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read())['sign_verify']; print(f\"{d['ms_per_step']*1e3:7.1f} {d['value']/1e6:6.2f}\")")
  echo "$us | $a | $b"
done
echo
echo "with a one-rank RCCL all-reduce (fz_allreduce_i64) in front of the stand-in, exchange on the second stream: stream priority high (default) | normal"
for us in 0 40; do
  a=$(python bench.py --no-cpu-baseline --single-rank-comm --exchange-standin-us $us --verify-per-step --full-out /dev/null 2>/dev/null | grep -a '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read())['sign_verify']; print(f\"{d['ms_per_step']*1e3:7.1f} {d['value']/1e6:6.2f}\")")
  b=$(FZ_BENCH_EXCHANGE_PRIORITY=normal python bench.py --no-cpu-baseline --single-rank-comm --exchange-standin-us $us --verify-per-step --full-out /dev/null 2>/dev/null | grep -a '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read())['sign_verify']; print(f\"{d['ms_per_step']*1e3:7.1f} {d['value']/1e6:6.2f}\")")
  echo "$us | $a | $b"
done
