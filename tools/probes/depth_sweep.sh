#!/bin/bash
# bench.py's headline over pipeline depth (batches per fz_ntt_multi launch) and streams, one box, one run:
#   per-launch fraction of the HBM peak with ONE launch in flight (HIP events; the device's own clock in brackets) and the
#   chip-level fraction / NTT rate with S streams.      usage: bash tools/probes/depth_sweep.sh > profiles/rNN_depth_sweep.txt
cd "$(dirname "$0")/../.."
echo "depth streams | kernel | one stream: frac  launch us (events)  [device clock: launch us, gap us] | S streams: G NTT/s  chip frac (events)  chip frac (device clock)  in flight"
for cfg in "1 1" "1 2" "1 4" "2 1" "2 2" "2 4" "4 1" "4 2" "4 3" "8 1" "8 2" "8 3" "16 1" "16 2"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --headline-only --no-cpu-baseline --depth $1 --streams $2 --steps 20 --warmup 5 --full-out /dev/null 2>/dev/null | tail -1 | python3 -c "
import json, sys
o = json.loads(sys.stdin.read()); r = o['roofline']; d = r.get('device_clock') or {}; c = r['chip']; cd = c.get('device_clock') or {}
print(f\"{$1:>5} {$2:>7} | {r['kernel']:<40} | {r['frac']:.3f} {r['avg_launch_us']:>7.2f}  [{d.get('launch_us', 0):.2f}, {d.get('gap_us', 0):.2f}] | {o['value'] / 1e9:.3f}  {c['frac']:.3f}  {cd.get('frac', 0):.3f}  {cd.get('in_flight', 0):.2f}\")"
done
