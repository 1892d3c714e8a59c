"""print the roofline fields of a bench.py JSON line (file argument, or stdin without one): the last line that starts with
{"metric" is taken, so a whole log can be piped in"""
import json
import sys

text = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
lines = [ln for ln in text.splitlines() if ln.startswith('{"metric"')]
if not lines:
    sys.exit("no bench.py line found")
d = json.loads(lines[-1])
r = d.get("roofline") or {}
chip = r.get("chip") or {}
print(f"value {d.get('value', 0) / 1e9:.4f} G {d.get('unit', '')}   ms_per_step {d.get('ms_per_step')}")
print(f"roofline: {r.get('kernel')}  {r.get('avg_launch_us')} us per launch  frac {r.get('frac')}  "
      f"traffic {r.get('traffic')} ({r.get('traffic_source', 'source not stated')}) vs {r.get('bytes_per_launch')} algorithmic bytes")
print(f"chip ({chip.get('streams')} streams): frac {chip.get('frac')}  launches timed {chip.get('launches_timed')}")
for k in ("sign_verify", "keygen_sign", "end_to_end"):
    if isinstance(d.get(k), dict):
        print(k, {kk: vv for kk, vv in d[k].items() if isinstance(vv, (int, float))})
