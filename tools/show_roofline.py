"""print the roofline fields of a bench.py JSON line (file argument, or stdin without one)"""
import json
import sys
d = json.loads((open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()).strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1:] , "value", round(d["value"] / 1e9, 4), "G; fwd", round(r["avg_launch_us"], 3), "us inv", round(r["inverse_avg_launch_us"], 3),
      "us frac", round(r["frac"], 4), "samples", r["launches_timed"], "region", round(r["region"]["avg_launch_us"], 3))
