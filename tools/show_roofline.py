"""print the roofline fields of a bench.py JSON line read from stdin (scratch helper)"""
import json
import sys
d = json.loads(sys.stdin.read())
r = d["roofline"]
print(sys.argv[1:] , "value", round(d["value"] / 1e9, 4), "G; fwd", round(r["avg_launch_us"], 3), "us inv", round(r["inverse_avg_launch_us"], 3),
      "us frac", round(r["frac"], 4), "samples", r["launches_timed"], "region", round(r["region"]["avg_launch_us"], 3))
