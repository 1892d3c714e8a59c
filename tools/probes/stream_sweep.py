"""bench.py's headline (batches of 4096 forward + inverse degree-256 transforms, cold, software-pipelined) on 1 .. 8 HIP streams
and with the HIP runtime's default number of hardware queues: one child process per point (GPU_MAX_HW_QUEUES is read at the
process's first HIP call; this process never touches the GPU).  Output: profiles/r04_multi_stream_sweep.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("streams  GPU_MAX_HW_QUEUES   G NTT/s   us/step   chip frac   us per launch in flight   one-stream frac", flush=True)
DEFAULT = "default"          # bench.py's own setting: max(8, 2 x streams)
for s, q in ((1, None), (2, None), (3, None), (4, None), (4, "4"), (4, "16"), (6, None), (8, None), (8, "8")):
    env = dict(os.environ)
    env.pop("GPU_MAX_HW_QUEUES", None)
    if q is not None:
        env["GPU_MAX_HW_QUEUES"] = q
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--headline-only", "--streams", str(s), "--full-out", "/dev/null"],
                       env=env, capture_output=True, text=True, timeout=300)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    if r.returncode != 0 or not line:
        print(f"{s:7d}  failed rc={r.returncode}: {r.stderr[-300:]}", flush=True)
        continue
    d = json.loads(line[-1])
    ro = d["roofline"]
    ch = ro.get("chip", {})
    print(f"{s:7d}  {q or DEFAULT:>17s}  {d['value'] / 1e9:8.3f}  {d['ms_per_step'] * 1e3:8.3f}   "
          f"{ch.get('frac', ro['frac']):9.3f}   {ch.get('launch_us_in_flight', ro['avg_launch_us']):23.2f}   {ro['frac']:15.3f}", flush=True)
print("\n(default = bench.py's own setting of GPU_MAX_HW_QUEUES, max(8, 2 x streams), when the caller has none; 4 = the HIP runtime's default)")
