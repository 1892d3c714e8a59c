"""per-dispatch duration and host launch cost of the B=4096 transform from the GPU's NUMA node and from the other one:
runs tools/probes/dispatch_dist.py as child processes under the two CPU sets (this process never touches the GPU)"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "fusion-cryptography_amd"))
from fusion_hip import numa

nodes = numa.gpu_numa_nodes()
if not nodes or nodes[0] < 0:
    sys.exit("no GPU NUMA node found")
local = nodes[0]
allowed = os.sched_getaffinity(0)
sets = {}
for d in sorted(os.listdir("/sys/devices/system/node")):
    if d.startswith("node") and d[4:].isdigit():
        cpus = numa._cpulist(open(f"/sys/devices/system/node/{d}/cpulist").read()) & allowed
        if cpus:
            sets[int(d[4:])] = cpus
print(f"GPU 0 hangs off NUMA node {local}; allowed CPUs per node: " + ", ".join(f"node {n}: {len(c)}" for n, c in sets.items()))
for rep in range(2):
    for n, cpus in sets.items():
        env = dict(os.environ, FZ_NO_PIN="1")
        out = subprocess.run([sys.executable, os.path.join(R, "tools", "dispatch_dist.py")], env=env, capture_output=True, text=True,
                             preexec_fn=lambda c=cpus: os.sched_setaffinity(0, c), timeout=120).stdout
        for line in out.splitlines():
            if line.startswith("pass 1") or line.startswith("pass 2"):
                where = "the GPU's node" if n == local else "the other socket"
                print(f"host threads on node {n} ({where}): {line}")
