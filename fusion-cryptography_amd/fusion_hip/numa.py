"""Host-thread placement: keep the process on the CPUs of ONE NUMA node -- the one the GPU hangs off.

A dispatch is a doorbell write from the launching core and a completion signal the command processor writes back.  On a
two-socket MI355X host the same 4096-row transform launch measures 4.4-4.8 us (begin -> end) and 9.6-10.4 us of host time
per launch while the launching thread stays on the socket the runtime was initialised on, and 5.1-7.5 us / 11.6-14 us once
the scheduler has moved it to the other socket (profiles/r02_numa_placement.txt): an unpinned benchmark is bimodal from
run to run.  WHICH node is second order (+-3 %, and not always in favour of the GPU's own node); the GPU's node is the
principled choice.  Nothing here touches the GPU: it reads sysfs (KFD topology -> PCI address -> node) and calls
sched_setaffinity, so it can -- and should -- run before the first HIP call (the runtime's own threads inherit the
affinity they are created under)."""
import glob
import os


def _cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_nodes():
    """NUMA node of every GPU the KFD topology lets this process read, in topology order (-1: unknown).  In a container
    that was handed one GPU of eight, the other seven are listed but unreadable: they are not devices of this process."""
    nodes = []
    for d in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*"), key=lambda p: int(os.path.basename(p))):
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(d, "properties")) if len(line.split()) >= 2)
        except OSError:
            continue                                       # not ours
        try:
            if int(props.get("simd_count", "0")) == 0:
                continue                                   # a CPU node
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
            bdf = "%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)
            nodes.append(int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read()))
        except (OSError, KeyError, ValueError):
            nodes.append(-1)
    return nodes


def pin_to_gpu_node(device_index=0):
    """restrict this process to the allowed CPUs of the GPU's NUMA node; -> a description, or None if nothing was changed
    (unknown topology, single node, FZ_NO_PIN=1)"""
    if os.environ.get("FZ_NO_PIN") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        if vis and all(t.strip().isdigit() for t in vis.split(",")):
            device_index = int(vis.split(",")[device_index])
        nodes = gpu_numa_nodes()
        node = nodes[device_index] if 0 <= device_index < len(nodes) else -1
        if node < 0:
            return None
        local = _cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())
        allowed = os.sched_getaffinity(0)
        want = allowed & local
        if not want or want == allowed:
            return None
        os.sched_setaffinity(0, want)
        return f"NUMA node {node} (the GPU's): {len(want)} of {len(allowed)} allowed CPUs"
    except (OSError, ValueError, IndexError):
        return None
