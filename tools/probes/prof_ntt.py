"""Minimal launcher for profiling: runs the forward and inverse NTT kernels a few times on a large
device-resident batch.  usage: prof_ntt.py [log2_batch] [reps] [secpar]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "fusion-cryptography_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import fusion_hip
from fusion_hip.numa import pin_to_gpu_node
pin_to_gpu_node(0)          # host threads on the GPU's NUMA node (before the first HIP call)
from oracle import oracle as O

logB = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
secpar = int(sys.argv[3]) if len(sys.argv) > 3 else 256
P = O.PARAMS[secpar]
d = P["d"]
ctx = fusion_hip.Context(P["q"], d, P["root"], P["inv_root"])
B = 1 << logB
din = fusion_hip.DeviceBuffer(ctx, B * d * 4)
dout = fusion_hip.DeviceBuffer(ctx, B * d * 4)
ctx.fill_synthetic_dev(din.ptr, B * d, 5)
for _ in range(reps):
    ctx.ntt_forward_dev(din.ptr, dout.ptr, B)
    ctx.ntt_inverse_dev(dout.ptr, din.ptr, B)
ctx.synchronize()
print("done", B, reps)
