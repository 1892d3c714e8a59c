"""Which streams of ONE process get a hardware queue of their own -- and what takes them away.  Four chains of bench.py's
pipelined steps (one context + stream each, 16 rotating batches of 4096 rows per chain, hipGraphs of 500 steps) are timed
together while the process also holds
  * K extra streams that ran one empty kernel each and are idle since,
  * an RCCL communicator of one rank, created through the C ABI BEFORE the chains' streams (1), AFTER they have run (2), or
    created and destroyed before they run (3),
  * torch in the process (1), the main context on a torch stream as in bench.py (2), a torch.distributed "nccl" group of one
    rank that has run collectives (3), or a "gloo" group (4).
One child process per point: GPU_MAX_HW_QUEUES is read at the first HIP call.
Finding (profiles/r04_hw_queue_oversubscription.txt): the HIP runtime gives hardware queues to streams in the order they are
created, up to GPU_MAX_HW_QUEUES, and then shares.  Idle streams cost nothing as long as every chain still gets a queue of its
own; but RCCL creates several streams per communicator, so chains created AFTER a communicator (fz_comm_create or a
torch.distributed "nccl" group) share queues: 1.75 G NTT/s instead of 2.25 G here, 0.84 G in bench.py's process.  Created
BEFORE any communicator the chains keep their queues: bench.py creates its chain contexts first thing.
usage: python tools/probes/hw_queue_probe.py            (the sweep)
       python tools/probes/hw_queue_probe.py child S K COMM TORCH    (one point)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(S, K, with_comm, with_torch=0):
    sys.path.insert(0, os.path.join(ROOT, "fusion-cryptography_amd"))
    if with_torch:                                      # torch in the process, its HIP context and caching allocator in use
        import torch
        torch.cuda.set_device(0)
        keep = torch.zeros(1 << 20, device="cuda")
        torch.cuda.synchronize()
        if with_torch >= 3:                             # a torch.distributed process group of one rank that has run a collective
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl" if with_torch == 3 else "gloo", rank=0, world_size=1)
            t = torch.ones(4, device="cuda") if with_torch == 3 else torch.ones(4)
            dist.all_reduce(t)
            dist.barrier()
            torch.cuda.synchronize()
    import fusion_hip
    from fusion_hip._lib import NttJob
    from fusion_hip.numa import pin_to_gpu_node
    pin_to_gpu_node(0)
    import fusion.fusion as F
    ps = F.PREFIX_PARAMETERS[256]
    q, d, B, NB, STEPS = ps["modulus"], ps["degree"], 4096, 16, 500
    main = fusion_hip.Context(q, d, ps["root"], ps["inv_root"])
    if with_torch >= 2:                                 # bench.py's arrangement: the main context on a torch stream made current
        tstream = torch.cuda.Stream(torch.device("cuda", 0))
        torch.cuda.set_stream(tstream)
        main.set_stream(tstream.cuda_stream)
    comm = fusion_hip.Comm(main, 1, 0, fusion_hip.comm_unique_id()) if with_comm == 1 else None
    extra = []
    for _ in range(K):                                  # streams that exist, have run something, and idle from now on
        c = fusion_hip.Context(q, d, ps["root"], ps["inv_root"])
        s = c.stream_create()
        c.set_stream(s)
        c.diag_empty_launch()
        c.synchronize()
        extra.append((c, s))
    chains = []
    step = B * d * 4
    for _ in range(S):
        c = fusion_hip.Context(q, d, ps["root"], ps["inv_root"])
        s = c.stream_create()
        c.set_stream(s)
        x, y, z = (fusion_hip.DeviceBuffer(c, NB * step) for _ in range(3))
        c.fill_synthetic_dev(x.ptr, NB * B * d, 7)
        pairs = [(NttJob * 2)(NttJob(x.ptr + (i % NB) * step, y.ptr + (i % NB) * step, B, 0),
                              NttJob(y.ptr + ((i - 1) % NB) * step, z.ptr + ((i - 1) % NB) * step, B, 1)) for i in range(NB)]
        for i in range(NB):
            c._lib.fz_ntt_multi(c._h, pairs[i], 2)
        c.synchronize()
        c.graph_begin()
        for i in range(STEPS):
            c._lib.fz_ntt_multi(c._h, pairs[i % NB], 2)
        chains.append((c, s, c.graph_end(), (x, y, z), pairs))

    if with_comm == 2:                                  # the communicator created AFTER the chains' streams exist and have run
        comm = fusion_hip.Comm(main, 1, 0, fusion_hip.comm_unique_id())
    if with_comm == 3:                                  # created before, destroyed before the chains run
        comm = fusion_hip.Comm(main, 1, 0, fusion_hip.comm_unique_id())
        comm.destroy()

    def replay(n):
        for _ in range(n):
            for c, _, g, _, _ in chains:
                g.launch()
        for c, *_ in chains:
            c.synchronize()
    replay(6)
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        replay(8)
        best = min(best, time.perf_counter() - t0)
    rate = S * 8 * STEPS * 2 * B / best
    print(f"{rate / 1e9:.3f}", flush=True)
    os._exit(0)                                         # (RCCL loaded through the C ABI: skip the interpreter's teardown)


def main():
    print("chains  idle extra streams  RCCL comm (1 before / 2 after / 3 destroyed)  torch (2 stream / 3 nccl group / 4 gloo group)  "
          "GPU_MAX_HW_QUEUES   G NTT/s", flush=True)
    points = [(1, 0, 0, 0, None), (4, 0, 0, 0, "4"), (4, 0, 0, 0, "5"), (4, 0, 0, 0, "8"), (4, 3, 0, 0, "8"), (4, 4, 0, 0, "8"),
              (4, 4, 0, 0, "16"), (4, 0, 1, 0, "8"), (4, 0, 1, 2, "8"), (4, 0, 1, 2, "16"), (4, 0, 2, 2, "8"), (4, 0, 3, 2, "8"),
              (4, 0, 0, 3, "8"), (4, 0, 2, 3, "8"), (4, 0, 0, 4, "8"), (4, 0, 2, 4, "8")]
    for S, K, C, T, hq in points:
        env = dict(os.environ)
        env.pop("GPU_MAX_HW_QUEUES", None)
        if hq:
            env["GPU_MAX_HW_QUEUES"] = hq
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(S), str(K), str(C), str(T)], env=env,
                           capture_output=True, text=True, timeout=200)
        out = [ln for ln in r.stdout.splitlines() if ln.strip().replace(".", "").isdigit()]
        val = out[-1] if out and r.returncode == 0 else f"failed rc={r.returncode} {r.stderr[-200:]!r}"
        print(f"{S:6d}  {K:18d}  {C:44d}  {T:47d}  {hq or 'runtime default (4)':>17s}   {val}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 0)
    else:
        main()
