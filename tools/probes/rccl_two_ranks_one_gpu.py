#!/usr/bin/env python3
"""Does this image's RCCL accept TWO ranks on ONE GPU?  (The builder's lease has one GPU, so fz_comm_create with nranks > 1 has
never run: if RCCL allows duplicate devices, the C ABI's collectives can at least be exercised across two processes here.)
Each rank: a context on device 0, fz_comm_create(nranks = 2) with the id handed over through a file, then -- only if the
communicator exists and counts two ranks -- fz_allreduce_i64 / fz_reduce_scatter_i64 / fz_broadcast_i32 on known data.
usage: python tools/probes/rccl_two_ranks_one_gpu.py            (spawns its two ranks; every step under a timeout)"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rank_main(rank, path):
    sys.path.insert(0, os.path.join(ROOT, "fusion-cryptography_amd"))
    import numpy as np
    import fusion_hip
    ctx = fusion_hip.Context(2147465729, 256, 3337519, pow(3337519, -1, 2147465729))
    ctx.set_stream(ctx.stream_create())
    if rank == 0:
        uid = fusion_hip.comm_unique_id()
        with open(path + ".tmp", "wb") as fh:
            fh.write(uid)
        os.rename(path + ".tmp", path)
    else:
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 30:
                print(f"rank {rank}: no unique id after 30 s", flush=True)
                return 2
            time.sleep(0.05)
        uid = open(path, "rb").read()
    try:
        comm = fusion_hip.Comm(ctx, 2, rank, uid)
    except fusion_hip.FusionHipError as e:
        print(f"rank {rank}: fz_comm_create(nranks=2) refused: {e}", flush=True)
        return 3
    n, r = comm.info()
    print(f"rank {rank}: communicator of {n} ranks (this is rank {r}); RCCL {fusion_hip.rccl_version()}, {fusion_hip.rccl_library()}", flush=True)
    count = 8192
    i64 = np.arange(count, dtype=np.int64)
    mine = (rank + 1) * (1 << 33) + i64 * (rank + 1) - 7
    want = 3 * (1 << 33) + i64 * 3 - 14
    buf = fusion_hip.DeviceBuffer.from_numpy(ctx, mine)
    ctx.allreduce_i64_dev(comm, buf.ptr, count)
    ctx.synchronize()
    ok_ar = bool(np.array_equal(buf.to_numpy(np.int64, (count,)), want))
    ctx.h2d(buf.ptr, mine)
    ctx.reduce_scatter_i64_dev(comm, buf.ptr, count // 2)
    ctx.synchronize()
    got = buf.to_numpy(np.int64, (count,))[rank * count // 2:(rank + 1) * count // 2]
    ok_rs = bool(np.array_equal(got, want[rank * count // 2:(rank + 1) * count // 2]))
    rows = np.arange(4096, dtype=np.int32).reshape(16, 256) * (7 if rank == 0 else 0)
    b32 = fusion_hip.DeviceBuffer.from_numpy(ctx, rows)
    ctx.broadcast_i32_dev(comm, b32.ptr, rows.size, 0)
    ctx.synchronize()
    ok_bc = bool(np.array_equal(b32.to_numpy(np.int32, (16, 256)), np.arange(4096, dtype=np.int32).reshape(16, 256) * 7))
    print(f"rank {rank}: allreduce_i64 {'ok' if ok_ar else 'WRONG'}, reduce_scatter_i64 {'ok' if ok_rs else 'WRONG'}, broadcast_i32 {'ok' if ok_bc else 'WRONG'}", flush=True)
    comm.destroy()
    return 0 if (ok_ar and ok_rs and ok_bc) else 4


def main():
    if len(sys.argv) == 3:
        sys.exit(rank_main(int(sys.argv[1]), sys.argv[2]))
    path = os.path.join(tempfile.mkdtemp(prefix="fz_uid_"), "uid")
    procs = [subprocess.Popen(["timeout", "-k", "5", "90", sys.executable, os.path.abspath(__file__), str(r), path],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    for r, p in enumerate(procs):
        out, _ = p.communicate()
        print(f"--- rank {r}: exit code {p.returncode}")
        print("\n".join(ln for ln in out.splitlines() if "amdgpu.ids" not in ln)[-3000:])


if __name__ == "__main__":
    main()
