#!/usr/bin/env python3
"""ShardedScheme.aggregate_verify_sharded / verify_sharded of ONE aggregate of N signers over W ranks (processes; gloo, the
ranks share this box's GPU) with hash_ag's serial sponge on every rank ("replicated") against rank 0 alone + broadcast of the
coefficient rows ("root") -- fusion/fusion.py:586-591, :632-652.  The wall time of either is the sponge's (serial by
construction: it does not shrink with W); what differs is how many host cores it occupies and what the other ranks wait for.
usage: sharded_modes.py [--world 4] [--n 1024] [--secpar 256]        (spawns its own ranks; rank 0 prints the table)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fusion-cryptography_amd"))
sys.path.insert(0, ROOT)


def arg(name, default, cast):
    return cast(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def rank_main(rank, world, port, n, secpar):
    import numpy as np
    import torch
    import torch.distributed as dist
    import fusion.fusion as F
    from fusion_hip.dist import ShardedScheme, TorchCollective, shard_range
    from fusion_hip.scheme import BatchScheme
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    params = F.fusion_setup(secpar, 2026)
    bs = BatchScheme(params, device=0, threads=max(1, 16 // world))
    seeds = [10_000 + 2 * i for i in range(n)]
    msgs = [f"synthetic message {i:06d}" for i in range(n)]
    lo, hi = shard_range(n, rank, world)
    sk, vk_l, vk_d = bs.keygen_batch(seeds[lo:hi], device=True, keep_vk=True)
    sig = bs.sign_batch(sk, vk_d, msgs[lo:hi], device=True)
    parts = [None] * world
    dist.all_gather_object(parts, vk_l)
    vk_all = np.concatenate(parts)
    rows = []
    ref = None
    for mode in ("replicated", "root"):
        sh = ShardedScheme(bs, rank, world, TorchCollective(bs.ctx, 0), alpha_mode=mode)
        sh.aggregate_verify_sharded(vk_all, msgs, sig)
        best_av, best_v = 1e30, 1e30
        for _ in range(4):
            dist.barrier()
            t0 = time.perf_counter()
            agg, verdict = sh.aggregate_verify_sharded(vk_all, msgs, sig)
            dist.barrier()
            best_av = min(best_av, time.perf_counter() - t0)
            assert verdict == (True, ""), verdict
            t0 = time.perf_counter()
            v = sh.verify_sharded(vk_all, msgs, agg)
            dist.barrier()
            best_v = min(best_v, time.perf_counter() - t0)
            assert v == (True, "")
        if ref is None:
            ref = agg
        assert np.array_equal(agg, ref)                     # both modes: the same aggregate
        t = torch.tensor([best_av, best_v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rows.append((mode, float(t[0]), float(t[1])))
    if rank == 0:
        for mode, av, v in rows:
            print(f"W={world:2d} N={n:5d}  {mode:10s}  aggregate+verify {av * 1e3:7.2f} ms ({n / av:9,.0f} signers/s)   verify {v * 1e3:7.2f} ms", flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    if "--rank" in sys.argv:
        return rank_main(arg("--rank", 0, int), arg("--world", 1, int), arg("--port", 0, int), arg("--n", 1024, int), arg("--secpar", 256, int))
    n, secpar = arg("--n", 1024, int), arg("--secpar", 256, int)
    worlds = [arg("--world", 0, int)] if "--world" in sys.argv else [1, 2, 4, 6]
    print(f"# secpar {secpar}, one aggregate of {n} signers, ranks = processes sharing GPU 0 over gloo, {os.cpu_count()} logical CPUs, "
          f"best of 4, max over ranks")
    for w in worlds:
        from bench import rendezvous_port          # outside the ephemeral range: see there
        port = rendezvous_port()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r), "--world", str(w), "--port", str(port),
                                   "--n", str(n), "--secpar", str(secpar)]) for r in range(w)]
        rc = [p.wait() for p in procs]
        if any(rc):
            sys.exit(f"a rank failed: {rc}")


if __name__ == "__main__":
    main()
