"""world_size-2 rehearsal of the multi-GPU exchange step on CPU (gloo): sharding + int64
all-reduce + centring gives the same aggregate as the unsharded oracle, for ragged splits."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, N, l, d, q, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "fusion-cryptography_amd"))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from fusion_hip.dist import shard_range, sharded_sum
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sig = O.splitmix_centered(11, N * l * d, q).reshape(N, l, d).astype(np.int64)
        alpha = O.splitmix_centered(12, N * d, q).reshape(N, 1, d).astype(np.int64)

        def partial(lo, hi):
            # exact products reduced mod q (centred), summed in int64 -- what fz_aggregate_partial does
            if hi == lo:
                return torch.zeros((l, d), dtype=torch.int64)
            prod = (sig[lo:hi] * alpha[lo:hi]) % q
            prod = np.where(prod > q // 2, prod - q, prod)
            return torch.from_numpy(prod.sum(axis=0))
        tot = sharded_sum(partial, N, rank, world).numpy()
        cent = tot % q
        cent = np.where(cent > q // 2, cent - q, cent).astype(np.int32)
        lo, hi = shard_range(N, rank, world)
        np.save(os.path.join(out_dir, f"r{rank}.npy"), cent)
        np.save(os.path.join(out_dir, f"range{rank}.npy"), np.array([lo, hi]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N", [1, 5, 8])
def test_sharded_aggregate_matches_oracle(N, tmp_path, coracle):
    import torch.multiprocessing as mp
    from oracle import oracle as O
    q, l, d, world = O.PRIME, 3, 64, 2
    port = 29500 + (os.getpid() + N) % 2000
    mp.spawn(_worker, args=(world, port, N, l, d, q, str(tmp_path)), nprocs=world, join=True)
    sig = O.splitmix_centered(11, N * l * d, q).reshape(N, l, d)
    alpha = O.splitmix_centered(12, N * d, q).reshape(N, d)
    want = coracle.aggregate_core(sig, alpha, q)
    covered = []
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"r{r}.npy"), want)
        covered.append(tuple(np.load(tmp_path / f"range{r}.npy")))
    assert covered[0][0] == 0 and covered[-1][1] == N and covered[0][1] == covered[1][0]


def test_shard_range_properties():
    sys.path.insert(0, os.path.join(ROOT, "fusion-cryptography_amd"))
    from fusion_hip.dist import shard_range
    for total in (0, 1, 7, 8, 1024, 2818):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [b[1] - b[0] for b in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)
