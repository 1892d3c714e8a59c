"""The N > 1 path with the library's own kernels on both sides of a REAL collective: two ranks (processes) on the test
box's GPU, each signing and partially aggregating its contiguous block of the signers
(fz_aggregate_target_partial_batch), one torch.distributed all-reduce of the int64 partials, then verification from the
sums (fz_verify_partials_batch_async) and centring (fz_reduce_i64) on every rank.  Checked against the oracle's aggregate
over ALL signers.  (fusion/fusion.py:670-676, :706-727; SURVEY.md 8e.  gloo because the ranks share one GPU; bench.py runs
the same step over RCCL through fz_allreduce_i64 when every rank has a GPU of its own.)"""
import os
import sys

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("secpar,n,world", [(256, 301, 2), (128, 77, 3)])
def test_sharded_aggregate_and_verify_with_a_real_collective(secpar, n, world, coracle, tmp_path):
    from _ranks import rendezvous_port, run_rank_processes
    port = rendezvous_port()
    run_rank_processes([[sys.executable, os.path.join(HERE, "_dist_worker.py"), str(r), str(world), str(port), str(secpar), str(n), str(tmp_path)]
                        for r in range(world)], tmp_path, 240)
    P = O.PARAMS[secpar]
    q, d, l = P["q"], P["d"], P["rank"]
    R = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    assert [int(r["lo"]) for r in R] == [sum(int(x["hi"] - x["lo"]) for x in R[:i]) for i in range(world)] and int(R[-1]["hi"]) == n
    sig = np.concatenate([r["sig"] for r in R])
    al = np.concatenate([r["al_hat"] for r in R])
    ch = np.concatenate([r["c_hat"] for r in R])
    vk = np.concatenate([r["vk"] for r in R])
    ref = coracle.aggregate_core(sig, al, q)
    total = sum(r["local"] for r in R)
    for r in R:
        assert np.array_equal(r["total"], total)                       # the collective summed the ranks' partials exactly
        assert np.array_equal(r["agg"], ref)                           # ... and every rank centres them to the oracle's aggregate
        assert r["verdict"].tolist() == [0] and r["verdict_bad"].tolist() == [3]
    assert coracle.verify_core(R[0]["A"], ref, vk[:, 0], vk[:, 1], ch, al, q, P["inv_root"], P["beta_vf"], d) == 0
