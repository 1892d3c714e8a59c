"""The asynchronous batch queue of the C ABI (fz_queue_*, fusion_hip.queue.BatchQueue): calls submitted from one thread are run
as coalesced batches by worker threads below Python; every call's keys and signatures must be the integers keygen / sign give
for that call alone -- the reference's golden arrays (fusion/fusion.py:338-373, :534-557) and BatchScheme on the same inputs."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("secpar", [128, 256])
def test_queue_reproduces_the_reference_arrays(secpar):
    import fusion.fusion as F
    from fusion_hip.queue import BatchQueue
    S = np.load(os.path.join(G, f"scheme_{secpar}.npz"))
    with open(os.path.join(G, "scheme.json")) as fh:
        m = json.load(fh)[str(secpar)]
    params = F.fusion_setup(secpar, m["setup_seed"])
    with BatchQueue(params, workers=2) as bq:
        # the four golden keys as one call, and as four calls of one key each (coalesced or not: the same rows)
        t_all = bq.submit_keygen_sign(m["key_seeds"], m["messages"], keep_sk=True)
        singles = [bq.submit_keygen_sign([s], [msg]) for s, msg in zip(m["key_seeds"], m["messages"])]
        r = bq.wait(t_all)
        assert r.n == 4 and np.array_equal(r.vk, S["vk"])
        assert np.array_equal(r.signatures(), S["sig"]) and np.array_equal(r.secret_keys(), S["sk_hat"])
        r.release()
        r.release()                                        # idempotent
        for i, t in enumerate(singles):
            ri = bq.wait(t)
            assert ri.n == 1 and np.array_equal(ri.vk[0], S["vk"][i]) and np.array_equal(ri.signatures()[0], S["sig"][i])
            assert ri.sk_ptr is None
            with pytest.raises(Exception):
                ri.secret_keys()                           # not kept
            ri.release(copy_vk=True)
            assert np.array_equal(ri.vk[0], S["vk"][i])


def test_many_calls_of_different_sizes_equal_batch_scheme():
    """60 calls of 1 .. 300 keys submitted back to back from one thread (most of them coalesced), some discarded: rows, order
    and ownership survive the cut into calls; release in any order; stats show the coalescing"""
    import fusion.fusion as F
    from fusion_hip.queue import BatchQueue, PackedMessages
    from fusion_hip.scheme import BatchScheme
    params = F.fusion_setup(256, 2026)
    bs = BatchScheme(params, threads=4)
    rng = np.random.default_rng(5)
    sizes = [int(x) for x in rng.integers(1, 300, size=60)]
    sizes[7] = 1
    calls = []
    for k, n in enumerate(sizes):
        seeds = [int(x) for x in rng.integers(0, 2**63, size=n)]
        if k == 3:
            seeds[0] = 2**64 - 2                           # the largest seed the clones take
        msgs = [f"call {k} message {i} " + "x" * int(rng.integers(0, 200)) for i in range(n)]
        if k == 5:
            msgs[0] = ""
            msgs[-1] = "é中 multi-byte"
        calls.append((seeds, msgs))
    with BatchQueue(params, workers=3, max_rows=1024) as bq:
        tickets = []
        for k, (seeds, msgs) in enumerate(calls):
            packed = PackedMessages(msgs) if k % 2 else msgs
            sd = np.array(seeds, dtype=np.uint64) if k % 3 == 0 else seeds
            tickets.append(bq.submit_keygen_sign(sd, packed, keep_sk=(k % 4 == 0), discard=(k % 10 == 9)))
        order = list(rng.permutation(len(calls)))
        for k in order:
            seeds, msgs = calls[k]
            r = bq.wait(tickets[k])
            sk, vk = bs.keygen_batch(seeds)
            assert np.array_equal(r.vk, vk), k                 # discarded calls still deliver their verification keys
            if k % 10 == 9:
                assert r.n == 0 and r.sig_ptr is None
            else:
                assert r.n == len(seeds)
                assert np.array_equal(r.signatures(), bs.sign_batch(sk, vk, msgs)), k
                if k % 4 == 0:
                    assert np.array_equal(r.secret_keys(), sk), k
                else:
                    assert not r.sk_ptr, k                     # even when another call of the same batch kept its secret keys
            r.release()
        done, batches, rows = bq.stats()
        assert done == len(calls) and rows == sum(sizes) and batches <= done
        bq.drain()


def test_queue_argument_errors_and_shutdown_with_results_outstanding():
    import fusion.fusion as F
    from fusion_hip import FusionHipError
    from fusion_hip.queue import BatchQueue
    params = F.fusion_setup(128, 3)
    bq = BatchQueue(params, workers=1, max_rows=64)
    with pytest.raises(FusionHipError) as e:
        bq.submit_keygen_sign(np.array([2**64 - 1], dtype=np.uint64), ["m"])       # seed + 1 would wrap
    assert e.value.code == -2
    with pytest.raises(ValueError):
        bq.submit_keygen_sign([-5], ["m"])                                          # random.seed(abs()): the Python sampler's
    with pytest.raises(ValueError):
        bq.submit_keygen_sign([1, 2], ["only one"])
    with pytest.raises(FusionHipError):
        bq.submit_keygen_sign(list(range(65)), ["m"] * 65)                          # more than max_rows
    with pytest.raises(FusionHipError):
        bq.wait(12345)                                                              # unknown ticket
    t = bq.submit_keygen_sign([1, 2, 3], ["a", "b", "c"])
    bq.close()                                                                      # finishes the call, frees its rows, joins
    bq.close()
    assert t == 1


def test_c_caller_drives_the_queue(tmp_path):
    """examples/queue_flow.c (strict C99, gcc, no HIP headers): 12 calls of 40 keys + signatures submitted from one thread,
    batches of at most 256 rows on two workers; every call's verification keys (pinned host buffers) and signatures (read
    back through the caller's own context) equal the direct entry points' rows for that call alone"""
    import subprocess
    from test_cabi_symbols import build_c_example
    r = subprocess.run([build_c_example(tmp_path, "queue_flow")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "queue_flow OK" in r.stdout, r.stdout + r.stderr


def test_release_right_behind_an_asynchronous_consumer_and_results_that_outlive_the_queue():
    """the queue's rows go back to a pool whose reuse is ordered on the WORKER's stream only, so a release
    right behind an asynchronous kernel that still reads them let the next coalesced batch overwrite them.  release(after=ctx)
    records the consumer's position (fz_queue_release_after) and the worker waits for it: an aggregation queued on the rows of
    every call -- behind a deliberate 2 ms delay on the consumer's stream -- followed at once by the release and by more calls
    of the same size (which take the same pooled blocks) must still give the aggregate of the ORIGINAL signatures.
     results still held when the queue closes keep a private copy of vk and release() on them is a no-op."""
    import fusion.fusion as F
    import fusion_hip
    from fusion_hip.queue import BatchQueue
    from fusion_hip.scheme import BatchScheme
    params = F.fusion_setup(256, 77)
    bs = BatchScheme(params, private_context=True)         # a non-default stream of its own: nothing orders it with the workers'
    ctx, l, d, n = bs.ctx, params.num_rows_sk, params.degree, 512
    rng = np.random.default_rng(11)
    al = ctx.ntt_forward(rng.integers(-1, 2, size=(n, d)).astype(np.int32))
    d_al = fusion_hip.DeviceBuffer.from_numpy(ctx, al)
    with BatchQueue(params, workers=1, max_rows=n) as bq:
        for rep in range(4):
            seeds = [1000 * rep + i for i in range(n)]
            msgs = [f"r{rep} m{i}" for i in range(n)]
            r = bq.wait(bq.submit_keygen_sign(seeds, msgs))
            want_sig = bs.sign_batch(*bs.keygen_batch(seeds), msgs)
            want = ctx.aggregate_core(want_sig, al)
            d_out = fusion_hip.DeviceBuffer(ctx, l * d * 4)
            ctx.diag_delay(2000)                               # the consumer is BUSY: the aggregation below starts 2 ms from now
            ctx.aggregate_core_dev(r.sig_ptr, d_al.ptr, d_out.ptr, n, l)
            r.release(after=ctx)                               # no synchronisation on this side
            nxt = [bq.submit_keygen_sign([5000 + 7 * k + i for i in range(n)], msgs) for k in range(2)]    # same sizes: the same pooled blocks
            for t in nxt:
                bq.wait(t).release()
            ctx.synchronize()
            assert np.array_equal(d_out.to_numpy(np.int32, (l, d)), want), rep
            d_out.free()
        held = bq.wait(bq.submit_keygen_sign([1, 2, 3], ["a", "b", "c"]))
        vk_before = held.vk.copy()
    # the queue is closed: the result is detached, not dangling
    assert held.sig_ptr is None and np.array_equal(held.vk, vk_before)
    held.release()
    held.release(after=ctx)
    d_al.free()
    bs.close()


@pytest.mark.parametrize("secpar", [128, 256])
def test_queued_aggregates_and_verifications_equal_the_reference(secpar):
    """aggregate() + verify(), and verify() alone, as queued calls (fz_queue_submit_aggregate_verify /
    fz_queue_submit_verify) -- many pending calls share ONE ragged launch for the partial sums and ONE for the verdicts.  On the
    reference's own flows (tests/golden/scheme_many_*: distinct signers, nested sub-aggregates of different sizes): every queued
    aggregate is the array the REFERENCE computed (fusion.py:655-677), every verdict the reference's (:680-728), tampered
    aggregates and swapped messages included; signatures as host arrays, as a DeviceArray and as the queue's own device rows."""
    import fusion.fusion as F
    import fusion_hip
    from fusion_hip.queue import BatchQueue
    from fusion_hip.scheme import BatchScheme
    S = np.load(os.path.join(G, f"scheme_many_{secpar}.npz"))
    with open(os.path.join(G, "scheme_many.json")) as fh:
        m = json.load(fh)[str(secpar)]
    params = F.fusion_setup(secpar, m["setup_seed"])
    bs = BatchScheme(params)
    msgs = m["messages"]
    with BatchQueue(params, workers=2, max_rows=4096) as bq:
        r = bq.wait(bq.submit_keygen_sign(m["key_seeds"], msgs))           # the queue's own keys and signatures (device rows)
        vk, sig = r.vk.copy(), r.signatures()
        assert np.array_equal(vk, S["vk"])
        d_sig = fusion_hip.DeviceArray.from_numpy(bs.ctx, sig)
        l, d = params.num_rows_sk, params.degree
        tickets = []
        for k, (lo, hi) in enumerate(m["subsets"] * 3):                     # 12 calls pending at once, sizes differ
            how = k % 3
            rows = (sig[lo:hi] if how == 0 else d_sig.ptr + lo * l * d * 4 if how == 1 else r.sig_ptr + lo * l * d * 4)
            tickets.append((lo, hi, bq.submit_aggregate_verify(vk[lo:hi], msgs[lo:hi], rows)))
        aggs = {}
        for lo, hi, t in tickets:
            agg, verdict = bq.wait_aggregate(t)
            info = m["agg"][f"{lo}_{hi}"]
            assert np.array_equal(agg, S[f"agg_{lo}_{hi}"]), (lo, hi)
            assert list(verdict) == info["verdict"]
            aggs[(lo, hi)] = agg
        # verify() alone: the reference's verdicts for the aggregate, a tampered one and swapped messages -- 12 calls pending
        vt = []
        for lo, hi in m["subsets"]:
            info = m["agg"][f"{lo}_{hi}"]
            agg = aggs[(lo, hi)]
            bad = agg.copy()
            bad[info["tampered_at"][0], info["tampered_at"][1]] += 1
            sw = list(msgs[lo:hi])
            sw[0], sw[-1] = sw[-1], sw[0]
            vt.append((bq.submit_verify(vk[lo:hi], msgs[lo:hi], agg), info["verdict"]))
            vt.append((bq.submit_verify(vk[lo:hi], msgs[lo:hi], bad), info["tampered_verdict"]))
            vt.append((bq.submit_verify(vk[lo:hi], sw, agg), info["swapped_messages_verdict"]))
        for t, want in vt:
            assert list(bq.wait_verdict(t)) == want
        calls, batches, _ = bq.stats()
        assert batches < calls                                              # calls WERE coalesced
        with pytest.raises(ValueError):
            bq.submit_verify(vk[:3], msgs[:2], aggs[tuple(m["subsets"][0])])
        r.release()
        d_sig.free()
    bs.close()


def test_queued_aggregates_mixed_with_keygen_sign_and_the_capacity_verdict():
    """calls of all three kinds interleaved from one thread (a batch is a run of calls of one kind: order is kept), 80 aggregates
    of 1 .. 40 signers pending at once (more than the 64 groups one ragged launch holds) against BatchScheme.aggregate_many /
    verify_many, and the reference's capacity check (fusion.py:686-687: "Too many keys.") through a queue whose capacity is 3"""
    import types
    import fusion.fusion as F
    from fusion_hip.queue import BatchQueue
    from fusion_hip.scheme import BatchScheme
    params = F.fusion_setup(256, 4242)
    bs = BatchScheme(params)
    rng = np.random.default_rng(3)
    sizes = [int(x) for x in rng.integers(1, 41, size=80)]
    n = sum(sizes)
    seeds = [31 * i + 5 for i in range(n)]
    msgs = [f"msg {i}" for i in range(n)]
    sk, vk = bs.keygen_batch(seeds)
    sig = bs.sign_batch(sk, vk, msgs)
    want = bs.aggregate_many(vk, msgs, sig, sizes)
    assert bs.verify_many(vk, msgs, want, sizes) == [(True, "")] * len(sizes)
    off = np.concatenate([[0], np.cumsum(sizes)])
    with BatchQueue(params, workers=2, max_rows=2048) as bq:
        tickets, others = [], []
        for g in range(len(sizes)):
            a, b = int(off[g]), int(off[g + 1])
            tickets.append(bq.submit_aggregate_verify(vk[a:b], msgs[a:b], sig[a:b]))
            if g % 16 == 7:                                                  # another kind in between
                others.append((bq.submit_keygen_sign(seeds[a:b], msgs[a:b]), a, b))
        for g, t in enumerate(tickets):
            agg, verdict = bq.wait_aggregate(t)
            assert np.array_equal(agg, want[g]) and verdict == (True, ""), g
        for t, a, b in others:
            res = bq.wait(t)
            assert np.array_equal(res.vk, vk[a:b]) and np.array_equal(res.signatures(), sig[a:b])
            res.release()
    small = types.SimpleNamespace(**{k: getattr(params, k) for k in dir(params) if not k.startswith("_")})
    small.capacity = 3
    with BatchQueue(small, workers=1) as bq:
        a, b = int(off[5]), int(off[5]) + 4
        agg4 = bs.aggregate(vk[a:b], msgs[a:b], sig[a:b])
        agg, verdict = bq.wait_aggregate(bq.submit_aggregate_verify(vk[a:b], msgs[a:b], sig[a:b]))
        assert np.array_equal(agg, agg4) and verdict == (False, "Too many keys.")
        assert bq.wait_verdict(bq.submit_verify(vk[a:b], msgs[a:b], agg4)) == (False, "Too many keys.")
        assert bq.wait_verdict(bq.submit_verify(vk[a:a + 3], msgs[a:a + 3], bs.aggregate(vk[a:a + 3], msgs[a:a + 3], sig[a:a + 3]))) == (True, "")
    bs.close()
