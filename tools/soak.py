#!/usr/bin/env python3
"""Randomised soak of every kernel against the CPU oracle: random batch sizes, both NTT schedules, both degrees,
fused and multi-launch paths, two contexts on two streams interleaved and ordered by events, repeated verify launches (re-armed
accumulators), graph replays.  usage: soak.py [seconds] [seed]   (needs an MI355X; prints a progress line per minute)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fusion-cryptography_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import fusion_hip  # noqa: E402
from oracle import oracle as O  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
orc = O.COracle()
q = O.PRIME
counts = {}


def bump(name):
    counts[name] = counts.get(name, 0) + 1


def make_ctx(secpar, kernel, env=None):
    """knobs are read once, at context creation: every variant is a context of its own"""
    env = dict(env or {})
    if kernel:
        env["FZ_NTT_KERNEL"] = kernel
    for k_, v_ in env.items():
        os.environ[k_] = v_
    P = O.PARAMS[secpar]
    c = fusion_hip.Context(q, P["d"], P["root"], P["inv_root"])
    for k_ in env:
        os.environ.pop(k_, None)
    c.set_stream(c.stream_create())
    return c


ctxs = {(sp, k): make_ctx(sp, k) for sp in (128, 256) for k in ("", "4", "16")}
# the radix-4 kernels with 1 / 2 / 4 row groups per wave and other workgroup shapes, whatever the batch size
for sp_ in (128, 256):
    ctxs[(sp_, "4r1")] = make_ctx(sp_, "4", {"FZ_NTT_ROWS": "1"})
    ctxs[(sp_, "4r2")] = make_ctx(sp_, "4", {"FZ_NTT_ROWS": "2"})
    ctxs[(sp_, "4r4")] = make_ctx(sp_, "4", {"FZ_NTT_ROWS": "4"})
KERNS = ["", "4", "16", "4r1", "4r2", "4r4"]
# the multi-launch forms of the fused kernels and other launch shapes of the one-pass aggregation
VARIANTS = [{}, {"FZ_UNFUSED": "1"}, {"FZ_AGG_DIRECT": "-1"}, {"FZ_AGG_DIRECT": "2"}, {"FZ_AGG_DIRECT": "4", "FZ_VERIFY_ORDERED": "1"},
            {"FZ_NO_IMAD": "1"}, {"FZ_VERIFY_CENT": "1"}, {"FZ_SHAKE_FORM": "2"}, {"FZ_SHAKE_FORM": "1"}, {"FZ_SHAKE_FORM": "3"},
            {"FZ_MATVEC_SLICES": "16"}, {"FZ_MATVEC_SLICES": "2"}, {"FZ_MATVEC_SLICES": "-1"}, {"FZ_POLYMUL_FORM": "2"}, {"FZ_POLYMUL_FORM": "1"}]
vctx = {(sp, i): make_ctx(sp, "", v) for sp in (128, 256) for i, v in enumerate(VARIANTS)}
DB = fusion_hip.DeviceBuffer
t_end = time.time() + budget
t_print = time.time() + 60
it = 0
while time.time() < t_end:
    it += 1
    sp = int(rng.choice([128, 256]))
    P = O.PARAMS[sp]
    d, root, inv = P["d"], P["root"], P["inv_root"]
    kern = str(rng.choice(KERNS))
    ctx = ctxs[(sp, kern)]
    other = ctxs[(sp, str(rng.choice(KERNS)))]
    rows = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 257, 1000, 4099, int(rng.integers(1, 20000))]))
    what = rng.choice(["ntt", "polymul", "scheme", "graph", "pointwise", "small", "batch_api", "multi", "challenge", "sampler", "ragged", "queue", "wide"])
    raw = rng.random() < 0.3
    x = (rng.integers(-2**31, 2**31, size=(rows, d), dtype=np.int64).astype(np.int32) if raw
         else O.splitmix_centered(int(rng.integers(1, 2**40)), rows * d).reshape(rows, d))
    if what == "ntt":
        # two contexts on two streams, interleaved launches, then compare both
        dx, dy, dz = DB.from_numpy(ctx, x), DB(ctx, x.nbytes), DB(ctx, x.nbytes)
        ex, ey = DB.from_numpy(other, x), DB(other, x.nbytes)
        for _ in range(int(rng.integers(1, 4))):
            ctx.ntt_forward_dev(dx.ptr, dy.ptr, rows)
            other.ntt_inverse_dev(ex.ptr, ey.ptr, rows)
            ctx.ntt_inverse_dev(dy.ptr, dz.ptr, rows)
        f = orc.ntt_forward(x, q, root).reshape(rows, d)
        if other is not ctx and rng.random() < 0.5:
            # ... and a dependency ACROSS the two streams (fz_event_*): the other context inverts what this one has just produced
            ew, ev = DB(other, x.nbytes), fusion_hip.Event(ctx)
            ctx.ntt_forward_dev(dx.ptr, dy.ptr, rows)
            ev.record(ctx)
            ev.wait(other)
            other.ntt_inverse_dev(dy.ptr, ew.ptr, rows)
            assert np.array_equal(ew.to_numpy(np.int32, (rows, d)), orc.ntt_inverse(f, q, inv).reshape(rows, d)), ("event", sp, kern, rows)
            ctx.synchronize()
            ew.free()
            ev.destroy()
        assert np.array_equal(dy.to_numpy(np.int32, (rows, d)), f), ("fwd", sp, kern, rows, raw)
        assert np.array_equal(ey.to_numpy(np.int32, (rows, d)), orc.ntt_inverse(x, q, inv).reshape(rows, d)), ("inv", sp, rows, raw)
        assert np.array_equal(dz.to_numpy(np.int32, (rows, d)), orc.ntt_inverse(f, q, inv).reshape(rows, d)), ("inv(fwd)", sp, kern, rows)
        for b in (dx, dy, dz, ex, ey):
            b.free()
        bump("ntt")
    elif what == "multi":
        # one dispatch over a random list of forward / inverse jobs (some in place, some empty)
        nj = int(rng.integers(1, 45))
        jobs, want, keep = [], [], []
        for j in range(nj):
            r = int(rng.choice([0, 1, 2, 7, 64, 65, int(rng.integers(1, 1500))]))
            inv_j = bool(rng.random() < 0.5)
            xx = rng.integers(-2**31, 2**31, size=(max(r, 1), d), dtype=np.int64).astype(np.int32)[:r]
            a = DB.from_numpy(ctx, xx) if r else DB(ctx, 16)
            b = a if rng.random() < 0.3 else DB(ctx, max(16, r * d * 4))
            keep += [a, b]
            jobs.append((a.ptr, b.ptr, r, inv_j))
            want.append((orc.ntt_inverse(xx, q, inv) if inv_j else orc.ntt_forward(xx, q, root)).reshape(r, d) if r else None)
        ctx.ntt_multi_dev(jobs)
        ctx.synchronize()
        for (a, b, r, inv_j), w_ in zip(jobs, want):
            if r:
                got = np.empty((r, d), np.int32)
                ctx.d2h(got, b)
                assert np.array_equal(got, w_), ("multi", sp, r, inv_j)
        for b in keep:
            b.free()
        bump("multi")
    elif what == "ragged":
        # many aggregates of different sizes in one launch (fz_aggregate_core_ragged / fz_aggregate_target_partial_ragged),
        # raw int32 signature rows, up to 150 aggregates (the group table of a launch holds 64), targets-only form too
        vc = vctx[(sp, int(rng.integers(0, len(VARIANTS))))]
        l = int(rng.choice([1, 3, P["rank"]]))
        G = int(rng.choice([1, 2, 5, 64, 65, int(rng.integers(1, 150))]))
        sizes = [int(v) for v in rng.choice([1, 2, 3, 7, 30], size=G)]
        if rng.random() < 0.3:
            sizes[int(rng.integers(0, G))] = int(rng.integers(100, 400 if l < 20 else 150))
        off = np.concatenate([[0], np.cumsum(sizes)])
        n = int(off[-1])
        sig = rng.integers(-2**31, 2**31, size=(n, l, d), dtype=np.int64).astype(np.int32)
        al, ch, L_, R_ = (rng.integers(-2**31, 2**31, size=(n, d), dtype=np.int64).astype(np.int32) for _ in range(4))
        d_sig, d_al, d_c, d_L, d_R = (DB.from_numpy(vc, a_) for a_ in (sig, al, ch, L_, R_))
        out32, part, tgt, tgt2 = DB(vc, G * l * d * 4), DB(vc, G * l * d * 8), DB(vc, G * d * 8), DB(vc, G * d * 8)
        vc.aggregate_core_ragged_dev(d_sig.ptr, d_al.ptr, off, l, out32.ptr)
        vc.aggregate_target_partial_ragged_dev(d_sig.ptr, d_al.ptr, d_L.ptr, d_R.ptr, d_c.ptr, off, l, part.ptr, l * d, tgt.ptr, d)
        vc.aggregate_target_partial_ragged_dev(0, d_al.ptr, d_L.ptr, d_R.ptr, d_c.ptr, off, l, 0, 0, tgt2.ptr, d)
        got32 = out32.to_numpy(np.int32, (G, l, d))
        got64 = part.to_numpy(np.int64, (G, l, d))
        t1, t2 = tgt.to_numpy(np.int64, (G, d)), tgt2.to_numpy(np.int64, (G, d))
        half = q // 2
        for g_ in sorted(set([0, G - 1] + [int(v) for v in rng.integers(0, G, size=4)])):
            a_, b_ = off[g_], off[g_ + 1]
            want = orc.aggregate_core(sig[a_:b_], al[a_:b_], q)
            assert np.array_equal(got32[g_], want), ("ragged aggregate", sp, l, G, g_)
            assert np.array_equal(((got64[g_] + half) % q - half).astype(np.int32), want), ("ragged partial", sp, l, G, g_)
            t = (L_[a_:b_].astype(object) * ch[a_:b_].astype(object) + R_[a_:b_].astype(object)) * al[a_:b_].astype(object)
            wt = np.array([int(v) % q for v in t.sum(axis=0)], dtype=np.int64)
            assert np.array_equal(t1[g_] % q, wt) and np.array_equal(t2[g_] % q, wt), ("ragged target", sp, G, g_)
        for b in (d_sig, d_al, d_c, d_L, d_R, out32, part, tgt, tgt2):
            b.free()
        bump("ragged")
    elif what == "sampler":
        # device MT19937 sampler == the C clone on the host (itself pinned by CPython's random), any seed / bound / degree
        from fusion_hip import hostpipe
        nn = int(rng.choice([1, 2, 31, 32, 33, 64, 65, int(rng.integers(1, 300))]))
        deg = int(rng.choice([4, 16, 64, 100, 256]))
        bound = int(rng.choice([1, 2, 52, 1000, 2**20 + 7, q // 2]))
        seeds = [int(v) for v in rng.integers(0, 2**63, size=nn, dtype=np.uint64)]
        if rng.random() < 0.5:
            seeds = [v % 2**32 for v in seeds]                                # one-word keys
        do = DB(ctx, nn * 2 * deg * 4)
        ctx.sample_secret_polys_dev(seeds, q, deg, bound, deg, do.ptr)
        assert np.array_equal(do.to_numpy(np.int32, (nn, 2, deg)), hostpipe.sample_secret_polys(seeds, q, deg, bound, deg)), ("sampler", nn, deg, bound)
        do.free()
        bump("sampler")
    elif what == "challenge":
        # device challenge pipeline (text of str(vk), SHAKE-256, decoder, NTT) == host pipeline + oracle NTT
        import fusion.fusion as F
        from fusion_hip import hostpipe
        if "hp_params" not in globals():
            globals()["hp_params"] = {s_: F.fusion_setup(s_, 11 + s_) for s_ in (128, 256)}
        prm = globals()["hp_params"][sp]
        HP = hostpipe.scheme_params(prm)
        nn = int(rng.choice([1, 2, 31, 32, 33, 64, 65, int(rng.integers(1, 400))]))
        vk = rng.integers(-(q // 2), q // 2 + 1, size=(nn, 2, d)).astype(np.int32)
        if rng.random() < 0.3:
            vk[rng.random(size=vk.shape) < 0.3] = 0                         # short decimal texts
        msgs = [f"soak {it} {i} " + "x" * int(rng.integers(0, 200)) for i in range(nn)]
        coefs, pre = hostpipe.challenge_coefficients(HP, vk[:, 0], vk[:, 1], msgs)
        # every form of the device pipeline: as chosen by batch size (a wave per signer here), a lane pair, a lane, a wave forced
        cctx = [ctx] + [vctx[(sp, i)] for i, v in enumerate(VARIANTS) if "FZ_SHAKE_FORM" in v]
        ctx = cctx[int(rng.integers(0, len(cctx)))]
        dv, dc = DB.from_numpy(ctx, vk), DB(ctx, nn * d * 4)
        if rng.random() < 0.5:
            ctx.challenge_dev(HP, dv.ptr, pre, nn, dc.ptr, transform=True)
        else:                                       # the messages hashed on the device as well
            blob, off = hostpipe._pack_messages(msgs)
            got_pre = ctx.challenge_msgs_dev(HP, dv.ptr, blob, off, nn, dc.ptr, want_prehash=True)
            assert np.array_equal(got_pre, pre), ("prehash", sp, nn)
        assert np.array_equal(dc.to_numpy(np.int32, (nn, d)), orc.ntt_forward(coefs, q, root).reshape(nn, d)), ("challenge", sp, nn)
        dv.free()
        dc.free()
        bump("challenge")
    elif what == "polymul":
        g = O.splitmix_centered(int(rng.integers(1, 2**40)), rows * d).reshape(rows, d)
        want = orc.ntt_inverse(orc.pw_mul(orc.ntt_forward(x, q, root), orc.ntt_forward(g, q, root), q), q, inv).reshape(rows, d)
        # both fused forms (the radix-4 kernel, the one on the 16-per-lane transforms) beside the one the batch size chooses
        pctx = [ctx] + [vctx[(sp, i)] for i, v in enumerate(VARIANTS) if "FZ_POLYMUL_FORM" in v]
        ctx = pctx[int(rng.integers(0, len(pctx)))]
        assert np.array_equal(ctx.poly_mul(x, g), want), ("polymul", sp, rows, raw)
        bump("polymul")
    elif what == "graph":
        rows = min(rows, 3000)
        x = x[:rows]
        dx, dy = DB.from_numpy(ctx, x), DB(ctx, x.nbytes)
        ctx.ntt_forward_dev(dx.ptr, dy.ptr, rows)
        ctx.synchronize()
        ctx.graph_begin()
        for _ in range(3):
            ctx.ntt_forward_dev(dx.ptr, dy.ptr, rows)
            ctx.ntt_inverse_dev(dy.ptr, dx.ptr, rows)
        ctx.ntt_forward_dev(dx.ptr, dy.ptr, rows)
        gr = ctx.graph_end()
        for _ in range(3):
            gr.launch()
        want = orc.ntt_forward(orc.ntt_inverse(orc.ntt_forward(x, q, root), q, inv), q, root).reshape(rows, d)
        assert np.array_equal(dy.to_numpy(np.int32, (rows, d)), want), ("graph", sp, kern, rows)
        gr.destroy()
        dx.free()
        dy.free()
        bump("graph")
    elif what == "pointwise":
        # ragged, possibly unaligned element counts through the host-pointer entries (raw int32 operands)
        count = int(rng.choice([1, 3, 4, 5, 255, 1023, int(rng.integers(1, 300000))]))
        a = rng.integers(-2**31, 2**31, size=count, dtype=np.int64).astype(np.int32)
        b = rng.integers(-2**31, 2**31, size=count, dtype=np.int64).astype(np.int32)
        ring = ctxs[(sp, "")]
        assert np.array_equal(ring.pw_mul(a, b), orc.pw_mul(a, b, q)), ("pw_mul", count)
        assert np.array_equal(ring.pw_add(a, b), orc.pw_add(a, b, q)), ("pw_add", count)
        assert np.array_equal(ring.pw_sub(a, b), orc.pw_sub(a, b, q)), ("pw_sub", count)
        assert np.array_equal(ring.pw_neg(a), orc.pw_neg(a, q)), ("pw_neg", count)
        l = int(rng.choice([1, 2, 9, 40, P["rank"]]))
        # few products (one workgroup per product), and every now and then enough of them for the sliced kernel (more than two
        # per CU) with a random slice count
        many = rng.random() < 0.15
        bt = int(rng.integers(520, 1400)) if many else int(rng.integers(1, 30))
        if many:
            l = min(l, 40)
            ctx = vctx[(sp, int(rng.integers(len(VARIANTS))))]
        A = rng.integers(-2**31, 2**31, size=(l, d), dtype=np.int64).astype(np.int32)
        S = rng.integers(-2**31, 2**31, size=(bt, l, d), dtype=np.int64).astype(np.int32)
        assert np.array_equal(ctx.matvec(A, S), orc.matvec(A, S, q)), ("matvec", sp, l, bt)
        mx, wt = ctx.norm_weight(x)
        rmx, rwt = orc.norm_weight(x, q)
        assert np.array_equal(mx, rmx) and np.array_equal(wt, rwt), ("norm_weight", sp, rows)
        bump("pointwise")
    elif what == "small":
        # other parameter sets: small primes / degrees (thread-per-polynomial kernels, general 6-op multiply)
        dd = int(rng.choice([2, 4, 8, 16, 32, 64, 128]))
        cands = [pp for pp in (257, 769, 12289, 40961, 65537, 786433, 2013265921) if (pp - 1) % (2 * dd) == 0]
        qq = int(rng.choice(cands))
        rt = next(r_ for r_ in (pow(g_, (qq - 1) // (2 * dd), qq) for g_ in range(2, 500)) if pow(r_, dd, qq) == qq - 1)
        irt = pow(rt, qq - 2, qq)
        cs = fusion_hip.Context(qq, dd, rt, irt)
        rr = int(rng.integers(1, 3000))
        xx = rng.integers(-2**31, 2**31, size=(rr, dd), dtype=np.int64).astype(np.int32)
        f = cs.ntt_forward(xx)
        assert np.array_equal(f, orc.ntt_forward(xx, qq, rt).reshape(rr, dd)), ("small fwd", qq, dd, rr)
        assert np.array_equal(cs.ntt_inverse(xx), orc.ntt_inverse(xx, qq, irt).reshape(rr, dd)), ("small inv", qq, dd, rr)
        yy = rng.integers(-2**31, 2**31, size=(rr, dd), dtype=np.int64).astype(np.int32)
        want = orc.ntt_inverse(orc.pw_mul(f, orc.ntt_forward(yy, qq, rt), qq), qq, irt).reshape(rr, dd)
        assert np.array_equal(cs.poly_mul(xx, yy), want), ("small polymul", qq, dd, rr)
        cs.close()
        bump("small")
    elif what == "queue":
        # the asynchronous batch queue under mixed load: keygen + sign calls, aggregate + verify calls (signatures as host rows, as
        # the queue's own device rows) and verify calls (one of them tampered) pending at once on 1-3 workers; every result against
        # BatchScheme on the same inputs; rows released behind an asynchronous consumer (release(after=))
        import fusion.fusion as F
        from fusion_hip import scheme as SCH
        from fusion_hip.queue import BatchQueue
        if "params" not in globals():
            globals()["params"] = {s_: F.fusion_setup(s_, 2026 + s_) for s_ in (128, 256)}
            globals()["bsch"] = {s_: SCH.BatchScheme(globals()["params"][s_]) for s_ in (128, 256)}
        prm, bs = globals()["params"][sp], globals()["bsch"][sp]
        l_ = prm.num_rows_sk
        ncalls = int(rng.integers(2, 7))
        sizes = [int(v) for v in rng.choice([1, 2, 3, 9, 40, 130], size=ncalls)]
        with BatchQueue(prm, workers=int(rng.integers(1, 4)), max_rows=int(rng.choice([130, 256, 4096])), host_threads=int(rng.integers(1, 5))) as bq:
            calls = []
            for n_ in sizes:
                seeds = [int(v) for v in rng.integers(1, 2**62, size=n_)]
                msgs = [f"soak {it} q {i} " + "y" * int(rng.integers(0, 60)) for i in range(n_)]
                calls.append((seeds, msgs, bq.submit_keygen_sign(seeds, msgs)))
            res, want, agg_t = [], [], []
            for seeds, msgs, t in calls:
                r = bq.wait(t)
                sk_b, vk_b = bs.keygen_batch(seeds)
                sig_b = bs.sign_batch(sk_b, vk_b, msgs)
                assert np.array_equal(r.vk, vk_b) and np.array_equal(r.signatures(), sig_b), ("queue keygen+sign", sp, len(seeds))
                rows_ = sig_b if rng.random() < 0.5 else r.sig_ptr
                agg_t.append(bq.submit_aggregate_verify(vk_b, msgs, rows_))
                res.append((r, vk_b, msgs, sig_b))
            ver_t = []
            for (r, vk_b, msgs, sig_b), t in zip(res, agg_t):
                agg, verdict = bq.wait_aggregate(t)
                w_ = bs.aggregate(vk_b, msgs, sig_b)
                assert np.array_equal(agg, w_) and verdict == (True, ""), ("queue aggregate", sp, len(msgs))
                bad = agg.copy()
                tamper = rng.random() < 0.4
                if tamper:
                    bad[int(rng.integers(0, l_)), int(rng.integers(0, d))] += 1
                ver_t.append((bq.submit_verify(vk_b, msgs, bad), tamper))
                if rng.random() < 0.5:                      # a consumer still reading the rows when they are released
                    d_al = DB.from_numpy(bs.ctx, np.ones((len(msgs), d), np.int32))
                    d_o = DB(bs.ctx, l_ * d * 4)
                    bs.ctx.aggregate_core_dev(r.sig_ptr, d_al.ptr, d_o.ptr, len(msgs), l_)
                    r.release(after=bs.ctx)
                    got = d_o.to_numpy(np.int32, (l_, d))
                    assert np.array_equal(got, orc.aggregate_core(sig_b, np.ones((len(msgs), d), np.int32), q)), ("release after", sp)
                    d_al.free()
                    d_o.free()
                else:
                    r.release()
            for t, tamper in ver_t:
                assert bq.wait_verdict(t) == ((False, "Target doesn't match image of aggregate signature.") if tamper else (True, "")), ("queue verify", sp, tamper)
        bump("queue")
    elif what == "batch_api":
        # array API == drop-in object API (which the GPU test-suite pins to the reference's golden strings)
        import fusion.fusion as F
        from fusion_hip import scheme as SCH
        if "params" not in globals():
            globals()["params"] = {s_: F.fusion_setup(s_, 2026 + s_) for s_ in (128, 256)}
            globals()["bsch"] = {s_: SCH.BatchScheme(globals()["params"][s_]) for s_ in (128, 256)}
        prm, bs = globals()["params"][sp], globals()["bsch"][sp]
        nn = int(rng.integers(1, 4))
        seeds = [int(v) for v in rng.integers(1, 2**31, size=nn)]
        msgs = [f"soak {it} {i}" for i in range(nn)]
        keys = [F.keygen(prm, s_) for s_ in seeds]
        sigs = [F.sign(prm, k_, m_) for k_, m_ in zip(keys, msgs)]
        agg = F.aggregate(prm, [k_[1] for k_ in keys], msgs, sigs)
        sk_b, vk_b = bs.keygen_batch(seeds)
        sig_b = bs.sign_batch(sk_b, vk_b, msgs)
        agg_b = bs.aggregate(vk_b, msgs, sig_b)
        assert np.array_equal(sig_b, np.stack([SCH.signature_from_object(prm, s_) for s_ in sigs])), ("batch sign", sp, nn)
        assert np.array_equal(agg_b, SCH.signature_from_object(prm, agg)), ("batch aggregate", sp, nn)
        assert bs.verify(vk_b, msgs, agg_b) == F.verify(prm, [k_[1] for k_ in keys], msgs, agg) == (True, ""), ("batch verify", sp)
        bump("batch_api")
    elif what == "wide":
        # the generic int64 path (csrc/fz_wide.hip): a random odd modulus of 33 .. 63 bits, random tables, against the reference's
        # loops and Python-integer arithmetic
        from fusion_hip.wide import WideContext
        bits = int(rng.integers(33, 64))
        qw = (int(rng.integers(1 << 62, (1 << 63) - 1)) >> (63 - bits)) | (1 << (bits - 1)) | 1
        dw = 1 << int(rng.integers(1, 11))
        nb = int(rng.integers(1, 6))
        hw = (qw - 1) // 2
        pyr = lambda n_: [int(rng.integers(-hw, hw + 1)) for _ in range(n_)]
        tab, itab = [int(rng.integers(0, qw)) for _ in range(dw)], [int(rng.integers(0, qw)) for _ in range(dw)]
        if qw % 2 == 1 and np.gcd(dw, qw) == 1:
            wc = WideContext(qw, dw, tab, itab)
            xs = [pyr(dw) for _ in range(nb)]
            X = np.array(xs, dtype=np.int64)
            yf, yi = wc.ntt_forward(X), wc.ntt_inverse(X)
            cw = lambda v: (v + hw) % qw - hw
            for b_ in range(nb):
                assert yf[b_].tolist() == O.py_ntt_forward(list(xs[b_]), qw, tab), ("wide fwd", qw, dw)
                # py_ntt_inverse computes n^-1 as pow(n, q - 2, q): only right for primes -- the loop is restated with pow(n, -1, q)
                v = list(xs[b_])
                t_, m_ = 1, dw
                while m_ > 1:
                    j1, h_ = 0, m_ // 2
                    for i_ in range(h_):
                        s_ = itab[h_ + i_]
                        for j_ in range(j1, j1 + t_):
                            u_, w_ = v[j_], v[j_ + t_]
                            v[j_], v[j_ + t_] = cw(u_ + w_), cw((u_ - w_) * s_)
                        j1 += 2 * t_
                    t_, m_ = 2 * t_, h_
                ninv = pow(dw, -1, qw)
                assert yi[b_].tolist() == [cw(c_ * ninv) for c_ in v], ("wide inv", qw, dw)
            a_, b2 = pyr(dw * nb), pyr(dw * nb)
            A_, B_ = np.array(a_, dtype=np.int64), np.array(b2, dtype=np.int64)
            assert wc.pw_mul(A_, B_).tolist() == [cw(x_ * y_) for x_, y_ in zip(a_, b2)], ("wide mul", qw)
            assert wc.pw_add(A_, B_).tolist() == [cw(x_ + y_) for x_, y_ in zip(a_, b2)] and wc.pw_neg(A_).tolist() == [-(x_ % qw) for x_ in a_]
        bump("wide")
    else:
        ctx = vctx[(sp, int(rng.integers(0, len(VARIANTS))))]          # fused / multi-launch forms, aggregation launch shapes
        l = int(rng.choice([1, 2, 7, P["rank"]]))
        n = int(rng.integers(1, 40)) if rng.random() < 0.8 else int(rng.integers(40, 700 if l < 20 else 160))
        G = int(rng.integers(1, 4))
        A = O.splitmix_centered(int(rng.integers(1, 2**40)), l * d).reshape(l, d)
        coef = rng.integers(-52, 53, size=(G * n, 2, l, d)).astype(np.int32)
        sk, vk = ctx.keygen_core(A, coef)
        rsk, rvk = orc.keygen_core(A, coef, q, root)
        assert np.array_equal(sk, rsk) and np.array_equal(vk, rvk), ("keygen", sp, l, n)
        c = np.zeros((G * n, d), np.int32)
        for i in range(G * n):
            c[i, rng.choice(d, min(d, 60), replace=False)] = rng.choice([-1, 1], min(d, 60))
        c_hat = ctx.ntt_forward(c)
        al_hat = ctx.ntt_forward(np.roll(c, 5, axis=1))
        sig = ctx.sign_core(sk, c_hat)
        assert np.array_equal(sig, orc.sign_core(sk, c_hat, q)), ("sign", sp, l, n)
        d_sig, d_al, d_c = DB.from_numpy(ctx, sig), DB.from_numpy(ctx, al_hat), DB.from_numpy(ctx, c_hat)
        d_L, d_R = DB.from_numpy(ctx, np.ascontiguousarray(vk[:, 0])), DB.from_numpy(ctx, np.ascontiguousarray(vk[:, 1]))
        d_A = DB.from_numpy(ctx, A)
        n_agg, n_t = G * l * d, G * d
        part = DB(ctx, (n_agg + n_t) * 8)
        verd = DB(ctx, G * 4)
        tamper = int(rng.integers(-1, G))                # -1: none
        for rep in range(3):                             # repeated launches: the verify accumulators must re-arm
            ctx.aggregate_target_partial_batch_dev(d_sig.ptr, d_al.ptr, d_L.ptr, d_R.ptr, d_c.ptr, part.ptr, l * d,
                                                   part.ptr + n_agg * 8, d, G, n, l)
            if tamper >= 0 and rep == 1:
                p = part.to_numpy(np.int64, (n_agg + n_t,))
                p[tamper * l * d + int(rng.integers(0, l * d))] += 1
                ctx.h2d(part.ptr, p)
            ctx.verify_partials_batch_async_dev(d_A.ptr, part.ptr, l * d, part.ptr + n_agg * 8, d, G, l, P["beta_vf"], d, verd.ptr)
            got = verd.to_numpy(np.int32, (G,)).tolist()
            want = [3 if (g_ == tamper and rep == 1) else 0 for g_ in range(G)]
            assert got == want, ("verify", sp, l, n, G, rep, got, want)
        p = part.to_numpy(np.int64, (n_agg + n_t,))
        half = q // 2
        agg = ((p[:n_agg] + half) % q - half).astype(np.int32).reshape(G, l, d)
        for g_ in range(G):
            assert np.array_equal(agg[g_], orc.aggregate_core(sig[g_ * n:(g_ + 1) * n], al_hat[g_ * n:(g_ + 1) * n], q)), ("agg", g_)
        # signing + aggregation + target sums in ONE pass: the same signatures and the same sums (as residues)
        d_sk, d_sig2, part2 = DB.from_numpy(ctx, sk), DB(ctx, sig.nbytes), DB(ctx, (n_agg + n_t) * 8)
        ctx.sign_aggregate_target_partial_batch_dev(d_sk.ptr, d_c.ptr, d_al.ptr, d_L.ptr, d_R.ptr, d_sig2.ptr, part2.ptr, l * d,
                                                    part2.ptr + n_agg * 8, d, G, n, l)
        assert np.array_equal(d_sig2.to_numpy(np.int32, sig.shape), sig), ("sign+aggregate: signatures", sp, l, n, G)
        p2 = part2.to_numpy(np.int64, (n_agg + n_t,))
        if tamper < 0:
            assert np.array_equal((p2 + half) % q, (p + half) % q), ("sign+aggregate: sums", sp, l, n, G)
        for b in (d_sk, d_sig2, part2):
            b.free()
        for b in (d_sig, d_al, d_c, d_L, d_R, d_A, part, verd):
            b.free()
        bump("scheme")
    if time.time() > t_print:
        print(f"[soak] {it} iterations ok: {counts}", flush=True)
        t_print = time.time() + 60
print(f"[soak] PASSED {it} iterations in {budget:.0f} s: {counts}", flush=True)
