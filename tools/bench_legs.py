"""Side legs of bench.py (the headline, its roofline fraction and the compact line live in bench.py itself).

Every function takes the `env` namespace bench.py builds (context, stream, rotating batches, rank / world helpers) and
returns a JSON-serialisable dict (None on ranks that sit a rank-0 leg out).  `sign_verify` runs by default -- it is the
second half of BASELINE's metric; the FULL_LEGS run under `bench.py --full` and only reach gpurun_out/bench_full.json.
Nothing here touches oracle/ (tests/test_cabi_symbols.py checks bench.py AND this file)."""
import ctypes
import sys
import time

FULL_LEGS = ["copy_floor", "multi_job", "pcie_inclusive", "sweep", "kernels", "end_to_end", "end_to_end_sharded"]


# ---- sign + aggregate + verify (algebra cores; synthetic keys/messages) -- every rank ----------------------------------
def sign_verify(e):
    """per rank: 1024 signatures in GROUPS aggregates of (1024 / GROUPS) x world signers; 8 operand sets (2.1 GB of keys and
    signatures) rotate so that nothing is cache-resident.  A step = signing + the aggregates' and the verification targets'
    int64 partial sums in ONE launch (fz_sign_aggregate_target_partial_batch; --sign-then-aggregate: round 3's two), the ONE
    int64 all-reduce (RCCL through the C ABI: fz_allreduce_i64, on a second, high-priority stream beside the next step's
    kernels), verification from the int64 sums (one launch for the aggregates of all 8 steps without a communicator, one per
    step behind the next step's kernels with one); the 8 steps are one graph replay.  At least as many aggregates as ranks, so
    that EVERY rank verifies.
    Reference arithmetic: fusion/fusion.py:557 (sign), :670-676 (aggregate), :690-727 (verify), :363-370 (keygen)."""
    torch, np, ctx, dist, dev = e.torch, e.np, e.ctx, e.dist, e.dev
    rank, world, l, d, P, args = e.rank, e.world, e.l, e.d, e.P, e.args
    comm, collective = e.comm, e.collective
    S, NSETS = 1024, 8
    rng = np.random.default_rng(1234 + rank)
    A = torch.empty((l, d), dtype=torch.int32, device=dev)                           # same on every rank
    ctx.fill_synthetic_dev(A.data_ptr(), l * d, 99)
    coef0 = torch.from_numpy(rng.integers(1, 53, size=(S, 2, l, d)).astype(np.int32) *
                             rng.choice(np.array([-1, 1], dtype=np.int32), size=(S, 2, l, d))).to(dev)

    def sparse(weight):
        c = np.zeros((S, d), np.int32)
        for i in range(S):
            c[i, rng.choice(d, weight, replace=False)] = rng.choice([-1, 1], weight)
        return torch.from_numpy(c).to(dev)
    cc0, aa0 = sparse(P["omega_ch"]), sparse(P["omega_ag"])
    # NSETS distinct operand sets: set i = the coefficients rotated by i positions.  A step works on ONE set, consecutive
    # steps on consecutive sets, so no step finds its keys or signatures in the 256 MB Infinity Cache.
    sets = []
    for i in range(NSETS):
        coef = torch.roll(coef0, shifts=i, dims=3).contiguous()
        sk_hat = torch.empty_like(coef)
        vk = torch.empty((S, 2, d), dtype=torch.int32, device=dev)
        ctx.keygen_core_dev(A.data_ptr(), coef.data_ptr(), sk_hat.data_ptr(), vk.data_ptr(), S, l)
        c_hat = torch.empty((S, d), dtype=torch.int32, device=dev)
        al_hat = torch.empty((S, d), dtype=torch.int32, device=dev)
        cc, aa = torch.roll(cc0, shifts=i, dims=1).contiguous(), torch.roll(aa0, shifts=3 * i + 1, dims=1).contiguous()
        ctx.ntt_forward_dev(cc.data_ptr(), c_hat.data_ptr(), S)
        ctx.ntt_forward_dev(aa.data_ptr(), al_hat.data_ptr(), S)
        torch.cuda.synchronize(dev)
        sets.append(dict(coef=coef, sk_hat=sk_hat, vk=vk, c_hat=c_hat, al_hat=al_hat, vkL=vk[:, 0].contiguous(),
                         vkR=vk[:, 1].contiguous(), sig=torch.empty((S, l, d), dtype=torch.int32, device=dev)))
    del coef0, cc0, aa0
    def measure(GROUPS):
        """the leg for one split of a rank's 1024 signatures into GROUPS aggregates"""
        per = S // GROUPS
        # int64 sums as RECORDS: [set][aggregate][l*d sums of the aggregate | d sums of its verification target] -- one all-reduce
        # per step covers a set's GROUPS records, and the records of all sets are uniformly strided, so ONE verification launch can
        # take every aggregate of the 8 steps (below)
        standin = args.exchange_standin_us if world == 1 else 0           # (a delay of known length in place of the all-reduce)
        overlap = (comm is not None or standin > 0) and not args.no_exchange_overlap
        # How many steps share ONE verification launch (a verification of 4-8 aggregates is a latency chain of 5 us on 4-8
        # workgroups; 32 of them are 7 us): all 8 when the exchange does not run on a second stream (68.7 instead of 72.5 us per
        # step with two launches per step).  With the overlap a captured graph whose exchange branch joins the compute branch only
        # rarely loses the branch's stream priority -- with a one-rank RCCL all-reduce + a 40 us stand-in per step: one launch per
        # 1 / 2 / 4 / 8 steps 69.8 / 93.5 / 99.9 / 91.7 us per step -- so there the default is one step per launch; a block of
        # steps is verified behind the NEXT step's kernels (its sums have had a step's time to arrive).  With more than one step per launch rank r verifies every aggregate of the
        # sets r, r + world, ...; --verify-per-step (= --verify-every 1): a launch per step over this rank's share of the
        # step's aggregates (round 3's form).
        vk = 1 if args.verify_per_step else (args.verify_every if args.verify_every > 0 else (1 if overlap else NSETS))
        vk = max(1, min(vk, NSETS))
        if world > NSETS:
            vk = 1
        batched = vk > 1
        rec = l * d + d
        nbuf = NSETS if (overlap or batched) else 1
        pool = torch.zeros(nbuf * GROUPS * rec, dtype=torch.int64, device=dev)
        g_lo, g_hi = e.shard_range(GROUPS, rank, world)      # per-step form: aggregates of every step verified by this rank
        my_sets = [s_ for s_ in range(NSETS) if s_ % world == rank] if batched else list(range(NSETS))
        d_verd = torch.full((NSETS * GROUPS,), -1, dtype=torch.int32, device=dev)        # verdict codes, read after the loops
        torch.cuda.synchronize(dev)                       # the fills ran on torch's stream; the kernels below run on the context's
        # The exchange step on a stream of its own (overlap): a second context issues fz_allreduce_i64 there behind an event per
        # step ("partials written", compute -> exchange), so the all-reduce of step i (a latency of tens of microseconds at 2-8
        # ranks, 0.7-1.4 MB) runs beside sign_core + the partial sums of step i + 1 instead of stalling the chip; the compute stream
        # waits for the sums where it verifies (once per 8 steps, or per step one step late with --verify-per-step).  Same
        # launches per signature, same results; round 3's form (everything on one stream): --no-exchange-overlap.
        # The exchange stream has HIGH priority (bench.py): at normal priority the all-reduce kernel's workgroups wait for slots behind
        # sign_core's 1024 -- 154 instead of 85 us per step with a one-rank RCCL all-reduce + a 40 us stand-in.
        # (Measured and dropped: the verification on the exchange stream as well -- beside sign_core, which saturates the memory
        # system, the verification's latency chain takes 4-10 times as long: 127-131 instead of 73-85 us per step, at either priority.)
        cx = ev_part = ev_sum = None
        if overlap:
            cx = e.exchange_ctx           # created with its stream at the start of the process (bench.py: hardware queues go to streams in order)
            ev_part = [e.fusion_hip.Event(ctx) for _ in range(NSETS)]
            ev_sum = [e.fusion_hip.Event(ctx) for _ in range(NSETS)]

        # Which collective: every rank verifies only ITS share of a step's aggregates, so a reduce-scatter (fz_reduce_scatter_i64:
        # block r of the sums to rank r, half the traffic and steps of the all-reduce) is enough whenever the aggregates divide
        # evenly over the ranks and verification is per step.  Chosen by a short calibration on the exchange stream -- 20 launches of
        # each, every rank, max over ranks -- unless --exchange says so; the all-reduce stays unless the other form is >= 20 % faster.
        use_rs, exch_cal = False, None
        rs_ok = comm is not None and not batched and GROUPS % world == 0 and (g_hi - g_lo) * world == GROUPS
        if args.exchange == "reduce-scatter" and not rs_ok:
            raise RuntimeError("--exchange reduce-scatter needs a communicator, one verification launch per step and aggregates that divide over the ranks")
        if rs_ok and args.exchange in ("auto", "reduce-scatter"):
            xc = cx if overlap else ctx
            cal = torch.zeros(GROUPS * rec, dtype=torch.int64, device=dev)
            torch.cuda.synchronize(dev)
            times = {}
            try:
                for name_, fn_ in (("all-reduce", lambda: xc.allreduce_i64_dev(comm, cal.data_ptr(), cal.numel())),
                                   ("reduce-scatter", lambda: xc.reduce_scatter_i64_dev(comm, cal.data_ptr(), (GROUPS // world) * rec))):
                    for _ in range(5):
                        fn_()
                    xc.synchronize()
                    e.barrier()
                    t0_ = time.perf_counter()
                    for _ in range(20):
                        fn_()
                    xc.synchronize()
                    times[name_] = e.max_over_ranks(time.perf_counter() - t0_) / 20 * 1e6
                ok_ = 1.0
            except e.fusion_hip.FusionHipError as exc:
                sys.stderr.write(f"rank {rank}: exchange calibration failed: {exc}\n")
                ok_ = 0.0
            if e.min_over_ranks(ok_) >= 1.0:
                exch_cal = {k_: round(v_, 2) for k_, v_ in times.items()}
                use_rs = args.exchange == "reduce-scatter" or times["reduce-scatter"] < 0.8 * times["all-reduce"]
            del cal

        def part_of(i):
            base = (i % nbuf) * GROUPS * rec
            return pool[base:base + GROUPS * rec]

        two_launch = args.sign_then_aggregate

        def sv_compute(i):
            s_, part = sets[i % NSETS], part_of(i)
            if two_launch:               # round 3's form: the signatures are written, then read back by the aggregation
                ctx.sign_core_dev(s_["sk_hat"].data_ptr(), s_["c_hat"].data_ptr(), s_["sig"].data_ptr(), S, l)
                # aggregate partials and the verification target's partials: one pass over this rank's signers, one launch
                ctx.aggregate_target_partial_batch_dev(s_["sig"].data_ptr(), s_["al_hat"].data_ptr(), s_["vkL"].data_ptr(),
                                                       s_["vkR"].data_ptr(), s_["c_hat"].data_ptr(), part.data_ptr(), rec,
                                                       part[l * d:].data_ptr(), rec, GROUPS, per, l)
                return
            # ONE pass: every signature is written as it is computed and enters its aggregate's sums (and the key pair the target's)
            # from registers -- fz_sign_aggregate_target_partial_batch
            ctx.sign_aggregate_target_partial_batch_dev(s_["sk_hat"].data_ptr(), s_["c_hat"].data_ptr(), s_["al_hat"].data_ptr(),
                                                        s_["vkL"].data_ptr(), s_["vkR"].data_ptr(), s_["sig"].data_ptr(), part.data_ptr(), rec,
                                                        part[l * d:].data_ptr(), rec, GROUPS, per, l)

        def sv_exchange(i):              # the ONE exchange step (RCCL over xGMI)
            part = part_of(i)
            if overlap:
                ev_part[i % NSETS].record(ctx)
                ev_part[i % NSETS].wait(cx)
                if comm is not None and use_rs:
                    cx.reduce_scatter_i64_dev(comm, part.data_ptr(), (GROUPS // world) * rec)
                elif comm is not None:
                    cx.allreduce_i64_dev(comm, part.data_ptr(), part.numel())
                if standin:
                    cx.diag_delay(standin)
                ev_sum[i % NSETS].record(cx)
            elif comm is not None or standin:
                if comm is not None and use_rs:
                    ctx.reduce_scatter_i64_dev(comm, part.data_ptr(), (GROUPS // world) * rec)
                elif comm is not None:
                    ctx.allreduce_i64_dev(comm, part.data_ptr(), part.numel())
                if standin:
                    ctx.diag_delay(standin)
            else:
                e.allreduce_sum_i64(part)

        def verify_records(first, count):      # verdicts straight from the int64 sums, left on the device: no host synchronisation
            if count > 0:
                ctx.verify_partials_batch_async_dev(A.data_ptr(), pool[first * rec:].data_ptr(), rec, pool[first * rec + l * d:].data_ptr(), rec,
                                                    count, l, P["beta_vf"], d, d_verd[first:].data_ptr())

        def sv_verify(lo, hi):           # the steps lo .. hi - 1 (their exchanges have been issued)
            if overlap:
                ev_sum[(hi - 1) % NSETS].wait(ctx)      # the exchange stream is in order: the last sums of the block arrive last
            if not batched:
                verify_records((lo % nbuf) * GROUPS + g_lo, g_hi - g_lo)
            elif world == 1:
                verify_records(lo * GROUPS, (hi - lo) * GROUPS)
            else:
                for s_ in range(lo, hi):
                    if s_ % world == rank:
                        verify_records(s_ * GROUPS, GROUPS)

        def sv_steps_once():             # NSETS steps
            done_ = 0
            for i in range(NSETS):
                sv_compute(i)
                sv_exchange(i)
                ready = i if overlap else i + 1          # with the overlap a block is verified behind the NEXT step's kernels
                if ready - done_ >= vk:
                    sv_verify(done_, ready)
                    done_ = ready
            if done_ < NSETS:
                sv_verify(done_, NSETS)

        def my_records():
            if batched:
                return [s_ * GROUPS + g_ for s_ in my_sets for g_ in range(GROUPS)]
            return [b_ * GROUPS + g_ for b_ in range(nbuf) for g_ in range(g_lo, g_hi)]

        def verdicts_ok():
            v = d_verd.tolist()
            return all(v[k] == 0 for k in my_records())

        sv_steps_once()
        e.barrier()
        torch.cuda.synchronize(dev)
        assert verdicts_ok(), f"verify verdicts {d_verd.tolist()}"
        # one graph = NSETS steps (every set once); refused together if any rank cannot capture (e.g. the collective)
        sv_graph, captured = None, 0.0
        if not args.no_graph and (world == 1 or comm is not None):      # (a stand-in delay is a kernel: capturable)
            try:
                ctx.graph_begin()
                try:
                    sv_steps_once()
                finally:
                    sv_graph = ctx.graph_end()
                captured = 1.0
            except e.fusion_hip.FusionHipError as exc:
                sys.stderr.write(f"rank {rank}: sign_verify capture failed: {exc}\n")
                sv_graph = None
        if e.min_over_ranks(captured) < 1.0:
            sv_graph = None

        def sv_round():
            if sv_graph is not None:
                sv_graph.launch()
            else:
                sv_steps_once()
        for _ in range(2):
            sv_round()
        for _ in range(40 if args.prewarm_ms > 0 else 0):      # count-based: every rank must issue the same collectives
            sv_round()
        e.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        sv_round()
        torch.cuda.synchronize(dev)
        t_once = max(time.perf_counter() - t0, 1e-6)
        sv_rounds = int(e.max_over_ranks(max(3.0, -(-2 * e.MIN_REGION_MS * 1e-3 // t_once))))
        e.barrier()
        t0 = time.perf_counter()
        for _ in range(sv_rounds):
            sv_round()
        e.barrier()
        dt = e.max_over_ranks(time.perf_counter() - t0)
        torch.cuda.synchronize(dev)
        assert verdicts_ok(), "a verification failed inside the timed region"
        sv_steps = sv_rounds * NSETS
        sv_bytes = S * ((3 * l + 1) + (l + 5)) * 4 * d
        sv = {"value": S * world * sv_steps / dt, "unit": "signatures signed+aggregated+verified per s",
              "signatures_per_rank": S, "aggregates": GROUPS, "signers_per_aggregate": per * world,
              "steps": sv_steps, "ms_per_step": dt / sv_steps * 1e3, "operand_sets_cycled": NSETS,
              # BASELINE configs[3] taken literally (ONE aggregate of 1024 signers, 128 per GPU at N = 8) cannot scale: 5.7 us of
              # aggregation + an all-reduce + 4.6 us of verification against 17 us on one GPU.  The leg keeps the work per GPU
              # fixed instead: every rank holds S signatures of each of the GROUPS aggregates, an aggregate spans all ranks
              "scaling": (f"weak: {GROUPS} aggregates of {per} x {world} signers, {S} signatures per GPU" if world > 1 else
                          f"weak (single rank: {GROUPS} aggregates of {per} signers)"),
              "launch": "hipGraph replay of %d steps (fz_graph_*)" % NSETS if sv_graph is not None else "one by one",
              "collective": collective,
              "exchange": ("on a second stream, overlapping the next step's kernels (fz_event_*)" if overlap else
                           "on the compute stream" if (comm is not None or standin) else "none" if world == 1 else "torch.distributed, host-ordered"),
              "exchange_standin_us": standin or None,
              "exchange_collective": ("ncclReduceScatter (fz_reduce_scatter_i64)" if use_rs else "ncclAllReduce (fz_allreduce_i64)") if comm is not None else None,
              "exchange_calibration_us": exch_cal,
              "verification": f"one launch per {vk} steps" if batched else "one launch per step",
              "sign_and_aggregate": "two launches (sign_core, then aggregate + target partials)" if two_launch else
                                    "one launch (fz_sign_aggregate_target_partial_batch: signatures written and aggregated in one pass)",
              # SURVEY 8d's bytes of the TWO kernels ((3l + 1) + (l + 5) rows per signature), for reference only: the fraction of the
              # peak is quoted on what the launches MOVE (moved_frac_per_gpu) -- round 4's hbm_frac_per_gpu counted bytes the fused
              # launch never moves and is gone
              "survey_8d_GB/s_per_gpu": sv_bytes * sv_steps / dt / 1e9,
              # what the launches actually move per signature: in one pass the l rows of a signature are written and never read
              # back ((3l + 4) rows instead of SURVEY 8d's (3l + 1) + (l + 5) for the two kernels the algorithmic figure counts)
              "moved_frac_per_gpu": S * ((3 * l + 4) if not two_launch else ((3 * l + 1) + (l + 5))) * 4 * d * sv_steps / dt / 1e9 / e.HBM_PEAK_GBS,
              "note": "algebra cores only: signing + aggregate and target partials (ONE pass, one launch: fz_sign_aggregate_target_partial_batch), int64 all-reduce, "
                      "verification from the int64 sums (ONE launch for the aggregates of all 8 steps unless --verify-per-step); every step works on "
                      "the next of 8 operand sets (2.1 GB), so keys and signatures come from HBM; host hashing of str(vk) excluded"}
        sv_graph_used = sv_graph is not None
        if sv_graph is not None:
            sv_graph.destroy()
        if overlap:
            ctx.synchronize()
            cx.synchronize()
            for ev in ev_part + ev_sum:
                ev.destroy()
        return sv, sv_graph_used, len(my_records())

    GROUPS = max(4, world)
    while S % GROUPS:
        GROUPS += 1
    per = S // GROUPS
    sv, sv_graph_used, n_verified = measure(GROUPS)
    if world == 1 and comm is None and not args.exchange_standin_us:
        # BASELINE configs[3]'s own shape on one GPU: ONE aggregate over all 1024 signers (the default above splits
        # the rank's 1024 signatures into four aggregates of 256 -- at N ranks: N aggregates of 128 x N signers, every rank verifying)
        one, _, _ = measure(1)
        sv["one_aggregate"] = {k_: one[k_] for k_ in ("value", "unit", "ms_per_step", "moved_frac_per_gpu", "aggregates", "signers_per_aggregate",
                                                       "steps", "verification", "sign_and_aggregate")}
    # BASELINE configs[2]: 1024 independent keygen + sign per step (keygen_core: 2*l transforms + two A.s products per
    # key; sign_core: sigma = L*c + R), no exchange step: ranks are independent
    def ks_step(i):
        s_ = sets[i % NSETS]
        ctx.keygen_core_dev(A.data_ptr(), s_["coef"].data_ptr(), s_["sk_hat"].data_ptr(), s_["vk"].data_ptr(), S, l)
        ctx.sign_core_dev(s_["sk_hat"].data_ptr(), s_["c_hat"].data_ptr(), s_["sig"].data_ptr(), S, l)
    for i in range(100 if args.prewarm_ms > 0 else 2):
        ks_step(i)
    e.barrier()
    ks_steps = NSETS * max(3, int(2 * e.MIN_REGION_MS / (NSETS * 0.15)) + 1)
    t0 = time.perf_counter()
    for i in range(ks_steps):
        ks_step(i)
    e.barrier()
    dt = e.max_over_ranks(time.perf_counter() - t0)
    ks_bytes = S * ((4 * l + 2) + (3 * l + 1)) * 4 * d
    sv["keygen_sign"] = {"value": S * world * ks_steps / dt, "unit": "keygen+sign per s", "per_rank": S, "steps": ks_steps,
                         "ms_per_step": dt / ks_steps * 1e3, "operand_sets_cycled": NSETS,
                         "algorithmic_GB/s_per_gpu": ks_bytes * ks_steps / dt / 1e9,
                         "hbm_frac_per_gpu": ks_bytes * ks_steps / dt / 1e9 / e.HBM_PEAK_GBS,
                         "note": "configs[2]: keygen_core + sign_core on 1024 distinct synthetic keys per rank; sign reads the "
                                 "sk_hat keygen has just written (174 MB: part of it may still sit in the Infinity Cache), "
                                 "coefficients come from HBM (8 sets rotated)"}
    if world > 1:            # what EVERY rank did in this leg, as the ranks themselves report it
        mine = {"rank": rank, "collective": collective, "aggregates_verified": n_verified, "verdicts_ok": True,
                "graph": sv_graph_used}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        sv["ranks"] = allr
        sv["every_rank_verified"] = all(r_["aggregates_verified"] > 0 for r_ in allr)
        sv["one_collective_path"] = len({r_["collective"] for r_ in allr}) == 1
    del sets
    return sv


# ---- the launch floor, same run: an empty dispatch and a plain copy of the bytes one launch moves -- rank 0 ------------
def copy_floor(e):
    if e.rank != 0:
        return None
    ctx, x, y = e.ctx, e.x, e.y
    e.prewarm(lambda: ctx.diag_empty_launch(), 20)
    t_empty = e.timed_on_stream(lambda: ctx.diag_empty_launch(), 400)
    k = [0]

    def cp():                                            # the rotation's own batches: cold reads like the headline's
        i = k[0] % len(e.rot_p)
        k[0] += 1
        ctx.diag_copy_dev(e.xs[i].data_ptr(), e.ys[i].data_ptr(), x.numel() * 4)
    e.prewarm(cp, 20)
    t_copy = e.timed_on_stream(cp, 400)
    fb = 8.0 * e.d * e.B
    return {"empty_dispatch_us": t_empty * 1e3, "copy_us": t_copy * 1e3, "copy_bytes": fb,
            "copy_frac": fb / (t_copy * 1e-3) / 1e9 / e.HBM_PEAK_GBS,
            "what": "back-to-back launches on the kernels' stream, HIP events around 400 of them: an empty 4096-workgroup "
                    "dispatch, and a 16-byte-per-lane copy of the 4 MiB in / 4 MiB out a B=4096 transform launch moves, "
                    "over the rotating batches (fz_diag_*)"}


# ---- many batches per dispatch (fz_ntt_multi): the same 4096-row batches, 1 / 2 / 4 / 8 of them per launch -- rank 0 ---
def multi_job(e):
    if e.rank != 0:
        return None
    torch, ctx, B, d = e.torch, e.ctx, e.B, e.d
    multi = {}
    for jobs in (1, 2, 4, 8, 16, 32):
        res = {}
        # rotate through the headline's batches in groups of `jobs`: cold inputs
        groups = [list(range(g, g + jobs)) for g in range(0, len(e.rot_p) - jobs + 1, jobs)]
        fjs = [[(e.xs[k].data_ptr(), e.ys[k].data_ptr(), B, False) for k in g] for g in groups]
        ijs = [[(e.ys[k].data_ptr(), e.zs[k].data_ptr(), B, True) for k in g] for g in groups]
        ctx.ntt_multi_dev(fjs[0])
        ctx.ntt_multi_dev(ijs[0])
        torch.cuda.synchronize(e.dev)
        assert all(torch.equal(e.zs[k], e.xs[k]) for k in groups[0]), "fz_ntt_multi round trip differs"
        for name, jls in (("fwd", fjs), ("inv", ijs)):
            it = [0]

            def fn(jls=jls, it=it):
                ctx.ntt_multi_dev(jls[it[0] % len(jls)])
                it[0] += 1
            e.prewarm(fn, 20, inner=10)
            ms = e.timed_on_stream(fn, 300)
            gbs = jobs * 8.0 * d * B / (ms * 1e-3) / 1e9
            res[name] = {"us_per_launch": round(ms * 1e3, 2), "GB/s": round(gbs, 1), "frac": round(gbs / e.HBM_PEAK_GBS, 4)}
        multi[f"{jobs}x{B}"] = res
    multi["what"] = ("one fz_ntt_multi dispatch over 1/2/4/8/16/32 independent batches of 4096 rows (the job table travels in the "
                     "kernel arguments); back-to-back launches, HIP events on the stream, inputs rotate through the headline's "
                     "64 batches (cold)")
    return multi


# ---- host-pointer path (PCIe-inclusive; never `value`) -- rank 0 -------------------------------------------------------
def pcie_inclusive(e):
    if e.rank != 0:
        return None
    hx = e.x.cpu().numpy().copy()
    e.ctx.ntt_forward(hx)                         # scratch growth and first-touch of the staging outside the timing
    times = []
    for _ in range(8):
        t0 = time.perf_counter()
        e.ctx.ntt_forward(hx)
        times.append(time.perf_counter() - t0)
    dt = min(times)
    return {"value": e.B / dt, "unit": "NTT/s", "ms_per_call": dt * 1e3, "ms_per_call_all": [round(t * 1e3, 3) for t in times],
            "what": "fz_ntt_forward_host on 4096x256 host rows (copy of the input array + H2D + kernel + D2H), best of 8 calls"}


# ---- large-batch asymptote of the same kernels -- rank 0 ---------------------------------------------------------------
def sweep(e):
    if e.rank != 0:
        return None
    torch, ctx, d = e.torch, e.ctx, e.d
    out = {}
    for logb in (14, 16, 18, 20, 22):                   # SURVEY 8d's sweep B = 2^12 (the headline) ... 2^22 (4 GiB in, 4 GiB out)
        nb = 1 << logb
        x1 = torch.empty((nb, d), dtype=torch.int32, device=e.dev)
        ctx.fill_synthetic_dev(x1.data_ptr(), nb * d, 7)              # generated on the device (up to 4 GiB)
        # cold inputs: cycle through enough (input, output) pairs (>= 2 GiB together) that no launch finds its
        # input in the 256 MB Infinity Cache or an L2 from an earlier repetition
        pairs = max(1, -(-(2 << 30) // (2 * x1.numel() * 4)))
        xs = [x1] + [x1.clone() for _ in range(pairs - 1)]
        ys = [torch.empty_like(x1) for _ in range(pairs)]
        ptrs = [(a.data_ptr(), b.data_ptr()) for a, b in zip(xs, ys)]
        for name, fn in (("fwd", ctx.ntt_forward_dev), ("inv", ctx.ntt_inverse_dev)):
            k = 0
            t_end = time.perf_counter() + 0.04          # 40 ms of the same launches first (clock ramp)
            while time.perf_counter() < t_end:
                for _ in range(3):
                    fn(ptrs[k % pairs][0], ptrs[k % pairs][1], nb)
                    k += 1
                torch.cuda.synchronize(e.dev)
            reps = 4 if logb >= 22 else 10 if logb >= 20 else 100
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(e.stream)
            for _ in range(reps):
                fn(ptrs[k % pairs][0], ptrs[k % pairs][1], nb)
                k += 1
            b_.record(e.stream)
            torch.cuda.synchronize(e.dev)
            ms = a.elapsed_time(b_) / reps
            gbs = 8.0 * d * nb / (ms * 1e-3) / 1e9
            out[f"{name}_B2^{logb}"] = {"us": round(ms * 1e3, 2), "GB/s": round(gbs, 1),
                                        "frac": round(gbs / e.HBM_PEAK_GBS, 4), "buffer_pairs_cycled": pairs}
        del xs, ys, x1, ptrs
    return out


# ---- every scheme kernel, cold operands, algorithmic bytes per unit from SURVEY 8d -- rank 0 ---------------------------
def kernels(e):
    if e.rank != 0:
        return None
    from tools.kernel_table import measure
    e.torch.cuda.empty_cache()
    out = measure(e.ctx, e.P, quick=False)
    out["what"] = ("per-launch averages over operand sets carved out of a 2.25 GiB pool (every launch reads bytes no "
                   "launch has touched for >= 2 GiB of other traffic): HBM, not cache bandwidth; HIP events on the kernels' stream")
    return out


# ---- end to end through the array API: device challenge pipeline + device algebra + the serial host sponge -- rank 0 ---
def end_to_end(e):
    if e.rank != 0:
        return None
    torch, np, F = e.torch, e.np, e.F
    from fusion_hip.scheme import BatchScheme
    params = F.fusion_setup(e.SECPAR, 2026)
    bs = BatchScheme(params, device=e.dev_index)
    bs.ctx.set_stream(e.stream.cuda_stream)
    n_e2e = 1024
    seeds = [10_000 + 2 * i for i in range(n_e2e)]
    msgs = [f"synthetic message {i:06d}" for i in range(n_e2e)]

    # every call is timed three times after one untimed call of the same size (first-use allocations, scratch growth);
    # the best is reported: these legs run host code on a shared machine and single shots scatter by 30-50 %
    def best_of(fn, keep=None, reps=3):
        best, out = 1e30, None
        for _ in range(reps):
            if out is not None and keep is not None:
                keep(out)                                          # release the previous repetition's results
            t0 = time.perf_counter()
            out = fn()
            best = min(best, time.perf_counter() - t0)
        return best, out

    def drop_keys(r):
        r[0].free()
        r[2].free()
    drop_keys(bs.keygen_batch(seeds, device=True, keep_vk=True))
    t_keygen, (sk_e, vk_e, vk_dev) = best_of(lambda: bs.keygen_batch(seeds, device=True, keep_vk=True), drop_keys)
    bs.sign_batch(sk_e, vk_dev, msgs, device=True).free()          # scratch growth outside the timing
    t_sign, sig_e = best_of(lambda: bs.sign_batch(sk_e, vk_dev, msgs, device=True), lambda r: r.free())
    t_agg, agg_e = best_of(lambda: bs.aggregate(vk_e, msgs, sig_e))
    t_ver, (ok, why) = best_of(lambda: bs.verify(vk_e, msgs, agg_e))
    assert ok, why
    bs.aggregate_verify(vk_e, msgs, sig_e)
    t_av, (agg_av, (ok, why)) = best_of(lambda: bs.aggregate_verify(vk_e, msgs, sig_e))     # one hash_ag for both
    assert ok and np.array_equal(agg_av, agg_e), why
    n_big = 16384
    seeds_b = [50_000 + 2 * i for i in range(n_big)]
    msgs_b = [f"synthetic message {i:06d}" for i in range(n_big)]
    sk_b, vk_b, vkd_b = bs.keygen_batch(seeds_b, device=True, keep_vk=True)
    bs.sign_batch(sk_b, vkd_b, msgs_b, device=True).free()
    t0 = time.perf_counter()
    bs.sign_batch(sk_b, vkd_b, msgs_b, device=True).free()
    t_sign_big = time.perf_counter() - t0
    sk_b.free()
    vkd_b.free()
    # many independent aggregates in one batch (aggregate_many / verify_many): one host thread per aggregate for its
    # sort + serial SHAKE-256, one launch for all aggregates, one for all verifications.  Same 1024 signatures as above.
    many = {}
    for g_, n_ in ((4, 256), (16, 64), (64, 16)):
        sizes = [n_] * g_
        bs.aggregate_many(vk_e, msgs, sig_e, sizes)                      # scratch growth outside the timing
        t_am, aggs = best_of(lambda: bs.aggregate_many(vk_e, msgs, sig_e, sizes))
        t_vm, verd = best_of(lambda: bs.verify_many(vk_e, msgs, aggs, sizes))
        assert all(v[0] for v in verd), verd
        many[f"{g_}x{n_}"] = {"aggregate_per_s": n_e2e / t_am, "verify_per_s": n_e2e / t_vm,
                              "sign_plus_verify_per_s": n_e2e / (t_sign + t_am + t_vm),
                              "aggregate_ms": t_am * 1e3, "verify_ms": t_vm * 1e3}
    sig_e.free()
    sk_e.free()
    vk_dev.free()
    out = {"signatures": n_e2e, "host_threads": bs.threads, "timing": "best of 3 calls after one untimed call of the same size",
           "keygen_per_s": n_e2e / t_keygen, "sign_per_s": n_e2e / t_sign, "sign_per_s_at_16384_signatures": n_big / t_sign_big,
           "aggregate_per_s": n_e2e / t_agg, "verify_per_s": n_e2e / t_ver,
           "sign_plus_verify_per_s": n_e2e / (t_sign + t_agg + t_ver),
           "aggregate_verify_per_s": n_e2e / t_av, "sign_plus_aggregate_verify_per_s": n_e2e / (t_sign + t_av),
           "many_aggregates": many,
           "note": "BatchScheme with device-resident keys and signatures: reference-exact MT19937 sampling, the per-signer challenge "
                   "pipeline and all algebra on the device; aggregate and verify are bounded by hash_ag, ONE serial SHAKE-256 over "
                   "~13.5 KB per signer on the host by construction (fusion.py:632-652)"}
    try:
        out["queue"] = _queue_leg(e, bs, params)
        out["queue_pairs_per_s"] = out["queue"].get("pairs_per_s")
    except Exception as exc:                              # newer than the rest of the leg: never its failure
        out["queue"] = {"error": repr(exc)}
    return out


def _queue_leg(e, bs, params):
    """keygen_batch + sign_batch at BASELINE's 1024 per call from ONE Python thread through the asynchronous batch queue
    (fusion_hip.queue.BatchQueue = fz_queue_* below Python) -- reference call pattern fusion.py:338-373, :534-557.  The same
    work per call as tools/probes/concurrent_batches.py (verification keys come back to the host, keys and signatures are dropped on
    the device), which needed 8-16 Python threads for 3.8-4.3 M pairs/s (profiles/r03_concurrent_batches.txt)."""
    from fusion_hip.queue import BatchQueue, PackedMessages
    np = e.np
    n, calls = 1024, 96
    out = {}
    for workers in (1, 2, 3):
        with BatchQueue(params, device=e.dev_index, workers=workers) as bq:
            seeds = [np.arange(n, dtype=np.uint64) * 2 + np.uint64(70_000 + 4096 * c) for c in range(calls)]
            msgs = PackedMessages([f"synthetic message {i:06d}" for i in range(n)])
            for c in range(8):                                # first-use allocations of every worker
                bq.submit_keygen_sign(seeds[c], msgs, discard=True)
            bq.drain()
            bq.collect_discarded()
            c0, b0, _ = bq.stats()
            best = 1e30
            for _ in range(3):
                t0 = time.perf_counter()
                for c in range(calls):
                    bq.submit_keygen_sign(seeds[c], msgs, discard=True)
                t_submit = time.perf_counter() - t0
                bq.drain()
                best = min(best, time.perf_counter() - t0)
                bq.collect_discarded()
            c1, b1, _ = bq.stats()
            out[f"workers={workers}"] = {"pairs_per_s": n * calls / best, "calls_per_batch": (c1 - c0) / max(1, b1 - b0),
                                         "submit_us_per_call": t_submit / calls * 1e6}
    best_w = max(out, key=lambda k: out[k]["pairs_per_s"])
    return {"pairs_per_s": out[best_w]["pairs_per_s"], "best": best_w, "calls": calls, "per_call": n, "by_workers": out,
            "what": "one Python thread submitting 1024-key + 1024-signature calls to the C-level batch queue (seeds as uint64 "
                    "arrays, messages packed once); vk back to pinned host buffers, device results dropped"}


# ---- end to end, SHARDED: aggregate() + verify() of ONE aggregate of 1024 signers spread over the ranks -- every rank --
def end_to_end_sharded(e):
    torch, np, F, dist = e.torch, e.np, e.F, e.dist
    from fusion_hip.dist import ShardedScheme, TorchCollective
    from fusion_hip.scheme import BatchScheme
    rank, world = e.rank, e.world
    params = F.fusion_setup(e.SECPAR, 2026)
    bs = BatchScheme(params, device=e.dev_index)
    bs.ctx.set_stream(e.stream.cuda_stream)
    n_all = 1024
    seeds = [10_000 + 2 * i for i in range(n_all)]
    msgs = [f"synthetic message {i:06d}" for i in range(n_all)]
    lo_, hi_ = e.shard_range(n_all, rank, world)
    sk_l, vk_l, vk_ld = bs.keygen_batch(seeds[lo_:hi_], device=True, keep_vk=True)
    sig_l = bs.sign_batch(sk_l, vk_ld, msgs[lo_:hi_], device=True)         # this rank's signatures stay in its HBM
    if world > 1:                                                          # verification keys are public: everyone gets all
        parts = [None] * world
        dist.all_gather_object(parts, vk_l)
        vk_all = np.concatenate(parts)
    else:
        vk_all = vk_l
    sh = ShardedScheme(bs, rank, world, TorchCollective(bs.ctx, e.dev_index))
    sh.aggregate_verify_sharded(vk_all, msgs, sig_l)                      # scratch growth, first-use tables
    e.barrier()
    t0 = time.perf_counter()
    agg_s, verdict_s = sh.aggregate_verify_sharded(vk_all, msgs, sig_l)
    e.barrier()
    t_sh = e.max_over_ranks(time.perf_counter() - t0)
    assert verdict_s == (True, ""), verdict_s
    t0 = time.perf_counter()
    v2 = sh.verify_sharded(vk_all, msgs, agg_s)
    e.barrier()
    t_vs = e.max_over_ranks(time.perf_counter() - t0)
    assert v2 == (True, ""), v2
    out = {"signers": n_all, "ranks": world, "signers_per_rank": hi_ - lo_,
           "aggregate_plus_verify_per_s": n_all / t_sh, "aggregate_plus_verify_ms": t_sh * 1e3,
           "verify_per_s": n_all / t_vs, "verify_ms": t_vs * 1e3,
           "collective": (f"torch.distributed all_reduce ({dist.get_backend()})" if world > 1 else "none (single rank)"),
           "what": "ShardedScheme.aggregate_verify_sharded / verify_sharded: hash_ag is one serial SHAKE-256 over all signers; every "
                   "rank transforms only its block of alpha, makes one pass over its block of signatures, then ONE all-reduce of "
                   "l*d + d int64; max over ranks"}
    for b in (sk_l, vk_ld, sig_l):
        b.free()
    return out
