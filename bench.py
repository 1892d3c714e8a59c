#!/usr/bin/env python3
"""bench.py -- headline benchmark of the algebra hot path on MI355X.

Workload (BASELINE.json configs[1]): secpar=256, one batch of 4096 independent degree-256
polynomials; a STEP is one pass of the hot path over that batch = forward NTT of the batch
followed by inverse NTT of the result (two kernel launches, inputs resident in HBM).  The K steps
(--steps) are recorded into a hipGraph beforehand (the library's fz_graph_* capture) and the graph is
replayed R times inside the timed region, R chosen so that the region lasts >= 20 ms whatever K is
(a 0.2 ms region measures graph start-up, not kernels): the same 2K kernels in the same order on one
stream, R times, no host round trip per launch (--no-graph launches them one by one).
`value` = NTTs per second over the whole job (forward and inverse each count as one NTT, summed over
all ranks); `ms_per_step` = elapsed / (R * K).  With --gpus N every rank owns its own batch (weak
scaling, no data-path collective for the transforms).

Launch: `python bench.py --gpus N` starts its own N ranks (child processes, before this process has
touched a GPU) when WORLD_SIZE is not set; under torch.distributed.run it uses the ranks it is given.

The same JSON line carries
  roofline      achieved algorithmic GB/s of the dominant kernel (forward NTT: 8*d bytes per polynomial)
                from per-dispatch begin/end events; `copy_floor` (an empty dispatch and a plain copy of the
                same bytes, same run); `multi_job` (fz_ntt_multi: 1/2/4/8 batches of 4096 rows per dispatch);
                `kernels` (every scheme kernel, cold operands: tools/kernel_table.py); the large-batch sweep.
  cpu_baseline  the pure-Python port of the reference's algebra path (oracle/oracle.py py_*) timed on one
                host core over a bounded sample of the same workload (rank 0, N=1), and on the box's cores.
  sign_verify   the second half of BASELINE's metric: signatures signed + aggregated + verified per second
                (algebra cores; synthetic keys/messages, operand sets rotated so that nothing is cache-resident),
                sharded over the ranks with ONE RCCL all-reduce of the int64 partials per step, issued through
                the C ABI (fz_allreduce_i64) and replayed from the library's graph together with the kernels.
  ranks         what every rank reported: device index, PCI bus id, world size and backend as RCCL saw them.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "fusion-cryptography_amd"))
sys.path.insert(0, ROOT)

METRIC = "batched NTT/s (deg-256, secpar=256) + aggregate sign+verify/sec at 1/2/4/8 GPU"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
B = 4096                       # BASELINE configs[1]
SECPAR = 256
MIN_REGION_MS = 20.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sign-verify", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the large-batch runs of the same kernels")
    ap.add_argument("--no-two-stream", action="store_true", help="skip the two-stream pipelined and cold-batch variants")
    ap.add_argument("--no-kernel-table", action="store_true", help="skip the cold per-kernel table of the scheme kernels")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the BatchScheme leg (host hashing included)")
    ap.add_argument("--no-graph", action="store_true", help="launch the timed steps one by one instead of replaying hipGraphs")
    ap.add_argument("--sample-every", type=int, default=1,
                    help="bind begin/end events to every k-th dispatch of each kernel in the instrumented pass (1: every "
                         "dispatch carries its own completion signal, as under rocprofv3 -- the two then agree within 3 %%; "
                         "k > 1 lets un-instrumented dispatches overlap the sampled one and reads ~0.5 us longer)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the pure-Python baseline sample")
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="untimed run of the same steps before the W warmup steps: after idle the GPU needs tens of "
                         "milliseconds of load to reach its steady clocks (measured: the first ~15 ms run 10-25 %% slower)")
    ap.add_argument("--watchdog-s", type=float, default=420.0,
                    help="if the legs after the headline have not finished by then, rank 0 prints the line with what it has "
                         "(\"watchdog\" says which leg was running) and every rank leaves WITH EXIT CODE 3: a hung collective must "
                         "not cost the headline, and must not pass for success either")
    ap.add_argument("--cpu-worker", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-signers", type=int, default=0, help=argparse.SUPPRESS)
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle's pure-Python port; rank 0 at N = 1 only)
# ---------------------------------------------------------------------------------------------------------
def _cpu_worker(job):
    """one host process of the CPU baseline (the ONLY place bench.py touches oracle/).
    (index, seconds): forward+inverse pure-Python NTTs for `seconds`; returns (transforms, seconds).
    (index, n, "scheme"): the metric's second half in the pure-Python port -- keygen, sign, aggregate and verify cores
    (fusion/fusion.py:363-370, :557, :670-676, :690-727) of ONE aggregate of n synthetic signers; returns the four phase
    times in seconds (the verdict must be 0: it is real work on valid signatures)."""
    from oracle import oracle as O
    P = O.PARAMS[SECPAR]
    q, d = P["q"], P["d"]
    tw, itw = O.py_twiddles(P["root"], q, d), O.py_twiddles(P["inv_root"], q, d)
    if len(job) == 3:
        import random
        index, n, _ = job
        l = P["rank"]
        rng = random.Random(9000 + index)
        A = O.splitmix_centered(99, l * d).reshape(l, d).tolist()

        def secret():      # beta_sk = 52, omega_sk = d (fusion.py:30-31, :116): every coefficient non-zero in +-[1, 52]
            return [(1 + rng.randrange(52)) * (1 - 2 * rng.randrange(2)) for _ in range(d)]

        def sparse_hat(weight):
            c = [0] * d
            for j in rng.sample(range(d), weight):
                c[j] = 1 - 2 * rng.randrange(2)
            return O.py_ntt_forward(c, q, tw)
        secrets = [([secret() for _ in range(l)], [secret() for _ in range(l)]) for _ in range(n)]
        c_hat = [sparse_hat(P["omega_ch"]) for _ in range(n)]
        al_hat = [sparse_hat(P["omega_ag"]) for _ in range(n)]
        t0 = time.perf_counter()
        keys = [O.py_keygen_core(A, sL, sR, q, tw) for sL, sR in secrets]
        t1 = time.perf_counter()
        sigs = [O.py_sign_core(k[0], k[1], c, q) for k, c in zip(keys, c_hat)]
        t2 = time.perf_counter()
        agg = O.py_aggregate_core(sigs, al_hat, q)
        t3 = time.perf_counter()
        code = O.py_verify_core(A, agg, [k[2] for k in keys], [k[3] for k in keys], c_hat, al_hat, q, itw, P["beta_vf"], d)
        t4 = time.perf_counter()
        assert code == 0, f"pure-Python verify returned {code}"
        return n, t1 - t0, t2 - t1, t3 - t2, t4 - t3
    index, seconds = job
    rows = 64
    x = O.splitmix_centered(20261003, B * d).reshape(B, d)[(index * rows) % B:][:rows].tolist()
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:          # bounded sample: as many rows as fit the budget
        row = list(x[(done // 2) % rows])
        O.py_ntt_inverse(O.py_ntt_forward(row, q, tw), q, itw)
        done += 2
    return done, time.perf_counter() - t0


def cpu_baseline(seconds):
    """Pure-Python port (lists of ints, one cent() per reference cent call): ONE core (the reference is
    single-threaded), then the same loop in one process per host core this job may use."""
    done, dt = _cpu_worker((0, seconds))
    ncpu = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = ncpu
    out = {"value": done / dt, "unit": "NTT/s", "cores": 1, "kind": "port",
           "sample": f"{done // 2} rows of the {B}-row batch: forward+inverse degree-256 NTT each, pure-Python port "
                     f"(oracle.py_ntt_forward/py_ntt_inverse), {dt:.1f} s on 1 core of {ncpu}"}
    try:
        import subprocess
        # BASELINE.md section 3 says "all host cores"; a one-GPU box of this pool is a 1/8 share (16 CPUs) of a
        # 256-CPU host that seven other jobs use at the same time, and its process guard counts children -- so the pool
        # is min(CPUs this process may run on, FZ_BENCH_CPU_WORKERS = 16); set the variable to os.cpu_count() on a
        # machine of one's own.  The count actually used is reported as `cores`.
        workers = max(1, min(usable, int(os.environ.get("FZ_BENCH_CPU_WORKERS", "16"))))
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(i), "--cpu-seconds",
                                   str(seconds / 3)], stdout=subprocess.PIPE, text=True) for i in range(workers)]
        res = []
        for pr in procs:
            try:
                text, _ = pr.communicate(timeout=seconds + 120)
                n, t = text.split()
                res.append((int(n), float(t)))
            except Exception:
                pr.kill()
        if not res:
            raise RuntimeError("no worker finished")
        out["all_cores"] = {"value": sum(n / t for n, t in res), "unit": "NTT/s", "cores": len(res),
                            "host_logical_cpus": ncpu, "usable_by_this_process": usable,
                            "why_not_all": "pool = min(schedulable CPUs, FZ_BENCH_CPU_WORKERS=16): a one-GPU box is a 16-CPU "
                                           "share of a shared 256-CPU host (see the comment in bench.py)",
                            "sample": f"{len(res)} processes x {seconds / 3:.1f} s of the same loop"}
    except Exception as e:                                   # the single-core figure stands on its own
        out["all_cores"] = {"error": repr(e)}
    # ---- the metric's second half: keygen/s, sign/s, aggregate-signatures/s, verify-signatures/s of the
    # pure-Python port on a sub-sample (64 signers on one core; `workers` x 16 signers on the pool), synthetic inputs as the
    # GPU legs use them (secrets in +-[1, 52], ternary challenges / aggregation coefficients of the parameter set's weights)
    def rates(n, tk, ts, ta, tv):
        return {"keygen_per_s": n / tk, "sign_per_s": n / ts, "aggregate_signatures_per_s": n / ta, "verify_signatures_per_s": n / tv,
                "sign_plus_verify_per_s": n / (ts + ta + tv), "keygen_plus_sign_per_s": n / (tk + ts)}
    try:
        n1 = int(os.environ.get("FZ_BENCH_CPU_SIGNERS", "64"))
        n, tk, ts, ta, tv = _cpu_worker((0, n1, "scheme"))
        sch = dict(rates(n, tk, ts, ta, tv), cores=1, kind="port", unit="per second",
                   sample=f"one aggregate of {n} synthetic signers at secpar {SECPAR}: keygen {tk:.2f} s, sign {ts:.2f} s, aggregate "
                          f"{ta:.2f} s, verify {tv:.3f} s on 1 core (oracle.py_keygen_core / py_sign_core / py_aggregate_core / "
                          f"py_verify_core: lists of Python ints, one cent() per reference cent call)")
        import subprocess
        workers = max(1, min(usable, int(os.environ.get("FZ_BENCH_CPU_WORKERS", "16"))))
        npool = max(2, n1 // 4)
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(i), "--cpu-signers", str(npool)],
                                  stdout=subprocess.PIPE, text=True) for i in range(workers)]
        res = []
        for pr in procs:
            try:
                text, _ = pr.communicate(timeout=600)
                res.append([float(x) for x in text.split()])
            except Exception:
                pr.kill()
        if res:
            # independent aggregates, one per process: the pool's rate of a phase = all signers / the slowest worker's phase time
            tot = sum(r[0] for r in res)
            sch["all_cores"] = dict(rates(tot, *[max(r[k] for r in res) for k in (1, 2, 3, 4)]), cores=len(res),
                                    sample=f"{len(res)} processes, one aggregate of {npool} signers each")
        out["scheme"] = sch
    except Exception as e:
        out["scheme"] = {"error": repr(e)}
    return out


# ---------------------------------------------------------------------------------------------------------
# --gpus N without a launcher: this process starts the N ranks itself and never touches a GPU
# ---------------------------------------------------------------------------------------------------------
def self_launch(args):
    import socket
    import subprocess
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n),
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    deadline = time.time() + float(os.environ.get("FZ_BENCH_TIMEOUT", "1500"))
    failed = None
    while any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0) and failed is None:
                failed = (r, p.returncode)
        if failed or time.time() > deadline:
            for p in procs:                              # exactly the children started above, nothing by pattern
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    line = procs[0].stdout.read() if procs[0].stdout else ""
    for r, p in enumerate(procs):
        p.wait()
        if p.returncode != 0 and failed is None:
            failed = (r, p.returncode)
    rescued = [ln for ln in line.splitlines() if ln.startswith('{"metric"') and '"watchdog"' in ln]
    if rescued:                                          # a leg after the headline hung: rank 0's watchdog printed what it had
        sys.stdout.write(rescued[-1] + "\n")
        sys.stdout.flush()
        sys.stderr.write("bench.py: a side leg hung and the watchdog fired (see the line's \"watchdog\" field): the headline is "
                         "in the line above, the run is a FAILURE -- exit code 3\n")
        sys.exit(3)
    if failed:
        sys.stderr.write(f"bench.py: rank {failed[0]} exited with code {failed[1]}; no result\n")
        sys.exit(1)
    if time.time() > deadline:
        sys.exit("bench.py: ranks did not finish in time")
    result = [ln for ln in line.splitlines() if ln.startswith('{"metric"')]      # a backend may chat on stdout (gloo does)
    if not result:
        sys.exit("bench.py: rank 0 printed no result line")
    sys.stdout.write(result[-1] + "\n")
    sys.stdout.flush()


def main():
    args = parse()
    if args.cpu_worker is not None:                      # child of cpu_baseline(): host only
        if args.cpu_signers:
            print(*_cpu_worker((args.cpu_worker, args.cpu_signers, "scheme")))
        else:
            n, t = _cpu_worker((args.cpu_worker, args.cpu_seconds))
            print(n, t)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)                         # before anything in this process touches a GPU

    # host threads on ONE NUMA node (the GPU's), before anything initialises the GPU (sysfs + sched_setaffinity only): a
    # launching thread the scheduler moves to the other socket after initialisation launches at 5.1-7.5 us per dispatch
    # instead of 4.4-4.8 (profiles/r02_numa_placement.txt) -- the run-to-run spread of roofline.frac before this
    from fusion_hip.numa import pin_to_gpu_node
    placement = pin_to_gpu_node(int(os.environ.get("LOCAL_RANK", "0")))

    import numpy as np
    import torch
    import torch.distributed as dist
    import fusion_hip
    from fusion_hip.dist import allreduce_sum_i64, shard_range
    import fusion.fusion as F                     # the drop-in's parameter sets (the oracle is the CPU baseline's only)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("FZ_BENCH_BACKEND", "nccl")
    if backend == "nccl" and world > ndev:
        sys.exit(f"--gpus {world} but only {ndev} GPU(s) visible (FZ_BENCH_BACKEND=gloo rehearses the N>1 path on fewer)")
    # one rank per GPU; the modulo only matters for the gloo rehearsal, where ranks share a device
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    ps = F.PREFIX_PARAMETERS[SECPAR]
    P = {"q": ps["modulus"], "d": ps["degree"], "root": ps["root"], "inv_root": ps["inv_root"], "rank": ps["num_rows_sk"],
         "omega_ch": ps["omega_ch"], "omega_ag": ps["omega_ag"], "capacity": ps["capacity"], "beta_vf": ps["beta_vf"]}
    q, d, l = P["q"], P["d"], P["rank"]
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"], device=dev_index)
    # a non-default stream, made torch's current one: graph capture needs it, and torch ops / RCCL order on it too
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(v):
        if world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def min_over_ranks(v):
        return -max_over_ranks(-v)

    # what every rank is, as the process group sees it (proof that N ranks on N devices took part)
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "device": props.name,
          "pci_bus_id": f"{getattr(props, 'pci_domain_id', 0):04x}:{getattr(props, 'pci_bus_id', -1):02x}:{getattr(props, 'pci_device_id', 0):02x}",
          "world_size_seen": dist.get_world_size() if world > 1 else 1,
          "backend": dist.get_backend() if world > 1 else "none (single rank)"}
    ranks = [me]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)

    # ---- NTT workload: inputs resident in HBM --------------------------------------------------
    # i.i.d. uniform centred residues, seeded per rank, generated where they are consumed (fz_fill_synthetic: the stream the
    # tests' host generator produces)
    x = torch.empty((B, d), dtype=torch.int32, device=dev)
    ctx.fill_synthetic_dev(x.data_ptr(), B * d, 20261003 + rank)
    torch.cuda.synchronize(dev)
    y = torch.empty_like(x)
    z = torch.empty_like(x)

    # the timed loop calls the C ABI directly with pre-built arguments: at ~4.5 us per kernel the Python
    # wrapper layers (attribute lookups, argument boxing) would otherwise be part of the measurement
    import ctypes
    lib, h = ctx._lib, ctx._h
    xp, yp, zp = (ctypes.c_void_p(t.data_ptr()) for t in (x, y, z))
    fz_fwd, fz_inv, nB = lib.fz_ntt_forward, lib.fz_ntt_inverse, ctypes.c_size_t(B)

    def step():
        return fz_fwd(h, xp, yp, nB) | fz_inv(h, yp, zp, nB)

    def prewarm(fn, ms, inner=50):
        """untimed: keep the device busy for `ms` so the timed region starts at steady clocks"""
        t_end = time.perf_counter() + ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(inner):
                fn()
            torch.cuda.synchronize(dev)

    def timed_on_stream(fn, reps):
        """average milliseconds of fn() over reps back-to-back calls, HIP events on the kernels' own stream"""
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(reps):
            fn()
        b_.record(stream)
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b_) / reps

    step()
    barrier()
    assert torch.equal(z, x), "INTT(NTT(x)) != x"
    prewarm(step, args.prewarm_ms)
    for _ in range(args.warmup):
        step()
    barrier()
    # The K steps are recorded once (fz_graph_*; graphs of <= 1000 steps) and replayed R times inside the timed region, R
    # from an untimed calibration so that the region lasts >= MIN_REGION_MS on every rank.  A replay costs the host and the
    # command processor ~10 us whatever it holds, so for K < 1000 one recording holds M = 1000 // K repetitions of the K
    # steps (R counts every repetition): --steps 20 and --steps 1000 then replay the same 1000-step recordings and agree.
    K = args.steps
    M = max(1, 1000 // K) if K < 1000 else 1
    chunk = min(K, 1000)
    graphs = []
    if not args.no_graph:
        sizes = [K * M] if K < 1000 else ([chunk] * (K // chunk)) + ([K % chunk] if K % chunk else [])
        for n_steps in sizes:
            if graphs and graphs[0][0] == n_steps:
                graphs.append(graphs[0])                 # the same recording, launched again
                continue
            ctx.graph_begin()
            rc = 0
            for _ in range(n_steps):
                rc |= step()
            g = ctx.graph_end()
            assert rc == 0, f"capture failed: {lib.fz_last_error()}"
            g.launch()                                   # untimed first replay (upload)
            graphs.append((n_steps, g))
        barrier()
    else:
        M = 1

    def k_steps():
        """M repetitions of the K steps"""
        if graphs:
            for _, g in graphs:
                g.launch()
        else:
            rc = 0
            for _ in range(args.steps):
                rc |= step()
            assert rc == 0, f"launch failed: {lib.fz_last_error()}"

    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    k_steps()
    torch.cuda.synchronize(dev)
    t_once = max(time.perf_counter() - t0, 1e-6)
    launches = int(max_over_ranks(max(1.0, -(-MIN_REGION_MS * 1e-3 // t_once))))     # the same count on every rank
    repeats = launches * M
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(launches):
        k_steps()
    ev1.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    total_steps = repeats * args.steps
    region_launch_us = ev0.elapsed_time(ev1) * 1e3 / (2 * total_steps)     # HIP events over the timed region / launches
    assert torch.equal(z, x), "INTT(NTT(x)) != x after the timed region"
    # the shader clock the chip holds while the timed region's launches run (a one-wave probe on a private stream beside a
    # few more milliseconds of them): the fp64-dense kernels are power-limited below the nominal 2.4 GHz
    shader_mhz = None
    try:
        for _ in range(max(1, launches // 4)):
            k_steps()
        shader_mhz = round(ctx.diag_shader_clock(500))
    except Exception as e:                                                              # a diagnostic: never the run's failure
        shader_mhz = f"failed: {type(e).__name__}: {e}"
    torch.cuda.synchronize(dev)
    for g in {id(g): g for _, g in graphs}.values():
        g.destroy()
    # Per-dispatch durations (kernel begin -> end, events bound to the dispatch on its own stream) cannot be taken
    # inside a graph: an instrumented pass of the same steps, launched one by one right after the timed region, with
    # EVERY dispatch carrying its events -- each then has its own completion signal and runs serialised, exactly the
    # condition rocprofv3 --kernel-trace puts every dispatch in, and the two averages agree (4.73 vs 4.61 us).
    # Three passes of 400 steps; the pass with the lowest mean is reported (all three are listed): a pass now and then has
    # a tail of 10-30 us samples from outside the kernel (the host's launch rate drops in the same pass; the median stays).
    # The roofline is an HBM roofline, so the passes that feed `achieved` read COLD operands: the steps rotate through 32
    # batches (x_i -> y_i -> z_i, 384 MiB together), no launch finds its input in a cache.  Three more passes on the ONE warm
    # batch of the timed region are listed beside them (`passes_warm`).
    n_inst, passes, passes_warm = 400, [], []
    nb_rot = 32
    rot = [(x.clone(), torch.empty_like(x), torch.empty_like(x)) for _ in range(nb_rot)]
    rot_p = [tuple(ctypes.c_void_p(t.data_ptr()) for t in trio) for trio in rot]
    rot_i = [0]

    def step_cold():
        a_, b_, c_ = rot_p[rot_i[0] % nb_rot]
        rot_i[0] += 1
        return fz_fwd(h, a_, b_, nB) | fz_inv(h, b_, c_, nB)
    for fn, out_list in ((step_cold, passes), (step, passes_warm)):
        for _ in range(3):
            prewarm(fn, 20.0)                           # dense launches first: a one-by-one pass leaves the device half idle
            ctx.profile_begin(2 * n_inst, args.sample_every)
            rc = 0
            for _ in range(n_inst):
                rc |= fn()
            assert rc == 0, f"launch failed: {lib.fz_last_error()}"
            us, kind = ctx.profile_end_samples(2 * n_inst)
            f_, i_ = us[kind == 0], us[kind == 1]
            assert len(f_) == len(i_) == (n_inst + args.sample_every - 1) // args.sample_every
            out_list.append({"fwd_avg_us": float(f_.mean()), "inv_avg_us": float(i_.mean()), "fwd_median_us": float(np.median(f_)),
                             "inv_median_us": float(np.median(i_)), "fwd_max_us": float(f_.max()), "fwd_count": int(len(f_))})
    torch.cuda.synchronize(dev)
    assert all(torch.equal(t[2], x) for t in rot), "INTT(NTT(x)) != x on the rotating batches"
    prof = min(passes, key=lambda p_: p_["fwd_avg_us"])
    prof_warm = min(passes_warm, key=lambda p_: p_["fwd_avg_us"])
    fwd_avg, inv_avg = prof["fwd_avg_us"] * 1e-3, prof["inv_avg_us"] * 1e-3      # ms
    elapsed = max_over_ranks(elapsed)
    value = 2.0 * B * total_steps * world / elapsed

    # ---- everything below is a side leg: the headline is measured.  A watchdog prints the line with whatever is there if
    # a leg hangs (a collective that never completes cannot be cancelled from Python) ------------------------------------
    floor = multi = cold = two_stream = pcie = kernels = sv = e2e = None
    sweep = {}
    stage = ["copy_floor"]
    done = threading.Event()

    def target_block():
        """north_star asks for >= 40 % of the HBM roofline on batched degree-256 NTT: from which batch size the COLD per-kernel
        table of this run meets it, and what a single dispatch of the headline's byte count can reach at all (a plain copy of the
        same 4 MiB in / 4 MiB out, same run)"""
        met = None
        try:
            rows = sorted((int(k.split("2^")[1]), v["frac"]) for k, v in (kernels or {}).items()
                          if k.startswith("ntt_forward B=2^") and isinstance(v, dict))
            for logb, frac in rows:
                if frac >= 0.40:
                    met = 1 << logb
                    break
        except Exception:
            pass
        return {"asked": 0.40, "met_from_rows": met,
                "met_with_batches_per_dispatch": next((int(k.split("x")[0]) for k, v in (multi or {}).items()
                                                       if isinstance(v, dict) and "fwd" in v and v["fwd"]["frac"] >= 0.40), None),
                "single_dispatch_ceiling": (floor or {}).get("copy_frac"),
                "what": "met_from_rows: smallest batch of the cold kernel table (2^12, 2^14, 2^16, 2^18 rows) whose forward transform reaches "
                        "40 % of 8 TB/s; met_with_batches_per_dispatch: batches of 4096 rows per fz_ntt_multi dispatch that reach it (warm); "
                        "single_dispatch_ceiling: the fraction a plain copy of the headline's bytes reaches in this run -- no kernel of "
                        "that byte count can do better, an EMPTY dispatch already costs 1.5-1.9 us (profiles/r03_ntt_variants_per_dispatch.txt)"}

    def rocprof_reference():
        """what rocprofv3 --kernel-trace measured for the same launch in the committed collection (profiles/): the third clock on
        the dominant kernel, next to `avg_launch_us` (begin / end events on every dispatch, launched one by one) and
        `region.avg_launch_us` (the un-instrumented graph replay)"""
        try:
            import csv
            with open(os.path.join(ROOT, "profiles", "r03_bench_rocprofv3_by_grid.csv")) as fh:
                rows = [r for r in csv.DictReader(fh) if r["kernel"].startswith("ntt_fwd4<8") and int(r["grid_threads"]) == B * 64]
            r = max(rows, key=lambda r_: int(r_["calls"]))
            us = float(r["avg_us"])
            return {"kernel": r["kernel"], "dispatches": int(r["calls"]), "avg_us": us, "frac": 8.0 * d * B / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                    "source": "profiles/r03_bench_rocprofv3_by_grid.csv (rocprofv3 --kernel-trace --stats over this script, committed; "
                              "most dispatches graph-replayed)"}
        except Exception:
            return None

    def build_line(watchdog=None):
        # HBM-side traffic of this launch from the PMC counters (rocprofv3 --pmc, separate passes; FETCH_SIZE corrected x2
        # for gfx950) cannot be collected live: the committed measurement of THIS round's kernel is reported, or null
        traffic, traffic_note = None, "no PMC file for this kernel under profiles/ (tools/collect_profiles.sh writes r03_pmc_ntt.json)"
        try:
            with open(os.path.join(ROOT, "profiles", "r03_pmc_ntt.json")) as fh:
                kernels_pmc = json.load(fh)["kernels"]
            # the bench's 4096-row launch (a 64-row call of the PCIe leg runs the same kernel with a smaller grid: take the largest)
            traffic = max(v["traffic_bytes_per_launch"] for k, v in kernels_pmc.items() if k.startswith("ntt_fwd4<8") and "B=4096" in k)
            traffic_note = "profiles/r03_pmc_ntt.json (PMC pass of the same launch, committed this round)"
        except Exception:
            pass
        fwd_bytes = 8.0 * d * B
        ach = fwd_bytes / (fwd_avg * 1e-3) / 1e9
        out = {
            "metric": METRIC, "value": value, "unit": "NTT/s", "n_gpus": world, "steps": args.steps, "repeats": repeats,
            "warmup": args.warmup, "ms_per_step": elapsed / total_steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "timed_region_ms": elapsed * 1e3,
            "config": {"workload": f"configs[1]: secpar={SECPAR}, batch of {B} degree-{d} forward+inverse NTTs per GPU",
                       "batch": B, "degree": d, "modulus": q, "kernels_per_step": 2,
                       "arithmetic": "exact integers carried in fp64 lanes (results bit-identical to the reference's int arithmetic); int32 in and out",
                       "launch": "one by one" if args.no_graph else f"hipGraphs (fz_graph_*) holding {M} x {args.steps} steps, {launches} replays in the timed region = {repeats} repetitions of the {args.steps} steps",
                       "prewarm_ms": args.prewarm_ms,
                       "host_threads_on": placement or "all allowed CPUs (no GPU-local NUMA node found, or FZ_NO_PIN=1)"},
            "ranks": ranks,
            "roofline": {"bound": "hbm", "inputs": "`achieved` / `avg_launch_us` / `passes`: per-dispatch events on steps that ROTATE through 32 batches "
                                                   "(384 MiB: every launch reads its input from HBM); `passes_warm` / `warm_*`: the same on the one "
                                                   "batch the timed region re-reads; `value` is the timed region (inputs resident in HBM, cache-warm "
                                                   "after the first step), `cold_batches` the same steps over the 32 batches",
                         "kernel": "ntt_fwd4<8, true, 1, 8> (forward NTT, B=4096: one row per wave, 8-wave workgroups)", "achieved": ach,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_note,
                         "bytes_per_launch": fwd_bytes, "butterflies_per_s": value * (d // 2) * 8, "avg_launch_us": fwd_avg * 1e3,
                         "inverse_avg_launch_us": inv_avg * 1e3, "launches_timed": prof["fwd_count"],
                         "median_launch_us": prof["fwd_median_us"], "passes": passes,
                         "passes_warm": passes_warm, "warm_avg_launch_us": prof_warm["fwd_avg_us"],
                         "warm_frac": fwd_bytes / (prof_warm["fwd_avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "timing": f"per-dispatch begin/end events (hipExtLaunchKernelGGL) on every {args.sample_every}th dispatch of "
                                   f"{n_inst} steps launched one by one right after the timed region (a hipGraph cannot carry them), the steps "
                                   f"rotating through {nb_rot} batches (cold inputs); mean over the launches of the best of 3 such passes",
                         "region": {"avg_launch_us": region_launch_us, "achieved": fwd_bytes / (region_launch_us * 1e-6) / 1e9,
                                    "frac": fwd_bytes / (region_launch_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                    "what": "HIP events around the timed region on the kernels' stream / launches "
                                            "(consecutive dispatches overlap their launch and drain phases)"},
                         "shader_mhz": {"timed_region": shader_mhz, "nominal": 2400,
                                        "what": "fz_diag_shader_clock beside the timed region's launches; per kernel: roofline.kernels[*].shader_mhz"},
                         "copy_floor": floor, "multi_job": multi, "kernels": kernels, "sweep": sweep,
                         "target": target_block(), "rocprofv3": rocprof_reference()},
            "cold_batches": cold, "two_stream_pipelined": two_stream, "sign_verify": sv, "end_to_end": e2e, "pcie_inclusive": pcie,
        }
        if watchdog:
            out["watchdog"] = watchdog
        return out

    def watchdog():
        if done.wait(args.watchdog_s):
            return
        _BAILING.set()
        if rank == 0:
            try:
                text = json.dumps(build_line(f"side legs not finished {args.watchdog_s:.0f} s after the headline; running: {stage[0]}"))
                print(text)
                sys.stdout.flush()
                try:
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    with open(os.path.join(ROOT, "gpurun_out", "bench_watchdog_line.json"), "w") as fh:
                        fh.write(text + "\n")
                except OSError:
                    pass
            except Exception:
                import traceback
                traceback.print_exc()
        sys.stderr.write(f"rank {rank}: watchdog fired in leg '{stage[0]}'\n")
        sys.stderr.flush()
        if rank == 0:
            time.sleep(3.0)                             # the other ranks' watchdogs fire at the same moment: let them leave first
        # A process that hung on the GPU must not report success: the line (with its "watchdog" field) is on stdout AND in
        # gpurun_out/bench_watchdog_line.json for launchers that drop the output of a failed run; the exit code says failure.
        os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()

    # ---- the launch floor, same run: an empty dispatch and a plain copy of the bytes one launch moves -----------
    if rank == 0:
        try:
            prewarm(lambda: ctx.diag_empty_launch(), 20)
            t_empty = timed_on_stream(lambda: ctx.diag_empty_launch(), 400)
            cp = lambda: ctx.diag_copy_dev(x.data_ptr(), y.data_ptr(), x.numel() * 4)
            prewarm(cp, 20)
            t_copy = timed_on_stream(cp, 400)
            fb = 8.0 * d * B
            floor = {"empty_dispatch_us": t_empty * 1e3, "copy_us": t_copy * 1e3, "copy_bytes": fb,
                     "copy_frac": fb / (t_copy * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "what": "back-to-back dependent launches on the kernels' stream, HIP events around 400 of them: an empty "
                             "4096-workgroup dispatch, and a 16-byte-per-lane copy of the 4 MiB in / 4 MiB out a B=4096 "
                             "transform launch moves (fz_diag_*)"}
        except Exception as exc:                      # a side leg must never take the headline down with it
            import traceback
            traceback.print_exc()
            floor = {"error": repr(exc)}

    # ---- many batches per dispatch (fz_ntt_multi): the same 4096-row batches, 1 / 2 / 4 / 8 of them per launch ---------
    stage[0] = "multi"
    if rank == 0:
        try:
            multi = {}
            nmax = 8
            xs = [x] + [x.clone() for _ in range(nmax - 1)]
            ys = [torch.empty_like(x) for _ in range(nmax)]
            for jobs in (1, 2, 4, 8):
                fj = [(xs[k].data_ptr(), ys[k].data_ptr(), B, False) for k in range(jobs)]
                ij = [(ys[k].data_ptr(), ys[k].data_ptr(), B, True) for k in range(jobs)]
                ctx.ntt_multi_dev(fj)
                ctx.ntt_multi_dev(ij)
                torch.cuda.synchronize(dev)
                assert all(torch.equal(ys[k], xs[k]) for k in range(jobs)), "fz_ntt_multi round trip differs"
                res = {}
                for name, jl in (("fwd", fj), ("inv", ij)):
                    # `ij` transforms in place, so its inputs change every launch: values stay arbitrary int32, timing is the same
                    fn = (lambda jl=jl: ctx.ntt_multi_dev(jl))
                    prewarm(fn, 20, inner=10)
                    ms = timed_on_stream(fn, 300)
                    gbs = jobs * 8.0 * d * B / (ms * 1e-3) / 1e9
                    res[name] = {"us_per_launch": round(ms * 1e3, 2), "GB/s": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
                multi[f"{jobs}x{B}"] = res
            multi["what"] = ("one fz_ntt_multi dispatch over 1/2/4/8 independent batches of 4096 rows (the job table travels in the "
                             "kernel arguments); back-to-back launches, HIP events on the stream, buffers re-used (cache-warm like the "
                             "headline step)")
            del xs, ys
        except Exception as exc:                      # a side leg must never take the headline down with it
            import traceback
            traceback.print_exc()
            multi = {"error": repr(exc)}

    # ---- the same steps over batches that are NOT cache-resident (informational) ----------------------------
    # The timed region above re-reads the same 4 MiB batch every step, so after the first step it lives in the L2s /
    # the 256 MB Infinity Cache.  Here the steps cycle through 32 batches (x, y, z: 384 MiB together).
    stage[0] = "cold"
    if rank == 0 and not args.no_two_stream:
        try:
            nb_c = nb_rot                                # the rotation the per-dispatch passes used
            zc = [t[2] for t in rot]
            pc = rot_p
            kc = max(nb_c, min(args.steps, 1000))
            torch.cuda.synchronize(dev)

            def run_cold(k):
                for i in range(k):
                    a_, b_, c_ = pc[i % nb_c]
                    fz_fwd(h, a_, b_, nB)
                    fz_inv(h, b_, c_, nB)
            gc = None
            if not args.no_graph:
                ctx.graph_begin()
                run_cold(kc)
                gc = ctx.graph_end()
            replay_c = gc.launch if gc else (lambda: run_cold(kc))
            prewarm(replay_c, args.prewarm_ms / 3, inner=1)
            reps_c = max(3, int(MIN_REGION_MS / (kc * 0.01)) + 1)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(reps_c):
                replay_c()
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            assert all(torch.equal(z_, x) for z_ in zc)
            cold = {"value": 2.0 * B * reps_c * kc / dt, "unit": "NTT/s", "ms_per_step": dt / (reps_c * kc) * 1e3,
                    "steps": reps_c * kc, "batches_cycled": nb_c,
                    "what": "the same step, cycling through 32 resident batches so that no step finds its input in a cache"}
            if gc:
                gc.destroy()
            del zc, pc
        except Exception as exc:                      # a side leg must never take the headline down with it
            import traceback
            traceback.print_exc()
            cold = {"error": repr(exc)}

    # ---- the same steps as two independent pipelines (two HIP streams / two branches of one graph; informational) ----
    stage[0] = "two_stream"
    if rank == 0 and not args.no_two_stream:
        try:
            side = torch.cuda.Stream(dev)
            cs = fusion_hip.Context(q, d, P["root"], P["inv_root"], device=dev_index)
            cs.set_stream(side.cuda_stream)
            y2, z2 = torch.empty_like(x), torch.empty_like(x)
            hs, y2p, z2p = cs._h, ctypes.c_void_p(y2.data_ptr()), ctypes.c_void_p(z2.data_ptr())
            k2 = min(args.steps, 1000) & ~1                    # steps per replay, half on each branch
            torch.cuda.synchronize(dev)

            def run2(k):
                for _ in range(k // 2):
                    fz_fwd(h, xp, yp, nB)
                    fz_inv(h, yp, zp, nB)
                    fz_fwd(hs, xp, y2p, nB)
                    fz_inv(hs, y2p, z2p, nB)
            g2 = None
            if not args.no_graph and k2 >= 2:
                g2 = torch.cuda.CUDAGraph()                      # fork/join across streams: torch's capture does the plumbing
                with torch.cuda.graph(g2, stream=stream):
                    side.wait_stream(stream)
                    run2(k2)
                    stream.wait_stream(side)
                replay = g2.replay
            else:
                k2 = 50

                def replay():
                    run2(k2)
            reps2 = max(10, int(MIN_REGION_MS / (k2 * 0.008)) + 1)      # a single replay would mostly measure its own start-up
            prewarm(replay, args.prewarm_ms / 3, inner=1 if g2 is not None else 50)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(reps2):
                replay()
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            assert torch.equal(z2, x) and torch.equal(z, x)
            two_stream = {"value": 2.0 * B * reps2 * k2 / dt, "unit": "NTT/s", "ms_per_step": dt / (reps2 * k2) * 1e3,
                          "steps": reps2 * k2,
                          "what": "steps issued alternately on two HIP streams with private output buffers"
                                  + (" (two branches of one hipGraph)" if g2 is not None else "")}
            del g2
            cs.close()
        except Exception as exc:                      # a side leg must never take the headline down with it
            import traceback
            traceback.print_exc()
            two_stream = {"error": repr(exc)}

    # ---- host-pointer path (PCIe-inclusive; informational, never `value`) -----------------------
    stage[0] = "pcie"
    if rank == 0:
        try:
            hx = x.cpu().numpy().copy()
            ctx.ntt_forward(hx)                         # scratch growth and first-touch of the staging outside the timing
            times = []
            for _ in range(8):
                t0 = time.perf_counter()
                ctx.ntt_forward(hx)
                times.append(time.perf_counter() - t0)
            dt = min(times)
            pcie = {"value": B / dt, "unit": "NTT/s", "ms_per_call": dt * 1e3, "ms_per_call_all": [round(t * 1e3, 3) for t in times],
                    "what": "fz_ntt_forward_host on 4096x256 host rows (copy of the input array + H2D + kernel + D2H), best of 8 calls"}
        except Exception as exc:                      # a side leg must never take the headline down with it
            import traceback
            traceback.print_exc()
            pcie = {"error": repr(exc)}

    # ---- large-batch asymptote of the same kernels (context for the roofline) ------------------
    stage[0] = "sweep"
    if rank == 0 and not args.no_sweep:
        try:
            for logb in (16, 18, 20):
                nb = 1 << logb
                x1 = torch.empty((nb, d), dtype=torch.int32, device=dev)
                ctx.fill_synthetic_dev(x1.data_ptr(), nb * d, 7)              # generated on the device (up to 1 GiB)
                # cold inputs: cycle through enough (input, output) pairs (>= 2 GiB together) that no launch finds its
                # input in the 256 MB Infinity Cache or an L2 from an earlier repetition
                pairs = max(1, -(-(2 << 30) // (2 * x1.numel() * 4)))
                xs = [x1] + [x1.clone() for _ in range(pairs - 1)]
                ys = [torch.empty_like(x1) for _ in range(pairs)]
                ptrs = [(a.data_ptr(), b.data_ptr()) for a, b in zip(xs, ys)]
                for name, fn in (("fwd", ctx.ntt_forward_dev), ("inv", ctx.ntt_inverse_dev)):
                    k = 0
                    t_end = time.perf_counter() + 0.04          # 40 ms of the same launches first (clock ramp, see --prewarm-ms)
                    while time.perf_counter() < t_end:
                        for _ in range(3):
                            fn(ptrs[k % pairs][0], ptrs[k % pairs][1], nb)
                            k += 1
                        torch.cuda.synchronize(dev)
                    reps = 10 if logb >= 20 else 100
                    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(stream)
                    for _ in range(reps):
                        fn(ptrs[k % pairs][0], ptrs[k % pairs][1], nb)
                        k += 1
                    b_.record(stream)
                    torch.cuda.synchronize(dev)
                    ms = a.elapsed_time(b_) / reps
                    gbs = 8.0 * d * nb / (ms * 1e-3) / 1e9
                    sweep[f"{name}_B2^{logb}"] = {"us": round(ms * 1e3, 2), "GB/s": round(gbs, 1),
                                                  "frac": round(gbs / HBM_PEAK_GBS, 4), "buffer_pairs_cycled": pairs}
                del xs, ys, x1, ptrs
        except Exception as exc:                      # a side leg must never take the headline down with it
            import traceback
            traceback.print_exc()
            sweep["error"] = repr(exc)

    # ---- every scheme kernel, cold operands, algorithmic bytes per unit from SURVEY 8d --------------------------
    stage[0] = "kernels"
    if rank == 0 and not args.no_kernel_table:
        try:
            from tools.kernel_table import measure
            torch.cuda.empty_cache()
            kernels = measure(ctx, P, quick=False)
            kernels["what"] = ("per-launch averages over operand sets carved out of a 2.25 GiB pool (every launch reads bytes no "
                               "launch has touched for >= 2 GiB of other traffic): HBM, not cache bandwidth; HIP events on the "
                               "kernels' stream; rocprofv3 per-kernel durations of the same launches: profiles/r02_*")
        except Exception as exc:                      # a side leg must never take the headline down with it
            import traceback
            traceback.print_exc()
            kernels = {"error": repr(exc)}
    barrier()

    # ---- sign + aggregate + verify (algebra cores; synthetic keys/messages) --------------------
    stage[0] = "sv"
    if not args.no_sign_verify:
        try:
            # per rank: 1024 signatures in GROUPS aggregates of (1024 / GROUPS) x world signers; 8 operand sets.  At least as many
            # aggregates as ranks, so that EVERY rank verifies (with 4 aggregates ranks 4-7 of an 8-GPU run would sit out the
            # verification step); world = 8: 8 aggregates of 128 x 8 = 1024 signers (capacity 2818)
            S, NSETS = 1024, 8
            GROUPS = max(4, world)
            while S % GROUPS:
                GROUPS += 1
            per = S // GROUPS
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
            # NSETS distinct operand sets (2.1 GB of keys + signatures): set i = the coefficients rotated by i positions.
            # A step works on ONE set, consecutive steps on consecutive sets, so no step finds its keys or signatures in
            # the 256 MB Infinity Cache (round 1 re-used one 262 MB set and read 79 % for sign_core out of the cache).
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
            # one flat int64 buffer: [GROUPS][l*d] aggregate partials followed by [GROUPS][d] target partials
            part = torch.zeros(GROUPS * (l * d + d), dtype=torch.int64, device=dev)
            part_t = part[GROUPS * l * d:]
            g_lo, g_hi = shard_range(GROUPS, rank, world)      # aggregates verified by this rank
            d_verd = torch.full((max(1, g_hi - g_lo),), -1, dtype=torch.int32, device=dev)   # verdict codes, read after the loop

            # the exchange step through the C ABI: rank 0's ncclUniqueId travels over the torch process group, every rank
            # joins with fz_comm_create; the all-reduce is then a node of the library's own graph.  If RCCL refuses (e.g.
            # the gloo rehearsal with ranks sharing one GPU), every rank falls back to torch.distributed together.
            comm, collective = None, "none (single rank: no exchange step)"
            if world > 1:
                uid = [None]
                if rank == 0:
                    try:
                        uid = [fusion_hip.comm_unique_id()]
                    except fusion_hip.FusionHipError as e:
                        uid = [repr(e)]
                dist.broadcast_object_list(uid, src=0)
                ok = 0.0
                if isinstance(uid[0], bytes) and backend == "nccl":
                    try:
                        comm = fusion_hip.Comm(ctx, world, rank, uid[0])
                        ok = 1.0 if comm.info()[0] == world else 0.0
                    except fusion_hip.FusionHipError as e:
                        sys.stderr.write(f"rank {rank}: fz_comm_create failed: {e}\n")
                if min_over_ranks(ok) < 1.0:
                    if comm is not None:
                        comm.destroy()
                    comm = None
                    collective = f"torch.distributed all_reduce ({backend}); fz_comm_* not usable in this run"
                else:
                    collective = f"fz_allreduce_i64 (ncclAllReduce int64 sum through the C ABI), comm of {comm.info()[0]} ranks"

            def sv_step(i):
                s_ = sets[i % NSETS]
                ctx.sign_core_dev(s_["sk_hat"].data_ptr(), s_["c_hat"].data_ptr(), s_["sig"].data_ptr(), S, l)
                # aggregate partials and the verification target's partials: one pass over this rank's signers, one launch
                ctx.aggregate_target_partial_batch_dev(s_["sig"].data_ptr(), s_["al_hat"].data_ptr(), s_["vkL"].data_ptr(),
                                                       s_["vkR"].data_ptr(), s_["c_hat"].data_ptr(), part.data_ptr(), l * d,
                                                       part_t.data_ptr(), d, GROUPS, per, l)
                if comm is not None:         # the ONE exchange step (RCCL over xGMI)
                    ctx.allreduce_i64_dev(comm, part.data_ptr(), part.numel())
                else:
                    allreduce_sum_i64(part)
                if g_hi > g_lo:          # verdicts straight from the int64 sums, left on the device: no host synchronisation
                    ctx.verify_partials_batch_async_dev(
                        A.data_ptr(), part[g_lo * l * d:].data_ptr(), l * d, part_t[g_lo * d:].data_ptr(), d,
                        g_hi - g_lo, l, P["beta_vf"], d, d_verd.data_ptr())

            for i in range(NSETS):
                sv_step(i)
                barrier()
                assert g_hi == g_lo or all(v == 0 for v in d_verd.tolist()), f"verify verdicts {d_verd.tolist()} on set {i}"
            # one graph = NSETS steps (every set once); refused together if any rank cannot capture (e.g. the collective)
            sv_graph, captured = None, 0.0
            if not args.no_graph and (world == 1 or comm is not None):
                try:
                    ctx.graph_begin()
                    try:
                        for i in range(NSETS):
                            sv_step(i)
                    finally:
                        sv_graph = ctx.graph_end()
                    captured = 1.0
                except fusion_hip.FusionHipError as e:
                    sys.stderr.write(f"rank {rank}: sign_verify capture failed: {e}\n")
                    sv_graph = None
            if min_over_ranks(captured) < 1.0:
                sv_graph = None

            def sv_round():
                if sv_graph is not None:
                    sv_graph.launch()
                else:
                    for i in range(NSETS):
                        sv_step(i)
            for _ in range(2):
                sv_round()
            for _ in range(40 if args.prewarm_ms > 0 else 0):      # count-based: every rank must issue the same collectives
                sv_round()
            barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            sv_round()
            torch.cuda.synchronize(dev)
            t_once = max(time.perf_counter() - t0, 1e-6)
            sv_rounds = int(max_over_ranks(max(3.0, -(-2 * MIN_REGION_MS * 1e-3 // t_once))))
            barrier()
            t0 = time.perf_counter()
            for _ in range(sv_rounds):
                sv_round()
            barrier()
            dt = max_over_ranks(time.perf_counter() - t0)
            assert g_hi == g_lo or all(v == 0 for v in d_verd.tolist()), "a verification failed inside the timed region"
            sv_steps = sv_rounds * NSETS
            sv = {"value": S * world * sv_steps / dt, "unit": "signatures signed+aggregated+verified per s",
                  "signatures_per_rank": S, "aggregates": GROUPS, "signers_per_aggregate": per * world,
                  "steps": sv_steps, "ms_per_step": dt / sv_steps * 1e3, "operand_sets_cycled": NSETS,
                  "launch": "hipGraph replay of %d steps (fz_graph_*)" % NSETS if sv_graph is not None else "one by one",
                  "collective": collective,
                  "algorithmic_GB/s_per_gpu": S * ((3 * l + 1) + (l + 5)) * 4 * d * sv_steps / dt / 1e9,
                  "hbm_frac_per_gpu": S * ((3 * l + 1) + (l + 5)) * 4 * d * sv_steps / dt / 1e9 / HBM_PEAK_GBS,
                  "note": "algebra cores only: sign_core, aggregate + target partials (one pass, one launch), int64 all-reduce, "
                          "verification from the int64 sums -- 3 kernel launches (+ the collective) per step; every step works on "
                          "the next of 8 operand sets (2.1 GB), so keys and signatures come from HBM; host hashing of str(vk) excluded"}
            sv_graph_used = sv_graph is not None
            if sv_graph is not None:
                sv_graph.destroy()
            # BASELINE configs[2]: 1024 independent keygen + sign per step (keygen_core: 2*l transforms + two A.s
            # products per key; sign_core: sigma = L*c + R), no exchange step: ranks are independent
            def ks_step(i):
                s_ = sets[i % NSETS]
                ctx.keygen_core_dev(A.data_ptr(), s_["coef"].data_ptr(), s_["sk_hat"].data_ptr(), s_["vk"].data_ptr(), S, l)
                ctx.sign_core_dev(s_["sk_hat"].data_ptr(), s_["c_hat"].data_ptr(), s_["sig"].data_ptr(), S, l)
            for i in range(100 if args.prewarm_ms > 0 else 2):
                ks_step(i)
            barrier()
            ks_steps = NSETS * max(3, int(2 * MIN_REGION_MS / (NSETS * 0.15)) + 1)
            t0 = time.perf_counter()
            for i in range(ks_steps):
                ks_step(i)
            barrier()
            dt = max_over_ranks(time.perf_counter() - t0)
            sv["keygen_sign"] = {"value": S * world * ks_steps / dt, "unit": "keygen+sign per s", "per_rank": S, "steps": ks_steps,
                                 "ms_per_step": dt / ks_steps * 1e3, "operand_sets_cycled": NSETS,
                                 "algorithmic_GB/s_per_gpu": S * ((4 * l + 2) + (3 * l + 1)) * 4 * d * ks_steps / dt / 1e9,
                                 "hbm_frac_per_gpu": S * ((4 * l + 2) + (3 * l + 1)) * 4 * d * ks_steps / dt / 1e9 / HBM_PEAK_GBS,
                                 "note": "configs[2]: keygen_core + sign_core on 1024 distinct synthetic keys per rank; sign reads the "
                                         "sk_hat keygen has just written (174 MB: part of it may still sit in the Infinity Cache), "
                                         "coefficients come from HBM (8 sets rotated)"}
            if world > 1:            # what EVERY rank did in this leg, as the ranks themselves report it
                mine = {"rank": rank, "collective": collective, "aggregates_verified": int(g_hi - g_lo), "verdicts_ok": True,
                        "graph": sv_graph_used}
                allr = [None] * world
                dist.all_gather_object(allr, mine)
                sv["ranks"] = allr
                sv["every_rank_verified"] = all(r_["aggregates_verified"] > 0 for r_ in allr)
                sv["one_collective_path"] = len({r_["collective"] for r_ in allr}) == 1
            if comm is not None:
                barrier()
                comm.destroy()
            del sets
        except Exception as exc:                      # a side leg must never take the headline down with it
            import traceback
            traceback.print_exc()
            sv = {"error": repr(exc)}

    # ---- end to end through the array API: host hashing (C pipeline) + device algebra ------------------
    stage[0] = "e2e"
    if rank == 0 and not args.no_sign_verify and not args.no_end_to_end:
        try:
            import fusion.fusion as F
            from fusion_hip.scheme import BatchScheme
            params = F.fusion_setup(SECPAR, 2026)
            bs = BatchScheme(params, device=dev_index)
            bs.ctx.set_stream(stream.cuda_stream)
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
            t_keygen, (sk_e, vk_e, vk_dev) = best_of(lambda: bs.keygen_batch(seeds, device=True, keep_vk=True), drop_keys)   # keys stay in HBM
            bs.sign_batch(sk_e, vk_dev, msgs, device=True).free()          # scratch growth outside the timing
            t_sign, sig_e = best_of(lambda: bs.sign_batch(sk_e, vk_dev, msgs, device=True), lambda r: r.free())       # signatures stay in HBM
            # the same signatures with the challenge pipeline on the HOST threads (round 1's path), for the split
            bs.device_hash = False
            t0 = time.perf_counter()
            bs.sign_batch(sk_e, vk_e, msgs, device=True).free()
            t_sign_host = time.perf_counter() - t0
            bs.device_hash = True
            t0 = time.perf_counter()
            pre_only = bs.challenges_dev(vk_dev, msgs)[0]
            torch.cuda.synchronize(dev)
            t_chal = time.perf_counter() - t0
            pre_only.free()
            t_agg, agg_e = best_of(lambda: bs.aggregate(vk_e, msgs, sig_e))
            t_ver, (ok, why) = best_of(lambda: bs.verify(vk_e, msgs, agg_e))
            assert ok, why
            bs.aggregate_verify(vk_e, msgs, sig_e)
            t_av, (agg_av, (ok, why)) = best_of(lambda: bs.aggregate_verify(vk_e, msgs, sig_e))     # one hash_ag for both, one pass over the signatures
            assert ok and np.array_equal(agg_av, agg_e), why
            # a larger batch of signatures: the device pipeline is a latency chain of ~108 Keccak permutations per signer,
            # the same ~0.7 ms for any batch up to 32 768 signers (two lanes per signer, one wave per SIMD)
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
            many["what"] = ("BatchScheme.aggregate_many / verify_many on the same 1024 signatures split into G aggregates of N signers: "
                            "G independent hash_ag sponges on G host threads, ONE ragged launch for the G aggregates, ONE for the G "
                            "verifications (reference call pattern: one aggregate()/verify() per aggregate, fusion.py:655, :680)")
            sig_e.free()
            sk_e.free()
            vk_dev.free()
            e2e = {"signatures": n_e2e, "host_threads": bs.threads, "timing": "best of 3 calls after one untimed call of the same size",
                   "keygen_per_s": n_e2e / t_keygen, "many_aggregates": many,
                   "sign_per_s": n_e2e / t_sign,
                   "sign_split": {"device_challenge_pipeline_ms": t_chal * 1e3, "whole_sign_batch_ms": t_sign * 1e3,
                                  "sign_per_s_with_host_challenge_pipeline": n_e2e / t_sign_host,
                                  "what": "sign_batch = upload of the message bytes + device: SHA3-256 of the messages, text of str(vk), "
                                          "SHAKE-256, decoder, NTT (fz_challenge_hat_msgs_dev) + sign_core; the host-pipeline figure runs "
                                          "pre-hash + serialiser + SHAKE + decoder on the host threads instead (round 1)"},
                   "sign_per_s_at_16384_signatures": n_big / t_sign_big,
                   "aggregate_per_s": n_e2e / t_agg, "verify_per_s": n_e2e / t_ver,
                   "sign_plus_verify_per_s": n_e2e / (t_sign + t_agg + t_ver),
                   "aggregate_verify_per_s": n_e2e / t_av, "sign_plus_aggregate_verify_per_s": n_e2e / (t_sign + t_av),
                   "aggregate_verify_what": "BatchScheme.aggregate_verify: aggregate() and verify() of its result with ONE hash_ag (the serial "
                                            "sponge runs once instead of twice) and one pass over the signatures",
                   "note": "BatchScheme with device-resident keys and signatures: reference-exact MT19937 sampling of the secret "
                           "polynomials, the per-signer challenge pipeline (message pre-hash included) and all algebra on the device; aggregate and "
                           "verify are bounded by hash_ag, ONE serial SHAKE-256 over ~13.5 KB per signer on the host by construction "
                           "(fusion.py:632-652)"}
        except Exception as exc:                      # a side leg must never take the headline down with it
            import traceback
            traceback.print_exc()
            e2e = {"error": repr(exc)}

    # ---- end to end, SHARDED: aggregate() + verify() of ONE aggregate of 1024 signers with the signers (and their signatures)
    # spread over the ranks (fusion_hip.dist.ShardedScheme: global sort + hash_ag on every rank, alpha scattered, ONE pass over
    # the local signatures, ONE all-reduce of int64 partials, verification from the sums) -- BASELINE configs[3]
    stage[0] = "e2e_sharded"
    e2e_sh = None
    if not args.no_sign_verify and not args.no_end_to_end:
        try:
            from fusion_hip.dist import ShardedScheme, TorchCollective
            from fusion_hip.scheme import BatchScheme
            params = F.fusion_setup(SECPAR, 2026)
            bs = BatchScheme(params, device=dev_index)
            bs.ctx.set_stream(stream.cuda_stream)
            n_all = 1024
            seeds = [10_000 + 2 * i for i in range(n_all)]
            msgs = [f"synthetic message {i:06d}" for i in range(n_all)]
            lo_, hi_ = shard_range(n_all, rank, world)
            sk_l, vk_l, vk_ld = bs.keygen_batch(seeds[lo_:hi_], device=True, keep_vk=True)
            sig_l = bs.sign_batch(sk_l, vk_ld, msgs[lo_:hi_], device=True)         # this rank's signatures stay in its HBM
            if world > 1:                                                          # verification keys are public: everyone gets all
                parts = [None] * world
                dist.all_gather_object(parts, vk_l)
                vk_all = np.concatenate(parts)
            else:
                vk_all = vk_l
            sh = ShardedScheme(bs, rank, world, TorchCollective(bs.ctx, dev_index))
            sh.aggregate_verify_sharded(vk_all, msgs, sig_l)                      # scratch growth, first-use tables
            barrier()
            t0 = time.perf_counter()
            agg_s, verdict_s = sh.aggregate_verify_sharded(vk_all, msgs, sig_l)
            barrier()
            t_sh = max_over_ranks(time.perf_counter() - t0)
            assert verdict_s == (True, ""), verdict_s
            t0 = time.perf_counter()
            v2 = sh.verify_sharded(vk_all, msgs, agg_s)
            barrier()
            t_vs = max_over_ranks(time.perf_counter() - t0)
            assert v2 == (True, ""), v2
            e2e_sh = {"signers": n_all, "ranks": world, "signers_per_rank": hi_ - lo_,
                      "aggregate_plus_verify_per_s": n_all / t_sh, "aggregate_plus_verify_ms": t_sh * 1e3,
                      "verify_per_s": n_all / t_vs, "verify_ms": t_vs * 1e3,
                      "collective": (f"torch.distributed all_reduce ({dist.get_backend()})" if world > 1 else "none (single rank)"),
                      "what": "ShardedScheme.aggregate_verify_sharded / verify_sharded: every rank sorts and hashes the whole key list "
                              "(hash_ag is one serial SHAKE-256 over all signers, the same on every rank), transforms only its block of "
                              "alpha, makes one pass over its block of signatures, then ONE all-reduce of l*d + d int64; max over ranks"}
            for b in (sk_l, vk_ld, sig_l):
                b.free()
        except Exception as exc:                      # a side leg must never take the headline down with it
            import traceback
            traceback.print_exc()
            e2e_sh = {"error": repr(exc)}
        if isinstance(e2e, dict):
            e2e["sharded"] = e2e_sh
        elif e2e is None and rank == 0:
            e2e = {"sharded": e2e_sh}

    if _BAILING.is_set():                               # the watchdog fired while a leg was (slowly) finishing: its line stands, and
        time.sleep(15.0)                                # its thread ends the process with code 3 -- never a second line, never rc 0
        os._exit(3)
    stage[0] = "cpu_baseline"
    done.set()                                          # the bounded CPU sample is not under the watchdog
    if rank == 0:
        out = build_line()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
            sch = out["cpu_baseline"].get("scheme") or {}
            if "sign_plus_verify_per_s" in sch and isinstance(out.get("sign_verify"), dict) and "value" in out["sign_verify"]:
                # the second half of the metric, CPU beside GPU: same cores -- sign, aggregate, verify
                # (and keygen + sign for configs[2]) -- in the pure-Python port, per signature
                out["sign_verify"]["cpu_baseline"] = {
                    "value": sch["sign_plus_verify_per_s"], "unit": "signatures signed+aggregated+verified per s", "cores": 1, "kind": "port",
                    "all_cores": (sch.get("all_cores") or {}).get("sign_plus_verify_per_s"), "all_cores_count": (sch.get("all_cores") or {}).get("cores"),
                    "sign_per_s": sch["sign_per_s"], "aggregate_signatures_per_s": sch["aggregate_signatures_per_s"],
                    "verify_signatures_per_s": sch["verify_signatures_per_s"], "sample": sch["sample"]}
                if isinstance(out["sign_verify"].get("keygen_sign"), dict):
                    out["sign_verify"]["keygen_sign"]["cpu_baseline"] = {
                        "value": sch["keygen_plus_sign_per_s"], "unit": "keygen+sign per s", "cores": 1, "kind": "port",
                        "all_cores": (sch.get("all_cores") or {}).get("keygen_plus_sign_per_s"), "keygen_per_s": sch["keygen_per_s"],
                        "sign_per_s": sch["sign_per_s"], "sample": sch["sample"]}
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        barrier()                                       # rank 0's single-rank legs are done: leave together
        dist.destroy_process_group()


_BAILING = threading.Event()       # set by the watchdog: from then on an exception in the main thread is a peer leaving

if __name__ == "__main__":
    try:
        main()
    except BaseException:
        if not _BAILING.is_set():
            raise
        time.sleep(10.0)           # the watchdog thread prints the line and ends the process (exit code 3: a hang is a failure)
        os._exit(3)
