#!/usr/bin/env python3
"""bench.py -- headline benchmark of the algebra hot path on MI355X.

Workload (BASELINE.json configs[1]): secpar=256, batches of 4096 independent degree-256 polynomials resident in HBM.
A STEP is one pass of the hot path over one batch: forward NTT of the batch, then inverse NTT of the result (reference
algebra/ntt.py:216-291 and :294-377).  Consecutive steps work on consecutive batches of a rotation of NBATCH batches
(x_i -> y_i -> z_i, 768 MiB together), so every forward transform reads its input from HBM, not from a cache: `value` is an
HBM number.  The steps are SOFTWARE-PIPELINED over D batches (--depth, 16: the job table's 32 entries): batches are independent, the forward transforms
of batches i+1 .. i+D and the inverse transforms of batches i-D+1 .. i share ONE launch (fz_ntt_multi: a table of 2 D jobs in
the kernel arguments = 2 D x 4096 transforms = D x 16 MiB of algorithmic bytes per launch: 256 MiB at D = 16); D steps are one
launch, a run of steps opens with a forward-only launch and closes with an inverse-only one, and every transform of every
batch is done exactly once (z == x is checked after every region).  `--depth 1` is round 4's form (one forward + one inverse
job per launch), `--two-launch` the un-pipelined one (fz_ntt_forward, then fz_ntt_inverse: two launches of 8 MiB per step;
by default the side leg `two_launch_step`).  The steps run as S CHAINS on S HIP streams (--streams, 2; a context + stream
each, created before anything else in the process, the rotating batches dealt in contiguous shares): S launches in flight,
which hides each launch's ramp and drain behind the other's body.  The K steps (--steps) are dealt to the chains, recorded
into one hipGraph per chain (fz_graph_*) and replayed R times inside the timed region, R chosen so that the region lasts
>= 20 ms whatever K is.  `value` = NTTs per second over the whole job (forward and inverse each count, all chains, summed
over all ranks); `ms_per_step` = elapsed / steps timed.  With --gpus N every rank owns its own batches (weak scaling, no
data-path collective for the transforms).

Output: the LAST stdout line is ONE compact strict-JSON object (< 4 KB: compact_line()); everything measured, with its
prose, goes to gpurun_out/bench_full.json.  The compact line carries
  roofline      the dominant kernel (the 2 D-job transform launch: 8*d algorithmic bytes per transform), per launch with ONE
                launch in flight: HIP events around a dense one-stream region of the same steps over the same rotating
                (cold) batches / the launches in it -- the figure rocprofv3 --kernel-trace --stats reproduces (it serialises
                streams); roofline.chip: the S-stream timed region of `value` itself (bytes of all launches / its
                duration), and roofline.chip.device_clock: the same figure from timestamps the KERNEL takes on the chip's
                100 MHz reference counter (fz_diag_stamps_*: every workgroup's entry and exit) -- launches of different
                streams really overlap, by the device's own clock, which no profiler trace can show
  cpu_baseline  the pure-Python port of the reference's algebra path (oracle/oracle.py py_*), one host core, bounded sample
  sign_verify / keygen_sign   the metric's second half (algebra cores, sharded over the ranks, ONE int64 all-reduce per step,
                on a second stream beside the next step's kernels when there is a communicator)
  end_to_end    keygen / sign / aggregate / verify per second through BatchScheme, hashing included (benchmarks/benchmarks.py:37-141)
  warm_replay   the un-pipelined step re-reading ONE batch (cache-resident): the side number, never `value`
  ranks         per rank: device, PCI bus id, ranks RCCL itself counted in the communicator, RCCL version, collective self-check
`--full` adds the side legs of tools/bench_legs.py (launch floor, fz_ntt_multi, PCIe, large-batch sweep, cold
per-kernel table, ShardedScheme end to end) to the full file.

Launch: `python bench.py --gpus N` starts its own N ranks (child processes, before this process has touched a GPU) when
WORLD_SIZE is not set; under torch.distributed.run it uses the ranks it is given.
"""
import argparse
import glob
import json
import math
import os
import re
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
MIN_REGION_MS = 20.0           # the secondary legs' timed regions (many legs: the default run stays within minutes)
ROOFLINE_REGION_MS = 250.0     # the one-stream region `roofline.frac` / `avg_launch_us` come from (~4 600 launches of 54 us)
HEADLINE_REGION_MS = 1000.0    # the region `value` comes from: long enough for the driver's clock and its GPU-busy sampler to see it
NBATCH = 64                    # batches in the rotation: 3 x 64 x 4 MiB = 768 MiB, three times the 256 MB Infinity Cache
LINE_LIMIT = 4096              # bytes of the compact line (tests/test_bench_line.py)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--full", action="store_true", help="also run the side legs of tools/bench_legs.py (full file only)")
    ap.add_argument("--single-rank-comm", action="store_true",
                    help="N = 1 only: create a one-rank RCCL communicator so that sign_verify runs its exchange-step code path")
    ap.add_argument("--exchange-standin-us", type=int, default=0,
                    help="N = 1 only: a one-wave kernel of this duration in place of the multi-GPU all-reduce in sign_verify "
                         "(fz_diag_delay): what the second stream hides of an exchange step of known latency")
    ap.add_argument("--sign-then-aggregate", action="store_true",
                    help="sign_verify: fz_sign_core, then fz_aggregate_target_partial_batch (round 3's two launches) instead of the one pass")
    ap.add_argument("--exchange", choices=("auto", "all-reduce", "reduce-scatter"), default="auto",
                    help="sign_verify: the exchange step's collective (auto: a short calibration picks the reduce-scatter if it is >= 20 %% faster)")
    ap.add_argument("--verify-every", type=int, default=0,
                    help="sign_verify: steps per verification launch (default: 8 without the exchange overlap, 1 with it)")
    ap.add_argument("--verify-per-step", action="store_true",
                    help="sign_verify: one verification launch per step (round 3's form) instead of one per 8 steps")
    ap.add_argument("--no-exchange-overlap", action="store_true",
                    help="sign_verify: the all-reduce on the compute stream (round 3's form) instead of a second stream")
    ap.add_argument("--headline-only", action="store_true",
                    help="timed region + per-dispatch roofline passes only (what tools/collect_profiles.sh runs under rocprofv3)")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams (a context each) that walk disjoint shares of the rotating batches side by side: launches of "
                         "different streams overlap on the chip (1: one launch in flight at a time)")
    ap.add_argument("--depth", type=int, default=16,
                    help="batches per launch of the software pipeline: D forward jobs (batches i+1..i+D) + D inverse jobs (batches "
                         "i-D+1..i) in one fz_ntt_multi dispatch = D x 16 MiB of algorithmic bytes (1: round 4's two-job launch)")
    ap.add_argument("--no-stamps", action="store_true", help="skip the device-timestamp pass (roofline.chip.device_clock)")
    ap.add_argument("--two-launch", action="store_true",
                    help="headline step as two launches (fz_ntt_forward, fz_ntt_inverse) instead of the software-pipelined one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sign-verify", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the end-to-end leg (BatchScheme: keygen / sign / aggregate / verify per second, hashing included)")
    ap.add_argument("--no-graph", action="store_true", help="launch the timed steps one by one instead of replaying hipGraphs")
    ap.add_argument("--sample-every", type=int, default=1,
                    help="bind begin/end events to every k-th dispatch in the instrumented passes (1: every dispatch carries "
                         "its own completion signal, as under rocprofv3 --kernel-trace)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the pure-Python baseline sample")
    ap.add_argument("--region-ms", type=float, default=HEADLINE_REGION_MS,
                    help="minimum duration of the headline's timed region: the K steps are replayed until it is reached")
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="untimed run of the same steps before the W warmup steps: after idle the GPU needs tens of "
                         "milliseconds of load to reach its steady clocks")
    ap.add_argument("--watchdog-s", type=float, default=420.0,
                    help="if the legs after the headline have not finished by then, rank 0 prints the line with what it has "
                         "(\"watchdog\" names the leg) and every rank leaves WITH EXIT CODE 3")
    ap.add_argument("--full-out", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"))
    ap.add_argument("--cpu-worker", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-signers", type=int, default=0, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------
# the compact line (pure function: tests/test_bench_line.py runs it on canned inputs)
# ---------------------------------------------------------------------------------------------------------
def _num(v, digits=6):
    """a JSON-safe number with `digits` significant digits (NaN / inf -> None: the line is STRICT JSON)"""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, int):
        return v
    try:
        f = float(v)
    except (TypeError, ValueError):
        return None
    if not math.isfinite(f):
        return None
    if f.is_integer() and abs(f) < 2.0 ** 53:
        return int(f)                                    # byte counts and the like stay exact
    return float(f"{f:.{digits}g}")


def _pick(src, keys, digits=6):
    out = {}
    for k in keys:
        if isinstance(src, dict) and k in src:
            v = src[k]
            out[k] = _num(v, digits) if isinstance(v, (int, float)) and not isinstance(v, bool) else v
    return out


def compact_line(full):
    """full result object -> the one line the driver parses: strict JSON, no prose, < LINE_LIMIT bytes.
    Counterpart of the reference harness' five numbers per N (benchmarks/benchmarks.py:144-171)."""
    out = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data"))
    out["config"] = _pick(full.get("config") or {}, ("workload", "batch", "degree", "modulus", "batches_rotated", "steps_per_launch",
                                                     "streams", "step", "launch"))
    roof = full.get("roofline") or {}
    out["roofline"] = _pick(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "bytes_per_launch",
                                   "units_per_launch", "avg_launch_us", "in_flight", "operands", "timing"), 5)
    if isinstance(out["roofline"].get("traffic_source"), str):      # `traffic` is NOT measured in this run: say where it comes from
        out["roofline"]["traffic_source"] = out["roofline"]["traffic_source"][:80]
    dck = ("frac", "in_flight", "launch_us", "gap_us", "span_us")
    if isinstance(roof.get("device_clock"), dict):
        out["roofline"]["device_clock"] = _pick(roof["device_clock"], dck, 4)
    if isinstance(roof.get("chip"), dict):
        out["roofline"]["chip"] = _pick(roof["chip"], ("streams", "achieved", "frac", "unit", "launch_us_in_flight", "launches_timed"), 5)
        if isinstance(roof["chip"].get("device_clock"), dict):
            out["roofline"]["chip"]["device_clock"] = _pick(roof["chip"]["device_clock"], dck, 4)
    cb = full.get("cpu_baseline")
    out["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample"), 5) if isinstance(cb, dict) else None
    if isinstance(cb, dict) and isinstance(cb.get("sample"), str):
        out["cpu_baseline"]["sample"] = cb["sample"][:160]
    for name, keys in (("sign_verify", ("value", "unit", "ms_per_step", "moved_frac_per_gpu", "aggregates", "signers_per_aggregate",
                                        "scaling", "collective", "cpu_value", "error")),
                       ("sign_verify_1x1024", ("value", "ms_per_step", "moved_frac_per_gpu", "aggregates", "signers_per_aggregate")),
                       ("keygen_sign", ("value", "unit", "ms_per_step", "hbm_frac_per_gpu", "cpu_value", "error")),
                       ("single_stream", ("value", "unit", "ms_per_step", "frac")),
                       ("two_launch_step", ("value", "unit", "ms_per_step", "frac")),
                       ("warm_replay", ("value", "unit", "ms_per_step")),
                       ("end_to_end", ("signatures", "keygen_per_s", "sign_per_s", "aggregate_per_s", "verify_per_s", "queue_pairs_per_s", "error"))):
        src = full.get(name)
        if isinstance(src, dict):
            sub = _pick(src, keys, 5)
            for k in ("collective", "error", "unit", "scaling"):
                if isinstance(sub.get(k), str):
                    sub[k] = sub[k][:96]
            if sub:
                out[name] = sub
    ranks = full.get("ranks")
    if isinstance(ranks, list):
        # what every rank reports alike once ("same"), what differs per rank ("each": a rank whose communicator or self-check
        # disagrees with the others shows up there): eight ranks in 600 bytes instead of 1500
        rows = [_pick(r, ("rank", "device_index", "pci_bus_id", "world_size_seen", "backend", "rccl_nranks", "rccl_version", "collective_check"))
                for r in ranks if isinstance(r, dict)]
        same = {k: v for k, v in (rows[0].items() if rows else ()) if k not in ("rank", "device_index", "pci_bus_id") and all(r.get(k) == v for r in rows)}
        out["ranks"] = {"n": len(rows), "same": same, "each": [{k: v for k, v in r.items() if k not in same} for r in rows]}
    for k in ("watchdog", "full"):
        if full.get(k):
            out[k] = str(full[k])[:200]
    text = json.dumps(out, allow_nan=False, separators=(",", ":"))
    # the limit is a contract: shed the optional blocks (never the required keys) rather than print an unparsable line
    for drop in ("warm_replay", "two_launch_step", "single_stream", "ranks", "keygen_sign", "sign_verify_1x1024", "end_to_end"):
        if len(text) < LINE_LIMIT:
            break
        out.pop(drop, None)
        text = json.dumps(out, allow_nan=False, separators=(",", ":"))
    return text


def newest_profile(pattern):
    """newest committed profiles/r<NN>_<pattern> by round number (never a hard-coded round), or None"""
    best = None
    for p in glob.glob(os.path.join(ROOT, "profiles", "r*_" + pattern)):
        m = re.match(r"r(\d+)_", os.path.basename(p))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), p)
    return best[1] if best else None


def pmc_traffic(kernel_prefix="ntt_fwd4<8", rows=None):
    """HBM-side bytes per launch of the dominant kernel from the newest committed PMC summary (rocprofv3 --pmc passes cannot
    be taken live: tools/collect_profiles.sh, tools/pmc_summary.py) -> (bytes or None, source).  rows: the launch's transforms
    (one kernel serves launches of several sizes: only the entry of THIS size counts)"""
    path = newest_profile("pmc_ntt.json")
    if not path:
        return None, "no profiles/r*_pmc_ntt.json"
    try:
        with open(path) as fh:
            ks = json.load(fh)["kernels"]
        vals = [v["traffic_bytes_per_launch"] for k, v in ks.items()
                if k.startswith(kernel_prefix) and "bench launch" in k and "traffic_bytes_per_launch" in v
                and (rows is None or v.get("rows_per_launch") == rows)]
        if not vals:
            return None, f"{os.path.relpath(path, ROOT)} has no entry for {kernel_prefix}"
        return max(vals), os.path.relpath(path, ROOT)
    except Exception as e:                                   # a stale or malformed file is not the run's failure
        return None, f"{os.path.relpath(path, ROOT)}: {e!r}"


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


def _pool(extra, workers, timeout):
    """`workers` child processes of this script (host only) -> the whitespace-split numbers each printed"""
    import subprocess
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(i)] + extra,
                              stdout=subprocess.PIPE, text=True) for i in range(workers)]
    res = []
    for pr in procs:
        try:
            text, _ = pr.communicate(timeout=timeout)
            res.append([float(v) for v in text.split()])
        except Exception:
            pr.kill()
    return res


def cpu_baseline(seconds):
    """Pure-Python port (lists of ints, one cent() per reference cent call): ONE core (the reference is
    single-threaded), then the same loop in one process per host core this job may use; then the metric's second half
     on a 64-signer aggregate."""
    done, dt = _cpu_worker((0, seconds))
    ncpu = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = ncpu
    out = {"value": done / dt, "unit": "NTT/s", "cores": 1, "kind": "port",
           "sample": f"{done // 2} rows of the {B}-row batch, forward+inverse degree-256 NTT each, pure-Python port, {dt:.1f} s on 1 core of {ncpu}"}
    # a one-GPU box of this pool is a 16-CPU share of a 256-CPU host that seven other jobs use, and its process guard counts
    # children: the pool is min(schedulable CPUs, FZ_BENCH_CPU_WORKERS = 16); the count actually used is reported as `cores`
    workers = max(1, min(usable, int(os.environ.get("FZ_BENCH_CPU_WORKERS", "16"))))
    try:
        res = _pool(["--cpu-seconds", str(seconds / 3)], workers, seconds + 120)
        if not res:
            raise RuntimeError("no worker finished")
        out["all_cores"] = {"value": sum(n / t for n, t in res), "unit": "NTT/s", "cores": len(res),
                            "host_logical_cpus": ncpu, "usable_by_this_process": usable,
                            "sample": f"{len(res)} processes x {seconds / 3:.1f} s of the same loop"}
    except Exception as e:                                   # the single-core figure stands on its own
        out["all_cores"] = {"error": repr(e)}

    def rates(n, tk, ts, ta, tv):
        return {"keygen_per_s": n / tk, "sign_per_s": n / ts, "aggregate_signatures_per_s": n / ta, "verify_signatures_per_s": n / tv,
                "sign_plus_verify_per_s": n / (ts + ta + tv), "keygen_plus_sign_per_s": n / (tk + ts)}
    try:
        n1 = int(os.environ.get("FZ_BENCH_CPU_SIGNERS", "64"))
        n, tk, ts, ta, tv = _cpu_worker((0, n1, "scheme"))
        sch = dict(rates(n, tk, ts, ta, tv), cores=1, kind="port", unit="per second",
                   sample=f"one aggregate of {n} synthetic signers at secpar {SECPAR}: keygen {tk:.2f} s, sign {ts:.2f} s, aggregate "
                          f"{ta:.2f} s, verify {tv:.3f} s on 1 core (oracle.py_keygen_core / py_sign_core / py_aggregate_core / py_verify_core)")
        npool = max(2, n1 // 4)
        res = _pool(["--cpu-signers", str(npool)], workers, 600)
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
def rendezvous_port():
    """A free TCP port for rank 0's store, chosen OUTSIDE the kernel's ephemeral range.  A port from bind(("127.0.0.1", 0)) lies
    inside that range, and a rank that starts connecting before rank 0 listens can be handed that very number as its SOURCE
    port (TCP self-connect): it then talks to itself, rank 0's bind fails with EADDRINUSE and the others wait for a store that
    never answers -- (tests/_ranks.py carries the same function)."""
    import random
    import socket
    lo, hi = 32768, 60999
    try:
        with open("/proc/sys/net/ipv4/ip_local_port_range") as fh:
            lo, hi = (int(x) for x in fh.read().split())
    except (OSError, ValueError):
        pass
    pool = range(20000, min(lo, 32000)) if lo > 21000 else range(hi + 1, 65000)
    for _ in range(200):
        port = random.choice(pool)
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", port))
            except OSError:
                continue
            return port
    raise RuntimeError("no free rendezvous port")


def self_launch(args):
    import subprocess
    n = args.gpus
    port = rendezvous_port()
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
        sys.exit(failed[1] if 0 < failed[1] < 126 else 1)      # (4: the C-ABI communicator is unusable; 5: a collective summed wrongly)
    if time.time() > deadline:
        sys.exit("bench.py: ranks did not finish in time")
    result = [ln for ln in line.splitlines() if ln.startswith('{"metric"')]      # a backend may chat on stdout (gloo does)
    if not result:
        sys.exit("bench.py: rank 0 printed no result line")
    sys.stdout.write(result[-1] + "\n")
    sys.stdout.flush()


def write_full(path, full):
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(full, fh, indent=1, default=repr)
            fh.write("\n")
        return os.path.relpath(path, ROOT)
    except OSError as e:
        return f"not written: {e!r}"


def main():
    args = parse()
    if args.cpu_worker is not None:                      # child of cpu_baseline(): host only
        if args.cpu_signers:
            print(*_cpu_worker((args.cpu_worker, args.cpu_signers, "scheme")))
        else:
            print(*_cpu_worker((args.cpu_worker, args.cpu_seconds)))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)                         # before anything in this process touches a GPU

    # every stream of the timed region on a hardware queue of its own: the HIP runtime maps a process's streams onto 4 hardware
    # queues by default (read when it starts: before the first HIP call), torch and the contexts' diagnostic streams take some,
    # and two chains that share a queue run one after the other (measured at 4 streams: two chains 15 ms, two 23.5 ms)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(8, 2 * args.streams)))
    # host threads on ONE NUMA node (the GPU's), before anything initialises the GPU (sysfs + sched_setaffinity only): a
    # launching thread the scheduler moves to the other socket launches at 5.1-7.5 us per dispatch instead of 4.4-4.8
    # (profiles/r02_numa_placement.txt)
    from fusion_hip.numa import pin_to_gpu_node
    placement = pin_to_gpu_node(int(os.environ.get("LOCAL_RANK", "0")))

    import ctypes
    import types
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
    dev_index = local_rank % ndev                 # one rank per GPU; the modulo only matters for the gloo rehearsal
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    ps = F.PREFIX_PARAMETERS[SECPAR]
    P = {"q": ps["modulus"], "d": ps["degree"], "root": ps["root"], "inv_root": ps["inv_root"], "rank": ps["num_rows_sk"],
         "omega_ch": ps["omega_ch"], "omega_ag": ps["omega_ag"], "capacity": ps["capacity"], "beta_vf": ps["beta_vf"]}
    q, d, l = P["q"], P["d"], P["rank"]
    # The headline's chains FIRST: a context + HIP stream each, and one empty kernel on each, BEFORE anything else in the
    # process creates streams.  The HIP runtime hands hardware queues to streams in the order they first need one; RCCL (a
    # torch.distributed "nccl" group, or fz_comm_create) takes several for its own streams, and chains created after it end up
    # sharing queues with them or with each other -- measured with a communicator of ONE rank: 0.84-1.75 G NTT/s instead of
    # 2.2 G (tools/probes/hw_queue_probe.py, its table under profiles/).
    D = 1 if args.two_launch else max(1, min(args.depth, 16))        # batches per launch of the software pipeline
    S = max(1, min(args.streams, NBATCH // (2 * D)))                  # a chain's rotation holds at least two launches' batches
    chain_ctx = []
    for _ in range(S):
        c_ = fusion_hip.Context(q, d, P["root"], P["inv_root"], device=dev_index)
        c_.set_stream(c_.stream_create())
        c_.diag_empty_launch()
        c_.synchronize()
        chain_ctx.append(c_)
    ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"], device=dev_index)
    # the main context: a non-default stream, made torch's current one (graph capture needs it, and torch ops order on it too)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    ctx.diag_empty_launch()
    # the exchange step's context (sign_verify: fz_allreduce_i64 on a stream of its own, beside the next step's kernels)
    exchange_ctx = fusion_hip.Context(q, d, P["root"], P["inv_root"], device=dev_index)
    # HIGH stream priority: beside sign_core's 1024 workgroups the all-reduce kernel's few otherwise wait for slots -- measured
    # with a one-rank RCCL all-reduce + a 40 us stand-in per step: 156 us per step at normal priority, 84 at high
    # (FZ_BENCH_EXCHANGE_PRIORITY=normal for the A/B)
    xprio = os.environ.get("FZ_BENCH_EXCHANGE_PRIORITY", "high")
    exchange_ctx.set_stream(exchange_ctx.stream_create(None if xprio == "normal" else xprio))
    exchange_ctx.diag_empty_launch()
    for c_ in (ctx, exchange_ctx):
        c_.synchronize()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif args.single_rank_comm:
        # rehearsal on ONE GPU of what an N > 1 run has in its process: a torch.distributed RCCL group that has run collectives
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:            # any free port: this is a group of one
            os.environ["MASTER_PORT"] = str(rendezvous_port())
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        t_ = torch.ones(4, device=dev)
        dist.all_reduce(t_)
        dist.barrier()
        torch.cuda.synchronize(dev)

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

    # ---- the exchange step's communicator, created FIRST so that the line can say what RCCL itself saw -----------------
    # rank 0's ncclUniqueId travels over the torch process group, every rank joins with fz_comm_create.  With the "nccl"
    # backend (a real multi-GPU run) a communicator that cannot be created, or that counts another number of ranks than the
    # process group, is a FAILURE: every rank leaves with exit code 4 (VERDICT r04 #3a: round 4 fell back to torch.distributed
    # and returned 0).  The fallback exists only for the rehearsal on fewer GPUs, FZ_BENCH_BACKEND=gloo, where ranks share a
    # device and RCCL must refuse.
    comm, collective = None, "none (single rank: no exchange step)"
    rccl_nranks, comm_error = None, None
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
                rccl_nranks = comm.info()[0]                        # ncclCommCount: what RCCL reports, not what we asked for
                ok = 1.0 if rccl_nranks == world else 0.0
                if not ok:
                    comm_error = f"RCCL counts {rccl_nranks} ranks, the process group {world}"
            except fusion_hip.FusionHipError as e:
                comm_error = f"fz_comm_create failed: {e}"
        elif backend == "nccl":
            comm_error = f"rank 0 could not produce an ncclUniqueId: {uid[0]}"
        if min_over_ranks(ok) < 1.0:
            if backend == "nccl":
                sys.stderr.write(f"bench.py rank {rank}: the C-ABI communicator is unusable ({comm_error or 'another rank failed'}); "
                                 "a multi-GPU run must carry its exchange step through fz_allreduce_i64 -- exit code 4 "
                                 "(FZ_BENCH_BACKEND=gloo rehearses the N > 1 path over torch.distributed on fewer GPUs)\n")
                sys.stderr.flush()
                dist.barrier()
                sys.exit(4)
            if comm is not None:
                comm.destroy()
            comm = None
            collective = f"torch.distributed all_reduce ({backend}); rehearsal: ranks share a GPU, fz_comm_* not usable"
        else:
            collective = f"fz_allreduce_i64 (ncclAllReduce int64 sum, C ABI), RCCL counts {rccl_nranks} ranks"
    if world == 1 and args.single_rank_comm:
        # rehearsal of the exchange step's code path on ONE GPU: a communicator of one rank (RCCL still launches through the same
        # call, on the exchange context's stream, inside the same capture); never the default
        comm = fusion_hip.Comm(ctx, 1, 0, fusion_hip.comm_unique_id())
        rccl_nranks = comm.info()[0]
        collective = f"fz_allreduce_i64 (ncclAllReduce int64 sum, C ABI), RCCL counts {rccl_nranks} rank (rehearsal)"
    try:
        rccl_version = fusion_hip.rccl_version() if (world > 1 or comm is not None) else None
    except Exception:
        rccl_version = None

    # ---- the collectives on KNOWN data, per rank, before anything is timed on them (VERDICT r04 #3c): one fz_allreduce_i64 and
    # one fz_reduce_scatter_i64 of a pattern whose sums are known in closed form -- element i of rank r holds
    # (r + 1) * 2^33 + i * (r + 1) - 7 (beyond int32 on purpose: the sums of the path are 64-bit) -- checked element by element
    # on every rank; with the torch.distributed rehearsal the same pattern goes through dist.all_reduce.  A wrong sum is exit 5.
    def collective_self_check():
        count = 4096 * world                               # divisible by the rank count (reduce-scatter blocks)
        i64 = np.arange(count, dtype=np.int64)
        mine = (rank + 1) * (1 << 33) + i64 * (rank + 1) - 7
        tri = world * (world + 1) // 2
        want = tri * (1 << 33) + i64 * tri - 7 * world
        out = {}
        if comm is not None:
            buf = fusion_hip.DeviceBuffer.from_numpy(ctx, mine)
            ctx.allreduce_i64_dev(comm, buf.ptr, count)
            ctx.synchronize()
            out["allreduce_i64"] = bool(np.array_equal(buf.to_numpy(np.int64, (count,)), want))
            ctx.h2d(buf.ptr, mine)
            per = count // world
            ctx.reduce_scatter_i64_dev(comm, buf.ptr, per)
            ctx.synchronize()
            got = buf.to_numpy(np.int64, (count,))[rank * per:(rank + 1) * per]
            out["reduce_scatter_i64"] = bool(np.array_equal(got, want[rank * per:(rank + 1) * per]))
            buf.free()
        elif world > 1:
            t = torch.from_numpy(mine.copy())
            t = t.to(dev) if dist.get_backend() == "nccl" else t
            dist.all_reduce(t)
            out["torch_all_reduce_i64"] = bool(np.array_equal(t.cpu().numpy(), want))
        return out
    coll_check = collective_self_check() if (world > 1 or comm is not None) else None
    if coll_check is not None and min_over_ranks(1.0 if all(coll_check.values()) else 0.0) < 1.0:
        sys.stderr.write(f"bench.py rank {rank}: collective self-check FAILED: {coll_check} -- exit code 5\n")
        sys.stderr.flush()
        if world > 1:
            dist.barrier()
        sys.exit(5)

    # what every rank is, as the process group and RCCL see it (proof that N ranks on N devices took part)
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "device": props.name,
          "pci_bus_id": f"{getattr(props, 'pci_domain_id', 0):04x}:{getattr(props, 'pci_bus_id', -1):02x}:{getattr(props, 'pci_device_id', 0):02x}",
          "world_size_seen": dist.get_world_size() if world > 1 else 1,
          "backend": dist.get_backend() if world > 1 else "none",
          "rccl_nranks": rccl_nranks, "rccl_version": rccl_version,
          "collective_check": None if coll_check is None else ("ok:" + "+".join(sorted(coll_check)) if all(coll_check.values()) else f"FAILED:{coll_check}")}
    ranks = [me]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)

    # ---- NTT workload: NBATCH batches resident in HBM ---------------------------------------------------------------
    # i.i.d. uniform centred residues, seeded per rank and batch, generated where they are consumed (fz_fill_synthetic: the
    # stream the tests' host generator produces)
    xs = torch.empty((NBATCH, B, d), dtype=torch.int32, device=dev)
    for i in range(NBATCH):
        ctx.fill_synthetic_dev(xs[i].data_ptr(), B * d, 20261003 + rank + 1000 * i)
    ys, zs = torch.empty_like(xs), torch.empty_like(xs)
    torch.cuda.synchronize(dev)
    x, y, z = xs[0], ys[0], zs[0]

    # the timed loop calls the C ABI directly with pre-built arguments: at ~4 us per kernel the Python wrapper layers
    # (attribute lookups, argument boxing) would otherwise be part of the measurement
    lib, h = ctx._lib, ctx._h
    nB = ctypes.c_size_t(B)
    rot_p = [tuple(ctypes.c_void_p(t[i].data_ptr()) for t in (xs, ys, zs)) for i in range(NBATCH)]
    NttJob = fusion_hip._lib.NttJob

    def jobs(*items):
        return (NttJob * len(items))(*[NttJob(i_, o_, B, inv) for i_, o_, inv in items])

    class Chain:
        """One HIP stream's share of the work: a context + stream and the batches of the rotation it walks, one after the other.
        pipe: ONE launch (fz_ntt_multi) = the forward transforms of the next g <= D batches (x_b -> y_b) + the inverse transforms
        (y_b -> z_b) of the batches the previous launch transformed forward; a run of steps opens with a forward-only launch and
        end() closes it with an inverse-only one (job tables are built once per distinct launch and kept).  two: fz_ntt_forward
        then fz_ntt_inverse (two launches per step); warm: the same two launches on ONE batch."""

        def __init__(self, c, idx):
            self.ctx, self.h, self.idx, self.k, self.prev, self.tabs = c, c._h, list(idx), 0, (), {}
            self.rp = [rot_p[i] for i in self.idx]

        def _launch(self, new, old):
            tab = self.tabs.get((new, old))
            if tab is None:
                rp = self.rp
                tab = self.tabs[(new, old)] = jobs(*([(rp[b_][0], rp[b_][1], 0) for b_ in new] + [(rp[b_][1], rp[b_][2], 1) for b_ in old]))
            return lib.fz_ntt_multi(self.h, tab, len(new) + len(old))

        def run(self, n_steps, mode_):
            """issue n_steps steps on this chain's stream (asynchronous) -> 0 or the first failing status ORed in"""
            rc, n = 0, len(self.rp)
            if mode_ == "pipe":
                while n_steps > 0:
                    g = min(D, n_steps)
                    new = tuple((self.k + t_) % n for t_ in range(g))
                    self.k = (self.k + g) % n
                    rc |= self._launch(new, self.prev)
                    self.prev = new
                    n_steps -= g
                return rc
            for _ in range(n_steps):
                a_, b_, c_ = self.rp[0 if mode_ == "warm" else self.k % n]
                self.k += 1
                rc |= lib.fz_ntt_forward(self.h, a_, b_, nB) | lib.fz_ntt_inverse(self.h, b_, c_, nB)
            return rc

        def end(self, mode_):
            """the inverse transforms still due: closes a run of pipelined steps"""
            if mode_ != "pipe" or not self.prev:
                return 0
            old, self.prev = self.prev, ()
            return self._launch((), old)

    # S chains: independent batches are in flight on S HIP streams at once (the contexts + streams created first thing above)
    # -- launches of different streams overlap on the chip, which hides each launch's ramp and drain
    bounds = [NBATCH * s_ // S for s_ in range(S + 1)]
    chains = [Chain(chain_ctx[s_], range(bounds[s_], bounds[s_ + 1])) for s_ in range(S)]
    solo = Chain(ctx, range(NBATCH))                    # every batch on ONE stream: the single-stream legs and the isolated launches
    mode = "two" if args.two_launch else "pipe"
    per_launch_steps = {"pipe": D, "two": 0.5, "warm": 0.5}          # steps one launch covers

    def prewarm_fn(fn, ms, inner=50):
        """untimed: keep the device busy for `ms` by calling fn (the side legs' form)"""
        t_end = time.perf_counter() + ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(inner):
                fn()
            torch.cuda.synchronize(dev)

    def prewarm(cs, mode_, ms):
        """untimed: keep the device busy for `ms` so the timed region starts at steady clocks (the run is closed afterwards)"""
        t_end = time.perf_counter() + ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(12):
                for c_ in cs:
                    c_.run(D, mode_)
            torch.cuda.synchronize(dev)
        for c_ in cs:
            c_.end(mode_)

    def timed_on_stream(fn, reps):
        """average milliseconds of fn() over reps back-to-back calls, HIP events on the kernels' own stream"""
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(reps):
            fn()
        b_.record(stream)
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b_) / reps

    def region(cs, mode_, K, min_ms=MIN_REGION_MS):
        """W warmup steps, then the K steps -- dealt to the chains `cs`, every chain's share recorded into its own hipGraph
        (M repetitions per recording for K < 1000) -- replayed for >= min_ms between two barriers.
        -> dict(elapsed s (max over ranks), steps, per-chain event ms, per-chain steps, replays, M, shader clock)"""
        n_c = len(cs)
        for j, c_ in enumerate(cs):
            c_.k = 0
            c_.run(args.warmup // n_c + (1 if j < args.warmup % n_c else 0), mode_)
            c_.end(mode_)
        barrier()
        M = max(1, 1000 // K) if K < 1000 else 1
        # steps per chain and replay round: the M x K steps of a round dealt evenly (the shares differ by one step at most
        # whatever K is); a share is issued as launches of D steps and, when D does not divide it, one shorter launch
        share = [(M * K) // n_c + (1 if j < (M * K) % n_c else 0) for j in range(n_c)]
        graphs = [[] for _ in cs]
        if not args.no_graph:
            for j, c_ in enumerate(cs):
                sizes = ([1000] * (share[j] // 1000) + ([share[j] % 1000] if share[j] % 1000 else [])) if share[j] else []
                made = {}
                for n_steps in sizes:
                    if n_steps in made:
                        graphs[j].append(made[n_steps])      # the same recording, launched again
                        continue
                    c_.k = 0
                    c_.ctx.graph_begin()
                    rc = c_.run(n_steps, mode_) | c_.end(mode_)        # n_steps steps = ceil(n_steps / D) + 1 launches when pipelined
                    g = c_.ctx.graph_end()
                    assert rc == 0, f"capture failed: {lib.fz_last_error()}"
                    g.launch()                               # untimed first replay (upload)
                    made[n_steps] = g
                    graphs[j].append(g)
            barrier()

        def k_steps():
            if not args.no_graph:
                for gl in graphs:                            # asynchronous: the chains' recordings run side by side
                    for g in gl:
                        g.launch()
            else:
                rc, grain = 0, max(1, int(per_launch_steps[mode_]))
                for t_ in range(0, max(share), grain):
                    for j, c_ in enumerate(cs):
                        if t_ < share[j]:
                            rc |= c_.run(min(grain, share[j] - t_), mode_)
                for c_ in cs:
                    rc |= c_.end(mode_)
                assert rc == 0, f"launch failed: {lib.fz_last_error()}"
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        k_steps()
        torch.cuda.synchronize(dev)
        t_once = max(time.perf_counter() - t0, 1e-6)
        launches = int(max_over_ranks(max(1.0, -(-min_ms * 1e-3 // t_once))))     # the same count on every rank
        barrier()
        t0 = time.perf_counter()
        for c_ in cs:
            c_.ctx.timer_start()                             # an event on every chain's own stream
        for _ in range(launches):
            k_steps()
        ev_ms = [c_.ctx.timer_stop_ms() for c_ in cs]        # (records the closing event and waits for it)
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        shader = None
        try:                                                 # the shader clock the chip holds beside these launches (a diagnostic)
            for _ in range(max(1, launches // 4)):
                k_steps()
            shader = round(ctx.diag_shader_clock(500))
        except Exception as e:
            shader = f"failed: {type(e).__name__}: {e}"
        torch.cuda.synchronize(dev)
        for gl in graphs:
            for g in {id(g): g for g in gl}.values():
                g.destroy()
        return {"elapsed": elapsed, "steps": launches * M * K, "ev_ms": ev_ms, "chain_steps": [launches * sh for sh in share],
                "replays": launches, "M": M, "shader_mhz": shader}

    def summary(r_, mode_):
        """value and the chip-level / per-launch figures of one region() result.  Launches are counted as FULL-launch equivalents
        (steps / steps per launch): the forward-only launch that opens a recording and the inverse-only one that closes it are two
        halves of one, and a recording of 1000 steps holds one such pair."""
        pls = per_launch_steps[mode_]
        byt = 8 * d * B * 2 * pls                              # algorithmic bytes per launch (2 * pls transforms of B rows)
        n_launch = r_["steps"] / pls
        interval = max(r_["ev_ms"]) * 1e3 / n_launch           # us between launches, all chains together
        per = [m_ * 1e3 / (st_ / pls) for m_, st_ in zip(r_["ev_ms"], r_["chain_steps"]) if st_]
        lat = sum(per) / len(per)                              # us per launch on its own stream
        return {"value": 2.0 * B * r_["steps"] * world / r_["elapsed"], "unit": "NTT/s", "ms_per_step": r_["elapsed"] / r_["steps"] * 1e3,
                "frac": byt / (interval * 1e-6) / 1e9 / HBM_PEAK_GBS, "launch_interval_us": interval, "launch_us": lat,
                "frac_per_launch": byt / (lat * 1e-6) / 1e9 / HBM_PEAK_GBS, "streams": len(r_["ev_ms"]), "shader_mhz": r_["shader_mhz"]}

    rc = solo.run(NBATCH, mode) | solo.end(mode)         # every batch once: z_i defined whatever the flags below skip
    assert rc == 0, f"launch failed: {lib.fz_last_error()}"
    barrier()
    assert torch.equal(zs, xs), "INTT(NTT(x)) != x"
    zs.zero_()
    prewarm(chains, mode, args.prewarm_ms)
    R = region(chains, mode, args.steps, args.region_ms)
    torch.cuda.synchronize(dev)
    assert torch.equal(zs, xs), "INTT(NTT(x)) != x after the timed region"
    elapsed, total_steps, launches, M, shader_mhz = R["elapsed"], R["steps"], R["replays"], R["M"], R["shader_mhz"]
    head = summary(R, mode)
    value = head["value"]
    # the same steps with ONE launch in flight at a time (one stream, all 64 batches): what a single launch of the dominant kernel
    # takes in a dense stream -- the figure a profiler can reproduce (rocprofv3 --kernel-trace serialises the dispatches of all
    # streams: in its trace of this script no two launches overlap)
    if S > 1:
        prewarm([solo], mode, 20.0)
        one = summary(region([solo], mode, args.steps, ROOFLINE_REGION_MS), mode)
    else:
        one = head

    # ---- the chip's own clock: every workgroup of every launch stamps the 100 MHz reference counter at entry
    # and exit (fz_diag_stamps_*); a launch = [min entry, max exit] over its workgroups.  Neither a kernel trace (it serialises
    # the streams) nor HIP events (host-side markers) can show launches of different streams overlapping; these can.
    def merged(iv):
        out_ = []
        for a_, b_ in sorted(iv):
            if out_ and a_ <= out_[-1][1]:
                out_[-1][1] = max(out_[-1][1], b_)
            else:
                out_.append([a_, b_])
        return out_

    def stamp_pass(cs, n_launch, rounds=3):
        """n_launch full launches per chain (+ the opening and the closing one) captured WITH stamp slots, then `rounds` times:
        reset, ONE replay of every chain's recording side by side, read.  -> the last round's table and all rounds' figures"""
        graphs = []
        for c_ in cs:
            c_.ctx.diag_stamps_begin(n_launch + 2, (n_launch + 2) * 2 * D * B)     # (at most one workgroup per row)
            c_.k = 0
            c_.ctx.graph_begin()
            rc_ = c_.run(n_launch * D, "pipe") | c_.end("pipe")
            graphs.append(c_.ctx.graph_end())
            c_.ctx.diag_stamps_stop()
            assert rc_ == 0, f"capture failed: {lib.fz_last_error()}"
        for g in graphs:
            g.launch()                                       # (upload)
        torch.cuda.synchronize(dev)
        per_launch_bytes = 8 * d * B * 2 * D
        rounds_out, table = [], []
        for _ in range(rounds):
            prewarm(cs, "pipe", 10.0)                        # steady clocks; these launches carry no slots
            for c_ in cs:
                c_.ctx.diag_stamps_reset()
            torch.cuda.synchronize(dev)
            for c_ in cs:
                c_.ctx.timer_start()
            for g in graphs:
                g.launch()
            ev = [c_.ctx.timer_stop_ms() for c_ in cs]
            recs = [c_.ctx.diag_stamps_read(n_launch + 2) for c_ in cs]
            t0_ = min(int(r_[0][r_[3] > 0].min()) for r_ in recs)
            iv, full_d, gaps, table = [], [], [], []
            for j, (st_, en_, ls_, wg_) in enumerate(recs):
                assert len(st_) == n_launch + 1 and (wg_ > 0).all(), (len(st_), wg_.tolist())
                st_u, en_u, ls_u = [(x_.astype(np.int64) - t0_) / 100.0 for x_ in (st_, en_, ls_)]     # ticks of 10 ns -> us
                iv += list(zip(st_u.tolist(), en_u.tolist()))
                full_d += (en_u - st_u)[1:-1].tolist()       # (the first launch is forward-only, the last inverse-only)
                gaps += (st_u[1:] - en_u[:-1]).tolist()
                table += [{"stream": j, "launch": k_, "start_us": round(float(st_u[k_]), 2), "end_us": round(float(en_u[k_]), 2),
                           "last_workgroup_in_us": round(float(ls_u[k_]), 2), "workgroups": int(wg_[k_])} for k_ in range(len(st_u))]
            span = max(b_ for _, b_ in iv) - min(a_ for a_, _ in iv)
            busy = sum(b_ - a_ for a_, b_ in merged(iv))
            total_b = per_launch_bytes * n_launch * len(cs)  # the two half launches of a chain = one full one: n_launch + 1 dispatches
            rounds_out.append({"span_us": span, "busy_us": busy, "in_flight": sum(b_ - a_ for a_, b_ in iv) / busy,
                               "frac": total_b / (span * 1e-6) / 1e9 / HBM_PEAK_GBS,
                               "launch_us": float(np.mean(full_d)), "launch_us_median": float(np.median(full_d)),
                               "gap_us": float(np.mean(gaps)), "event_ms": ev, "frac_by_events": total_b / (max(ev) * 1e-3) / 1e9 / HBM_PEAK_GBS})
        for g in graphs:
            g.destroy()
        best = sorted(rounds_out, key=lambda r_: r_["frac"])[len(rounds_out) // 2]      # the median round
        return dict(best, rounds=rounds_out, launches_per_stream=n_launch + 1, streams=len(cs), bytes_per_launch=per_launch_bytes,
                    table=table, clock="s_memrealtime (100 MHz reference counter, one for the chip), ticks of 10 ns",
                    what="per launch: [min entry, max exit] over its workgroups' own stamps; span = first entry to last exit of the replay; "
                         "in_flight = sum of launch durations / time at least one launch was running; frac = algorithmic bytes / span / 8 TB/s")

    stamps_chip = stamps_one = None
    if mode == "pipe" and not args.no_stamps:
        try:
            stamps_chip = stamp_pass(chains, 32)
            stamps_one = stamp_pass([solo], 32) if S > 1 else stamps_chip
            torch.cuda.synchronize(dev)
            assert torch.equal(zs, xs), "INTT(NTT(x)) != x on the stamped passes"
        except fusion_hip.FusionHipError as e:
            stamps_chip = stamps_one = {"error": repr(e)}

    # ---- context for the roofline figures (which come from the timed region itself, below): begin/end events bound to EVERY
    # dispatch (hipExtLaunchKernelGGL) of instrumented passes on ONE stream over all 64 batches, launched one by one right after
    # the timed region (a hipGraph cannot carry the events) -- the duration of a launch that starts on an idle memory system.
    n_inst, n_pass = 200, 2
    dom_kind = 0 if args.two_launch else 2              # the dominant launch: the forward kernel, or the multi-job launch
    per_inst = 2 if args.two_launch else 1
    dom_all, inv_all, passes = [], [], []
    for _ in range(n_pass):
        prewarm([solo], mode, 20.0)                      # dense launches first: a one-by-one pass leaves the device half idle
        rc = solo.run(D, mode)                           # (the forward-only launch that opens the run: not instrumented)
        ctx.profile_begin(per_inst * n_inst + 2, args.sample_every)
        for _ in range(n_inst):
            rc |= solo.run(D, mode)
        assert rc == 0, f"launch failed: {lib.fz_last_error()}"
        us, kind = ctx.profile_end_samples(per_inst * n_inst + 2)
        rc = solo.end(mode)
        assert rc == 0, f"launch failed: {lib.fz_last_error()}"
        f_, i_ = us[kind == dom_kind], us[kind == 1]
        assert len(f_) >= n_inst // args.sample_every - 1, (len(f_), len(us))
        dom_all.append(f_)
        inv_all.append(i_)
        passes.append({"avg_us": float(f_.mean()), "median_us": float(np.median(f_)), "max_us": float(f_.max()), "count": int(len(f_)),
                       "inverse_avg_us": float(i_.mean()) if len(i_) else None})
    torch.cuda.synchronize(dev)
    assert torch.equal(zs, xs), "INTT(NTT(x)) != x on the instrumented passes"
    dom_all = np.concatenate(dom_all)
    inv_all = np.concatenate(inv_all)
    iso_us = float(dom_all.mean())
    # 8*d algorithmic bytes per transform; one dominant launch = D x 4096 forward + D x 4096 inverse transforms
    units_per_launch = B if args.two_launch else 2 * D * B
    dom_bytes = 8 * d * units_per_launch
    # THE per-launch fraction (`roofline.frac`): algorithmic bytes per launch / the launch's average duration with ONE launch in
    # flight -- HIP events on the kernel's stream around a dense single-stream region of the same steps / its launches.  This is
    # the figure rocprofv3 --kernel-trace --stats reproduces (it serialises dispatches), profiles/ + DESIGN.md section 6.
    # `roofline.chip`: the headline's timed region itself -- `streams` launches in flight on as many HIP streams; algorithmic bytes
    # moved / the region's duration by HIP events on every stream = `value` x 2048 B: what the chip achieves on batches of 4096;
    # `roofline.chip.device_clock`: the same from the kernels' own timestamps.
    ach = one["frac"] * HBM_PEAK_GBS
    if args.two_launch:
        kernel_name = "ntt_fwd4<8, true, 1, 8>"
        launch_text = "fz_ntt_forward + fz_ntt_inverse: two launches per step"
    else:
        cus = props.multi_processor_count                  # the launcher's rule (fz_launch_ntt_multi): row groups per wave by the launch's rows
        rows_ = 2 * D * B
        nr_ = 1 if rows_ <= 24 * cus else (2 if rows_ <= 48 * cus else 4)
        tab_ = 4 if 2 * D <= 4 else (8 if 2 * D <= 8 else 32)
        if chain_ctx[0].diag_ntt_schedule(rows_) == 16:                # the library's own crossover (fz_diag_ntt_schedule)
            kernel_name = f"ntt_jobs16<8, true, FzJobsN<{tab_}>>"
        else:
            kernel_name = f"ntt_jobs4<8, true, {nr_}, {(8 if rows_ >= 8 * cus else 4 if rows_ >= 4 * cus else 1) if nr_ == 1 else 2}, FzJobsN<{tab_}>>"
        launch_text = (f"software-pipelined over {D} batches: forward of batches i+1..i+{D} + inverse of batches i-{D - 1}..i in one "
                       f"fz_ntt_multi launch ({2 * D} jobs) per {D} steps")
    traffic, traffic_src = pmc_traffic(kernel_name.split(", FzJobsN")[0], units_per_launch)      # prefix: whatever follows the launch shape

    def stamp_brief(sp):
        if not isinstance(sp, dict) or "frac" not in sp:
            return sp
        return {k_: sp[k_] for k_ in ("frac", "in_flight", "launch_us", "gap_us", "span_us", "frac_by_events", "launches_per_stream", "streams")}

    full = {
        "metric": METRIC, "value": value, "unit": "NTT/s", "n_gpus": world, "steps": args.steps, "repeats": launches * M,
        "warmup": args.warmup, "ms_per_step": elapsed / total_steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "timed_region_ms": elapsed * 1e3,
        "config": {"workload": f"configs[1]: secpar={SECPAR}, {B} degree-{d} forward+inverse NTTs per step per GPU",
                   "batch": B, "degree": d, "modulus": q, "batches_rotated": NBATCH, "steps_per_launch": per_launch_steps[mode],
                   "streams": S, "step": launch_text,
                   "parallelism": f"{world} independent rank(s); per rank {S} HIP stream(s), each walking its own {NBATCH // S} of the {NBATCH} batches",
                   "arithmetic": "exact integers carried in fp64 lanes (bit-identical to the reference's int arithmetic); int32 in and out",
                   "launch": ("one by one" if args.no_graph else f"{S} hipGraph(s) of {M}x{args.steps} steps together, {launches} replays") +
                             f": timed region {elapsed * 1e3:.0f} ms (>= {args.region_ms:.0f} ms asked)",
                   "prewarm_ms": args.prewarm_ms,
                   "host_threads_on": placement or "all allowed CPUs (no GPU-local NUMA node found, or FZ_NO_PIN=1)"},
        "ranks": ranks,
        "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": one["frac"], "traffic": traffic, "traffic_source": traffic_src,
                     "bytes_per_launch": dom_bytes, "units_per_launch": units_per_launch,
                     "avg_launch_us": one["launch_us"], "in_flight": 1,
                     "operands": f"cold: rotation of {NBATCH} batches",
                     "timing": "HIP events around a dense one-stream region of the same steps / launches (one launch in flight)",
                     "one_stream": one,
                     "device_clock": stamp_brief(stamps_one),
                     "chip": {"streams": S, "achieved": head["frac"] * HBM_PEAK_GBS, "frac": head["frac"], "unit": "GB/s",
                              "launch_us_in_flight": head["launch_us"], "launch_interval_us": head["launch_interval_us"],
                              "launches_timed": int(round(total_steps / per_launch_steps[mode])),
                              "per_chain_event_ms": R["ev_ms"], "per_chain_steps": R["chain_steps"],
                              "device_clock": stamp_brief(stamps_chip),
                              "what": "the timed region of `value`: `streams` launches in flight, one per HIP stream; algorithmic bytes of "
                                      "all launches / the region's duration by HIP events on every stream (the slowest chain's); device_clock: "
                                      "the same quantity over a 33-launch replay per stream from the kernels' own entry / exit timestamps"},
                     "isolated": {"avg_launch_us": iso_us, "median_launch_us": float(np.median(dom_all)),
                                  "frac": dom_bytes / (iso_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "launches": int(len(dom_all)),
                                  "inverse_avg_launch_us": float(inv_all.mean()) if len(inv_all) else None, "passes": passes,
                                  "what": f"begin/end events bound to EVERY dispatch (hipExtLaunchKernelGGL) of {n_pass} passes of {n_inst} "
                                          "launches issued one by one from Python on ONE stream after the timed region: the host paces them, "
                                          "so a launch may start on an idle memory system; context, not the fraction"},
                     "shader_mhz": shader_mhz, "butterflies_per_s": value * (d // 2) * 8},
        "device_timestamps": {"chip": stamps_chip, "one_stream": stamps_one},
    }

    # ---- everything below is a side leg: the headline is measured.  A watchdog prints the line with whatever is there if
    # a leg hangs (a collective that never completes cannot be cancelled from Python) ------------------------------------
    stage = ["warm_replay"]
    done = threading.Event()

    def emit(watchdog=None):
        if watchdog:
            full["watchdog"] = watchdog
        full["full"] = write_full(args.full_out, full)
        text = compact_line(full)
        print(text)
        sys.stdout.flush()
        return text

    def watchdog():
        if done.wait(args.watchdog_s):
            return
        _BAILING.set()
        if rank == 0:
            try:
                text = emit(f"side legs not finished {args.watchdog_s:.0f} s after the headline; running: {stage[0]}")
                try:
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
        os._exit(3)                                     # a process that hung on the GPU must not report success
    threading.Thread(target=watchdog, daemon=True).start()

    def leg(name, fn):
        """a side leg must never take the headline down with it"""
        stage[0] = name
        try:
            return fn()
        except Exception as exc:
            import traceback
            traceback.print_exc()
            return {"error": repr(exc)}

    env = types.SimpleNamespace(
        args=args, ctx=ctx, torch=torch, np=np, dist=dist, dev=dev, dev_index=dev_index, stream=stream, lib=lib, h=h,
        xs=xs, ys=ys, zs=zs, x=x, y=y, z=z, rot_p=rot_p, B=B, d=d, q=q, l=l, P=P, F=F, fusion_hip=fusion_hip, rank=rank, world=world,
        barrier=barrier, max_over_ranks=max_over_ranks, min_over_ranks=min_over_ranks, prewarm=prewarm_fn,
        timed_on_stream=timed_on_stream, exchange_ctx=exchange_ctx, comm=comm, collective=collective, backend=backend, shard_range=shard_range,
        allreduce_sum_i64=allreduce_sum_i64, HBM_PEAK_GBS=HBM_PEAK_GBS, MIN_REGION_MS=MIN_REGION_MS, SECPAR=SECPAR)

    if not args.headline_only:
        # the single-stream forms over the same batches: one launch in flight at a time
        def solo_leg(mode_, what):
            prewarm([solo], mode_, 20.0)
            out_ = summary(region([solo], mode_, args.steps), mode_)
            out_["what"] = what
            return out_
        if S > 1:
            full["single_stream"] = dict(one, what="the headline's steps on ONE stream (one launch in flight at a time) over the rotating batches")
        if not args.two_launch:
            full["two_launch_step"] = leg("two_launch_step", lambda: solo_leg(
                "two", "ONE stream, two launches per step (fz_ntt_forward, then fz_ntt_inverse) over the rotating batches"))
        full["warm_replay"] = leg("warm_replay", lambda: solo_leg(
            "warm", "ONE stream, two launches per step on ONE batch re-read every step (cache-resident; round 3's `value`)"))
        from tools import bench_legs as L
        if not args.no_sign_verify:
            sv = leg("sign_verify", lambda: L.sign_verify(env))
            full["sign_verify"] = sv
            if isinstance(sv, dict) and "keygen_sign" in sv:
                full["keygen_sign"] = sv.pop("keygen_sign")
            if isinstance(sv, dict) and "one_aggregate" in sv:
                full["sign_verify_1x1024"] = sv.pop("one_aggregate")
        # the metric's second half END TO END, hashing included: keygen / sign / aggregate / verify per second
        # through BatchScheme -- the reference harness' per-function timings (benchmarks/benchmarks.py:37-141).  The algebra cores
        # above run at ~19 M signatures/s; aggregate() and verify() of ONE aggregate are bounded by hash_ag, one serial SHAKE-256 on
        # the host by construction (fusion.py:632-652) -- the line carries both so that neither is read for the other.
        if not args.no_end_to_end:
            full["end_to_end"] = leg("end_to_end", lambda: L.end_to_end(env))
        if args.full:
            for name in L.FULL_LEGS:
                if name != "end_to_end" or args.no_end_to_end:
                    full[name] = leg(name, lambda: getattr(L, name)(env))
    if comm is not None:
        barrier()
        comm.destroy()

    if _BAILING.is_set():                               # the watchdog fired while a leg was (slowly) finishing: its line stands, and
        time.sleep(15.0)                                # its thread ends the process with code 3 -- never a second line, never rc 0
        os._exit(3)
    stage[0] = "cpu_baseline"
    done.set()                                          # the bounded CPU sample is not under the watchdog
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not args.headline_only:
            cb = cpu_baseline(args.cpu_seconds)
            full["cpu_baseline"] = cb
            sch = cb.get("scheme") or {}
            # the second half of the metric, CPU beside GPU, per signature on one core
            if isinstance(full.get("sign_verify"), dict) and "sign_plus_verify_per_s" in sch:
                full["sign_verify"]["cpu_value"] = sch["sign_plus_verify_per_s"]
            if isinstance(full.get("keygen_sign"), dict) and "keygen_plus_sign_per_s" in sch:
                full["keygen_sign"]["cpu_value"] = sch["keygen_plus_sign_per_s"]
        emit()
    if world > 1:
        barrier()                                       # rank 0's single-rank legs are done: leave together
        dist.destroy_process_group()
    elif args.single_rank_comm:
        dist.destroy_process_group()
    # and then an ORDINARY interpreter exit.  Rounds 3-4 left through os._exit(0) here because a test process had aborted at exit
    # with `double free or corruption`: the C ABI had bound /opt/rocm's RCCL by soname into the global scope and torch then
    # mapped its own copy -- two RCCLs and two rocm_smi in one process, whose static destructors freed one object twice
    # (profiles/r05_rccl_exit_matrix.txt has the backtrace).  fz_comm_* now binds the copy that is already mapped, or the one
    # beside the HIP runtime the process runs on: one copy whatever the import order, and nothing to hide.


_BAILING = threading.Event()       # set by the watchdog: from then on an exception in the main thread is a peer leaving

if __name__ == "__main__":
    try:
        main()
    except BaseException:
        if not _BAILING.is_set():
            raise
        time.sleep(10.0)           # the watchdog thread prints the line and ends the process (exit code 3: a hang is a failure)
        os._exit(3)
