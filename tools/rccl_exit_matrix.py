#!/usr/bin/env python3
"""Which RCCL does a process end up with, and does it leave cleanly?  (VERDICT / ADVICE r04: `double free or corruption` at
process exit after a GPU test file had passed.)

Every scenario runs ONCE in a process of its own under an LD_PRELOADed SIGABRT / SIGSEGV handler that prints the native
backtrace (tools/microbench/abort_bt.c, built here with gcc), and reports: exit code, which librccl / libamdhip64 /
librocm_smi / libroctx files the process had mapped when it finished its work, what fz_rccl_library() says, and the tail of
stderr.  Scenarios differ in the ORDER in which fusion_hip, torch and RCCL enter the process and in how communicators are
released -- the variables the exit-time abort can depend on.  The LEGACY scenarios re-create round 4's binding
(dlopen by soname with RTLD_GLOBAL) through ctypes before the library binds, to show the failing state next to the fixed one.

    python tools/rccl_exit_matrix.py [--out profiles/r05_rccl_exit_matrix.txt]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fusion-cryptography_amd")

PRELUDE = r"""
import os, sys, json, ctypes
sys.path.insert(0, %r)
def maps():
    keep = ("librccl", "libamdhip64", "librocm_smi", "libroctx", "libhsa-runtime", "rocprofiler-register")
    with open("/proc/self/maps") as fh:
        return sorted({ln.split()[-1] for ln in fh if "/" in ln and any(k in ln for k in keep)})
def report(tag):
    import fusion_hip
    try:
        lib = fusion_hip.rccl_library()
    except Exception as e:
        lib = repr(e)
    print("REPORT " + json.dumps({"tag": tag, "maps": maps(), "rccl_library": lib, "runtime": fusion_hip.runtime_report()}), flush=True)
def legacy_bind():
    # round 4's rccl_bind(): by soname, into the global scope
    ctypes.CDLL("librccl.so.1", mode=ctypes.RTLD_GLOBAL)
def fz_round(n_comm=1, destroy=True, twice_destroy=False):
    import numpy as np
    import fusion_hip
    ctx = fusion_hip.Context(2147465729, 256, 3337519, pow(3337519, -1, 2147465729))
    s = ctx.stream_create()
    ctx.set_stream(s)
    buf = fusion_hip.DeviceBuffer.from_numpy(ctx, np.arange(4096, dtype=np.int64))
    comms = []
    for _ in range(n_comm):
        c = fusion_hip.Comm(ctx, 1, 0, fusion_hip.comm_unique_id())
        assert c.info()[0] == 1
        ctx.allreduce_i64_dev(c, buf.ptr, 4096)
        ctx.synchronize()
        comms.append(c)
        if destroy:
            c.destroy()
            if twice_destroy:
                c.destroy()
                ctx._lib.fz_comm_destroy(c._c)          # NULL after destroy(): a no-op
    assert (buf.to_numpy(np.int64, (4096,)) == np.arange(4096)).all()
    return ctx, comms
def torch_round(nccl=False):
    import torch
    x = torch.ones(1024, device="cuda")
    if nccl:
        import socket
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        dist.all_reduce(x)
        torch.cuda.synchronize()
        return dist
    torch.cuda.synchronize()
    return None
""" % PKG

SCENARIOS = [
    ("fz only: one comm, destroyed", "ctx, cs = fz_round(); report('a')"),
    ("fz only: two comms one after the other, destroyed (+ destroy twice)", "ctx, cs = fz_round(2, True, True); report('b')"),
    ("fz only: one comm, NOT destroyed (left to process exit)", "ctx, cs = fz_round(1, False); report('c')"),
    ("import torch first, then fz comm", "import torch; torch_round(); ctx, cs = fz_round(); report('d')"),
    ("fz comm first, then import torch + a cuda op", "ctx, cs = fz_round(); torch_round(); report('e')"),
    ("fz comm first, then torch.distributed nccl group of one rank", "ctx, cs = fz_round(); d = torch_round(True); report('f'); d.destroy_process_group()"),
    ("torch nccl group first, then fz comm (bench.py --single-rank-comm order)",
     "d = torch_round(True); ctx, cs = fz_round(); report('g'); d.destroy_process_group()"),
    ("torch nccl group first, fz comm, process group NOT destroyed", "d = torch_round(True); ctx, cs = fz_round(); report('h')"),
    ("LEGACY bind (soname, RTLD_GLOBAL) first, fz comm, then import torch + cuda op (round 4's failing order)",
     "import fusion_hip; fusion_hip.load_library(); legacy_bind(); ctx, cs = fz_round(); torch_round(); report('i')"),
    ("LEGACY bind first, fz comm, then torch nccl group", "import fusion_hip; fusion_hip.load_library(); legacy_bind(); ctx, cs = fz_round(); d = torch_round(True); report('j'); d.destroy_process_group()"),
    ("LEGACY bind, two comms, no torch at all", "import fusion_hip; fusion_hip.load_library(); legacy_bind(); ctx, cs = fz_round(2); report('k')"),
]

ABORT_BT = r"""
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
static void handler(int sig) {
    void *frames[64];
    const char head[] = "\n== native backtrace at fatal signal ==\n";
    int n = backtrace(frames, 64);
    if (write(2, head, sizeof head - 1) < 0) {}
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void install(void) {
    signal(SIGABRT, handler);
    signal(SIGSEGV, handler);
}
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "rccl_exit_matrix.txt"))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="fz_abort_bt_")
    src, so = os.path.join(tmp, "abort_bt.c"), os.path.join(tmp, "abort_bt.so")
    with open(src, "w") as fh:
        fh.write(ABORT_BT)
    preload = None
    try:
        subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", src, "-o", so])
        preload = so
    except Exception as e:                       # the matrix still runs, without native backtraces
        print(f"(no abort handler: {e!r})")
    lines = []
    for i, (name, body) in enumerate(SCENARIOS):
        if args.only and args.only not in name:
            continue
        env = dict(os.environ)
        if preload:
            env["LD_PRELOAD"] = preload
        r = subprocess.run(["timeout", "-k", "5", "150", sys.executable, "-c", PRELUDE + body + "\nprint('WORK DONE', flush=True)\n"],
                           capture_output=True, text=True, env=env)
        rep = [ln[7:] for ln in r.stdout.splitlines() if ln.startswith("REPORT ")]
        done = "WORK DONE" in r.stdout
        lines.append(f"[{i}] {name}\n    exit code {r.returncode}{'' if done else '  (work NOT finished)'}")
        if rep:
            o = json.loads(rep[-1])
            lines.append(f"    fz_rccl_library: {o['rccl_library']}")
            for m in o["maps"]:
                lines.append(f"      mapped {m}")
        err = [ln for ln in r.stderr.splitlines() if "amdgpu.ids" not in ln]
        if r.returncode != 0 or any("backtrace" in ln or "double free" in ln or "corruption" in ln for ln in err):
            lines += ["    stderr: " + ln for ln in err[-40:]]
        print("\n".join(lines[-3:])[:400], flush=True)
    text = "\n".join(lines) + "\n"
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        fh.write(text)
    print(text)


if __name__ == "__main__":
    main()
