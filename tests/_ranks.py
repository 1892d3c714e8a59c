"""Start the ranks of a multi-process test, wait for all of them, and say WHICH rank failed or hung, with its output.
A rank that dies takes the others with it at once (they would sit in a collective until the limit otherwise, and the test would
report the survivor's timeout instead of the cause)."""
import os
import subprocess
import time

import pytest


def rendezvous_port():
    """A free TCP port for rank 0's store, chosen OUTSIDE the kernel's ephemeral range.  A port from bind(("127.0.0.1", 0)) lies
    inside that range, and a rank that starts connecting before rank 0 listens can be handed that very number as its SOURCE
    port (TCP self-connect): it then talks to itself, rank 0's bind fails with EADDRINUSE and the others wait for a store that
    never answers -- seen once in this round's GPU suite as a rank that "did not finish"."""
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


def run_rank_processes(argvs, log_dir, limit=300.0):
    """argvs: one command line per rank; output goes to <log_dir>/rank<r>.log (a pipe nobody reads can fill and block a rank)"""
    logs = [open(os.path.join(str(log_dir), f"rank{r}.log"), "w+") for r in range(len(argvs))]
    procs = [subprocess.Popen(argv, stdout=logs[r], stderr=subprocess.STDOUT) for r, argv in enumerate(argvs)]

    def tails():
        out = []
        for r, fh in enumerate(logs):
            fh.flush()
            fh.seek(0)
            out.append(f"--- rank {r} (exit code {procs[r].poll()})\n{fh.read()[-3000:]}")
        return "\n".join(out)
    deadline = time.time() + limit
    try:
        while any(p.poll() is None for p in procs):
            failed = [r for r, p in enumerate(procs) if p.poll() not in (None, 0)]
            if failed or time.time() > deadline:
                if failed:
                    time.sleep(1.0)                          # the others may be about to report the same cause
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                for p in procs:
                    p.wait()
                pytest.fail((f"rank {failed[0]} failed" if failed else f"the ranks did not finish in {limit:.0f} s") + "\n" + tails())
            time.sleep(0.05)
        assert all(p.returncode == 0 for p in procs), tails()
    finally:
        for fh in logs:
            fh.close()
