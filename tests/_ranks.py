"""Start the ranks of a multi-process test, wait for all of them, and say WHICH rank failed or hung, with its output.
A rank that dies takes the others with it at once (they would sit in a collective until the limit otherwise, and the test would
report the survivor's timeout instead of the cause)."""
import os
import subprocess
import time

import pytest


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
