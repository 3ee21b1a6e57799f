"""Launcher for the multi-rank tests: every rank is a FRESH child process (`python tests/rank_worker.py kind rank world port
out.json args...`), never a fork of the pytest process -- which, in the GPU suite, holds HIP state and tens of GB of device
memory by the time these tests run.  Results come back as one JSON file per rank under tmp_path; every child has a hard
timeout and its stderr tail goes into the assertion message."""
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "rank_worker.py")


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(kind: str, world: int, args, tmp_path, timeout: float = 600.0, env=None):
    """Start `world` children, wait for all of them (hard deadline: the stragglers are killed by PID), return the list of
    per-rank result dicts.  Raises AssertionError carrying every failed rank's output tail."""
    port = free_port()
    e = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONFAULTHANDLER="1")
    e.update(env or {})
    procs = []
    for r in range(world):
        out = os.path.join(str(tmp_path), f"{kind}_rank{r}.json")
        log = open(os.path.join(str(tmp_path), f"{kind}_rank{r}.log"), "w+")
        p = subprocess.Popen([sys.executable, WORKER, kind, str(r), str(world), str(port), out] + [str(a) for a in args],
                             cwd=ROOT, env=e, stdout=log, stderr=subprocess.STDOUT)
        procs.append((p, out, log))
    deadline = time.time() + timeout
    fails = []
    for r, (p, out, log) in enumerate(procs):
        try:
            rc = p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
            rc = "timeout"
        if rc != 0 or not os.path.exists(out):
            log.seek(0)
            detail = open(out).read() if os.path.exists(out) else "(no result file)"
            fails.append(f"rank {r}: rc={rc} result={detail}\n{log.read()[-3000:]}")
        log.close()
    assert not fails, "\n".join(fails)
    return [json.load(open(out)) for _, out, _ in procs]
