"""One-node worker launcher with a watchdog: N fresh child processes (one per GPU), rendezvous on 127.0.0.1.

Stands where the reference starts its own workers (``/root/reference/scripts/train_final.sh:13``: ``torchrun --standalone --nproc_per_node N``);
``bench.py --gpus N`` started directly uses it.  Host-only code (nothing here touches the GPU, so it is also what the CPU tests drive):

* every worker is polled; as soon as ONE exits non-zero (or dies on a signal) the rest are terminated (SIGTERM, then SIGKILL) and the launcher
  returns non-zero - without it rank 0 sits in RCCL / the c10d store until their ~10-minute timeouts;
* the whole run has a wall-clock limit;
* the rendezvous port is one this process found free a moment before the workers start (it is released again before they bind it: another process
  can still take it in between - a worker that cannot bind it then fails fast, at which point the rule above ends the run);
* stdout of rank 0 is drained continuously by a reader thread (a worker that prints more than a pipe buffer must not block in write()) and returned;
  stderr of every rank goes to the launcher's stderr.
Workers are always fresh child processes: a process that has initialised the GPU is never re-executed."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Tuple


def free_port() -> int:
    s = socket.socket()
    s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_workers(n: int, cmd: List[str], env: Optional[Dict[str, str]] = None, wall_limit_s: float = 3000.0, poll_s: float = 0.2,
                grace_s: float = 5.0) -> Tuple[int, str]:
    """Runs ``cmd`` n times with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set.  Returns (exit code, rank 0's stdout): 0 only if every
    worker exited 0 within the limit."""
    base = dict(os.environ if env is None else env)
    port = free_port()
    procs: List[subprocess.Popen] = []
    chunks: List[str] = []
    reader: Optional[threading.Thread] = None
    t0, rc, why = time.time(), 0, ""
    try:
        for r in range(n):   # (inside the try: a spawn failure for rank k must not leak ranks 0..k-1)
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                     HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
            if r == 0:
                reader = threading.Thread(target=lambda f=procs[0].stdout: chunks.extend(iter(lambda: f.read(65536), "")), daemon=True)
                reader.start()
        while True:
            codes = [q.poll() for q in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                rc, why = 1, "; ".join(f"worker rank {r} exited with {c}" for r, c in bad)
                break
            if all(c == 0 for c in codes):
                break
            if time.time() - t0 > wall_limit_s:
                rc, why = 1, f"wall-clock limit of {wall_limit_s:.0f} s reached"
                break
            time.sleep(poll_s)
    finally:
        live = [q for q in procs if q.poll() is None]
        for q in live:
            q.terminate()
        t1 = time.time()
        while any(q.poll() is None for q in live) and time.time() - t1 < grace_s:
            time.sleep(0.05)
        for q in live:
            if q.poll() is None:
                q.kill()
    if reader is not None:
        reader.join(timeout=grace_s)     # (rank 0 has exited or was killed: its pipe reaches end of file)
    out0 = "".join(chunks)
    if why:
        print(f"launch: {why}; the other workers were stopped", file=sys.stderr)
    return rc, out0
