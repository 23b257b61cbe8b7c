"""One process per GPU without a launcher: start the N ranks of a script, wait for them with a deadline.

Used by bench.py and tools/score_many.py when they are started as plain `python script.py --gpus N` (under
`python -m torch.distributed.run` the ranks come from the environment and nothing here runs).  This module
imports nothing that touches a GPU: the parent process only starts and watches its ranks."""
import os
import socket
import subprocess
import sys
import tempfile
import time


def spawn_ranks(script, argv, n, timeout_s, name=None, inherit_stderr=False):
    """Start ranks 0..n-1 of `script argv` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in
    the environment), wait for them, exit with the status of the FIRST rank that failed (128 + s for a rank killed by
    signal s; 0 if none did).  Rank 0 inherits stdout and stderr, so its
    output is this command's output.  The other ranks' stderr goes to a file each, replayed with a rank prefix when a
    rank fails or the run times out.  A rank that dies takes the others with it (they would wait in a collective for
    ever); a run that exceeds `timeout_s` is killed as a whole and exits with status 124.  inherit_stderr: every
    rank writes to this command's stderr (a tool whose ranks each report on their own share of the work)."""
    name = name or os.path.basename(script)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        ef = None if r == 0 or inherit_stderr else tempfile.TemporaryFile(mode="w+")
        errs.append(ef)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL, stderr=ef))
    rc = 0
    terminated = []   # ranks this launcher stopped because another rank had failed
    deadline = time.monotonic() + timeout_s
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    # the FIRST rank that fails decides the exit status (the ranks terminated because of it end with
                    # -SIGTERM and are not failures of their own); a rank killed by signal s reports 128 + s, as a shell does
                    if p not in terminated:
                        rc = rc or (128 - code if code < 0 else code)
                    for q in pending:
                        if q not in terminated:
                            terminated.append(q)
                            q.terminate()
            if pending and time.monotonic() > deadline:
                sys.stderr.write("%s: %d rank(s) still running after %.0f s: killing the run\n" % (name, len(pending), timeout_s))
                rc = rc or 124
                for q in pending:
                    q.kill()
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        if rc:
            for r, ef in enumerate(errs):
                if ef is not None:
                    ef.seek(0)
                    tail = ef.read()[-4000:]
                    if tail.strip():
                        tag = "rank %d, stopped after another rank failed" % r if procs[r] in terminated else "rank %d" % r
                        sys.stderr.write("".join("[%s] %s\n" % (tag, l) for l in tail.splitlines()))
    sys.exit(rc)
