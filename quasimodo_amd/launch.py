"""One process per GPU, started by a parent that has touched no GPU.

`spawn_ranks` starts `world` children of one command with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT set (rendezvous on 127.0.0.1), watches them, relays what rank 0 writes to its standard output,
stops the others as soon as one fails or the time limit passes, and reports the failing ranks with the tails
of their output.  Children are fresh processes (never a re-exec of a process that has initialised the GPU).

Used by `bench.py --gpus N` when it is not already running under a launcher, and by
`quasimodo_amd.multigpu.extract_many_sharded`.
"""
import os
import socket
import subprocess
import sys
import tempfile
import time

DEFAULT_TIMEOUT = float(os.environ.get("QM_RANK_TIMEOUT", "3600"))   # seconds; a rendezvous or collective that hangs must not hang the caller for ever


class RankFailure(RuntimeError):
    def __init__(self, msg, bad, returncodes):
        super().__init__(msg)
        self.bad = bad
        self.returncodes = returncodes


def free_port():
    """A port nobody listens on right now.  (Another process can still take it before the children bind:
    spawn_ranks then starts the ranks once more on another port.)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
               LOCAL_WORLD_SIZE=str(world))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL between processes)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    return env


def _tail(path, n=3000):
    try:
        with open(path, "rb") as fh:
            fh.seek(0, 2)
            size = fh.tell()
            fh.seek(max(0, size - n))
            return fh.read().decode("utf-8", "replace")
    except OSError:
        return ""


def _port_was_taken(e):
    s = str(e)
    return "EADDRINUSE" in s or "Address already in use" in s or "address already in use" in s


def spawn_ranks(cmd, world, timeout=None, env=None, port=None, poll=0.05):
    """Runs `cmd` (argv list) once per rank.  Returns rank 0's standard output (str).
    Raises RankFailure when a rank exits non-zero or the time limit (default QM_RANK_TIMEOUT, 3600 s) passes.
    When the port was probed here (port=None) and the ranks fail because somebody took it between the probe and their
    bind, they are started once more on a fresh port; a second failure is raised like any other."""
    if world < 1:
        raise ValueError("world must be >= 1")
    timeout = DEFAULT_TIMEOUT if timeout is None else timeout
    if port is None:
        try:
            return _spawn_once(cmd, world, timeout, env, free_port(), poll)
        except RankFailure as e:
            if not _port_was_taken(e):
                raise
        return _spawn_once(cmd, world, timeout, env, free_port(), poll)
    return _spawn_once(cmd, world, timeout, env, port, poll)


def _spawn_once(cmd, world, timeout, env, port, poll):
    with tempfile.TemporaryDirectory(prefix="qmvt_ranks_") as tmp:
        outs = [os.path.join(tmp, "rank%d.out" % r) for r in range(world)]
        errs = [os.path.join(tmp, "rank%d.err" % r) for r in range(world)]
        procs = []
        try:
            for r in range(world):
                with open(outs[r], "wb") as fo, open(errs[r], "wb") as fe:
                    procs.append(subprocess.Popen(list(cmd), env=rank_env(r, world, port, env), stdout=fo, stderr=fe))
            t0 = time.monotonic()
            bad, timed_out = [], False
            while True:
                codes = [p.poll() for p in procs]
                bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
                if bad or all(c == 0 for c in codes):
                    break
                if time.monotonic() - t0 > timeout:
                    bad = [r for r, c in enumerate(codes) if c is None]
                    timed_out = True
                    break
                time.sleep(poll)
        finally:
            # a rank that died leaves its peers waiting in a collective: stop exactly the processes started here
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for p in procs:
                p.wait()
        codes = [p.returncode for p in procs]
        if bad:
            tails = ["--- rank %d (exit %s) ---\n%s%s" % (r, codes[r], _tail(outs[r], 1000), _tail(errs[r])) for r in bad]
            raise RankFailure("rank(s) %s %s:\n%s" % (bad, "timed out after %.0f s" % timeout if timed_out else "failed", "\n".join(tails)),
                              bad, codes)
        with open(outs[0], "rb") as fh:
            return fh.read().decode("utf-8", "replace")


def main_relay(cmd, world, timeout=None):
    """For command-line tools: run the ranks, print rank 0's output, exit code 0 / 1."""
    try:
        out = spawn_ranks(cmd, world, timeout=timeout)
    except RankFailure as e:
        sys.stderr.write(str(e) + "\n")
        return 1
    sys.stdout.write(out)
    sys.stdout.flush()
    return 0
