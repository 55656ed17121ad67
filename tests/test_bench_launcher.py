"""`python bench.py --gpus N` without an external launcher: the parent starts one process per GPU, relays rank 0's one
JSON line, stops everybody when a rank fails and exits non-zero.  The step is injected, so
nothing here needs a GPU; quasimodo_amd.launch is the code under test."""
import json
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT


def _run(inject, extra=(), env_extra=None, timeout=90):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    env["QM_BENCH_INJECT"] = "bench_inject:" + inject
    env["PYTHONPATH"] = os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", "")
    env.update(env_extra or {})
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "7", "--warmup", "2", *extra],
                       env=env, capture_output=True, text=True, timeout=timeout)
    return p, time.monotonic() - t0


def test_bench_gpus_n_starts_its_own_ranks_and_prints_one_line():
    p, _ = _run("step")
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out == {"metric": "injected", "n_gpus": 3, "steps": 7, "warmup": 2, "rank": 0}


def test_bench_a_failing_rank_stops_the_others_and_fails_the_run():
    p, dt = _run("fail_on_rank1")
    assert p.returncode != 0 and dt < 60          # the sleeping ranks were stopped, not waited for
    assert "rank 1 breaks on purpose" in p.stderr and "rank(s) [1] failed" in p.stderr
    assert p.stdout.strip() == ""


def test_bench_a_hanging_run_times_out():
    p, dt = _run("hang", env_extra={"QM_BENCH_TIMEOUT": "3"})
    assert p.returncode != 0 and dt < 60 and "timed out" in p.stderr


def test_bench_under_an_external_launcher_every_process_is_a_rank():
    """the torch.distributed.run form: WORLD_SIZE is set, nothing is spawned"""
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999",
               QM_BENCH_INJECT="bench_inject:step", PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and json.loads(p.stdout)["n_gpus"] == 3
    env["WORLD_SIZE"] = "2"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], env=env, capture_output=True, text=True, timeout=60)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_spawn_ranks_env_and_relay(tmp_path):
    from quasimodo_amd.launch import RankFailure, spawn_ranks
    script = tmp_path / "r.py"
    script.write_text("import os, sys\nr = int(os.environ['RANK'])\nprint('rank', r, os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK'])\n"
                      "sys.exit(3 if r == 2 and os.environ.get('BREAK') else 0)\n")
    out = spawn_ranks([sys.executable, str(script)], 4, timeout=60)
    assert out == "rank 0 4 0\n"
    with pytest.raises(RankFailure) as ei:
        spawn_ranks([sys.executable, str(script)], 4, timeout=60, env=dict(os.environ, BREAK="1"))
    assert ei.value.bad == [2] and ei.value.returncodes[2] == 3


def test_spawn_ranks_starts_once_more_when_the_port_was_taken(tmp_path):
    """The probed port can be taken before the children bind (the docstring promised a retry nobody made):
    ranks that fail with "Address already in use" are started once more on another port; any other failure, and a second
    one, is raised."""
    from quasimodo_amd.launch import RankFailure, spawn_ranks
    flag = tmp_path / "seen"
    prog = ("import os, sys\n"
            "flag = %r\n"
            "if os.environ['RANK'] == '0' and not os.path.exists(flag):\n"
            "    open(flag, 'w').write(os.environ['MASTER_PORT'])\n"
            "    sys.stderr.write('RuntimeError: The server socket has failed to listen on any local network address. "
            "port: 1, useIpv6: false, code: -98, name: EADDRINUSE, message: address already in use\\n')\n"
            "    sys.exit(1)\n"
            "print('port', os.environ['MASTER_PORT'])\n") % str(flag)
    out = spawn_ranks([sys.executable, "-c", prog], 2, timeout=60)
    assert out.startswith("port ") and out.split()[1] != flag.read_text()          # the second start used another port
    always = "import sys; sys.stderr.write('EADDRINUSE\\n'); sys.exit(1)"
    with pytest.raises(RankFailure):
        spawn_ranks([sys.executable, "-c", always], 2, timeout=60)
    flag.unlink()
    with pytest.raises(RankFailure):                                                  # a port the caller chose is not replaced
        spawn_ranks([sys.executable, "-c", prog], 2, timeout=60, port=29871)
