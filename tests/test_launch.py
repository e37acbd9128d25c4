"""cbgbench_amd.launch: the one-node rank launcher (FileStore rendezvous, no port chosen ahead of time) that bench.py --gpus N,
sample_cli and train_cli go through; the rank split it serves is the reference's pocket loop, sample.py:159.  CPU, gloo."""
import os
import subprocess
import sys
import threading

import pytest

from cbgbench_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from cbgbench_amd import sharding
rank, world, local = sharding.init_process_group("gloo")
mode = sys.argv[1]
if mode == "fail" and rank == 1:
    raise RuntimeError("rank one gives up (the traceback the launcher must show)")
if mode == "collide" and rank == 0 and not os.path.exists(sys.argv[2]):
    open(sys.argv[2], "w").close()
    sys.stderr.write("fake: EADDRINUSE\n")
    sys.exit(7)
mine = sharding.shard_indices(9, rank, world)
sharding.barrier()
t, u = sharding.reduce_max_sum(1.0 + rank, len(mine))
if rank == 0:
    print("RESULT", world, t, u, flush=True)
dist.destroy_process_group()
"""


@pytest.fixture()
def rank_script(tmp_path):
    p = tmp_path / "rank_script.py"
    p.write_text(RANK_SCRIPT.format(root=ROOT))
    return str(p)


def _cli(n, script, *args, timeout=300):
    env = dict(os.environ)
    for k in ("MASTER_ADDR", "MASTER_PORT", "CBGX_RDZV_FILE", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    return subprocess.run([sys.executable, "-m", "cbgbench_amd.launch", "--nproc", str(n), script] + list(args), capture_output=True,
                          text=True, timeout=timeout, cwd=ROOT, env=env)


def test_two_ranks_meet_without_a_port(rank_script):
    p = _cli(2, rank_script, "ok")
    assert p.returncode == 0, p.stderr[-2000:]
    assert "RESULT 2 2.0 9.0" in p.stdout


def test_four_launches_at_once_do_not_collide(rank_script):
    """the collision case of round 5: several 2-rank jobs started in the same instant on one box"""
    res = [None] * 4

    def go(i):
        res[i] = _cli(2, rank_script, "ok")

    th = [threading.Thread(target=go, args=(i,)) for i in range(4)]
    for t in th: t.start()
    for t in th: t.join()
    for p in res:
        assert p.returncode == 0, p.stderr[-2000:]
        assert "RESULT 2 2.0 9.0" in p.stdout


def test_failing_rank_ends_the_job_and_its_traceback_is_shown(rank_script):
    p = _cli(2, rank_script, "fail", timeout=120)      # rank 0 would wait in the barrier for ever: the launcher ends it
    assert p.returncode != 0
    assert "rank 1 of 2 exited with code" in p.stderr and "rank one gives up" in p.stderr
    assert "RESULT" not in p.stdout


def test_address_collision_is_retried(rank_script, tmp_path):
    """a rank that dies naming EADDRINUSE (a library-owned data socket) costs one relaunch, not the job"""
    p = _cli(2, rank_script, "collide", str(tmp_path / "once"))
    assert p.returncode == 0, p.stderr[-2000:]
    assert "launching again" in p.stderr and "RESULT 2 2.0 9.0" in p.stdout


def test_rank_env_drops_an_outer_rendezvous():
    e = launch.rank_env(1, 2, "/tmp/x/store", base={"MASTER_PORT": "1", "MASTER_ADDR": "h", "KEEP": "1"})
    assert "MASTER_PORT" not in e and "MASTER_ADDR" not in e and e["KEEP"] == "1"
    assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["CBGX_RDZV_FILE"]) == ("1", "1", "2", "/tmp/x/store")
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_many_ranks_without_a_rendezvous_is_a_launch_error(monkeypatch):
    from cbgbench_amd import sharding
    for k in ("MASTER_ADDR", "MASTER_PORT", "CBGX_RDZV_FILE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("LOCAL_RANK", "0")
    with pytest.raises(RuntimeError, match="cbgbench_amd.launch"):
        sharding.init_process_group("gloo")
