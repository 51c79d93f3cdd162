"""``python bench.py --gpus N`` without an external launcher must start its own ranks: the parent spawns a child
``torch.distributed.run`` before it touches any GPU and relays rank 0's JSON line.  Driven here on CPU with the ``selftest`` workload
(gloo, world size 2): the same argument parsing, launcher, rendezvous, barrier-bracketed timed region and max-over-ranks reduction the
GPU workloads use, with no kernels inside."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "selftest", "--backend", "gloo", "--steps", "2", "--warmup", "1", *extra],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def test_self_launch_two_ranks_gloo():
    p = _run("--gpus", "2")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                       # exactly ONE JSON line on stdout, whatever the children printed
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["rank_sum"] == 3.0               # ranks 0 and 1 both took part in the collective
    assert line["value"] > 0 and line["ms_per_step"] > 0


def test_single_rank_needs_no_launcher():
    p = _run("--gpus", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_failing_child_gives_nonzero_exit_and_no_line():
    p = _run("--gpus", "2", "--selftest-fail-rank", "1")
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_rank_count_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "selftest", "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode != 0 and "ranks" in p.stderr


def test_force_pg_runs_the_collectives_through_a_group_of_one_rank():
    """--force-pg (round 6): the process group is initialised with ONE rank -- a rendezvous of bench.py's own when no launcher set one up -- and the timed region's barriers,
    its max-over-ranks all-reduce and the workload's all-reduce run through it.  On a GPU box with --backend nccl this is the RCCL rehearsal (tests/test_gpu_rccl_world1.py)."""
    p = _run("--gpus", "1", "--force-pg")
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["process_group"] == "gloo x1" and line["config"]["rank_sum"] == 1.0
