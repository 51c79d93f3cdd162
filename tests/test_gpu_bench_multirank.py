"""The N > 1 branch of bench.py on real hardware (round-2 review, weak #12: "has never executed anywhere"): `python bench.py --gpus 2`
with no external launcher starts its two ranks itself; on a one-GPU box both ranks share cuda:0 (--same-device, gloo for the barrier and
the max-over-ranks reduction -- RCCL refuses two ranks on one device).  Everything else is the production path: per-rank seeds, the
sharded CIFAR10 workload on the HIP engine, barrier-bracketed timing, rank 0's single JSON line."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_two_ranks_self_launched_on_the_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", "--batch", "64", "--steps", "1",
                        "--warmup", "1", "--fid-samples", "600", "--no-cpu-baseline", "--no-roofline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["batch_per_gpu"] == 64
    assert line["config"]["sharding"].startswith("batch x2")
    assert line["value"] > 0 and abs(line["value"] - 2 * 64 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-3      # whole-job images / max-over-ranks time
    # the N > 1 default line: no SD3 / validate objects (they are `--workload` runs there), but config 3's sharded job with its all-reduce across the ranks
    assert "sd3" not in line and "validate" not in line
    f = line["fid50k"]
    assert f["share_of"] == 2 and f["images"] == 300 and f["s"]["allreduce"] > 0 and f["fid"] == "blocked" and f["value"] > 0
