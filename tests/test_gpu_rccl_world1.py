"""RCCL once before an 8-GPU node does it for us (round-5 review, item 3): every collective of the N > 1 path through a process group of ONE
rank on the one GPU a test box has -- `init_process_group("nccl", device_id=dev)`, `timed_region`'s barriers and its device-tensor
`all_reduce(MAX)`, and the 33.6 MB fp64 `all_reduce(SUM)` of `calc_fid_sharded` (`fid_stats.ActivationStats.all_reduce`) on the device.
The rank is a CHILD of this process (a GPU-initialised process must not exec; a child is the allowed form).  Replaces the reference's
single-process `nn.DataParallel` (deps/score_sde_pytorch/models/utils.py:93) and its one-process FID (src/CIFAR10NaturalInference.py:311-312)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_bench_line_through_an_rccl_group_of_one_rank():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--force-pg", "--backend", "nccl", "--batch", "64", "--steps", "1", "--warmup", "1",
                        "--fid-samples", "600", "--fid-share-of", "1", "--no-sd3", "--no-validate", "--no-cpu-baseline", "--no-roofline"],
                       env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["config"]["process_group"] == "nccl x1"
    assert line["n_gpus"] == 1 and line["value"] > 0
    f = line["fid50k"]
    assert f["collective"] == "nccl x1" and f["s"]["allreduce"] > 0 and f["images"] == 600 and f["fid"] == "blocked"


SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from naturaldiffusion_amd.fid_stats import ActivationStats
from naturaldiffusion_amd.shard import max_over_ranks, gather_images
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
st = ActivationStats(2048, device=dev)
g = torch.Generator(device=dev).manual_seed(5)
st.update(torch.randn(300, 2048, device=dev, generator=g))
n, s1, s2 = st.n.clone(), st.s1.clone(), st.s2.clone()
st.all_reduce()                                   # 1 + 2048 + 2048^2 doubles = 33.6 MB through RCCL, in place on the device
assert st.s2.is_cuda and st.s2.dtype == torch.float64
assert torch.equal(st.n, n) and torch.equal(st.s1, s1) and torch.equal(st.s2, s2)      # a sum over one rank is the operand, bit for bit
assert max_over_ranks(1.25, device=dev) == 1.25
imgs = torch.arange(5 * 12, dtype=torch.uint8, device=dev).reshape(5, 12)
full = gather_images(imgs, torch.tensor([4, 0, 2, 1, 3], device=dev), 5)
assert torch.equal(full[torch.tensor([4, 0, 2, 1, 3], device=dev)], imgs)
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
"""


def test_statistics_all_reduce_on_the_device_through_rccl():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = _env()
    env.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29731")
    p = subprocess.run([sys.executable, "-c", SCRIPT.format(root=str(ROOT))], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_WORLD1_OK" in p.stdout, p.stderr[-3000:]
