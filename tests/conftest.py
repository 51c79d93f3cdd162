import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # the CPU oracle works on small tensors: a 256-thread OpenMP team (GPU box host) makes every op slower
    import torch
    torch.set_num_threads(min(8, os.cpu_count() or 1))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def repo_root():
    return REPO


# Collection order of the `-m gpu` suite (the driver runs it with -x: the first failure hides everything after it).  The bit-exact contract
# of the hot path goes first, the rows whose oracle is pinned by the reference's own classes next, the rows whose oracle is an unpinned
# restatement (diffusers / torchvision absent here: MMDiT, AutoencoderKL, Inception-V3) last.
_GPU_ORDER = ["test_gpu_ni_step", "test_philox", "test_gpu_ncsnpp", "test_gpu_accuracy", "test_gpu_conv_gn", "test_gpu_gemm_epilogue",
              "test_gpu_ddpm", "test_gpu_dit", "test_gpu_fid50k", "test_gpu_bench_multirank", "test_gpu_rccl_world1", "test_gpu_vae", "test_gpu_inception", "test_gpu_mmdit", "test_gpu_sd3_shard"]


def pytest_collection_modifyitems(config, items):
    def key(item):
        mod = Path(str(item.fspath)).stem
        return _GPU_ORDER.index(mod) if mod in _GPU_ORDER else len(_GPU_ORDER) - 4.5      # unknown modules: before the unpinned rows
    # only the gpu-marked items are re-ordered (the CPU suite keeps pytest's own order, in front); stable: the order inside a module is kept
    gpu = [it for it in items if it.get_closest_marker("gpu") is not None]
    cpu = [it for it in items if it.get_closest_marker("gpu") is None]
    items[:] = cpu + sorted(gpu, key=key)
