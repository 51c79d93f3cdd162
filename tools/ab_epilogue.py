"""Same-box A/B of the GEMM epilogues on the NCSN++ forward at B=512 and a few plain GEMMs: packed bf16 (0) vs fp32 slab (1)."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
argv, sys.argv = sys.argv, sys.argv[:1]
import tools.bench_gemm as BG   # noqa: E402
from naturaldiffusion_amd._lib import lib, check
from naturaldiffusion_amd.ncsnpp import NCSNppEngine
from naturaldiffusion_amd.synth import synthetic_flat_params
for mode in (1, 0):
    check(lib.natinf_set_gemm_epilogue(mode), "set")
    for (v, M, N, K0, K1, taps, res) in [(16, 65536, 256, 64, 0, 1, 0), (16, 32768, 1536, 1536, 0, 1, 0), (16, 131072, 256, 2304, 0, 9, 16), (9, 524288, 128, 1152, 0, 9, 32)]:
        ms, tf, err = BG.run(v, M, N, K0, K1, taps, res, iters=20, check_ref=(taps == 1 and M <= 65536))
        print(f"epilogue {'fp32-slab' if mode else 'packed'} {BG.NAMES[v]} {(M, N, K0, taps)}: {ms*1e3:.1f} us {tf:.0f} TF/s err {err}", flush=True)
eng = NCSNppEngine(synthetic_flat_params(0), max_batch=512)
x = torch.randn(512, 3, 32, 32, device="cuda"); t = torch.rand(512, device="cuda") * 999
outs = {}
for rep in range(2):
    for mode in (1, 0):
        check(lib.natinf_set_gemm_epilogue(mode), "set")
        for _ in range(2): outs[mode] = eng(x, t)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): eng(x, t)
        torch.cuda.synchronize()
        print(f"NCSN++ forward B=512, epilogue {'fp32-slab' if mode else 'packed'}: {(time.perf_counter() - t0) * 100:.2f} ms", flush=True)
d = (outs[0] - outs[1]).abs().max().item() / outs[1].abs().max().item()
print(f"packed vs fp32-slab output: max rel diff {d:.3e}")
check(lib.natinf_set_gemm_epilogue(0), "set")
