"""Same-process A/B of a gemm knob (argv[1], default natinf_set_gemm_w128) on DiT-XL/2 forwards at B = 8 / 16 / 64 (GPU box)."""
import sys, time, torch
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
from naturaldiffusion_amd._lib import lib, check
from naturaldiffusion_amd.dit import DiTEngine, flatten_state_dict, XL2
from naturaldiffusion_amd.synth import synthetic_dit_state_dict
KNOB = sys.argv[1] if len(sys.argv) > 1 else "natinf_set_gemm_w128"
flat = flatten_state_dict(synthetic_dit_state_dict(XL2["depth"], XL2["hidden"], seed=0), XL2["depth"], XL2["hidden"])
for B in (8, 16, 64):
    eng = DiTEngine(flat, B, **XL2)
    z = torch.randn(B, 4, 32, 32, device="cuda"); t = torch.full((B,), 500.0, device="cuda"); y = torch.zeros(B, dtype=torch.int32, device="cuda")
    for rep in range(2):
        for v in (0, 1):
            check(getattr(lib, KNOB)(v), "set")
            for _ in range(3): eng(z, t, y)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): eng(z, t, y)
            torch.cuda.synchronize()
            print(f"B={B} {KNOB}({v}): {(time.perf_counter() - t0) * 100:.3f} ms per forward", flush=True)
    del eng
