#!/usr/bin/env python3
"""SD3-medium MMDiT engine at 1024x1024 (4096 image + 333 text tokens): forward time and TFLOP/s by batch (GPU box)."""
import json, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd.mmdit import MMDiTEngine, SD3_MEDIUM, param_layout
from naturaldiffusion_amd.synth import synthetic_mmdit_flat

def main():
    fp8 = "fp8" in sys.argv
    batches = [int(a) for a in sys.argv[1:] if a != "fp8"] or [2, 8]
    flat = synthetic_mmdit_flat(grid=64, seed=0, **SD3_MEDIUM)
    out = {"params": int(flat.numel())}
    D, L, tx, tc = 1536, 24, 4096, 333
    T = tx + tc
    fl = L * (2.0 * T * 3 * D * D + 4.0 * T * T * D + 2.0 * T * D * D + 2.0 * T * 8 * D * D) - (2.0 * tc * 9 * D * D) + 2.0 * tc * 4096 * D
    for B in batches:
        eng = MMDiTEngine(flat, max_batch=B, grid=64, ctx_tokens=tc, fp8=fp8, **SD3_MEDIUM)
        z = torch.randn(B, 16, 128, 128, device="cuda"); t = torch.full((B,), 500.0, device="cuda")
        e = torch.randn(B, tc, 4096, device="cuda"); p = torch.randn(B, 2048, device="cuda")
        for _ in range(2): o = eng.forward(z, t, e, p)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 3
        for _ in range(n): o = eng.forward(z, t, e, p)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        assert torch.isfinite(o).all()
        out[f"B{B}"] = {"ms": round(dt * 1e3, 2), "seq_s": round(B / dt, 2), "TFLOPs": round(fl * B / dt / 1e12, 1), "ws_GB": round(eng.workspace_bytes / 1e9, 2)}
        del eng
    print(json.dumps(out, indent=1))
main()
