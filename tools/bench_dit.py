#!/usr/bin/env python3
"""DiT-XL/2 engine: error against the CPU oracle and forward throughput by batch (GPU box)."""
import json, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import dit_oracle as D
from naturaldiffusion_amd.dit import DiTEngine, flatten_state_dict

GFLOP = None
def main():
    P = D.make_params(28, 1152, seed=3)
    flat = flatten_state_dict(P, 28, 1152)
    out = {}
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 4, 32, 32, generator=g); t = torch.tensor([999.0, 3.0]); y = torch.tensor([1000, 207])
    ref = D.forward(P, x, t, y, 16).numpy()
    for B in (2, 16, 64, 128, 256):
        eng = DiTEngine(flat, max_batch=B)
        if B == 2:
            o = eng(x.cuda(), t.cuda(), y.cuda()).cpu().numpy()
            out["rel_err"] = float(np.abs(o - ref).max() / np.abs(ref).max())
        xb = torch.randn(B, 4, 32, 32, device="cuda"); tb = torch.full((B,), 500.0, device="cuda"); yb = torch.zeros(B, dtype=torch.int32, device="cuda")
        for _ in range(2): eng(xb, tb, yb)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 5
        for _ in range(n): eng(xb, tb, yb)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        # 2*MAC: per token 24 D^2 (qkv 6, proj 2, mlp 16) + attention 4 T D per token; 256 tokens, 28 blocks
        fl = 28 * 256 * (24 * 1152 * 1152 + 4 * 256 * 1152) * B
        out[f"B{B}"] = {"ms": round(dt * 1e3, 3), "img_s": round(B / dt, 1), "TFLOPs": round(fl / dt / 1e12, 1), "ws_GB": round(eng.workspace_bytes / 1e9, 2)}
        del eng
    print(json.dumps(out, indent=1))
main()
