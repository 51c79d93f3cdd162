"""DiT-XL/2 forward at B = 16 (the Validate script's CFG pair of a step as one forward): natinf_gemm_profile table -- mean launch duration of every matmul-shaped
launch where it runs, tiles per launch against the 256 CUs -- plus the per-kernel rocprof-free split (everything else = forward - sum).  (GPU box)
usage: python tools/dit_gemm_profile.py [B]"""
import ctypes, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import dit_oracle as D
from naturaldiffusion_amd._lib import lib, check
from naturaldiffusion_amd.dit import DiTEngine, flatten_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
P = D.make_params(28, 1152, seed=3)
eng = DiTEngine(flatten_state_dict(P, 28, 1152), max_batch=B)
xb = torch.randn(B, 4, 32, 32, device="cuda"); tb = torch.full((B,), 500.0, device="cuda"); yb = torch.zeros(B, dtype=torch.int32, device="cuda")
for _ in range(3): eng(xb, tb, yb)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): eng(xb, tb, yb)
torch.cuda.synchronize(); fwd = (time.perf_counter() - t0) / 10
check(lib.natinf_gemm_profile(1), "profile")
for _ in range(5): eng(xb, tb, yb)
torch.cuda.synchronize(); check(lib.natinf_gemm_profile(0), "profile")
buf = ctypes.create_string_buffer(1 << 16)
check(min(0, lib.natinf_gemm_profile_read(buf, len(buf))), "read")
print(f"DiT-XL/2 forward, B = {B}: {fwd * 1e3:.3f} ms")
print(f"{'M N K K1 taps batch kernel':52s} {'per fwd':>7s} {'us':>8s} {'TF/s':>7s} {'256x256 tiles':>13s} {'ms/fwd':>7s}")
tot = 0.0
for r in buf.value.decode().splitlines():
    f = r.split(); n, ms = int(f[7]), float(f[8]); M, N, K, bt = int(f[0]), int(f[1]), int(f[2]) + int(f[3]), int(f[5])
    us = ms / n * 1e3
    tot += ms / 5
    print(f"{' '.join(f[:7]):52s} {n // 5:7d} {us:8.1f} {2.0 * M * N * K * bt / us / 1e6:7.0f} {((M + 255) // 256) * ((N + 255) // 256) * bt:13d} {ms / 5:7.3f}")
print(f"sum of matmul launches {tot:.3f} ms of {fwd * 1e3:.3f} (attention, LayerNorm-modulate, embeddings, gaps: the rest)")
