"""Tile variants on DiT-XL/2's GEMM shapes at Validate's batch (M = 16 x 256 = 4,096 rows): time and TFLOP/s per variant (GPU box)."""
import sys
sys.argv = sys.argv[:1]
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
import tools.bench_gemm as BG
shapes = [(4096, 3456, 1152), (4096, 1152, 1152), (4096, 4608, 1152), (4096, 1152, 4608), (2048, 1152, 1152), (2048, 3456, 1152), (2048, 4608, 1152)]
vs = [0, 29, 17, 9, 8]
print(f"{'shape':>24} " + " ".join(f"{BG.NAMES[v]:>14}" for v in vs))
for (M, N, K) in shapes:
    cells = []
    for v in vs:
        try:
            ms, tf, _ = BG.run(v, M, N, K, 0, 1, 0, iters=20)
            cells.append(f"{ms*1e3:6.1f}us{tf:5.0f}")
        except Exception as e:
            cells.append("     n/a")
    print(f"{str((M, N, K)):>24} " + " ".join(f"{c:>14}" for c in cells), flush=True)
