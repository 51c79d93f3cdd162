"""Tile variants on the small-M plain GEMM shapes of the MMDiT text stream (M = 8 x 333 rows): time and TFLOP/s per variant (GPU box)."""
import sys
sys.argv = sys.argv[:1]
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
import tools.bench_gemm as BG
shapes = [(2664, 1536, 1536), (2664, 4608, 1536), (2664, 6144, 1536), (2664, 1536, 6144)]
vs = [0, 17, 4, 6, 8, 9, 26, 27]
print(f"{'shape':>24} " + " ".join(f"{BG.NAMES[v]:>13}" for v in vs))
for (M, N, K) in shapes:
    cells = []
    for v in vs:
        try:
            ms, tf, _ = BG.run(v, M, N, K, 0, 1, 0, iters=20)
            cells.append(f"{ms*1e3:6.1f}us{tf:5.0f}")
        except Exception as e:
            cells.append("     n/a")
    print(f"{str((M, N, K)):>24} " + " ".join(f"{c:>13}" for c in cells), flush=True)
