"""3x3-conv shapes: LDS-resident-patch kernels (14, 15) against the tap-by-tap DMA kernels (16, 13, 9) (GPU box)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
argv, sys.argv = sys.argv, sys.argv[:1]
import tools.bench_gemm as BG   # noqa: E402
for _ in range(2):
    for (M, N, K, res, vs) in [(131072, 256, 2304, 16, (16, 14)), (131072, 256, 4608, 16, (16, 14)), (524288, 128, 1152, 32, (13, 9, 15)), (524288, 128, 2304, 32, (13, 9, 15)),
                               (524288, 256, 2304, 32, (16, 14))]:
        cells = []
        for v in vs:
            ms, tf, _ = BG.run(v, M, N, K, 0, 9, res, iters=20)
            cells.append(f"{BG.NAMES[v]} {ms*1e3:6.1f} us {tf:5.0f} TF/s")
        print(f"{(M, N, K)}: " + " | ".join(cells), flush=True)
