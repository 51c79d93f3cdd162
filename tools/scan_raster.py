"""Tile rasterisation of wide-N GEMMs: plain row-major (0) against groups of 4 / 8 / 16 row-tiles (GPU box)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
argv, sys.argv = sys.argv, sys.argv[:1]
import tools.bench_gemm as BG   # noqa: E402
from naturaldiffusion_amd._lib import lib
for _ in range(2):
    for (M, N, K) in [(32768, 6144, 1536), (32768, 3072, 1536), (32768, 1536, 6144), (65536, 4608, 1152), (8192, 8192, 8192)]:
        cells = []
        for gsz in (0, 4, 8, 16):
            lib.natinf_set_gemm_raster(gsz)
            ms, tf, err = BG.run(16, M, N, K, 0, 1, 0, iters=20)
            cells.append(f"g={gsz}: {ms*1e3:6.1f} us {tf:5.0f} TF/s")
        print(f"{(M, N, K)}: " + " | ".join(cells), flush=True)
lib.natinf_set_gemm_raster(8)
print("check g=8:", BG.run(16, 2304, 2560, 256, 0, 1, 0, iters=2, check_ref=True)[2], BG.run(16, 2000, 2312, 128, 0, 1, 0, iters=2, check_ref=True)[2])
