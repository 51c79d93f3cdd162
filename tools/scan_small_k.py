"""Variant scan on the short-K NIN shapes of the attention blocks (GPU box; run() warms up each case)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
argv, sys.argv = sys.argv, sys.argv[:1]
import tools.bench_gemm as BG   # noqa: E402
for _ in range(2):
    for (M, N, K) in [(131072, 512, 256), (131072, 256, 256), (131072, 256, 1152), (131072, 128, 1152), (32768, 256, 2304), (8192, 256, 2304)]:
        cells = []
        for v in (9, 16, 17, 8, 10):
            ms, tf, _ = BG.run(v, M, N, K, 0, 1, 0, iters=30)
            cells.append(f"{BG.NAMES[v]} {ms*1e3:6.1f} us")
        print(f"{(M, N, K)}: " + " | ".join(cells), flush=True)
