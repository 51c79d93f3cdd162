"""What bounds the K loop of the 256x256 hand-pipelined kernel: time per 64-wide K-tile with everything (16), without the
LDS-DMA after the first tile (24: operands stale) and without the MFMAs (25: DMA + fragment reads + waits only).  GPU box."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
argv, sys.argv = sys.argv, sys.argv[:1]
import tools.bench_gemm as BG   # noqa: E402
BG.NAMES.update({24: "no-DMA", 25: "no-MFMA"})
for (M, N) in [(65536, 256), (32768, 1536)]:
    for v in (16, 24, 25):
        t = {}
        for K in (1024, 4096):
            t[K] = BG.run(v, M, N, K, 0, 1, 0, iters=20)[0] * 1e3
        tiles = (M // 256) * (N // 256) / 256
        print(f"{(M, N)} {BG.NAMES[v]:>12}: {t[1024]:7.1f} us at K=1024, {t[4096]:7.1f} us at K=4096 -> {(t[4096]-t[1024])/48/tiles:5.2f} us per K-tile and tile", flush=True)

# 3x3 convolution (taps inner: 8 of 9 A reads and all weight reads are L2 hits): K = 9 * C
for (M, N, res) in [(131072, 256, 16), (524288, 256, 32)]:
    for v in (16, 24, 25):
        t = {}
        for C_ in (128, 512):
            t[C_] = BG.run(v, M, N, 9 * C_, 0, 9, res, iters=20)[0] * 1e3
        tiles = (M // 256) * (N // 256) / 256
        print(f"conv {(M, N)} {BG.NAMES[v]:>12}: {t[128]:7.1f} us at C=128, {t[512]:7.1f} us at C=512 -> {(t[512]-t[128])/54/tiles:5.2f} us per K-tile and tile", flush=True)
