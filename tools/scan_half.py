"""Hand-pipelined kernels with every wave issuing its own LDS-DMA pieces (16, 13) against one issuing wave per SIMD (26, 27).  GPU box."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
argv, sys.argv = sys.argv, sys.argv[:1]
import tools.bench_gemm as BG   # noqa: E402
for v in (26, 27):
    print(f"check {BG.NAMES[v]}: gemm {BG.run(v, 1024, 512, 256, 128, 1, 0, iters=2, check_ref=True)[2]:.1e} {BG.run(v, 1000, 392, 192, 0, 1, 0, iters=2, check_ref=True)[2]:.1e}"
          f"  conv {BG.check_conv(v, 4, 16, 256, 256, 128):.1e} {BG.check_conv(v, 2, 32, 64, 512, 0):.1e}")
for _ in range(2):
    for (M, N, K, taps, res, vs) in [(131072, 256, 2304, 9, 16, (16, 26)), (131072, 256, 4608, 9, 16, (16, 26)), (524288, 256, 2304, 9, 32, (16, 26)), (32768, 1536, 1536, 1, 0, (16, 26)),
                                     (32768, 6144, 1536, 1, 0, (16, 26)), (524288, 128, 1152, 9, 32, (13, 27)), (524288, 128, 2304, 9, 32, (13, 27))]:
        cells = []
        for v in vs:
            ms, tf, _ = BG.run(v, M, N, K, 0, taps, res, iters=20)
            cells.append(f"{BG.NAMES[v]} {ms*1e3:6.1f} us {tf:5.0f} TF/s")
        print(f"{(M, N, K, taps)}: " + " | ".join(cells), flush=True)
