"""K-loop schedules of k_gemm_w128 side by side (a -DNATINF_DEV library, NATINF_LIB selects it): 26 = the eight-wave 256x256 tile, 29 = the shipped schedule (W128SchB),
30 = W128SchA (two release barriers, requests between the fragment reads), 31 = W128SchP (+ L2 prefetch), 32 = W128SchX (no requests after the prologue: wrong results,
what the requests cost).  usage: ab_w128_sched.py [variant ids] (GPU box)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
vs = [int(a) for a in sys.argv[1:]] or [26, 29, 30, 31, 32]
sys.argv = sys.argv[:1]
import tools.bench_gemm as BG   # noqa: E402
shapes = [(8192, 8192, 8192), (32768, 6144, 1536), (32768, 1536, 1536), (32768, 1536, 6144)]
print(f"{'M,N,K':>22} " + " ".join(f"{BG.NAMES.get(v, 'v%d' % v):>13}" for v in vs) + "   (TFLOP/s; two passes)")
for rep in range(2):
    for (M, N, K) in shapes:
        row = []
        for v in vs:
            ms, tf, err = BG.run(v, M, N, K, 0, 1, 0, iters=20, check_ref=(rep == 0 and v != 32 and M * N <= 1 << 26))
            row.append(f"{tf:8.0f}" + (f"/{err:.0e}" if err is not None else "     "))
        print(f"{str((M, N, K)):>22} " + " ".join(f"{r:>13}" for r in row), flush=True)
