"""Where one 256x256 tile's time goes: shader-clock stamps of block 0 (natinf_debug_timestamps) for a K-scan (GPU box).
usage: tile_timeline.py [variant]"""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
argv, sys.argv = sys.argv, sys.argv[:1]
import tools.bench_gemm as BG   # noqa: E402
from naturaldiffusion_amd._lib import lib
v = int(argv[1]) if len(argv) > 1 else 16
ts = torch.zeros(16, dtype=torch.int64, device="cuda")
lib.natinf_debug_timestamps(ts.data_ptr())
names = ["start", "tile0 landed", "main loop", "p0 slab", "p0 sweeps", "p0 stores", "p1 slab", "p1 sweeps", "p1 stores"]
for (M, N, K) in [(65536, 256, 64), (65536, 256, 1024), (65536, 256, 2304), (32768, 1536, 1536)]:
    ms, tf, _ = BG.run(v, M, N, K, 0, 1, 0, iters=20)
    torch.cuda.synchronize()
    t = ts.cpu().tolist()
    ts.zero_()
    if t[5] == 0:        # packed epilogue: stamps 2 (epilogue start), 3 (register phase + slab written), 4 (copy-out issued)
        print(f"{(M, N, K)}: {ms*1e3:.1f} us/launch, {tf:.0f} TF/s; block 0 shader clocks: first K-tile landed {t[1]-t[0]}, main loop {t[2]-t[1]}, "
              f"packed epilogue: registers + slab {t[3]-t[2]}, copy-out {t[4]-t[3]}", flush=True)
        continue
    d = [t[i + 1] - t[i] for i in range(8)]
    print(f"{(M, N, K)}: {ms*1e3:.1f} us/launch, {tf:.0f} TF/s; block 0 shader clocks: total {t[8]-t[0]}")
    print("   " + "  ".join(f"{names[i+1]}: {d[i]}" for i in range(8)), flush=True)
lib.natinf_debug_timestamps(None)
