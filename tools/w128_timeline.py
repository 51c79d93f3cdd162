"""Where one 256x256 tile of k_gemm_w128 (one block per tile) spends its time: shader-clock stamps of block 0 (a -DNATINF_DEV library, NATINF_LIB) (GPU box)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.argv = sys.argv[:1]
import tools.bench_gemm as BG   # noqa: E402
from naturaldiffusion_amd._lib import lib, check
check(lib.natinf_set_gemm_w128(3), "set")
ts = torch.zeros(16, dtype=torch.int64, device="cuda")
lib.natinf_debug_timestamps(ts.data_ptr())
for (M, N, K) in [(32768, 1536, 1536), (32768, 6144, 1536), (32768, 1536, 6144), (8192, 8192, 8192)]:
    ms, tf, _ = BG.run(29, M, N, K, 0, 1, 0, iters=20)
    torch.cuda.synchronize()
    t = ts.cpu().tolist(); ts.zero_()
    nk = K // 64
    print(f"{(M, N, K)}: {ms*1e3:.1f} us/launch, {tf:.0f} TF/s; block 0, shader clocks: until K-tile 0 landed {t[1]-t[0]}, K loop {t[5]-t[1]} = {(t[5]-t[1])/nk:.0f} per K-tile, "
          f"first half epilogue {t[6]-t[5]} (registers + slab {t[3]-t[2]}, copy-out {t[4]-t[3]} of the SECOND half), second half {t[7]-t[6]}; total {t[7]-t[0]}", flush=True)
lib.natinf_debug_timestamps(None)
