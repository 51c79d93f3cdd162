"""The head_dim-64 attention kernels alone at the SD3 shape (8 sequences x 24 heads x 4,429 tokens): time and useful TFLOP/s per natinf_set_flash_mode value.  usage: bench_flash.py [modes...]"""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd._lib import lib, check, ptr, stream_ptr
B, H, T = 8, 24, 4429
Tp, D = (T + 127) // 128 * 128, H * 64
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(B, Tp, D, device="cuda", generator=g).bfloat16(); k = torch.randn(B, Tp, D, device="cuda", generator=g).bfloat16()
vT = torch.randn(B, D, Tp, device="cuda", generator=g).bfloat16(); o = torch.empty(B, Tp, D, device="cuda", dtype=torch.bfloat16)
def run(n):
    for _ in range(n):
        check(lib.natinf_attention_hd64_bf16(ptr(q), ptr(k), D, Tp * D, ptr(vT), ptr(o), D, Tp * D, B, H, Tp, T, 0.125, stream_ptr()), "attn")
modes = [int(v) for v in sys.argv[1:]] or [0, 3]        # (modes 1, 2: -DNATINF_DEV builds)
ref = None
for rep in range(3):                                    # interleaved rounds in ONE process (cdna guide rule 24)
    for m in modes:
        check(lib.natinf_set_flash_mode(m), "mode")
        run(2); torch.cuda.synchronize(); t0 = time.perf_counter(); run(10); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        if rep == 0:
            cur = o[:, :T].float().clone()
            if ref is None: ref = cur
            err = float((cur - ref).abs().max() / ref.abs().max())
        print(f"mode {m} round {rep}: {dt*1e6:.0f} us  {4.0*T*T*64*H*B/dt/1e12:.0f} TF/s useful" + (f"  max rel diff vs mode {modes[0]}: {err:.2e}" if rep == 0 else ""), flush=True)
