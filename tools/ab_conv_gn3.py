"""Same-process A/B of k_conv_gn3 (natinf_set_conv_gn_w128(7)) against k_conv_gn2 (0) on the engine's 32x32 / 16x16 shapes at B = 512:
ab_conv_gn3.py [rounds].  Interleaved rounds (A B A B ...), 20 launches each; prints ms and TFLOP/s (2*M*N*(9*cin + c1)) per kernel and the ratio."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd._lib import lib, check, ptr, stream_ptr

SHAPES = ((32, 512, 128, 128, 0, 0), (32, 512, 128, 128, 0, 1), (32, 512, 128, 128, 256, 0), (32, 512, 128, 128, 384, 0), (32, 512, 256, 128, 0, 0),
          (32, 512, 384, 128, 0, 0), (32, 512, 256, 256, 0, 0), (32, 512, 256, 256, 256, 0),
          (16, 512, 256, 256, 0, 0), (16, 512, 256, 256, 0, 1), (16, 512, 256, 256, 512, 0), (16, 512, 512, 256, 0, 0), (16, 512, 128, 256, 0, 0))


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    for sh in range(3):
        check(lib.natinf_set_conv_gn_w128_min_k(sh, 0), "min_k")            # every K on k_conv_gn3 (the library's defaults keep short K on k_conv_gn2)
    dev = "cuda"
    for res, B, cin, N, c1, resid in SHAPES:
        M = B * res * res
        x = torch.randn(B, res, res, cin, device=dev).bfloat16()
        sc = torch.rand(B, cin, device=dev) + 0.5; sh = torch.randn(B, cin, device=dev) * 0.3
        w = (torch.randn(N, 9 * cin + c1, device=dev) / (9 * cin) ** 0.5).bfloat16()
        a1 = torch.randn(M, c1, device=dev).bfloat16() if c1 else None
        r = torch.randn(M, N, device=dev).bfloat16() if resid else None
        bias = torch.randn(N, device=dev); out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        part = torch.zeros(M // 128, N // 4, 2, device=dev)
        wf = torch.zeros_like(w)
        args = (res, B, N, cin, c1, ptr(x), ptr(sc), ptr(sh), ptr(w), ptr(wf), ptr(a1), ptr(bias), ptr(r), 0.7071, ptr(out), ptr(part))
        best = {0: 1e9, 7: 1e9}
        outs = {}
        for rd in range(rounds):
            for mask in (0, 7):
                check(lib.natinf_set_conv_gn_w128(mask), "knob")
                check(lib.natinf_debug_conv_gn(*args, 3, stream_ptr()), "warm")
                torch.cuda.synchronize(); t0 = time.perf_counter()
                check(lib.natinf_debug_conv_gn(*args, 20, stream_ptr()), "run")
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
                best[mask] = min(best[mask], dt)
                outs[mask] = out.clone()
        lib.natinf_set_conv_gn_w128(7)
        fl = 2.0 * M * N * (9 * cin + c1)
        same = torch.equal(outs[0].view(torch.int16), outs[7].view(torch.int16))
        print(f"res {res} cin {cin} N {N} c1 {c1} resid {resid}: gn2 {best[0] * 1e3:.3f} ms {fl / best[0] / 1e12:7.1f} TF/s | gn3 {best[7] * 1e3:.3f} ms {fl / best[7] / 1e12:7.1f} TF/s | "
              f"x{best[0] / best[7]:.3f} | same bytes {same}", flush=True)


if __name__ == "__main__":
    main()
