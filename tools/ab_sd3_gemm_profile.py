"""Same-box A/B of a knob on the SD3-size MMDiT forward (8 sequences) WITH the per-launch table of natinf_gemm_profile: forward time, and the mean launch duration of
every matmul-shaped launch where it runs (HIP events around each launch), per knob value; outputs compared bit for bit.  A fresh engine per value (the text stream's
HIP stream is created at an engine's first forward).
    python tools/ab_sd3_gemm_profile.py natinf_set_mmdit_text_stream 0 1 [fp8]        # round-5 review item 2: the isolated-vs-engine gap, text stream on / off, per kernel
    python tools/ab_sd3_gemm_profile.py natinf_set_gemm_epilogue 0 1                  # the general fp32-slab epilogue everywhere (1): what each packed / direct epilogue buys"""
import ctypes, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd._lib import lib, check
from naturaldiffusion_amd.mmdit import MMDiTEngine, SD3_MEDIUM
from naturaldiffusion_amd.synth import synthetic_mmdit_flat
fp8 = "fp8" in sys.argv
fn = getattr(lib, sys.argv[1]); vals = [int(v) for v in sys.argv[2:] if v != "fp8"] or [0, 1]
cfg = dict(SD3_MEDIUM)
flat = synthetic_mmdit_flat(64, seed=0, **cfg)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(8, 16, 128, 128, device="cuda", generator=g); t = torch.rand(8, device="cuda", generator=g) * 1000
e = torch.randn(8, 333, cfg["joint_dim"], device="cuda", generator=g); p = torch.randn(8, cfg["pooled_dim"], device="cuda", generator=g)
tables, outs = {}, {}
for rep in range(2):
    for v in vals:
        check(fn(v), "set")
        eng = MMDiTEngine(flat, max_batch=8, grid=64, ctx_tokens=333, fp8=fp8, **cfg)
        for _ in range(2): o = eng.forward(x, t, e, p)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): o = eng.forward(x, t, e, p)
        torch.cuda.synchronize()
        print(f"{sys.argv[1]}({v}){' fp8' if fp8 else ''}: {(time.perf_counter() - t0) * 200:.2f} ms per forward of 8 sequences", flush=True)
        outs[v] = o.clone()
        if rep == 1:
            check(lib.natinf_gemm_profile(1), "profile")
            for _ in range(3): eng.forward(x, t, e, p)
            torch.cuda.synchronize()
            check(lib.natinf_gemm_profile(0), "profile")
            buf = ctypes.create_string_buffer(1 << 16)
            check(min(0, lib.natinf_gemm_profile_read(buf, len(buf))), "read")
            tables[v] = {" ".join(r.split()[:7]): (int(r.split()[7]), float(r.split()[8])) for r in buf.value.decode().splitlines()}
        del eng
pass
print("outputs identical:", all(torch.equal(outs[vals[0]], outs[v]) for v in vals))
shape = lambda tag: " ".join(tag.split()[:6])
by = {v: {} for v in vals}
for v in vals:
    for tag, (n, ms) in tables[v].items():
        o = by[v].setdefault(shape(tag), [0, 0.0, set()]); o[0] += n; o[1] += ms; o[2].add(tag.split()[6])
print(f"{'M N K K1 taps batch':30s} launches " + " ".join(f"{'us(%d)' % v:>9s} {'TF/s(%d)' % v:>9s} {'kernel(%d)' % v:24s}" for v in vals))
for sh in by[vals[0]]:
    f = sh.split(); flops = 2.0 * int(f[0]) * int(f[1]) * (int(f[2]) + int(f[3])) * int(f[5])
    cells = []
    for v in vals:
        n, ms, ks = by[v].get(sh, (0, 0.0, set()))
        cells.append(f"{ms / max(n, 1) * 1e3:9.1f} {flops / (ms / max(n, 1) * 1e-3) / 1e12 if ms else 0:9.0f} {'+'.join(sorted(ks)):24s}")
    print(f"{sh:30s} {by[vals[0]][sh][0]:7d}  " + " ".join(cells))
