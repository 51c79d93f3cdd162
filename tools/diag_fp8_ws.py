"""Which workspace bytes does the fp8 MMDiT forward read before it writes them?  (tests/test_gpu_mmdit.py's SD3-width fp8 case differs from run to run when earlier
tests of the same process left finite data in the allocator's blocks; in a fresh process -- zeros -- and with NaN-filled blocks it does not.)  The engine's workspace
is a tensor of the wrapper: fill a byte range with finite garbage, run, compare with the zero-filled run, bisect the range."""
import json, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from naturaldiffusion_amd.mmdit import MMDiTEngine, flatten_state_dict
from test_gpu_mmdit import _sd3_width_case

fp8 = "bf16" not in sys.argv
B = 8
cfg, P, (x, t, e, p), ref = _sd3_width_case()
eng = MMDiTEngine(flatten_state_dict(P, 64, **cfg), max_batch=B, grid=64, ctx_tokens=333, fp8=fp8, **cfg)
rep = lambda v: v.cuda().repeat(B, *([1] * (v.dim() - 1)))
X, T, E, Pq = rep(x), rep(t), rep(e), rep(p)
n = eng._ws.numel()
g = torch.Generator(device="cuda").manual_seed(1)
junk = torch.randint(0, 0x60, (n,), dtype=torch.uint8, device="cuda", generator=g)


def run(lo, hi):
    eng._ws.zero_()
    eng._ws[lo:hi] = junk[lo:hi]
    return eng.forward(X, T, E, Pq).clone()


base = run(0, 0)
assert torch.equal(base, run(0, 0)), "not deterministic on a zeroed workspace"
full = run(0, n)
print("workspace bytes", n, "per sequence", n // B, "| garbage everywhere changes the output:", not torch.equal(full, base),
      "max rel", ((full - base).abs().max() / base.abs().max()).item(), flush=True)
found = []


def bisect(lo, hi):
    if torch.equal(run(lo, hi), base):
        return
    if hi - lo <= 4096:
        found.append((lo, hi)); return
    mid = (lo + hi) // 2 // 256 * 256
    bisect(lo, mid); bisect(mid, hi)
    # (a difference that needs garbage on BOTH sides of mid shows up as "neither half": report the parent range then)
    if not any(lo <= a and b <= hi for a, b in found):
        found.append((lo, hi))


if not torch.equal(full, base):
    bisect(0, n)
# merge neighbours, report per-sequence offsets (Ctx::at: ws + off * B)
found.sort()
merged = []
for a, b in found:
    if merged and a <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], b)
    else:
        merged.append([a, b])
Tx, Tc, Tp, D, Jd, Pd, C, L = 4096, 333, 4480, 1536, 4096, 2048, 16, 2
nmod = (12 * (L - 1) + 8 + 2) * D
names = [("x", Tx * D * 4), ("e", Tc * D * 4), ("hx", Tx * D * 2), ("he", Tc * D * 2), ("qk", Tp * 2 * D * 2), ("vT", D * Tp * 2), ("o", Tp * D * 2), ("fx", Tx * 4 * D * 2),
         ("fe", Tc * 4 * D * 2), ("mod", nmod * 4), ("tf", 512), ("a1", D * 2), ("a2", D * 2), ("pl", Pd * 2), ("cf", D * 4), ("cs", D * 2), ("tx", Tc * Jd * 2), ("tok", Tx * C * 4 * 4)]
if fp8:
    names += [("hx8", Tx * D), ("sx", Tx * 4), ("o8", Tp * D), ("omx", Tp * D // 32), ("f8", Tx * 4 * D), ("fmx", Tx * 4 * D // 32), ("mx_slack", 2048)]
off, table = 0, []
for nm, sz in names:
    sz = (sz + 255) // 256 * 256
    table.append((nm, off, off + sz)); off += sz
print("arena bytes per sequence by this table:", off, flush=True)
out = []
for a, b in merged:
    pa, pb = a / B, b / B
    hit = [f"{nm}+{int(pa - lo)} (of {hi - lo})" for nm, lo, hi in table if lo <= pa < hi]
    out.append({"ws_bytes": [a, b], "per_sequence_offset": [pa, pb], "buffer": hit})
    print(out[-1], flush=True)
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / f"diag_fp8_ws_{'fp8' if fp8 else 'bf16'}.json").write_text(json.dumps(out, indent=1))
