"""Two libnatinf engines on two HIP streams with compute-unit masks: does the bit-level
non-reproducibility need shared compute units, a shared L2, or neither?"""
import ctypes, sys, torch
sys.path.insert(0, "/root/repo")
from naturaldiffusion_amd._lib import lib
from naturaldiffusion_amd.ncsnpp import NCSNppEngine
from naturaldiffusion_amd.synth import synthetic_flat_params
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << j for j in range(32) if (32 * w + j) in bits) for w in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
flat = synthetic_flat_params(0)
B = 64
eA = NCSNppEngine(flat, max_batch=B)
eB = NCSNppEngine(flat, max_batch=B)
x = torch.randn(B, 3, 32, 32, device="cuda"); x2 = torch.randn(B, 3, 32, 32, device="cuda"); t = torch.rand(B, device="cuda") * 999
ref = eA(x, t).clone(); torch.cuda.synchronize()
allb = set(range(256))
cases = {
    "no mask": (None, None),
    "both on all 256 bits": (allb, allb),
    "A bits i%8<4, B bits i%8>=4": ({i for i in allb if i % 8 < 4}, {i for i in allb if i % 8 >= 4}),
    "A bits 0-127, B bits 128-255": (set(range(128)), set(range(128, 256))),
    "A bits i%8==0, B bits i%8==0": ({i for i in allb if i % 8 == 0}, {i for i in allb if i % 8 == 0}),
    "A bits 0-31, B bits 0-31": (set(range(32)), set(range(32))),
    "A alone, bits i%8<4": ({i for i in allb if i % 8 < 4}, "idle"),
}
for name, (ma, mb) in cases.items():
    sa = torch.cuda.Stream() if ma is None else masked_stream(ma)
    idle = mb == "idle"
    sb = torch.cuda.Stream() if (mb is None or idle) else masked_stream(mb)
    bad = 0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); ev0.record()
    for it in range(40):
        if not idle:
            with torch.cuda.stream(sb): eB(x2, t)
        with torch.cuda.stream(sa): o = eA(x, t)
        torch.cuda.synchronize()
        bad += int(not torch.equal(o, ref))
    ev1.record(); torch.cuda.synchronize()
    print(f"{name:34s}: {bad:2d}/40 forwards differ   ({ev0.elapsed_time(ev1) / 40:.2f} ms per pair)", flush=True)
