"""Two batches of 512 on two HIP streams: both streams on the whole GPU against each stream on its own half of every XCD's compute units
(hipExtStreamCreateWithCUMask; mask bit i = XCD i % 8, tools/probe/cumask_probe.hip).  ms per 512 images, forward only."""
import ctypes, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd.ncsnpp import NCSNppEngine
from naturaldiffusion_amd.synth import synthetic_flat_params
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << j for j in range(32) if (32 * w + j) in bits) for w in range(8)])
    s = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words) == 0
    return torch.cuda.ExternalStream(s.value)
p = synthetic_flat_params(0)
ea, eb = NCSNppEngine(p, max_batch=512), NCSNppEngine(p, max_batch=512)
xa = torch.randn(512, 3, 32, 32, device="cuda"); xb = torch.randn(512, 3, 32, 32, device="cuda"); t = torch.rand(512, device="cuda") * 999
ra, rb = ea(xa, t).clone(), eb(xb, t).clone()
cases = {"both on the whole GPU": (torch.cuda.Stream(), torch.cuda.Stream()),
         "A on CUs 0-15, B on CUs 16-31 of every XCD": (masked_stream(set(range(128))), masked_stream(set(range(128, 256)))),
         "A on 0-19, B on 12-31 (8 shared)": (masked_stream(set(range(160))), masked_stream(set(range(96, 256)))),
         "A whole GPU, B on CUs 16-31": (torch.cuda.Stream(), masked_stream(set(range(128, 256))))}
for rep in range(2):
    for name, (sa, sb) in cases.items():
        def par(n):
            for _ in range(n):
                with torch.cuda.stream(sa): oa = ea(xa, t)
                with torch.cuda.stream(sb): ob = eb(xb, t)
            return oa, ob
        par(2); torch.cuda.synchronize(); t0 = time.perf_counter(); oa, ob = par(8); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 16 * 1e3
        print(f"{name:46s}: {dt:.3f} ms per 512 images; identical {torch.equal(oa, ra) and torch.equal(ob, rb)}", flush=True)
