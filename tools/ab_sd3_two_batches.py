"""SD3 28-step NI: 4-image batches one after the other on one stream vs consecutive batches alternating between two HIP streams (two MMDiT engine
handles).  usage: ab_sd3_two_batches.py [--fp8]"""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from naturaldiffusion_amd.coeff import load_sd3_csv
from naturaldiffusion_amd.mmdit import MMDiTEngine, SD3_MEDIUM
from naturaldiffusion_amd.sampler import SD3NI
from naturaldiffusion_amd.synth import synthetic_mmdit_flat
fp8 = "--fp8" in sys.argv
dev = "cuda:0"; n, tc, nstep = 4, 333, 28
W = load_sd3_csv("/root/repo/weights/" + ("sd3_step_28_weight_sharp.csv" if fp8 else "sd3_step_28_weight.csv"))
u = np.linspace(1.0, 3 * 0.001 / (1 + 2 * 0.001), nstep)
sig = np.append(3 * u / (1 + 2 * u), 0.0).astype(np.float32)
sigmas, timesteps = torch.from_numpy(sig).to(dev), torch.from_numpy(sig[:-1] * 1000).to(dev)
flat = synthetic_mmdit_flat(grid=64, seed=0, **SD3_MEDIUM)
g = torch.Generator(device=dev).manual_seed(10)
noises = torch.randn(n, 16, 128, 128, device=dev, dtype=torch.float16, generator=g)
text = torch.randn(2 * n, tc, 4096, device=dev, generator=g); pooled = torch.randn(2 * n, 2048, device=dev, generator=g)
zflat = noises.reshape(-1)
lanes = []
for i in range(2):
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        lanes.append((MMDiTEngine(flat, max_batch=2 * n, grid=64, ctx_tokens=tc, device=dev, fp8=fp8, **SD3_MEDIUM), SD3NI(W, sigmas, noises.numel(), device=dev, cfg=7.0), st))
torch.cuda.synchronize()
def one(eng, ni):
    x = ni.first_input(zflat)
    for k in range(nstep):
        xx = x.view(n, 16, 128, 128)
        v = eng.forward(torch.cat([xx, xx]), timesteps[k].expand(2 * n), text, pooled)
        mean, x = ni.step(k, x, v[:n].reshape(-1), v[n:].reshape(-1), zflat, want_next=k + 1 < nstep)
    return mean
def run(nstr, steps):
    outs = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        e, s, st = lanes[i % nstr]
        with torch.cuda.stream(st):
            outs.append(one(e, s))
    torch.cuda.synchronize()
    return time.perf_counter() - t0, outs
run(2, 2)
t1, o1 = run(1, 4); t2, o2 = run(2, 4); t1b, _ = run(1, 4); t2b, o2b = run(2, 4)
print(f"fp8 {fp8}: one stream {4 * n / t1:.4f} / {4 * n / t1b:.4f} images/s, two streams {4 * n / t2:.4f} / {4 * n / t2b:.4f}; identical outputs: "
      f"{all(torch.equal(a, b) for a, b in zip(o1, o2))} {all(torch.equal(a, b) for a, b in zip(o2, o2b))}")
