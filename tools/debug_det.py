import sys, torch
sys.path.insert(0, "/root/repo")
from naturaldiffusion_amd.ncsnpp import NCSNppEngine
from naturaldiffusion_amd.synth import synthetic_flat_params
flat = synthetic_flat_params(0)
B = 64
e = [NCSNppEngine(flat, max_batch=B) for _ in range(2)]
xs = [torch.randn(B, 3, 32, 32, device="cuda") for _ in range(2)]; t = torch.rand(B, device="cuda") * 999
ref = [e[i](xs[i], t).clone() for i in range(2)]
torch.cuda.synchronize()
st = [torch.cuda.Stream() for _ in range(2)]
N = 30
outs = [[torch.empty_like(ref[0]) for _ in range(N)] for _ in range(2)]
torch.cuda.synchronize()
for it in range(N):
    for i in range(2):
        with torch.cuda.stream(st[i]): e[i](xs[i], t, out=outs[i][it])
torch.cuda.synchronize()
bad = sum(1 for it in range(N) for i in range(2) if not torch.equal(outs[i][it], ref[i]))
print("mismatching forwards:", bad, "/", 2 * N)
