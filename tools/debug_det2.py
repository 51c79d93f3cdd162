"""Two-engine reproducibility counter against an arbitrary checkout (argv[1] = repository root to import from): used to show that the session-start build
had the same non-reproducibility."""
import sys, torch
root = sys.argv[1]
sys.path.insert(0, root)
from naturaldiffusion_amd.ncsnpp import NCSNppEngine
from naturaldiffusion_amd.synth import synthetic_flat_params
import naturaldiffusion_amd._lib as L
print("library:", L.LIB_PATH)
flat = synthetic_flat_params(0)
B = 64
e = [NCSNppEngine(flat, max_batch=B) for _ in range(2)]
xs = [torch.randn(B, 3, 32, 32, device="cuda") for _ in range(2)]; t = torch.rand(B, device="cuda") * 999
ref = [e[i](xs[i], t).clone() for i in range(2)]
torch.cuda.synchronize()
st = [torch.cuda.Stream() for _ in range(2)]
N = 20
outs = [[torch.empty_like(ref[0]) for _ in range(N)] for _ in range(2)]
torch.cuda.synchronize()
for it in range(N):
    for i in range(2):
        with torch.cuda.stream(st[i]): e[i](xs[i], t, out=outs[i][it])
torch.cuda.synchronize()
bad = sum(1 for it in range(N) for i in range(2) if not torch.equal(outs[i][it], ref[i]))
print("mismatching forwards:", bad, "/", 2 * N)
