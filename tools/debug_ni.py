"""GPU-side diagnosis of ni_step vs oracle (prints where the first bit difference appears)."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import ni_oracle as O
from naturaldiffusion_amd.sampler import CifarNI, SD3NI, ValidateNI
from naturaldiffusion_amd._lib import lib

dev = torch.device("cuda:0")
fx = np.load(ROOT / "tests/golden/cifar_form.npz")
NAME = sys.argv[1] if len(sys.argv) > 1 else "step_5_weight_00"
C, B, node = O.load_coeff_npz(ROOT / f"weights/{NAME}.npz")
ref = fx[f"k4_{NAME}_xs"]
STD = fx[f"k4_{NAME}_stds"]
noise = torch.from_numpy(ref[0])
model = O.analytic_vp_model()
ni = CifarNI(C, B, node, noise.numel(), device=dev, stds=STD)
x = noise.to(dev).reshape(-1)
nz = noise.to(dev).reshape(-1)
hs = []
xc = noise
for k in range(node.shape[0] - 1):
    labels = torch.full((2,), ni.labels[k])
    out = model(xc, labels)
    out_ref_labels = torch.ones(2) * node[k, 0] * 999
    print(k, "labels equal:", torch.equal(labels, out_ref_labels), "std host", ni.std[k], float(O.vp_std_f32(node[k, 0])))
    xg = ni.step(k, x, out.to(dev).reshape(-1), nz)
    torch.cuda.synchronize()
    # oracle for the same inputs
    std = torch.tensor(float(STD[k]))
    x0 = O.x0_from_score(xc, O.score_from_model_out(out, std), node[k, 1], node[k, 2])
    hs.append(x0)
    xo = O.cifar_weighted_sum(C[k], hs) + noise * float(np.float32(B[k, 0]))
    hg = ni.hist[k].cpu().view_as(x0)
    d_h = (hg != x0).sum().item()
    d_x = (xg.cpu().view_as(xo) != xo).sum().item()
    i0 = rows_n = ni.rows.rows[k].n
    print(f"step {k}: n_terms {rows_n} hist mismatches {d_h} (max rel {((hg-x0).abs()/x0.abs().clamp_min(1e-300)).max().item():.3e}); x_next mismatches {d_x} (max abs {(xg.cpu().view_as(xo)-xo).abs().max().item():.3e}); vs golden {(xo.numpy()!=ref[k+1]).sum()}")
    if d_h:
        # localise: score
        s_o = O.score_from_model_out(out, std)
        i = (hg != x0).flatten().nonzero()[0].item()
        print("   first bad idx", i, "out", out.flatten()[i].item(), "x", xc.flatten()[i].item(), "score", s_o.flatten()[i].item(), "x0 ref", x0.flatten()[i].item(), "gpu", hg.flatten()[i].item())
    x = xg
    xc = xo
