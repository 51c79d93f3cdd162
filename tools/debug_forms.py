"""on the GPU box: is it the fixture (host dependence) or the kernel?"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
torch.set_num_threads(4)
from oracle import ni_oracle as O
sd3 = np.load(ROOT / "tests/golden/sd3_form.npz")
val = np.load(ROOT / "tests/golden/validate_form.npz")
ts, sg = O.sd3_sigma_schedule(28)
print("schedule equal:", np.array_equal(sg.numpy(), sd3["sigmas"]), np.array_equal(ts.numpy(), sd3["timesteps"]))
noises = torch.from_numpy(sd3["noises"])
vel = O.analytic_velocity_model()
W = O.load_sd3_csv(ROOT / "weights/sd3_step_28_weight.csv")
means = O.sd3_ni(vel, noises, W, sg, ts, return_all=True)
sc = lambda z: (z / 1.5305) + 0.0609
print("oracle-on-box vs golden (plain):", int((sc(means[-1]).numpy() != sd3["final_plain_scaled"]).sum()))
# kernel vs oracle on box, step by step
from naturaldiffusion_amd.sampler import SD3NI
dev = torch.device("cuda:0")
ni = SD3NI(W, sg, noises.numel(), device=dev)
fn = noises.to(dev).reshape(-1)
x = ni.first_input(fn)
seq = []
mean_o = torch.zeros_like(noises)
for k in range(28):
    xo = sg[k] * noises + (1 - sg[k]) * mean_o
    dx = int((x.cpu().view_as(xo) != xo).sum())
    vt = vel(xo, ts[k], True); vn = vel(xo, ts[k], False)
    mean, xn = ni.step(k, x, vt.to(dev).reshape(-1), vn.to(dev).reshape(-1), fn, want_next=k + 1 < 28)
    x0n = xo - sg[k] * vn; x0t = xo - sg[k] * vt
    seq.append(x0n + 7.0 * (x0t - x0n))
    mean_o = O.sd3_weighted_mean(seq, W)
    dm = int((mean.cpu().view_as(mean_o) != mean_o).sum())
    dh = int((ni.hist[k].cpu().view_as(mean_o) != seq[-1]).sum())
    if dx or dm or dh or k in (0, 27):
        print(f"sd3 step {k}: x_in mismatches {dx}, hist {dh}, mean {dm}")
    x = xn if xn is not None else x
# validate
from test_oracle_ni import _fake_dit_eps
z0 = torch.from_numpy(val["rng_z0"]); steps = [torch.from_numpy(a) for a in val["rng_steps"]]
C, B, node = O.load_coeff_npz(ROOT / "results/ddpm/ddpm_sympy_024.npz")
zs = O.validate_ni(_fake_dit_eps(), z0, steps, C, B, node, return_all=True)
print("validate oracle-on-box vs golden:", int(((zs[-1] / 0.18215).numpy() != val["ni_ddpm_sympy"]).sum()))
tb = O.ddim_skip_tables(24)
import hashlib
print("tables sha", hashlib.sha256(tb["xt2x0"].tobytes()).hexdigest()[:10], hashlib.sha256(tb["eps2x0"].tobytes()).hexdigest()[:10])
