"""print checksums of the stand-in model's intermediates (to diff between hosts)"""
import sys, hashlib
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import ni_oracle as O
torch.set_num_threads(4)
sha = lambda t: hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()[:12]
fx = np.load(ROOT / "tests/golden/cifar_form.npz")
NAME = "step_10_weight_42"
C, B, node = O.load_coeff_npz(ROOT / f"weights/{NAME}.npz")
xs = fx[f"k4_{NAME}_xs"]; STD = fx[f"k4_{NAME}_stds"]
print(torch.backends.cpu.get_cpu_capability(), torch.__version__, np.__version__)
for k in (2, 3, 4):
    x = torch.from_numpy(xs[k])
    lab = torch.full((2,), float(np.float32(node[k, 0]) * np.float32(999)))
    t = lab / 999
    a = (1.0 / (1.0 + 4.0 * t * t))[:, None, None, None]
    sg = torch.sqrt(1.0 - a * a)
    num = sg * (x - a * 0.25)
    den = a * a * (0.5 * 0.5) + sg * sg
    bump = x / (1.0 + x * x)
    out = num / den + 0.05 * bump
    print(k, "t", sha(t), "a", sha(a), "sg", sha(sg), "num", sha(num), "den", sha(den), "bump", sha(bump), "out", sha(out), "model", sha(O.analytic_vp_model()(x, lab)))
    std = torch.tensor(float(STD[k]))
    sc = (-out) / std
    x0 = O.x0_from_score(x, sc, node[k, 1], node[k, 2])
    print("   score", sha(sc), "x0", sha(x0), "alpha", repr(node[k, 1]), repr(node[k, 2]), "s2", repr(float(torch.tensor(node[k,2], dtype=torch.float64)**2)))
