"""Synthetic NCSN++ weights for benchmarking / smoke runs (the reference ships no checkpoint:
``checkpoint_8.pth`` is a missing large blob).  Recipe: one CPU generator seeded with
``seed``; in engine parameter order, matrices / filters ~ fan-avg uniform, norm scales 1, biases 0, each
then perturbed by ``perturb * randn`` so that the reference's zero-initialised layers are exercised.
Host-side data generation only; returns the flat fp32 vector ``NCSNppEngine`` consumes."""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

from .ncsnpp import param_layout


def synthetic_state_dict(seed: int = 0, perturb: float = 0.01) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shp in param_layout():
        if len(shp) >= 2:
            rf = int(np.prod(shp[2:])) if len(shp) > 2 else 1
            fan_in, fan_out = (shp[0], shp[1]) if ".NIN_" in name else (shp[1] * rf, shp[0] * rf)
            w = (torch.rand(shp, generator=g) * 2 - 1) * math.sqrt(3.0 / ((fan_in + fan_out) / 2))
        elif name.endswith(".weight"):
            w = torch.ones(shp)
        else:
            w = torch.zeros(shp)
        out[name] = (w + perturb * torch.randn(shp, generator=g)).contiguous()
    return out


def synthetic_flat_params(seed: int = 0, perturb: float = 0.01) -> torch.Tensor:
    return torch.cat([t.reshape(-1) for t in synthetic_state_dict(seed, perturb).values()])
