"""Oracle: the `ddpm` score network (CIFAR10, VP continuous) forward, torch CPU fp32.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Functional restatement of
``deps/score_sde_pytorch/models/ddpm.py:39-181`` under ``configs/vp/ddpm/cifar10_continuous.py`` (nf 128, ch_mult (1,2,2,2),
TWO res-blocks per level, attention at 16 px, resamp_with_conv, conditional, centered data, scale_by_sigma False) -- the network of
the checkpoint the reference's own docstring names (src/CIFAR10NaturalInference.py:416, vp/cifar10_ddpm_continuous/checkpoint_8.pth):

* res-block   ``ResnetBlockDDPM``  models/layers.py:615-661  (no 1/sqrt(2) rescale; NIN shortcut when the width changes)
* attention   ``AttnBlock``        models/layers.py:558-583  (x + h, no rescale)
* resampling  ``Downsample`` (3x3 conv, stride 2, 'SAME' padding emulated by F.pad (0,1,0,1)) / ``Upsample`` (nearest 2x + 3x3 conv),
              models/layers.py:586-612
* embedding   models/layers.py:515-530

Pinned by tests/golden/ddpm_forward.npz: the reference's ``DDPM`` class run on ``make_params`` (tests/golden/make_golden.py group
``ddpm``).  Parameters carry the reference module's state-dict keys (``all_modules.<i>.<leaf>``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .ncsnpp_oracle import timestep_embedding, _same, bf16_round        # noqa: F401  (shared embedding / rounding model)

NF = 128
CH_MULT = (1, 2, 2, 2)
NUM_RES = 2
ATTN_RES = (16,)
IMG = 32
TEMB = NF * 4
GN_EPS = 1e-6


@dataclass
class Mod:
    idx: int
    kind: str               # 'lin' | 'conv' | 'res' | 'attn' | 'down' | 'up' | 'gn'
    cin: int = 0
    cout: int = 0
    res: int = 0            # input resolution


def plan() -> List[Mod]:
    """``all_modules`` order (ddpm.py:60-110)."""
    mods: List[Mod] = []
    add = lambda **kw: mods.append(Mod(idx=len(mods), **kw))
    add(kind="lin", cin=NF, cout=TEMB)
    add(kind="lin", cin=TEMB, cout=TEMB)
    add(kind="conv", cin=3, cout=NF, res=IMG)
    hs_c = [NF]
    ch, res = NF, IMG
    for lvl, mult in enumerate(CH_MULT):
        for _ in range(NUM_RES):
            add(kind="res", cin=ch, cout=NF * mult, res=res)
            ch = NF * mult
            if res in ATTN_RES:
                add(kind="attn", cin=ch, cout=ch, res=res)
            hs_c.append(ch)
        if lvl != len(CH_MULT) - 1:
            add(kind="down", cin=ch, cout=ch, res=res)
            res //= 2
            hs_c.append(ch)
    add(kind="res", cin=ch, cout=ch, res=res)
    add(kind="attn", cin=ch, cout=ch, res=res)
    add(kind="res", cin=ch, cout=ch, res=res)
    for lvl in reversed(range(len(CH_MULT))):
        for _ in range(NUM_RES + 1):
            add(kind="res", cin=ch + hs_c.pop(), cout=NF * CH_MULT[lvl], res=res)
            ch = NF * CH_MULT[lvl]
        if res in ATTN_RES:
            add(kind="attn", cin=ch, cout=ch, res=res)
        if lvl != 0:
            add(kind="up", cin=ch, cout=ch, res=res)
            res *= 2
    assert not hs_c
    add(kind="gn", cin=ch, cout=ch, res=res)
    add(kind="conv", cin=ch, cout=3, res=res)
    return mods


def param_shapes() -> Dict[str, tuple]:
    out: Dict[str, tuple] = {}
    for m in plan():
        p = f"all_modules.{m.idx}."
        if m.kind == "lin":
            out[p + "weight"] = (m.cout, m.cin); out[p + "bias"] = (m.cout,)
        elif m.kind == "conv":
            out[p + "weight"] = (m.cout, m.cin, 3, 3); out[p + "bias"] = (m.cout,)
        elif m.kind in ("down", "up"):
            out[p + "Conv_0.weight"] = (m.cout, m.cin, 3, 3); out[p + "Conv_0.bias"] = (m.cout,)
        elif m.kind == "gn":
            out[p + "weight"] = (m.cin,); out[p + "bias"] = (m.cin,)
        elif m.kind == "attn":
            out[p + "GroupNorm_0.weight"] = (m.cin,); out[p + "GroupNorm_0.bias"] = (m.cin,)
            for i in range(4):
                out[p + f"NIN_{i}.W"] = (m.cin, m.cin); out[p + f"NIN_{i}.b"] = (m.cin,)
        elif m.kind == "res":
            out[p + "GroupNorm_0.weight"] = (m.cin,); out[p + "GroupNorm_0.bias"] = (m.cin,)
            out[p + "Conv_0.weight"] = (m.cout, m.cin, 3, 3); out[p + "Conv_0.bias"] = (m.cout,)
            out[p + "Dense_0.weight"] = (m.cout, TEMB); out[p + "Dense_0.bias"] = (m.cout,)
            out[p + "GroupNorm_1.weight"] = (m.cout,); out[p + "GroupNorm_1.bias"] = (m.cout,)
            out[p + "Conv_1.weight"] = (m.cout, m.cout, 3, 3); out[p + "Conv_1.bias"] = (m.cout,)
            if m.cin != m.cout:
                out[p + "NIN_0.W"] = (m.cin, m.cout); out[p + "NIN_0.b"] = (m.cout,)
    return out


def make_params(seed: int = 0, perturb: float = 0.01) -> Dict[str, torch.Tensor]:
    """synthetic weights, the recipe of ncsnpp_oracle.make_params: fan-avg uniform matrices / filters, unit norm scales, zero biases,
    every tensor then perturbed by ``perturb * randn`` (the reference zero-initialises Conv_1 / NIN_3 / the last conv)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shp in param_shapes().items():
        if len(shp) >= 2:
            recf = int(np.prod(shp[2:])) if len(shp) > 2 else 1
            fan_in, fan_out = (shp[0], shp[1]) if ".NIN_" in name else (shp[1] * recf, shp[0] * recf)
            w = (torch.rand(shp, generator=g) * 2 - 1) * math.sqrt(3.0 / ((fan_in + fan_out) / 2))
        elif name.endswith(".weight"):
            w = torch.ones(shp)
        else:
            w = torch.zeros(shp)
        out[name] = (w + perturb * torch.randn(shp, generator=g)).contiguous()
    return out


def _gn(x, P, pre):
    return F.group_norm(x, 32, P[pre + ".weight"], P[pre + ".bias"], eps=GN_EPS)


def res_block(x, temb, P, pre, m: Mod, rnd=_same):
    """layers.py:643-661 (dropout is the identity in eval mode)."""
    h = rnd(F.silu(_gn(x, P, pre + "GroupNorm_0")))
    h = F.conv2d(h, rnd(P[pre + "Conv_0.weight"]), P[pre + "Conv_0.bias"], padding=1)
    h = rnd(h + F.linear(rnd(F.silu(temb)), rnd(P[pre + "Dense_0.weight"]), P[pre + "Dense_0.bias"])[:, :, None, None])
    h = rnd(F.silu(_gn(h, P, pre + "GroupNorm_1")))
    h = F.conv2d(h, rnd(P[pre + "Conv_1.weight"]), P[pre + "Conv_1.bias"], padding=1)
    if m.cin != m.cout:
        x = torch.einsum("bchw,co->bohw", rnd(x), rnd(P[pre + "NIN_0.W"])) + P[pre + "NIN_0.b"][None, :, None, None]
    return rnd(x + h)


def attn_block(x, P, pre, rnd=_same):
    """layers.py:569-583."""
    n, c, hh, ww = x.shape
    h = rnd(_gn(x, P, pre + "GroupNorm_0"))
    tok = h.permute(0, 2, 3, 1).reshape(n, hh * ww, c)
    nin = lambda t, i: t @ rnd(P[pre + f"NIN_{i}.W"]) + P[pre + f"NIN_{i}.b"]
    q, k, v = rnd(nin(tok, 0)), rnd(nin(tok, 1)), rnd(nin(tok, 2))
    w = rnd(torch.softmax(torch.einsum("bqc,bkc->bqk", q, k) * (int(c) ** (-0.5)), dim=-1))
    o = nin(rnd(torch.einsum("bqk,bkc->bqc", w, v)), 3)
    return rnd(x + o.reshape(n, hh, ww, c).permute(0, 3, 1, 2))


def downsample(x, P, pre, rnd=_same):
    """layers.py:600-612: 'SAME' padding of a stride-2 3x3 convolution = one zero row / column at the bottom / right."""
    return rnd(F.conv2d(F.pad(rnd(x), (0, 1, 0, 1)), rnd(P[pre + "Conv_0.weight"]), P[pre + "Conv_0.bias"], stride=2))


def upsample(x, P, pre, rnd=_same):
    """layers.py:586-597."""
    h = F.interpolate(rnd(x), scale_factor=2, mode="nearest")
    return rnd(F.conv2d(h, rnd(P[pre + "Conv_0.weight"]), P[pre + "Conv_0.bias"], padding=1))


@torch.no_grad()
def forward(P: Dict[str, torch.Tensor], x: torch.Tensor, labels: torch.Tensor, taps: Optional[Dict[int, torch.Tensor]] = None, rnd=_same) -> torch.Tensor:
    """ddpm.py:112-181 (conditional, centered, scale_by_sigma False)."""
    mods = plan()
    rec = (lambda i, t: taps.__setitem__(i, t)) if taps is not None else (lambda i, t: None)
    it = iter(mods)
    key = lambda m: f"all_modules.{m.idx}."
    m = next(it); temb = F.linear(rnd(timestep_embedding(labels, NF)), rnd(P[key(m) + "weight"]), P[key(m) + "bias"]); rec(m.idx, temb)
    m = next(it); temb = F.linear(rnd(F.silu(temb)), rnd(P[key(m) + "weight"]), P[key(m) + "bias"]); rec(m.idx, temb)
    m = next(it); h = rnd(F.conv2d(rnd(x), rnd(P[key(m) + "weight"]), P[key(m) + "bias"], padding=1)); rec(m.idx, h)
    hs = [h]
    res = IMG
    for lvl in range(len(CH_MULT)):
        for _ in range(NUM_RES):
            m = next(it); h = res_block(hs[-1], temb, P, key(m), m, rnd); rec(m.idx, h)
            if res in ATTN_RES:
                m = next(it); h = attn_block(h, P, key(m), rnd); rec(m.idx, h)
            hs.append(h)
        if lvl != len(CH_MULT) - 1:
            m = next(it); h = downsample(hs[-1], P, key(m), rnd); rec(m.idx, h)
            res //= 2
            hs.append(h)
    h = hs[-1]
    m = next(it); h = res_block(h, temb, P, key(m), m, rnd); rec(m.idx, h)
    m = next(it); h = attn_block(h, P, key(m), rnd); rec(m.idx, h)
    m = next(it); h = res_block(h, temb, P, key(m), m, rnd); rec(m.idx, h)
    for lvl in reversed(range(len(CH_MULT))):
        for _ in range(NUM_RES + 1):
            m = next(it); h = res_block(torch.cat([h, hs.pop()], dim=1), temb, P, key(m), m, rnd); rec(m.idx, h)
        if res in ATTN_RES:
            m = next(it); h = attn_block(h, P, key(m), rnd); rec(m.idx, h)
        if lvl != 0:
            m = next(it); h = upsample(h, P, key(m), rnd); rec(m.idx, h)
            res *= 2
    assert not hs
    m = next(it); h = F.group_norm(h, 32, P[key(m) + "weight"], P[key(m) + "bias"], eps=GN_EPS); rec(m.idx, h)
    h = rnd(F.silu(h))
    m = next(it); h = F.conv2d(h, rnd(P[key(m) + "weight"]), P[key(m) + "bias"], padding=1); rec(m.idx, h)
    assert next(it, None) is None
    return h


def model_fn_from_params(P: Dict[str, torch.Tensor], rnd=_same):
    def fn(x, labels):
        return forward(P, x.detach().to("cpu", torch.float32), labels.detach().to("cpu", torch.float32), rnd=rnd).to(x.device)
    return fn


def flops_per_image() -> float:
    tot = 0.0
    for m in plan():
        hw = m.res * m.res
        if m.kind == "conv":
            tot += 2.0 * hw * 9 * m.cin * m.cout
        elif m.kind == "lin":
            tot += 2.0 * m.cin * m.cout
        elif m.kind == "down":
            tot += 2.0 * (hw // 4) * 9 * m.cin * m.cout
        elif m.kind == "up":
            tot += 2.0 * (hw * 4) * 9 * m.cin * m.cout
        elif m.kind == "res":
            tot += 2.0 * hw * 9 * m.cin * m.cout + 2.0 * hw * 9 * m.cout * m.cout + 2.0 * TEMB * m.cout
            if m.cin != m.cout:
                tot += 2.0 * hw * m.cin * m.cout
        elif m.kind == "attn":
            tot += 4 * 2.0 * hw * m.cin * m.cin + 2 * 2.0 * hw * hw * m.cin
    return tot
