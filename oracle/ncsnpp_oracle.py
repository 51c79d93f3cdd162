"""Oracle: NCSN++ / DDPM++ (CIFAR10, VP continuous) forward, torch CPU fp32.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Functional restatement of
``deps/score_sde_pytorch/models/ncsnpp.py:232-381`` under the configuration
``configs/vp/cifar10_ddpmpp_continuous.py:41-64`` (nf 128, ch_mult (1,2,2,2), 4
BigGAN res-blocks per level, attention at 16 px, positional embedding, fir=False,
skip_rescale=True, progressive none, centered data, scale_by_sigma False):

* res-block  -- ``models/layerspp.py:242-274``
* attention  -- ``models/layerspp.py:75-91`` with NIN ``models/layers.py:546-555``
* embedding  -- ``models/layers.py:515-530``
* up / down  -- ``models/up_or_down_sampling.py:59-69``

Parameters are a flat ``{name: tensor}`` dict carrying the reference module's own
state-dict keys (``all_modules.<i>.<leaf>``), so the very same dict loads into the
reference ``NCSNpp`` (``tests/golden/make_golden.py`` does exactly that to pin
this file).  ``plan()`` is the one description of the network's topology; the
GPU engine's host side builds its own plan and a CPU test checks the two agree.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

NF = 128
CH_MULT = (1, 2, 2, 2)
NUM_RES = 4
ATTN_RES = (16,)
IMG = 32
TEMB = NF * 4
GN_EPS = 1e-6


@dataclass
class Mod:
    idx: int                # index into all_modules
    kind: str               # 'lin' | 'conv' | 'res' | 'attn' | 'gn'
    cin: int = 0
    cout: int = 0
    up: bool = False
    down: bool = False
    res: int = 0            # input resolution


def plan() -> List[Mod]:
    """Module list in ``all_modules`` order (ncsnpp.py:66-230)."""
    mods: List[Mod] = []
    add = lambda **kw: mods.append(Mod(idx=len(mods), **kw))
    add(kind="lin", cin=NF, cout=TEMB)
    add(kind="lin", cin=TEMB, cout=TEMB)
    add(kind="conv", cin=3, cout=NF, res=IMG)
    skip_ch = [NF]
    ch, res = NF, IMG
    for lvl, mult in enumerate(CH_MULT):
        for _ in range(NUM_RES):
            add(kind="res", cin=ch, cout=NF * mult, res=res)
            ch = NF * mult
            if res in ATTN_RES:
                add(kind="attn", cin=ch, cout=ch, res=res)
            skip_ch.append(ch)
        if lvl != len(CH_MULT) - 1:
            add(kind="res", cin=ch, cout=ch, down=True, res=res)
            res //= 2
            skip_ch.append(ch)
    add(kind="res", cin=ch, cout=ch, res=res)
    add(kind="attn", cin=ch, cout=ch, res=res)
    add(kind="res", cin=ch, cout=ch, res=res)
    for lvl in reversed(range(len(CH_MULT))):
        for _ in range(NUM_RES + 1):
            add(kind="res", cin=ch + skip_ch.pop(), cout=NF * CH_MULT[lvl], res=res)
            ch = NF * CH_MULT[lvl]
        if res in ATTN_RES:
            add(kind="attn", cin=ch, cout=ch, res=res)
        if lvl != 0:
            add(kind="res", cin=ch, cout=ch, up=True, res=res)
            res *= 2
    assert not skip_ch
    add(kind="gn", cin=ch, cout=ch, res=res)
    add(kind="conv", cin=ch, cout=3, res=res)
    return mods


def param_shapes() -> Dict[str, tuple]:
    """name -> shape, in the reference's ``state_dict()`` order for parameters."""
    out: Dict[str, tuple] = {}
    for m in plan():
        p = f"all_modules.{m.idx}."
        if m.kind == "lin":
            out[p + "weight"] = (m.cout, m.cin); out[p + "bias"] = (m.cout,)
        elif m.kind == "conv":
            out[p + "weight"] = (m.cout, m.cin, 3, 3); out[p + "bias"] = (m.cout,)
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
            if m.cin != m.cout or m.up or m.down:
                out[p + "Conv_2.weight"] = (m.cout, m.cin, 1, 1); out[p + "Conv_2.bias"] = (m.cout,)
    return out


def make_params(seed: int = 0, perturb: float = 0.01, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic weights (no checkpoint is shipped):
    fan-avg uniform for every matrix / filter, N(1, .) / N(0, .) affine terms, all
    drawn from one CPU generator in ``param_shapes()`` order, then ``+ perturb*randn``
    so the reference's zero-initialised (``init_scale=0``) layers contribute."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shp in param_shapes().items():
        if len(shp) >= 2:
            recf = int(np.prod(shp[2:])) if len(shp) > 2 else 1
            if ".NIN_" in name:
                fan_in, fan_out = shp[0], shp[1]
            else:
                fan_in, fan_out = shp[1] * recf, shp[0] * recf
            lim = math.sqrt(3.0 * gain / ((fan_in + fan_out) / 2))
            w = (torch.rand(shp, generator=g) * 2 - 1) * lim
        elif name.endswith("GroupNorm_0.weight") or name.endswith("GroupNorm_1.weight") or \
                (name.endswith(".weight") and len(shp) == 1):
            w = torch.ones(shp)
        else:
            w = torch.zeros(shp)
        out[name] = (w + perturb * torch.randn(shp, generator=g)).contiguous()
    return out


# --------------------------------------------------------------------------- #
def timestep_embedding(labels: torch.Tensor, dim: int = NF, max_pos: int = 10000) -> torch.Tensor:
    """layers.py:515-530."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(max_pos) / (half - 1)))
    arg = labels.float()[:, None] * freq[None, :]
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)


def _gn(x, P, pre):
    return F.group_norm(x, min(x.shape[1] // 4, 32), P[pre + ".weight"], P[pre + ".bias"], eps=GN_EPS)


def _up(x):
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def _down(x):
    n, c, h, w = x.shape
    return x.reshape(n, c, h // 2, 2, w // 2, 2).mean(dim=(3, 5))


def _same(t):
    return t


def bf16_round(t):
    """operand-rounding model of the gfx950 engine: a value as it sits in a bf16 tensor / MFMA operand"""
    return t.to(torch.bfloat16).to(torch.float32)


def res_block(x, temb, P, pre, m: Mod, rnd=_same):
    """layerspp.py:242-274.  ``rnd`` (default: identity) is applied to every matmul operand and stored activation: with
    ``bf16_round`` the function models WHERE the engine rounds to bf16 (fp32 accumulation and statistics everywhere), which
    separates operand rounding from every other difference between the engine and the fp32 reference."""
    h = rnd(F.silu(_gn(x, P, pre + "GroupNorm_0")))
    if m.up:
        h, x = _up(h), _up(x)
    elif m.down:
        h, x = rnd(_down(h)), rnd(_down(x))
    h = F.conv2d(h, rnd(P[pre + "Conv_0.weight"]), P[pre + "Conv_0.bias"], padding=1)
    h = rnd(h + F.linear(rnd(F.silu(temb)), rnd(P[pre + "Dense_0.weight"]), P[pre + "Dense_0.bias"])[:, :, None, None])
    h = rnd(F.silu(_gn(h, P, pre + "GroupNorm_1")))
    h = F.conv2d(h, rnd(P[pre + "Conv_1.weight"]), P[pre + "Conv_1.bias"], padding=1)
    if m.cin != m.cout or m.up or m.down:
        x = F.conv2d(x, rnd(P[pre + "Conv_2.weight"]), P[pre + "Conv_2.bias"])
    return rnd((x + h) / np.sqrt(2.0))


def attn_block(x, P, pre, rnd=_same):
    """layerspp.py:75-91."""
    n, c, hh, ww = x.shape
    h = rnd(_gn(x, P, pre + "GroupNorm_0"))
    tok = h.permute(0, 2, 3, 1).reshape(n, hh * ww, c)
    nin = lambda t, i: t @ rnd(P[pre + f"NIN_{i}.W"]) + P[pre + f"NIN_{i}.b"]
    q, k, v = rnd(nin(tok, 0)), rnd(nin(tok, 1)), rnd(nin(tok, 2))
    w = rnd(torch.softmax(torch.einsum("bqc,bkc->bqk", q, k) * (int(c) ** (-0.5)), dim=-1))
    o = nin(rnd(torch.einsum("bqk,bkc->bqc", w, v)), 3)
    o = o.reshape(n, hh, ww, c).permute(0, 3, 1, 2)
    return rnd((x + o) / np.sqrt(2.0))


@torch.no_grad()
def forward(P: Dict[str, torch.Tensor], x: torch.Tensor, labels: torch.Tensor,
            taps: Optional[Dict[int, torch.Tensor]] = None, rnd=_same) -> torch.Tensor:
    """ncsnpp.py:232-381 under the fixed configuration; ``taps`` (optional) receives the
    output of every ``all_modules`` entry keyed by its index; ``rnd``: see ``res_block`` (identity = the reference)."""
    mods = plan()
    rec = (lambda i, t: taps.__setitem__(i, t)) if taps is not None else (lambda i, t: None)
    it = iter(mods)
    m = next(it); temb = F.linear(rnd(timestep_embedding(labels)), rnd(P[f"all_modules.{m.idx}.weight"]), P[f"all_modules.{m.idx}.bias"]); rec(m.idx, temb)
    m = next(it); temb = F.linear(rnd(F.silu(temb)), rnd(P[f"all_modules.{m.idx}.weight"]), P[f"all_modules.{m.idx}.bias"]); rec(m.idx, temb)
    m = next(it); h = rnd(F.conv2d(rnd(x), rnd(P[f"all_modules.{m.idx}.weight"]), P[f"all_modules.{m.idx}.bias"], padding=1)); rec(m.idx, h)
    hs = [h]
    res = IMG
    for lvl in range(len(CH_MULT)):
        for _ in range(NUM_RES):
            m = next(it); h = res_block(hs[-1], temb, P, f"all_modules.{m.idx}.", m, rnd); rec(m.idx, h)
            if res in ATTN_RES:
                m = next(it); h = attn_block(h, P, f"all_modules.{m.idx}.", rnd); rec(m.idx, h)
            hs.append(h)
        if lvl != len(CH_MULT) - 1:
            m = next(it); h = res_block(hs[-1], temb, P, f"all_modules.{m.idx}.", m, rnd); rec(m.idx, h)
            res //= 2
            hs.append(h)
    h = hs[-1]
    m = next(it); h = res_block(h, temb, P, f"all_modules.{m.idx}.", m, rnd); rec(m.idx, h)
    m = next(it); h = attn_block(h, P, f"all_modules.{m.idx}.", rnd); rec(m.idx, h)
    m = next(it); h = res_block(h, temb, P, f"all_modules.{m.idx}.", m, rnd); rec(m.idx, h)
    for lvl in reversed(range(len(CH_MULT))):
        for _ in range(NUM_RES + 1):
            m = next(it); h = res_block(torch.cat([h, hs.pop()], dim=1), temb, P, f"all_modules.{m.idx}.", m, rnd); rec(m.idx, h)
        if res in ATTN_RES:
            m = next(it); h = attn_block(h, P, f"all_modules.{m.idx}.", rnd); rec(m.idx, h)
        if lvl != 0:
            m = next(it); h = res_block(h, temb, P, f"all_modules.{m.idx}.", m, rnd); rec(m.idx, h)
            res *= 2
    assert not hs
    m = next(it); h = F.group_norm(h, 32, P[f"all_modules.{m.idx}.weight"], P[f"all_modules.{m.idx}.bias"], eps=GN_EPS); rec(m.idx, h)
    h = rnd(F.silu(h))
    m = next(it); h = F.conv2d(h, rnd(P[f"all_modules.{m.idx}.weight"]), P[f"all_modules.{m.idx}.bias"], padding=1); rec(m.idx, h)
    assert next(it, None) is None
    return h


def model_fn_from_params(P: Dict[str, torch.Tensor], rnd=_same):
    """``model_fn(x, labels)`` closure for the NI oracle (CPU fp32)."""
    def fn(x, labels):
        return forward(P, x.detach().to("cpu", torch.float32), labels.detach().to("cpu", torch.float32), rnd=rnd).to(x.device)
    return fn


def flops_per_image() -> float:
    """2*MAC count of conv / linear / attention matmuls (21.69 GFLOP)."""
    tot = 0.0
    for m in plan():
        hw = m.res * m.res
        if m.kind == "conv":
            tot += 2.0 * hw * 9 * m.cin * m.cout
        elif m.kind == "lin":
            tot += 2.0 * m.cin * m.cout
        elif m.kind == "res":
            ohw = hw * 4 if m.up else (hw // 4 if m.down else hw)
            tot += 2.0 * ohw * 9 * m.cin * m.cout + 2.0 * ohw * 9 * m.cout * m.cout + 2.0 * TEMB * m.cout
            if m.cin != m.cout or m.up or m.down:
                tot += 2.0 * ohw * m.cin * m.cout
        elif m.kind == "attn":
            tot += 4 * 2.0 * hw * m.cin * m.cin + 2 * 2.0 * hw * hw * m.cin
    return tot
