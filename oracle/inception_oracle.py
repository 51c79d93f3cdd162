"""TEST INFRASTRUCTURE -- CPU restatement of the FID Inception-V3 pool3 forward.  **PARITY UNPINNED.**

What it restates: ``pytorch_fid.inception.InceptionV3([BLOCK_INDEX_BY_DIM[2048]])`` as the reference calls it
(/root/reference/src/CIFAR10NaturalInference.py:44-86 ``get_activation`` / ``calc_fid``: uint8 HWC images / 255 -> NCHW ->
``model(batch)[0]`` -> [n, 2048]).  ``pytorch_fid`` (and torchvision, whose ``inception_v3`` it subclasses) is an un-vendored,
un-pinned pip dependency of the reference (requirements.txt) and is NOT installed in this image; its weights
(pt_inception-2015-12-05-6726825d.pth) are a download.  No reference test or fixture touches this arithmetic, so the
restatement follows the published architecture (torchvision ``Inception3`` module names; the three FID patches of
``pytorch_fid``: ``count_include_pad=False`` average pools in Mixed_5b-5d / 6b-6e / 7b, a MAX pool in Mixed_7c's pool branch,
bilinear 299x299 resize with ``align_corners=False`` and the 2x-1 input scaling) and says so here: parity against it is
"engine == this file", not "engine == pytorch_fid".  tools/capture_inception.py is the probe that would pin it on a machine
that has the package and the weights.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-3

# (name, cin, cout, (kh, kw), stride, (ph, pw)) of every BasicConv2d on the pool3 path, in torchvision registration order
def conv_specs() -> List[Tuple[str, int, int, Tuple[int, int], int, Tuple[int, int]]]:
    s: List = [("Conv2d_1a_3x3", 3, 32, (3, 3), 2, (0, 0)), ("Conv2d_2a_3x3", 32, 32, (3, 3), 1, (0, 0)), ("Conv2d_2b_3x3", 32, 64, (3, 3), 1, (1, 1)),
               ("Conv2d_3b_1x1", 64, 80, (1, 1), 1, (0, 0)), ("Conv2d_4a_3x3", 80, 192, (3, 3), 1, (0, 0))]

    def incA(n, cin, pf):
        s.extend([(f"{n}.branch1x1", cin, 64, (1, 1), 1, (0, 0)), (f"{n}.branch5x5_1", cin, 48, (1, 1), 1, (0, 0)), (f"{n}.branch5x5_2", 48, 64, (5, 5), 1, (2, 2)),
                  (f"{n}.branch3x3dbl_1", cin, 64, (1, 1), 1, (0, 0)), (f"{n}.branch3x3dbl_2", 64, 96, (3, 3), 1, (1, 1)), (f"{n}.branch3x3dbl_3", 96, 96, (3, 3), 1, (1, 1)),
                  (f"{n}.branch_pool", cin, pf, (1, 1), 1, (0, 0))])

    def incB(n, cin):
        s.extend([(f"{n}.branch3x3", cin, 384, (3, 3), 2, (0, 0)), (f"{n}.branch3x3dbl_1", cin, 64, (1, 1), 1, (0, 0)),
                  (f"{n}.branch3x3dbl_2", 64, 96, (3, 3), 1, (1, 1)), (f"{n}.branch3x3dbl_3", 96, 96, (3, 3), 2, (0, 0))])

    def incC(n, cin, c7):
        s.extend([(f"{n}.branch1x1", cin, 192, (1, 1), 1, (0, 0)),
                  (f"{n}.branch7x7_1", cin, c7, (1, 1), 1, (0, 0)), (f"{n}.branch7x7_2", c7, c7, (1, 7), 1, (0, 3)), (f"{n}.branch7x7_3", c7, 192, (7, 1), 1, (3, 0)),
                  (f"{n}.branch7x7dbl_1", cin, c7, (1, 1), 1, (0, 0)), (f"{n}.branch7x7dbl_2", c7, c7, (7, 1), 1, (3, 0)), (f"{n}.branch7x7dbl_3", c7, c7, (1, 7), 1, (0, 3)),
                  (f"{n}.branch7x7dbl_4", c7, c7, (7, 1), 1, (3, 0)), (f"{n}.branch7x7dbl_5", c7, 192, (1, 7), 1, (0, 3)),
                  (f"{n}.branch_pool", cin, 192, (1, 1), 1, (0, 0))])

    def incD(n, cin):
        s.extend([(f"{n}.branch3x3_1", cin, 192, (1, 1), 1, (0, 0)), (f"{n}.branch3x3_2", 192, 320, (3, 3), 2, (0, 0)),
                  (f"{n}.branch7x7x3_1", cin, 192, (1, 1), 1, (0, 0)), (f"{n}.branch7x7x3_2", 192, 192, (1, 7), 1, (0, 3)),
                  (f"{n}.branch7x7x3_3", 192, 192, (7, 1), 1, (3, 0)), (f"{n}.branch7x7x3_4", 192, 192, (3, 3), 2, (0, 0))])

    def incE(n, cin):
        s.extend([(f"{n}.branch1x1", cin, 320, (1, 1), 1, (0, 0)),
                  (f"{n}.branch3x3_1", cin, 384, (1, 1), 1, (0, 0)), (f"{n}.branch3x3_2a", 384, 384, (1, 3), 1, (0, 1)), (f"{n}.branch3x3_2b", 384, 384, (3, 1), 1, (1, 0)),
                  (f"{n}.branch3x3dbl_1", cin, 448, (1, 1), 1, (0, 0)), (f"{n}.branch3x3dbl_2", 448, 384, (3, 3), 1, (1, 1)),
                  (f"{n}.branch3x3dbl_3a", 384, 384, (1, 3), 1, (0, 1)), (f"{n}.branch3x3dbl_3b", 384, 384, (3, 1), 1, (1, 0)),
                  (f"{n}.branch_pool", cin, 192, (1, 1), 1, (0, 0))])
    incA("Mixed_5b", 192, 32); incA("Mixed_5c", 256, 64); incA("Mixed_5d", 288, 64)
    incB("Mixed_6a", 288)
    incC("Mixed_6b", 768, 128); incC("Mixed_6c", 768, 160); incC("Mixed_6d", 768, 160); incC("Mixed_6e", 768, 192)
    incD("Mixed_7a", 768)
    incE("Mixed_7b", 1280); incE("Mixed_7c", 2048)
    return s


def param_layout() -> List[Tuple[str, Tuple[int, ...]]]:
    """state-dict names (torchvision ``Inception3``; the pt_inception checkpoint uses the same) and shapes, pool3 path only."""
    out = []
    for n, cin, cout, (kh, kw), _, _ in conv_specs():
        out.extend([(f"{n}.conv.weight", (cout, cin, kh, kw)), (f"{n}.bn.weight", (cout,)), (f"{n}.bn.bias", (cout,)),
                    (f"{n}.bn.running_mean", (cout,)), (f"{n}.bn.running_var", (cout,))])
    return out


def make_params(seed: int = 0) -> Dict[str, torch.Tensor]:
    """synthetic weights (the real ones are a download): He-uniform filters, BatchNorm statistics that are not the identity."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for name, shp in param_layout():
        if name.endswith("conv.weight"):
            fan_in = shp[1] * shp[2] * shp[3]
            P[name] = (torch.rand(shp, generator=g) * 2 - 1) * math.sqrt(6.0 / fan_in)
        elif name.endswith("bn.weight"):
            P[name] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif name.endswith("running_var"):
            P[name] = 0.5 + torch.rand(shp, generator=g)
        else:
            P[name] = 0.1 * torch.randn(shp, generator=g)
    return P


def n_params() -> int:
    return sum(int(torch.tensor(s).prod()) for _, s in param_layout())


def flops_per_image() -> float:
    """2 * MAC of the convolutions at 299 x 299."""
    sizes = {}
    x = torch.zeros(1, 3, 299, 299)
    tot = [0.0]

    def rec(name, y, spec):
        _, cin, cout, (kh, kw), _, _ = spec
        tot[0] += 2.0 * y.shape[2] * y.shape[3] * cout * cin * kh * kw
    forward(None, x, _shape_only=rec, resize=False)
    return tot[0]


def forward(P, x: torch.Tensor, taps: dict | None = None, _shape_only=None, resize: bool = True) -> torch.Tensor:
    """x: float [B, 3, H, W] in [0, 1] (what the reference passes: uint8 / 255, NCHW).  Returns pool3 features [B, 2048]."""
    specs = {s[0]: s for s in conv_specs()}

    def conv(name, h):
        s = specs[name]
        _, cin, cout, (kh, kw), stride, (ph, pw) = s
        if _shape_only is not None:
            ho = (h.shape[2] + 2 * ph - kh) // stride + 1
            wo = (h.shape[3] + 2 * pw - kw) // stride + 1
            y = torch.zeros(h.shape[0], cout, ho, wo)
            _shape_only(name, y, s)
            return y
        y = F.conv2d(h, P[f"{name}.conv.weight"], None, stride=stride, padding=(ph, pw))
        y = F.batch_norm(y, P[f"{name}.bn.running_mean"], P[f"{name}.bn.running_var"], P[f"{name}.bn.weight"], P[f"{name}.bn.bias"], False, 0.0, BN_EPS)
        y = F.relu(y)
        if taps is not None:
            taps[name] = y
        return y
    avg = lambda h: F.avg_pool2d(h, 3, 1, 1, count_include_pad=False)

    if resize:
        x = F.interpolate(x, size=(299, 299), mode="bilinear", align_corners=False)
    x = 2 * x - 1
    h = conv("Conv2d_1a_3x3", x); h = conv("Conv2d_2a_3x3", h); h = conv("Conv2d_2b_3x3", h)
    h = F.max_pool2d(h, 3, 2)
    h = conv("Conv2d_3b_1x1", h); h = conv("Conv2d_4a_3x3", h)
    h = F.max_pool2d(h, 3, 2)
    for n in ("Mixed_5b", "Mixed_5c", "Mixed_5d"):
        b1 = conv(f"{n}.branch1x1", h)
        b5 = conv(f"{n}.branch5x5_2", conv(f"{n}.branch5x5_1", h))
        b3 = conv(f"{n}.branch3x3dbl_3", conv(f"{n}.branch3x3dbl_2", conv(f"{n}.branch3x3dbl_1", h)))
        bp = conv(f"{n}.branch_pool", avg(h))
        h = torch.cat([b1, b5, b3, bp], 1)
        if taps is not None:
            taps[n] = h
    n = "Mixed_6a"
    b3 = conv(f"{n}.branch3x3", h)
    bd = conv(f"{n}.branch3x3dbl_3", conv(f"{n}.branch3x3dbl_2", conv(f"{n}.branch3x3dbl_1", h)))
    h = torch.cat([b3, bd, F.max_pool2d(h, 3, 2)], 1)
    if taps is not None:
        taps[n] = h
    for n in ("Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e"):
        b1 = conv(f"{n}.branch1x1", h)
        b7 = conv(f"{n}.branch7x7_3", conv(f"{n}.branch7x7_2", conv(f"{n}.branch7x7_1", h)))
        bd = h
        for i in range(1, 6):
            bd = conv(f"{n}.branch7x7dbl_{i}", bd)
        bp = conv(f"{n}.branch_pool", avg(h))
        h = torch.cat([b1, b7, bd, bp], 1)
        if taps is not None:
            taps[n] = h
    n = "Mixed_7a"
    b3 = conv(f"{n}.branch3x3_2", conv(f"{n}.branch3x3_1", h))
    b7 = h
    for i in range(1, 5):
        b7 = conv(f"{n}.branch7x7x3_{i}", b7)
    h = torch.cat([b3, b7, F.max_pool2d(h, 3, 2)], 1)
    if taps is not None:
        taps[n] = h
    for n in ("Mixed_7b", "Mixed_7c"):
        b1 = conv(f"{n}.branch1x1", h)
        t = conv(f"{n}.branch3x3_1", h)
        b3 = torch.cat([conv(f"{n}.branch3x3_2a", t), conv(f"{n}.branch3x3_2b", t)], 1)
        t = conv(f"{n}.branch3x3dbl_2", conv(f"{n}.branch3x3dbl_1", h))
        bd = torch.cat([conv(f"{n}.branch3x3dbl_3a", t), conv(f"{n}.branch3x3dbl_3b", t)], 1)
        pooled = avg(h) if n == "Mixed_7b" else F.max_pool2d(h, 3, 1, 1)          # pytorch_fid FIDInceptionE_1 / FIDInceptionE_2
        bp = conv(f"{n}.branch_pool", pooled)
        h = torch.cat([b1, b3, bd, bp], 1)
        if taps is not None:
            taps[n] = h
    return F.adaptive_avg_pool2d(h, (1, 1)).flatten(1)
