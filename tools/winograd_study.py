"""Winograd F(2x2, 3x3) for the fused GroupNorm + SiLU + 3x3 convolution (round-5 review, item 6): what the 2.25x fewer multiplications would cost in accuracy.
CPU study (torch, float64 reference) of ONE layer shape of the network -- 16x16, cin = 512, N = 256 (K = 4,608; reference deps/score_sde_pytorch/models/layerspp.py:242-274,
Conv_0 of an up-path block) -- with the engine's operand model: the normalised + SiLU'd activation and the weights are rounded to a 16-bit type, products accumulate in fp32.
  direct       : operands rounded to bf16 (what k_conv_gn2 / k_conv_gn3 multiply)
  wino bf16    : input tiles d -> B^T d B and filters g -> G g G^T computed in fp32 from the bf16-rounded operands, THEN rounded to bf16 (the MFMA's operand type), 16 GEMMs, A^T m A in fp32
  wino fp16    : the same with IEEE half as the transformed operands' type (v_mfma_f32_16x16x32_f16 runs at the bf16 rate)
  wino fp16 raw: transformed from UNROUNDED fp32 operands, rounded once to fp16 (the best a fused input transform could do)
Errors are against the float64 convolution of the unrounded operands, relative to the output's max magnitude (the per-module tap metric of tests/test_gpu_ncsnpp.py) and RMS.
usage: python tools/winograd_study.py [seed]   (about a minute on 8 cores)"""
import sys
import numpy as np
import torch
import torch.nn.functional as F

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
g = torch.Generator().manual_seed(seed)
Bn, C, N, R = 4, 512, 256, 16
x = torch.randn(Bn, C, R, R, generator=g, dtype=torch.float64) * 1.3 + 0.2
h = F.silu(x)                                                      # act(GroupNorm(x)) up to the affine map: a unit-scale activation
w = (torch.rand(N, C, 3, 3, generator=g, dtype=torch.float64) * 2 - 1) * np.sqrt(3.0 / (9 * C))      # the engine's synthetic init (synth.py)
ref = F.conv2d(h, w, padding=1)
rnd = {"bf16": lambda t: t.float().bfloat16().double(), "fp16": lambda t: t.float().half().double(), "none": lambda t: t}
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def direct(op):
    return F.conv2d(rnd[op](h).float(), rnd[op](w).float(), padding=1).double()        # fp32 accumulate


def wino(op_in, op_t):
    hp = F.pad(rnd[op_in](h), (1, 1, 1, 1))
    tiles = hp.unfold(2, 4, 2).unfold(3, 4, 2)                                            # [B, C, 8, 8, 4, 4]: 4x4 input tiles, stride 2
    V = rnd[op_t](torch.einsum("ij,bcyxjk,lk->bcyxil", BT, tiles, BT))                    # B^T d B, rounded to the MFMA operand type
    U = rnd[op_t](torch.einsum("ij,ncjk,lk->ncil", G, rnd[op_in](w), G))                  # G g G^T
    Mm = torch.einsum("bcyxil,ncil->bnyxil", V.float(), U.float()).double()               # the sixteen GEMMs over c, fp32 accumulate
    Y = torch.einsum("ij,bnyxjk,lk->bnyxil", AT, Mm, AT)                                  # A^T m A: [B, N, 8, 8, 2, 2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(Bn, N, R, R)


scale = ref.abs().max().item()
rows = [("direct, bf16 operands (shipped)", direct("bf16")), ("direct, fp16 operands", direct("fp16")),
        ("winograd F(2x2,3x3), bf16 transformed operands", wino("bf16", "bf16")), ("winograd, fp16 transformed operands (from bf16 inputs)", wino("bf16", "fp16")),
        ("winograd, fp16 transformed from unrounded fp32", wino("none", "fp16")), ("winograd, exact arithmetic (sanity)", wino("none", "none"))]
print(f"16x16, cin = {C}, N = {N}, B = {Bn}, seed {seed}; output max |y| = {scale:.3f}, rms = {ref.pow(2).mean().sqrt().item():.3f}")
print(f"{'form':58s} {'max err / max|y|':>18s} {'rms err / rms y':>16s}")
for name, y in rows:
    e = (y - ref)
    print(f"{name:58s} {e.abs().max().item() / scale:18.3e} {(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item():16.3e}")
