"""Round-4 review, item 1(a): where does the fp8 engine's eight-sequences-vs-one difference at SD3 width come from?

The engine's per-token activation scales do not depend on the batch, so the candidates are the PLAN choices that do: split-K on the
under-filled long-K GEMMs (one sequence only), the round model's tile choice for small-M GEMMs, and the tile a GEMM lands on through
`mt256 * nt * batch >= NUM_CU`.  Each changes the order of the fp32 summation over K, and in the fp8 engine a changed last bit of an fp32
sum can flip an e4m3 rounding downstream.  This script runs tests/test_gpu_mmdit.py's SD3-width case (two blocks, one sequence against the
same sequence eight times) under each knob setting and prints, per setting: max |out8[i] - out1| / max |oracle|, whether the bytes are
equal, and both runs' error against the fp32 oracle.

    python tools/diag_fp8_batch.py            -> gpurun_out/diag_fp8_batch.json
"""
import json, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from naturaldiffusion_amd._lib import lib, check
from naturaldiffusion_amd.mmdit import MMDiTEngine, flatten_state_dict
from test_gpu_mmdit import _sd3_width_case

V_AUTO, V_DMA_128x128_P, V_W128 = 0, 17, 29
cfg, P, (x, t, e, p), ref = _sd3_width_case()
flat = flatten_state_dict(P, 64, **cfg)
rep = lambda v: v.cuda().repeat(8, *([1] * (v.dim() - 1)))
refmax = ref.abs().max().item()


def run(fp8, **knobs):
    check(lib.natinf_set_gemm_splitk(knobs.get("splitk", 1)), "splitk")
    check(lib.natinf_set_gemm_round_model(knobs.get("round_model", 1)), "round_model")
    check(lib.natinf_set_gemm_w128(knobs.get("w128", 1)), "w128")
    check(lib.natinf_set_gemm_variant(knobs.get("variant", V_AUTO)), "variant")
    check(lib.natinf_set_mmdit_text_stream(knobs.get("text_stream", 1)), "text_stream")
    e1 = MMDiTEngine(flat, max_batch=1, grid=64, ctx_tokens=333, fp8=fp8, **cfg)
    o1 = e1.forward(x.cuda(), t.cuda(), e.cuda(), p.cuda()).cpu()
    del e1
    e8 = MMDiTEngine(flat, max_batch=8, grid=64, ctx_tokens=333, fp8=fp8, **cfg)
    o8 = e8.forward(rep(x), rep(t), rep(e), rep(p)).cpu()
    del e8
    d = max(((o8[i] - o1[0]).abs().max() / refmax).item() for i in range(8))
    same8 = all(torch.equal(o8[i], o8[0]) for i in range(8))
    return dict(knobs=knobs, fp8=fp8, eight_vs_one=d, bytes_equal=bool(torch.equal(o8[0], o1[0])), eight_rows_identical=same8,
                one_vs_oracle=((o1 - ref).abs().max() / refmax).item(), eight_vs_oracle=((o8[0:1] - ref).abs().max() / refmax).item())


settings = [dict(),                                                                                  # the shipped plan
            dict(splitk=0),
            dict(round_model=0),
            dict(splitk=0, round_model=0),
            dict(splitk=0, variant=V_DMA_128x128_P),                                                 # every plain GEMM of both batches on one tile
            dict(splitk=0, variant=V_DMA_128x128_P, w128=0),                                         # ... and the fp8 GEMMs on the eight-wave tile
            dict(splitk=1, variant=V_DMA_128x128_P),
            dict(splitk=0, variant=V_W128)]
out = []
for fp8 in (True, False):
    for k in settings:
        r = run(fp8, **k)
        out.append(r)
        print(json.dumps(r), flush=True)
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "diag_fp8_batch.json").write_text(json.dumps(out, indent=1))
