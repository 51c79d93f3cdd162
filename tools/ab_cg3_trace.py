"""Per-shape A/B of k_conv_gn3 against k_conv_gn2 INSIDE the network: run under rocprofv3 --kernel-trace, 4 forwards at B = 512 with natinf_set_conv_gn_w128(0), then 4 with
(7) and every K on k_conv_gn3 (min_k 0).  `ab_cg3_trace.py run` is the profiled program; `ab_cg3_trace.py table <kernel_trace.csv>` prints ms per launch of both by layer shape."""
import csv, sys, ctypes as C
from collections import defaultdict
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
NF = 4
if sys.argv[1] == "run":
    import torch
    from naturaldiffusion_amd._lib import lib, check
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine
    from naturaldiffusion_amd.synth import synthetic_flat_params
    eng = NCSNppEngine(synthetic_flat_params(0), max_batch=512)
    x = torch.randn(512, 3, 32, 32, device="cuda"); t = torch.rand(512, device="cuda") * 999
    for mask in (0, 7):
        check(lib.natinf_set_conv_gn_w128(mask), "mask")
        for sh in range(3): check(lib.natinf_set_conv_gn_w128_min_k(sh, 0), "min_k")
        for _ in range(NF): eng(x, t)
        torch.cuda.synchronize()
else:
    from naturaldiffusion_amd._lib import lib
    h = C.c_void_p(); lib.natinf_ncsnpp_create(C.byref(h), 0)
    buf = C.create_string_buffer(1 << 16)
    lib.natinf_set_conv_gn_w128(0)
    lib.natinf_ncsnpp_describe_gemms(h, 512, buf, len(buf))
    shapes = []
    for l in buf.value.decode().strip().split("\n"):
        M, N, K0, K1, taps, batch, k = l.split()
        shapes.append((int(M), int(N), int(K0), int(K1), k))
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "k_gemm" in r["Kernel_Name"] or "k_conv_gn" in r["Kernel_Name"] or "k_head_conv" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = len(shapes)
    assert len(rows) == 2 * NF * per, (len(rows), per)
    agg = defaultdict(lambda: [[0, 0.0], [0, 0.0]])
    for i, r in enumerate(rows):
        ph = i // (NF * per); s = shapes[i % per]
        if not s[4].startswith("conv_gn") or s[0] < 131072: continue
        a = agg[s][ph]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    print(f"{'M':>7} {'N':>4} {'K0':>5} {'K1':>4} {'epi':>10} {'gn2 ms':>8} {'gn3 ms':>8} {'gn2/gn3':>8} {'calls/fwd':>9}")
    for s, (a, b) in sorted(agg.items(), key=lambda kv: -kv[1][0][1]):
        print(f"{s[0]:7d} {s[1]:4d} {s[2]:5d} {s[3]:4d} {s[4]:>10} {a[1]/a[0]:8.3f} {b[1]/b[0]:8.3f} {a[1]/a[0]/(b[1]/b[0]):8.3f} {a[0]//NF:9d}")
