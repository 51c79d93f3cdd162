"""Attribute rocprofv3 kernel-trace rows to GEMM layer shapes.
usage: analyze_trace.py <kernel_trace.csv> [B]"""
import csv, sys, ctypes as C
from collections import defaultdict
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd._lib import lib
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
h = C.c_void_p(); lib.natinf_ncsnpp_create(C.byref(h), 0)
buf = C.create_string_buffer(1 << 16)
lib.natinf_ncsnpp_describe_gemms(h, B, buf, len(buf))
shapes = []
for l in buf.value.decode().strip().split("\n"):
    M, N, K0, K1, taps, batch, k = l.split()       # k = variant/e<epilogue code, 0 = fp32 slab>
    shapes.append((int(M), int(N), int(K0) + int(K1), int(batch), k))
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_gemm" in r["Kernel_Name"] or "k_conv_gn" in r["Kernel_Name"] or "k_head_conv" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
assert len(rows) % len(shapes) == 0, (len(rows), len(shapes))
agg = defaultdict(lambda: [0, 0.0, 0.0])
for i, r in enumerate(rows):
    s = shapes[i % len(shapes)]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    a = agg[s]; a[0] += 1; a[1] += d; a[2] += 2.0 * s[0] * s[1] * s[2] * s[3]
tot_t = sum(a[1] for a in agg.values()); tot_f = sum(a[2] for a in agg.values())
print(f"{'M':>7} {'N':>5} {'K':>5} {'batch':>5} {'kernel':>16} {'calls':>6} {'ms/call':>8} {'TF/s':>7} {'%time':>6}")
for s, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{s[0]:7d} {s[1]:5d} {s[2]:5d} {s[3]:5d} {s[4]:>16} {a[0]:6d} {a[1]/a[0]*1e3:8.3f} {a[2]/a[1]/1e12:7.1f} {100*a[1]/tot_t:6.1f}")
print(f"total GEMM {tot_t*1e3:.1f} ms, {tot_f/tot_t/1e12:.1f} TF/s issued")
