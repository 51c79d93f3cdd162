"""Per-launch table of ONE Inception forward from a rocprofv3 kernel trace (tools/bench_inception.py under --kernel-trace): argv[1] = *_kernel_trace.csv.
Launches are matched to the plan's convolutions in order (conv_specs of naturaldiffusion_amd/inception.py), so every k_conv_ring row carries its shape and rate."""
import csv, re, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
rows = list(csv.DictReader(open(sys.argv[1])))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 500
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_inc_stem' in r['Kernel_Name']]
seq = [r for r in rows[idx[-1]:] if 'ncsn' in r['Kernel_Name']]
# engine op order = build order of inception_engine.inc (not torchvision registration order): replay it
convs = []
def conv(H, cin, cout, kh, kw, stride, ph, pw):
    Ho = (H + 2 * ph - kh) // stride + 1 if kh > 1 or stride > 1 else H
    Wo = (H + 2 * pw - kw) // stride + 1 if kw > 1 or stride > 1 else H
    convs.append((H, cin, cout, kh, kw, stride, Ho, Wo)); return Ho
convs.append((299, 3, 32, 3, 3, 2, 149, 149))
conv(149, 32, 32, 3, 3, 1, 0, 0); conv(147, 32, 64, 3, 3, 1, 1, 1); conv(73, 64, 80, 1, 1, 1, 0, 0); conv(73, 80, 192, 3, 3, 1, 0, 0)
def A(cin, pf):
    for a in ((cin, 64, 1, 1, 0, 0), (cin, 48, 1, 1, 0, 0), (48, 64, 5, 5, 2, 2), (cin, 64, 1, 1, 0, 0), (64, 96, 3, 3, 1, 1), (96, 96, 3, 3, 1, 1), (cin, pf, 1, 1, 0, 0)):
        conv(35, a[0], a[1], a[2], a[3], 1, a[4], a[5])
A(192, 32); A(256, 64); A(288, 64)
conv(35, 288, 384, 3, 3, 2, 0, 0); conv(35, 288, 64, 1, 1, 1, 0, 0); conv(35, 64, 96, 3, 3, 1, 1, 1); conv(35, 96, 96, 3, 3, 2, 0, 0)
def C(c7):
    for a in ((768, 192, 1, 1, 0, 0), (768, c7, 1, 1, 0, 0), (c7, c7, 1, 7, 0, 3), (c7, 192, 7, 1, 3, 0), (768, c7, 1, 1, 0, 0), (c7, c7, 7, 1, 3, 0), (c7, c7, 1, 7, 0, 3),
              (c7, c7, 7, 1, 3, 0), (c7, 192, 1, 7, 0, 3), (768, 192, 1, 1, 0, 0)):
        conv(17, a[0], a[1], a[2], a[3], 1, a[4], a[5])
C(128); C(160); C(160); C(192)
conv(17, 768, 192, 1, 1, 1, 0, 0); conv(17, 192, 320, 3, 3, 2, 0, 0); conv(17, 768, 192, 1, 1, 1, 0, 0); conv(17, 192, 192, 1, 7, 1, 0, 3); conv(17, 192, 192, 7, 1, 1, 3, 0); conv(17, 192, 192, 3, 3, 2, 0, 0)
def E_(cin):
    for a in ((cin, 320, 1, 1, 0, 0), (cin, 384, 1, 1, 0, 0), (384, 384, 1, 3, 0, 1), (384, 384, 3, 1, 1, 0), (cin, 448, 1, 1, 0, 0), (448, 384, 3, 3, 1, 1), (384, 384, 1, 3, 0, 1),
              (384, 384, 3, 1, 1, 0), (cin, 192, 1, 1, 0, 0)):
        conv(8, a[0], a[1], a[2], a[3], 1, a[4], a[5])
E_(1280); E_(2048)
ci = 0; tot = 0.0; by = {}
for r in seq:
    n = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('ncsn::', '')
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d; by[n] = by.get(n, 0.0) + d
    extra = ''
    if 'k_conv_ring' in n or 'k_gemm' in n:
        if ci < len(convs):
            H, cin, cout, kh, kw, st, Ho, Wo = convs[ci]; ci += 1
            fl = 2.0 * B * Ho * Wo * cout * cin * kh * kw * 2          # two filter terms
            extra = f"{H:4d} {cin:5d}->{cout:4d} {kh}x{kw}/{st}  M {B * Ho * Wo:9d}  {fl / d / 1e6:7.0f} TF/s (both terms)"
    print(f"{n:34s} {d:9.1f} us  {extra}")
print("total", round(tot / 1e3, 2), "ms;", {k: round(v / 1e3, 2) for k, v in sorted(by.items(), key=lambda kv: -kv[1])})
