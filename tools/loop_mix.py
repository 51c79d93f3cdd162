"""Static instruction mix inside the loops of a kernel of the built library: loop_mix.py <mangled-name substring> [top N]"""
import re, sys, collections
from pathlib import Path
s = (Path(__file__).resolve().parent.parent / "naturaldiffusion_amd/csrc/build/ncsnpp-hip-amdgcn-amd-amdhsa-gfx950.s").read_text()
code = s[:s.index("amdhsa.kernels:")]
for m0 in re.finditer(r"^(_Z\w+):\s*; @", code, flags=re.M):
    name = m0.group(1)
    if sys.argv[1] not in name: continue
    body = code[m0.end():code.index("s_endpgm", m0.end())]
    inloop = False; cnt = collections.Counter()
    for ln in body.split("\n"):
        m = re.match(r"(\.LBB\d+_\d+):", ln)
        if m: inloop = "in Loop" in ln
        elif "in Loop:" in ln: inloop = True
        t = ln.strip().split(" ")[0] if ln.startswith("\t") else ""
        if inloop and t and not t.startswith((".", ";")): cnt[t] += 1
    nm = sum(v for k, v in cnt.items() if "mfma" in k)
    valu = sum(v for k, v in cnt.items() if k.startswith("v_") and "mfma" not in k)
    print(name[:70], "mfma", nm, "valu", valu, "ratio %.2f" % (valu / max(nm, 1)), "s_nop", cnt["s_nop"], "salu", sum(v for k, v in cnt.items() if k.startswith("s_")))
    for k, v in cnt.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 0): print("   ", v, k)
