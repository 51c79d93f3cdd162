"""Compare the two files tools/debug_pad.py wrote (default and -DNATINF_ASM_PAD builds): outputs and per-module taps bit for bit."""
import sys, torch
sys.path.insert(0, "/root/repo")
from naturaldiffusion_amd.ncsnpp import module_table
mods = module_table()
a = torch.load("/tmp/pad_default.pt"); b = torch.load("/tmp/pad_pad.pt")
print("outputs bit-identical:", torch.equal(a["out"], b["out"]), " max|d|", float((a["out"] - b["out"]).abs().max()))
for k in sorted(a["taps"]):
    if not torch.equal(a["taps"][k], b["taps"][k]):
        d = (a["taps"][k].float() - b["taps"][k].float()).abs()
        per = d.amax(dim=(1, 2, 3))
        print("first differing module", mods[k][:7], "images differing", int((per > 0).sum()), "of", per.numel(), "max|d|", float(d.max()),
              "elements", int((d > 0).sum()), "of", d.numel())
        break
else:
    print("all taps bit-identical")
