"""register / spill summary of the kernels whose name contains argv[1] (from the -save-temps ISA listing)"""
import re, subprocess, sys
from pathlib import Path
s = (Path(__file__).resolve().parent.parent / "naturaldiffusion_amd/csrc/build/ncsnpp-hip-amdgcn-amd-amdhsa-gfx950.s").read_text()
md = s[s.index('amdhsa.kernels:'):]
for blk in re.split(r"\n  - \.", md)[1:]:
    get = lambda k: re.search(r"\." + k + r":\s*(\S+)", blk).group(1)
    n = get("name")
    dn = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    if sys.argv[1] in dn:
        print(dn[:70], 'vgpr', get('vgpr_count'), 'spill', get('vgpr_spill_count'), 'sgpr', get('sgpr_count'), 'sspill', get('sgpr_spill_count'), 'scratch', get('private_segment_fixed_size'))
