"""Debug: VAE decode at r=32, B=1 vs B=4, per forced GEMM variant, vs the oracle (GPU box)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import vae_oracle as V
from naturaldiffusion_amd.vae import VAEDecoder, flatten_state_dict
from naturaldiffusion_amd._lib import lib
P = V.make_params(4, seed=1)
dec = VAEDecoder(flatten_state_dict(P, 4), max_batch=4, latent_ch=4, latent_res=32)
z = torch.randn(4, 4, 32, 32, generator=torch.Generator().manual_seed(0))
ref = V.decode(P, z[2:3]); sc = ref.abs().max()
for v in [int(a) for a in sys.argv[1:]] or [0, 17, 8, 18, 16, 4, 10]:
    lib.natinf_set_gemm_variant(v)
    out = dec(z.cuda()).cpu()
    solo = dec(z[2:3].cuda()).cpu()
    print(f"variant {v}: batch-vs-oracle {((out[2:3]-ref).abs().max()/sc).item():.3e}  solo-vs-oracle {((solo-ref).abs().max()/sc).item():.3e}  "
          f"solo-vs-batch {((solo-out[2:3]).abs().max()/sc).item():.3e}", flush=True)
