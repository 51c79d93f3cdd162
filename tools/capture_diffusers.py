"""Probe-and-capture for the two denoisers the reference takes from un-vendored ``diffusers`` (reference call sites
src/SD3NaturalInference.py:175-176 (pipeline load), :210-213 (``pipe.transformer``), :238-240 (``vae.decode``)).

Run on the GPU box (``gpurun -- python3 tools/capture_diffusers.py``).  If ``diffusers`` imports AND an SD3-medium checkpoint is on
the machine, one ``pipe.transformer`` forward and one ``vae.decode`` are captured on seeded inputs into
``gpurun_out/diffusers_capture.npz`` (inputs, outputs, package version, checkpoint path) -- the fixture that would pin
oracle/mmdit_oracle.py / oracle/vae_oracle.py; copy it to tests/golden/ and the parity tests pick it up.  Otherwise the script prints
ONE line starting with ``blocked:`` naming what is missing and exits 0; DESIGN.md section 2 records the outcome.  Nothing under
naturaldiffusion_amd/ imports this file."""
import glob
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "gpurun_out"


def find_checkpoint():
    roots = [os.environ.get("SD3_MODEL_PATH", ""), os.path.expanduser("~/.cache/huggingface/hub"), "/root/.cache/huggingface/hub",
             os.environ.get("HF_HOME", ""), "/data", "/models", "/mnt", "/opt/models", str(ROOT / "deps")]
    pats = ("**/stable-diffusion-3-medium*/**/model_index.json", "**/models--stabilityai--stable-diffusion-3*/**/model_index.json",
            "**/sd3_medium*.safetensors")
    for r in roots:
        if not r or not os.path.isdir(r):
            continue
        for p in pats:
            hit = glob.glob(os.path.join(r, p), recursive=True)
            if hit:
                return hit[0]
    return None


def main():
    OUT.mkdir(exist_ok=True)
    report = {"diffusers": None, "checkpoint": None}
    try:
        import diffusers
        report["diffusers"] = diffusers.__version__
    except Exception as e:                                  # noqa: BLE001 -- any import failure is "absent"
        report["diffusers_error"] = f"{type(e).__name__}: {e}"
    ck = find_checkpoint()
    report["checkpoint"] = ck
    if report["diffusers"] is None or ck is None:
        missing = [n for n, ok in (("the diffusers package", report["diffusers"] is not None), ("an SD3-medium checkpoint", ck is not None)) if not ok]
        report["status"] = "blocked: " + " and ".join(missing) + " not on this machine (no network to fetch them); MMDiT / AutoencoderKL parity stays unpinned"
        (OUT / "diffusers_capture.json").write_text(json.dumps(report, indent=1))
        print(report["status"])
        return 0
    import numpy as np
    import torch
    from diffusers import StableDiffusion3Pipeline
    src = os.path.dirname(ck) if ck.endswith("model_index.json") else ck
    load = StableDiffusion3Pipeline.from_pretrained if os.path.isdir(src) else StableDiffusion3Pipeline.from_single_file
    pipe = load(src, torch_dtype=torch.float32, text_encoder=None, text_encoder_2=None, text_encoder_3=None, tokenizer=None, tokenizer_2=None,
                tokenizer_3=None)
    g = torch.Generator().manual_seed(1234)
    tr, vae = pipe.transformer.eval(), pipe.vae.eval()
    x = torch.randn(1, 16, 128, 128, generator=g)
    t = torch.tensor([500.0])
    e = torch.randn(1, 333, 4096, generator=g)
    p = torch.randn(1, 2048, generator=g)
    dev = "cuda:0" if torch.cuda.is_available() else "cpu"
    with torch.no_grad():
        y = tr.to(dev)(hidden_states=x.to(dev), timestep=t.to(dev), encoder_hidden_states=e.to(dev), pooled_projections=p.to(dev), return_dict=False)[0].cpu()
        z = torch.randn(1, 16, 32, 32, generator=g)
        img = vae.to(dev).decode(z.to(dev), return_dict=False)[0].cpu()
    np.savez_compressed(OUT / "diffusers_capture.npz", x=x.numpy(), t=t.numpy(), text=e.numpy(), pooled=p.numpy(), mmdit_out=y.numpy(), vae_z=z.numpy(),
                        vae_img=img.numpy(), diffusers_version=np.array(report["diffusers"]), checkpoint=np.array(src))
    report["status"] = f"captured: pipe.transformer {tuple(y.shape)} and vae.decode {tuple(img.shape)} -> gpurun_out/diffusers_capture.npz"
    (OUT / "diffusers_capture.json").write_text(json.dumps(report, indent=1))
    print(report["status"])
    return 0


if __name__ == "__main__":
    sys.exit(main())
