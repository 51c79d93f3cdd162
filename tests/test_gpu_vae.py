"""HIP AutoencoderKL decoder engine (include/natinf_vae.h) against oracle/vae_oracle.py -- a restatement of the published
decoder architecture; PARITY UNPINNED with respect to the reference's un-vendored ``diffusers`` (see the oracle's header)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 4e-2          # max |engine - oracle| / max |oracle|: bf16 operands through ~40 convolutions vs an fp32 oracle


@pytest.mark.parametrize("latent_ch,r,B", [(4, 8, 2), (4, 16, 3), (16, 16, 1)])
def test_decoder_matches_oracle(latent_ch, r, B):
    from oracle import vae_oracle as V
    from naturaldiffusion_amd.vae import VAEDecoder, flatten_state_dict
    P = V.make_params(latent_ch, seed=3)
    if r == 16:                                          # a whole-AutoencoderKL style dict: post_quant_conv in front of the decoder
        g = torch.Generator().manual_seed(9)
        P["post_quant_conv.weight"] = torch.eye(latent_ch) + 0.2 * torch.randn(latent_ch, latent_ch, generator=g)
        P["post_quant_conv.bias"] = 0.1 * torch.randn(latent_ch, generator=g)
    dec = VAEDecoder(flatten_state_dict(P, latent_ch), max_batch=B, latent_ch=latent_ch, latent_res=r)
    z = torch.randn(B, latent_ch, r, r, generator=torch.Generator().manual_seed(r))
    ref = V.decode(P, z)
    out = dec(z.cuda()).cpu()
    assert out.shape == ref.shape and torch.isfinite(out).all()
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    assert err <= TOL, err


def test_full_size_decode_is_batch_independent_and_close_to_oracle():
    """32x32 latents -> 256x256 images (the ValidateNaturalInference size), one image checked against the oracle."""
    from oracle import vae_oracle as V
    from naturaldiffusion_amd.vae import VAEDecoder, flatten_state_dict
    P = V.make_params(4, seed=1)
    dec = VAEDecoder(flatten_state_dict(P, 4), max_batch=4, latent_ch=4, latent_res=32)
    g = torch.Generator().manual_seed(0)
    z = torch.randn(4, 4, 32, 32, generator=g)
    out = dec(z.cuda()).cpu()
    assert out.shape == (4, 3, 256, 256) and torch.isfinite(out).all()
    ref = V.decode(P, z[2:3])
    assert ((out[2:3] - ref).abs().max() / ref.abs().max()).item() <= TOL
    solo = dec(z[2:3].cuda()).cpu()
    assert ((solo - out[2:3]).abs().max() / ref.abs().max()).item() <= 1e-2


def test_argument_errors():
    from oracle import vae_oracle as V
    from naturaldiffusion_amd.vae import VAEDecoder, flatten_state_dict
    flat = flatten_state_dict(V.make_params(4, seed=0), 4)
    with pytest.raises(ValueError):
        VAEDecoder(flat[:-1], max_batch=1, latent_res=8)
    with pytest.raises(ValueError):
        VAEDecoder(flat, max_batch=1, latent_res=12)
    dec = VAEDecoder(flat, max_batch=1, latent_res=8)
    with pytest.raises(ValueError):
        dec(torch.zeros(2, 4, 8, 8).cuda())


def test_validate_script_decodes_and_writes_the_image_grid(tmp_path, monkeypatch):
    """src/ValidateNaturalInference.py:231-236 end of a sampler: latents / 0.18215 -> vae.decode -> 2x4 image grid on disk,
    with the decoder engine loaded from an AutoencoderKL-style safetensors file (synthetic weights)."""
    from PIL import Image
    from safetensors.torch import save_file
    from oracle import vae_oracle as V8, ni_oracle as O
    from naturaldiffusion_amd import ValidateNaturalInference as V
    P = V8.make_params(4, seed=2)
    sd = {"decoder." + k: v.contiguous() for k, v in P.items()}
    sd["post_quant_conv.weight"] = torch.eye(4).reshape(4, 4, 1, 1).contiguous()
    sd["post_quant_conv.bias"] = torch.zeros(4)
    (tmp_path / "vae").mkdir()
    save_file(sd, str(tmp_path / "vae" / "diffusion_pytorch_model.safetensors"))

    base = O.analytic_eps_model()

    class FakeDiT:
        def forward(self, z, t, y):
            e = base(z, int(t[0])) * (0.9 if bool((y == 1000).all()) else 1.1)
            return torch.cat([e, torch.zeros_like(e)], dim=1)
    monkeypatch.setattr(V, "denoiser_factory", lambda: FakeDiT())
    monkeypatch.setattr(V, "vae_path", str(tmp_path / "vae"))
    monkeypatch.setattr(V, "root_path", tmp_path)
    monkeypatch.setattr(V, "device", "cuda:0")
    (tmp_path / "results" / "ddim").mkdir(parents=True)
    import shutil
    shutil.copy(V.__file__.rsplit("/", 2)[0] + "/results/ddim/ddim_024.npz", tmp_path / "results" / "ddim" / "ddim_024.npz")
    z = V.natural_inference("ddim", 24)
    img = Image.open(tmp_path / "results" / "validation" / "ddim_024__seed_0__natural.png")
    assert img.size == (4 * 258 + 2, 2 * 258 + 2)
    # the pixels are the decoder engine's output for those latents
    from naturaldiffusion_amd.vae import VAEDecoder, flatten_state_dict
    ref = V8.decode(P, (z / 0.18215).cpu())
    got = torch.from_numpy(__import__("numpy").array(img)).permute(2, 0, 1)[:, 2:258, 2:258].float() / 255 * 2 - 1
    assert (got - ref[0].clamp(-1, 1)).abs().max().item() <= 0.12
