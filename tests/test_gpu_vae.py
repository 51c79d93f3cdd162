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
    # a sample's result does not depend on its batch neighbours: bit-identical when both runs use the same GEMM tile variant;
    # the automatic choice differs with M (different fp32 summation orders -> bf16 rounding flips, both within TOL of the oracle)
    solo = dec(z[2:3].cuda()).cpu()
    assert ((solo - ref).abs().max() / ref.abs().max()).item() <= TOL
    from naturaldiffusion_amd._lib import lib
    try:
        lib.natinf_set_gemm_variant(17)
        assert torch.equal(dec(z[2:3].cuda()).cpu(), dec(z.cuda()).cpu()[2:3])
    finally:
        lib.natinf_set_gemm_variant(0)


def test_sd3_size_decode_matches_oracle():
    """128x128 latents with 16 channels -> 1024x1024 images (SD3, src/SD3NaturalInference.py:238-243): 16,384-token mid-block
    attention, GroupNorm statistics folded from 4,096 tile partials per sample; sample 1 of a batch of 2 against the oracle."""
    from oracle import vae_oracle as V
    from naturaldiffusion_amd.vae import VAEDecoder, flatten_state_dict
    P = V.make_params(16, seed=5)
    dec = VAEDecoder(flatten_state_dict(P, 16), max_batch=2, latent_ch=16, latent_res=128)
    z = torch.randn(2, 16, 128, 128, generator=torch.Generator().manual_seed(4))
    out = dec(z.cuda()).cpu()
    assert out.shape == (2, 3, 1024, 1024) and torch.isfinite(out).all()
    ref = V.decode(P, z[1:2])
    assert ((out[1:2] - ref).abs().max() / ref.abs().max()).item() <= TOL
    solo = dec(z[1:2].cuda()).cpu()
    assert ((solo - ref).abs().max() / ref.abs().max()).item() <= TOL


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
    """src/ValidateNaturalInference.py:231-236 end of a sampler: latents / 0.18215 -> vae.decode -> 1x8 image row on disk (save_image nrow=8),
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
    assert img.size == (8 * 258 + 2, 258 + 2)
    # the pixels are the decoder engine's output for those latents
    from naturaldiffusion_amd.vae import VAEDecoder, flatten_state_dict
    ref = V8.decode(P, (z / 0.18215).cpu())
    got = torch.from_numpy(__import__("numpy").array(img)).permute(2, 0, 1)[:, 2:258, 2:258].float() / 255 * 2 - 1
    assert (got - ref[0].clamp(-1, 1)).abs().max().item() <= 0.12


def test_sd3_script_decodes_with_the_engine_and_writes_the_row_of_images(tmp_path, monkeypatch):
    """src/SD3NaturalInference.py:238-243: final latents / scaling_factor + shift_factor -> vae.decode -> one row of images on
    disk.  Fake pipe (analytic velocity field, 16x16 latents), the decoder engine built from ``pipe.vae.state_dict()``."""
    import numpy as np
    from PIL import Image
    from oracle import vae_oracle as V
    from naturaldiffusion_amd import SD3NaturalInference as S
    P = V.make_params(16, seed=2)

    class Sched:
        def set_timesteps(self, n, device=None):
            from oracle import ni_oracle as O
            self.timesteps, self.sigmas = O.sd3_sigma_schedule(n)

    class Vae:
        class config:
            scaling_factor, shift_factor, latent_channels = 1.5305, 0.0609, 16

        def state_dict(self):
            return {"decoder." + k: v for k, v in P.items()}

    class Pipe:
        scheduler, vae = Sched(), Vae()

        def encode_prompt(self, prompt, **k):
            return (1.0, 0.0, None, None)

        def transformer(self, hidden_states, timestep, encoder_hidden_states, pooled_projections, return_dict=False):
            return [(hidden_states * (0.3 + 0.1 * encoder_hidden_states)).to(hidden_states.dtype)]
    monkeypatch.setattr(S, "results_path", tmp_path)
    pipe = S.use_native_vae(Pipe(), 2, latent_side=16)
    noises = torch.randn(2, 16, 16, 16, generator=torch.Generator().manual_seed(3)).half().cuda()
    finals = S.sd_natural_inference_tx(pipe=pipe, device="cuda:0", noises=noises, n=2, weight_names=("sd3_step_28_weight.csv",))
    img = np.array(Image.open(tmp_path / "results/sd3/sgl_sd3_step_28_weight.png"))
    assert img.shape == (128, 256, 3)
    ref = V.decode(P, finals[0].cpu().float() / 1.5305 + 0.0609)
    ref8 = ((ref * 0.5 + 0.5).clamp(0, 1) * 255).round().permute(0, 2, 3, 1).numpy()
    want = np.hstack(list(ref8))
    assert np.abs(img.astype(np.float64) - want).max() <= 255 * 0.5 * TOL * float(ref.abs().max()) + 1
