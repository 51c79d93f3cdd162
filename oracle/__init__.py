"""CPU oracle for the Natural Inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``naturaldiffusion_amd/`` may import,
call or execute anything in this package: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and
only as the checker.  The oracle is a plain torch-CPU / numpy restatement of the
reference's arithmetic (dtype promotions and operation order included); every
function cites the reference ``file:line`` it follows.

Parity status: PINNED for the sampler half (A1-A10 of SURVEY.md section 8) and
for the NCSN++ denoiser (A4) by the fixtures under ``tests/golden/`` that
``tests/golden/make_golden.py`` captured from the reference itself, imported
on CPU in the build container; ``dit_oracle`` is pinned to the reference's own
``DiT`` class for everything DiT-specific.  PARITY UNPINNED: ``mmdit_oracle``
and ``vae_oracle`` (the SD3 MMDiT and the AutoencoderKL decoder are
``diffusers``' -- un-vendored, un-pinned, absent here; the files restate the
published architectures) and the three ``timm`` building blocks inside DiT
(see DESIGN.md section 2).
"""
