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
on CPU in the build container.  The SD3 MMDiT / DiT-XL/2 denoiser arithmetic
lives in un-vendored, un-pinned third-party packages (``diffusers``, ``timm``)
and is "parity unpinned" (see DESIGN.md).
"""
