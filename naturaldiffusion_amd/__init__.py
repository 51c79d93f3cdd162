"""naturaldiffusion_amd -- MI355X (gfx950) native Natural Inference sampling engine.

Drop-in for the hot path of blairstar/NaturalDiffusion: the coefficient-matrix recurrence
(``ni_step`` HIP kernels behind the C ABI of ``include/natinf.h``) and the denoiser it wraps.
Host code is Python on PyTorch-ROCm (device memory, streams); all arithmetic of the path runs in
``libnatinf.so``.  There is NO CPU / eager fallback: importing :mod:`naturaldiffusion_amd._lib`
raises if the library is missing, and every op raises if it cannot run on the GPU.
"""
from .coeff import load_coeff_npz, load_sd3_csv, SparseRows  # noqa: F401

__all__ = ["load_coeff_npz", "load_sd3_csv", "SparseRows"]
