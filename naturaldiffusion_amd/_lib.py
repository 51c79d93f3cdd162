"""ctypes binding of libnatinf.so (include/natinf.h, include/natinf_ncsnpp.h).

The product path fails loudly: no library -> ImportError with the build command; a negative
return code -> RuntimeError naming the entry point.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); two streams that share a queue run one after the other.
# The engines count on concurrency between a few streams (the two lanes of the CIFAR10 pipeline, the MMDiT's text stream beside its image stream) next to whatever
# streams the host framework made: ask for eight unless the user chose.  It only takes effect when set before the process's first HIP call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("NATINF_LIB", _HERE / "libnatinf.so"))

if not LIB_PATH.exists():
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C {_HERE / 'csrc'}` "
        "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
        "naturaldiffusion_amd has no CPU or eager-PyTorch fallback.")

lib = C.CDLL(str(LIB_PATH))

_p, _i32, _i64, _f32, _f64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double

SIGNATURES = {
    "natinf_abi_version": (C.c_int, []),
    "natinf_strerror": (C.c_char_p, [_i32]),
    "natinf_probe": (C.c_int, []),
    "natinf_step_f64hist": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i32, _f64, _i32, _f64, _f64, _f32, _f32, _i64, _p]),
    "natinf_step_f32hist": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i32, _f32, _i32, _f32, _f32, _f32, _f32, _i64, _p]),
    "natinf_weighted_sum_f64": (C.c_int, [_p, _p, _p, _p, _i32, _i64, _p]),
    "natinf_randn_philox_f32": (C.c_int, [_p, _i64, _i64, _p, _i64, _i64, C.c_uint64, _p]),
    "natinf_to_pixel_u8": (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "natinf_step_f32prod": (C.c_int, [_p, _p, _p, _f32, _i64, _i64, _p, _p, _p, _p, _p, _i32, _f32, _p, _p, _i32,
                                      _i32, _f32, _f32, _i64, _p]),
    "natinf_weighted_sum_f32prod": (C.c_int, [_p, _p, _p, _p, _i32, _i64, _p]),
    "natinf_step_f16chain": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _f32, _f32, _i32, _f32, _f32, _f32,
                                       _f32, _i32, _i64, _p]),
    "natinf_weighted_mean_f16": (C.c_int, [_p, _p, _p, _p, _i32, _f32, _i64, _p]),
    "natinf_flow_input_f16": (C.c_int, [_p, _p, _p, _f32, _f32, _i64, _p]),
    # include/natinf_ncsnpp.h
    "natinf_ncsnpp_param_count": (C.c_int64, []),
    "natinf_ncsnpp_workspace_bytes": (C.c_int64, [_p, _i32]),
    "natinf_ncsnpp_packed_bytes": (C.c_int64, []),
    "natinf_ncsnpp_handle_param_count": (_i64, [_p]),
    "natinf_ncsnpp_handle_packed_bytes": (_i64, [_p]),
    "natinf_ncsnpp_create": (C.c_int, [C.POINTER(_p), _i32]),
    "natinf_ncsnpp_describe": (C.c_int, [_p, C.c_char_p, _i32]),
    "natinf_ncsnpp_destroy": (C.c_int, [_p]),
    "natinf_ncsnpp_load": (C.c_int, [_p, _p, _i64, _p, _i64, _p]),
    "natinf_ncsnpp_share": (C.c_int, [_p, _p]),
    "natinf_ncsnpp_forward": (C.c_int, [_p, _p, _p, _p, _i32, _p, _i64, _p]),
    "natinf_ncsnpp_debug_tap": (C.c_int, [_p, _i32, _p, _i64, _p]),
    "natinf_ncsnpp_describe_gemms": (C.c_int, [_p, _i32, C.c_char_p, _i32]),
    "natinf_debug_gemm": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _i32, _f32, _i32, _p]),
    "natinf_set_gemm_variant": (C.c_int, [_i32]),
    "natinf_debug_timestamps": (C.c_int, [_p]),
    "natinf_set_gemm_epilogue": (C.c_int, [_i32]),
    "natinf_debug_gemm_fused": (C.c_int, [_i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _f32, _i32, _p, _i32, _p, _p, _i32, _p]),
    "natinf_set_gemm_pref512": (C.c_int, [_i32]),
    "natinf_set_gemm_half_issue": (C.c_int, [_i32]),
    "natinf_set_gemm_raster": (C.c_int, [_i32]),
    "natinf_set_gemm_round_model": (C.c_int, [_i32]),
    "natinf_set_gemm_w128": (C.c_int, [_i32]),
    "natinf_set_fuse_gn": (C.c_int, [_i32]),
    "natinf_set_conv_gn_wide": (C.c_int, [_i32]),
    "natinf_set_attn_block": (C.c_int, [_i32]),
    "natinf_set_conv_gn_w128": (C.c_int, [_i32]),
    "natinf_set_conv_gn_w128_min_k": (C.c_int, [_i32, _i32]),
    "natinf_set_conv_gn_regw": (C.c_int, [_i32]),
    "natinf_set_fuse_up": (C.c_int, [_i32]),
    "natinf_set_fuse_head": (C.c_int, [_i32]),
    "natinf_set_fuse_gn8": (C.c_int, [_i32]),
    "natinf_set_fuse_gn4": (C.c_int, [_i32]),
    "natinf_set_fuse_fin": (C.c_int, [_i32]),
    "natinf_set_attn_proj": (C.c_int, [_i32]),
    "natinf_set_attn_waves8": (C.c_int, [_i32]),
    "natinf_set_attn_qkv": (C.c_int, [_i32]),
    "natinf_set_conv_gn_warm": (C.c_int, [_i32]),
    "natinf_set_attn256": (C.c_int, [_i32]),
    "natinf_set_conv_gn8_tile": (C.c_int, [_i32]),
    "natinf_set_gemm_splitk": (C.c_int, [_i32]),
    "natinf_debug_set_splitk_workspace": (C.c_int, [_p, _i32]),
    "natinf_debug_conv_gn": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _f32, _p, _p, _i32, _p]),
    "natinf_debug_conv_gn_up": (C.c_int, [_i32]),
    "natinf_ncsnpp_profile": (C.c_int, [_p, _i32]),
    "natinf_ncsnpp_profile_read": (C.c_int, [_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "natinf_debug_quant_fp8_rows": (C.c_int, [_p, _p, _p, _i32, _i32, _p]),
    "natinf_debug_gemm_fp8": (C.c_int, [_i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _p]),
    # include/natinf_dit.h
    "natinf_dit_create": (C.c_int, [C.POINTER(_p), _i32, _i32, _i32, _i32]),
    "natinf_dit_destroy": (C.c_int, [_p]),
    "natinf_dit_param_count": (C.c_int64, [_p]),
    "natinf_dit_packed_bytes": (C.c_int64, [_p]),
    "natinf_dit_workspace_bytes": (C.c_int64, [_p, _i32]),
    "natinf_dit_load": (C.c_int, [_p, _p, _i64, _p, _i64, _p]),
    "natinf_dit_forward": (C.c_int, [_p, _p, _p, _p, _p, _i32, _p, _i64, _p]),
    # include/natinf_mmdit.h
    "natinf_mmdit_create": (C.c_int, [C.POINTER(_p), _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "natinf_mmdit_destroy": (C.c_int, [_p]),
    "natinf_mmdit_param_count": (C.c_int64, [_p]),
    "natinf_mmdit_packed_bytes": (C.c_int64, [_p]),
    "natinf_mmdit_workspace_bytes": (C.c_int64, [_p, _i32]),
    "natinf_mmdit_load": (C.c_int, [_p, _p, _i64, _p, _i64, _p]),
    "natinf_mmdit_forward": (C.c_int, [_p, _p, _p, _p, _p, _p, _i32, _p, _i64, _p]),
    "natinf_attention_hd64_bf16": (C.c_int, [_p, _p, _i32, _i64, _p, _p, _i32, _i64, _i32, _i32, _i32, _i32, C.c_float, _p]),
    "natinf_inception_create": (C.c_int, [_p, _i32, _i32]),
    "natinf_inception_destroy": (C.c_int, [_p]),
    "natinf_set_inception_conv": (C.c_int, [_i32]),
    "natinf_debug_conv_ring": (C.c_int, [_i32] * 14 + [_p, _p, _p, _p, _p, _p, _i32, _p]),
    "natinf_inception_param_count": (_i64, [_p]),
    "natinf_inception_packed_bytes": (_i64, [_p]),
    "natinf_inception_workspace_bytes": (_i64, [_p, _i32]),
    "natinf_inception_load": (C.c_int, [_p, _p, _i64, _p, _i64, _p]),
    "natinf_inception_forward": (C.c_int, [_p, _p, _i32, _p, _i32, _p, _i64, _p]),
    "natinf_gemm_profile": (C.c_int, [_i32]),
    "natinf_gemm_profile_read": (C.c_int, [_p, _i32]),
    "natinf_set_mmdit_stream16": (C.c_int, [_i32]),
    "natinf_set_dit_stream16": (C.c_int, [_i32]),
    "natinf_set_mmdit_text_flat": (C.c_int, [_i32]),
    "natinf_attention_profile": (C.c_int, [_i32]),
    "natinf_set_mmdit_text_stream": (C.c_int, [_i32]),
    "natinf_set_flash_mode": (C.c_int, [_i32]),
    "natinf_attention_profile_read": (C.c_int, [_p, _p]),
    # include/natinf_vae.h
    "natinf_vae_create": (C.c_int, [C.POINTER(_p), _i32, _i32]),
    "natinf_vae_destroy": (C.c_int, [_p]),
    "natinf_vae_param_count": (C.c_int64, [_p]),
    "natinf_vae_packed_bytes": (C.c_int64, [_p]),
    "natinf_vae_workspace_bytes": (C.c_int64, [_p, _i32]),
    "natinf_vae_load": (C.c_int, [_p, _p, _i64, _p, _i64, _p]),
    "natinf_vae_decode": (C.c_int, [_p, _p, _p, _i32, _p, _i64, _p]),
}

for _name, (_res, _args) in SIGNATURES.items():
    try:
        _fn = getattr(lib, _name)
    except AttributeError as e:                       # header and library out of sync
        raise ImportError(f"{LIB_PATH} does not export {_name}; rebuild it") from e
    _fn.restype, _fn.argtypes = _res, _args

SD3_CFG_ON_VELOCITY = 1


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed: {lib.natinf_strerror(rc).decode()} ({rc})")


def stream_ptr(stream=None) -> int:
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return int(s.cuda_stream)


def ptr(t) -> int:
    return 0 if t is None else int(t.data_ptr())


def require_gpu() -> None:
    """Raise unless a gfx950 device is visible and the code object loads on it."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("naturaldiffusion_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False "
                           "and there is no CPU fallback")
    check(lib.natinf_probe(), "natinf_probe")
