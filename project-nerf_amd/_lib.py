"""ctypes binding of libnerf_hip.so (C ABI declared in include/nerf_hip.h).

The library is loaded lazily and exactly once; a missing library is a hard error (there is
no CPU or PyTorch fallback for the hot path).  ``torch`` must be imported first so that the
HIP runtime the process already uses (torch's bundled libamdhip64.so.7) satisfies our
DT_NEEDED entry instead of a second copy.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnerf_hip.so")

c_f32p = ctypes.c_void_p
c_ptr = ctypes.c_void_p
i64 = ctypes.c_int64
i32 = ctypes.c_int
f32 = ctypes.c_float
size_t = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/nerf_hip.h one to one
PROTOTYPES = {
    "nerf_last_error": (ctypes.c_char_p, []),
    "nerf_abi_version": (i32, []),
    "nerf_set_option": (i32, [ctypes.c_char_p, i32]),
    "nerf_get_option": (i32, [ctypes.c_char_p, ctypes.POINTER(i32)]),
    "nerf_sample_rays": (i32, [c_ptr, c_ptr, c_ptr, i64, i32, f32, f32, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_gather_rays": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i64, i32, i32, i32, f32, f32, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_gather_batch": (i32, [c_ptr, c_ptr, c_ptr, i64, i32, i32, i32, f32, f32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_train_batch": (i32, [c_ptr, c_ptr, i32, i32, i32, f32, f32, c_ptr, ctypes.c_uint64, ctypes.c_uint64, i64, i32, f32, f32,
                               i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_train_batch_shard": (i32, [c_ptr, c_ptr, i32, i32, i32, f32, f32, c_ptr, ctypes.c_uint64, ctypes.c_uint64, i64, i64, i32, f32, f32,
                                     i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_active_mask": (i32, [c_ptr, i64, c_ptr, i32, f32, c_ptr, c_ptr, c_ptr]),
    "nerf_sample_compact": (i32, [c_ptr, c_ptr, c_ptr, i64, i32, f32, f32, c_ptr, i32, f32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_sample_compact_jitter": (i32, [c_ptr, c_ptr, ctypes.c_uint64, ctypes.c_uint64, i64, i32, f32, f32, c_ptr, i32, f32,
                                         c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_sample_compact_jitter_shard": (i32, [c_ptr, c_ptr, ctypes.c_uint64, ctypes.c_uint64, i64, i64, i32, f32, f32, c_ptr, i32, f32,
                                               c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_sample_compact_jitter_chain": (i32, [c_ptr, c_ptr, ctypes.c_uint64, ctypes.c_uint64, i64, i64, i32, f32, f32, c_ptr, i32, f32,
                                               c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i32, c_ptr, ctypes.c_uint32, c_ptr]),
    "nerf_composite_fwd_indexed": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i64, i64, i32, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_composite_bwd_indexed": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i64, c_ptr, c_ptr, c_ptr, i64, i32, c_ptr, c_ptr, c_ptr]),
    "nerf_sample_pdf": (i32, [c_ptr, c_ptr, c_ptr, i64, i32, i32, c_ptr, c_ptr]),
    "nerf_grid_lattice": (i32, [f32, i32, c_ptr, c_ptr]),
    "nerf_grid_update": (i32, [c_ptr, c_ptr, c_ptr, i64, f32, i32, f32, c_ptr, c_ptr]),
    "nerf_fourier_encode": (i32, [c_ptr, i64, i32, i32, c_ptr, c_ptr]),
    "nerf_composite_fwd": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i64, c_ptr, i64, i32,
                                 c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_composite_bwd": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i64, c_ptr, c_ptr, c_ptr, c_ptr,
                                 c_ptr, i64, i32, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_render_rays_workspace_bytes": (size_t, [i64, i32]),
    "nerf_render_rays_fwd": (i32, [c_ptr, c_ptr, c_ptr, i64, i32, f32, f32, c_ptr, i64, i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_mlp_packed_bytes": (size_t, []),
    "nerf_mlp_pack": (i32, [c_ptr, c_ptr, c_ptr]),
    "nerf_mlp_pack_streams": (i32, [c_ptr, c_ptr, i32, c_ptr]),
    "nerf_mlp_stash_bytes": (size_t, [i64]),
    "nerf_mlp_fwd": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, i64, i32, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_mlp_fwd_encoded": (i32, [c_ptr, c_ptr, c_ptr, i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_mlp_bwd_workspace_bytes": (size_t, [i64]),
    "nerf_mlp_bwd": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i64, c_ptr, c_ptr, c_ptr]),
    "nerf_mlp_bwd_dgrad": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i64, c_ptr, c_ptr]),
    "nerf_mlp_bwd_wgrad": (i32, [c_ptr, c_ptr, i64, c_ptr, c_ptr]),
    "nerf_mlp_bwd_dgrad_ex": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i64, c_ptr, c_ptr, c_ptr]),
    "nerf_composite_mse_bwd": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i64, c_ptr, f32, i64, i32, c_ptr, c_ptr,
                                     c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_mlp_wgrad_part_split": (i64, []),
    "nerf_mlp_bwd_wgrad_part": (i32, [c_ptr, c_ptr, i64, c_ptr, i32, c_ptr]),
    "nerf_hash_encode_fwd": (i32, [c_ptr, i64, c_ptr, i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, f32, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_hash_encode_fwd_f16": (i32, [c_ptr, i64, c_ptr, i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, f32, c_ptr, c_ptr, c_ptr]),
    "nerf_hash_encode_fwd_nat": (i32, [c_ptr, i64, c_ptr, c_ptr, i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, f32, c_ptr, i32, c_ptr]),
    "nerf_hash_encode_fwd_nat_tables": (i32, [c_ptr, i64, c_ptr, i32, i64, i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, f32, c_ptr, i64, i32, c_ptr]),
    "nerf_f32_to_f16": (i32, [c_ptr, c_ptr, i64, c_ptr]),
    "nerf_hash_encode_bwd": (i32, [c_ptr, i64, i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, f32, c_ptr, c_ptr, c_ptr]),
    "nerf_hash_encode_bwd_levels": (i32, [c_ptr, i64, i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, f32, c_ptr, c_ptr, i32, i32, c_ptr]),
    "nerf_hash_encode_bwd_workspace_bytes": (size_t, [i64, i32]),
    "nerf_hash_encode_bwd_ws": (i32, [c_ptr, i64, i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, f32, c_ptr, c_ptr, i32, i32, c_ptr, size_t, c_ptr]),
    "nerf_hash_encode_bwd_ws_store": (i32, [c_ptr, i64, i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, f32, c_ptr, c_ptr, i32, i32, c_ptr, size_t, c_ptr]),
    "nerf_hash_encode_bwd_tables_workspace_bytes": (size_t, [i64, i32, i32]),
    "nerf_hash_encode_bwd_ws_store_tables": (i32, [c_ptr, i64, i32, i64, i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, f32, c_ptr, i64, c_ptr, c_ptr, size_t, c_ptr]),
    "nerf_hash_encode_bwd_ws_store_tables_spec": (i32, [c_ptr, i64, i32, i64, i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, f32, c_ptr, i64, c_ptr, c_ptr, size_t,
                                                        c_ptr, c_ptr]),
    "nerf_hash_encode_fwd_f16_hist": (i32, [c_ptr, i64, c_ptr, i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, f32, c_ptr, c_ptr, size_t, c_ptr]),
    "nerf_hash_encode_bwd_ws_slots": (i32, [c_ptr, i64, i32, ctypes.POINTER(c_ptr), ctypes.POINTER(c_ptr)]),
    "nerf_hash_encode_bwd_ws_store_precounted": (i32, [c_ptr, i64, i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, f32, c_ptr, c_ptr, size_t, c_ptr]),
    "nerf_hash_encode_bwd_spec_begin": (i32, [c_ptr, c_ptr]),
    "nerf_hash_encode_bwd_ws_store_spec": (i32, [c_ptr, i64, i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, f32, c_ptr, c_ptr, c_ptr, size_t, c_ptr, c_ptr]),
    "nerf_imlp_bwd_amax": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_hash_encode_bwd_spec_status": (c_ptr, [c_ptr]),
    "nerf_imlp_bwd_lm": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_hash_encode_bwd_input": (i32, [c_ptr, i64, c_ptr, i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, f32, c_ptr, c_ptr, c_ptr]),
    "nerf_hash_encode_bwd_input_f16": (i32, [c_ptr, i64, c_ptr, i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, f32, c_ptr, c_ptr, c_ptr]),
    "nerf_hash_encode_bwd_input_f16_accum": (i32, [c_ptr, i64, c_ptr, i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, f32, c_ptr, c_ptr, c_ptr]),
    "nerf_hash_encode_bwd_input_lm_f16": (i32, [c_ptr, i64, c_ptr, i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, f32, c_ptr, c_ptr, i32, c_ptr]),
    "nerf_imlp_packed_bytes": (size_t, []),
    "nerf_imlp_workspace_bytes": (size_t, [i64]),
    "nerf_imlp_hash_operand_offset": (size_t, [i64]),
    "nerf_imlp_pack": (i32, [c_ptr, c_ptr, c_ptr]),
    "nerf_imlp_fwd": (i32, [c_ptr, c_ptr, c_ptr, i64, c_ptr, c_ptr, i32, c_ptr]),
    "nerf_imlp_fwd_encoded": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, i64, c_ptr, c_ptr, i32, c_ptr]),
    "nerf_imlp_bwd": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i64, c_ptr, c_ptr, c_ptr]),
    "nerf_adam_step": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, i64, i32, f32, f32, f32, f32, f32, c_ptr, c_ptr]),
    "nerf_tv_normsq": (i32, [c_ptr, c_ptr, i64, f32, f32, c_ptr, c_ptr]),
    "nerf_tv_normsq_accum": (i32, [c_ptr, c_ptr, i64, f32, f32, c_ptr, c_ptr]),
    "nerf_tv_normsq_accum_tables": (i32, [c_ptr, c_ptr, i64, i32, f32, f32, c_ptr, c_ptr]),
    "nerf_composite_mse_reg_bwd": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i64, c_ptr, f32, c_ptr, f32, i64, i32, c_ptr, c_ptr,
                                         c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_sample_compact_ordered_scratch_bytes": (size_t, [i64, i32]),
    "nerf_sample_compact_ordered": (i32, [c_ptr, c_ptr, c_ptr, i32, ctypes.c_uint64, ctypes.c_uint64, i64, i64, i32, f32, f32, c_ptr, i32, f32,
                                          c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, size_t, c_ptr]),
    "nerf_p4_param_count": (i64, []),
    "nerf_p4_packed_bytes": (size_t, []),
    "nerf_p4_workspace_bytes": (size_t, [i64]),
    "nerf_p4_workspace_offset": (size_t, [i64, i32]),
    "nerf_p4_pack": (i32, [c_ptr, c_ptr, c_ptr]),
    "nerf_p4_sample_inputs": (i32, [c_ptr, c_ptr, c_ptr, i64, i32, f32, f32, ctypes.c_uint64, ctypes.c_uint64, i64, c_ptr, c_ptr, c_ptr]),
    "nerf_p4_deform_fwd": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i64, c_ptr, c_ptr, i32, c_ptr]),
    "nerf_p4_canon_fwd": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, i64, c_ptr, c_ptr, i32, c_ptr]),
    "nerf_p4_canon_bwd": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_p4_deform_bwd": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "nerf_tv_codes_bytes": (size_t, [i64]),
    "nerf_tv_normsq_codes": (i32, [c_ptr, c_ptr, i64, i32, f32, f32, c_ptr, i32, c_ptr, c_ptr]),
    "nerf_clip_adamw_small": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, i64, i32, f32, f32, f32, f32, f32, f32, f32, c_ptr, i32, c_ptr]),
    "nerf_adamw_clip_step_tv": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, i64, i32, f32, f32, f32, f32, f32, c_ptr, f32, f32, c_ptr, i64, f32, i64,
                                      f32, i64, i64, f32, c_ptr, c_ptr]),
    "nerf_tv_normsq_codes_piece": (i32, [c_ptr, c_ptr, i64, i64, i32, f32, f32, c_ptr, i32, c_ptr, c_ptr]),
    "nerf_adamw_clip_step_tv_piece": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, i64, i32, f32, f32, f32, f32, f32, c_ptr, f32, f32, c_ptr, f32, i64, i32,
                                            c_ptr, c_ptr]),
    "nerf_adamw_clip_step": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, i64, i32, f32, f32, f32, f32, f32, c_ptr, f32, f32, c_ptr]),
    "nerf_adamw_clip_step_shadow": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, i64, i32, f32, f32, f32, f32, f32, c_ptr, f32, f32, c_ptr, c_ptr]),
}

ABI_VERSION = 3      # the NERF_ABI_VERSION of include/nerf_hip.h this table was written against

_lib = None


class NerfHipError(RuntimeError):
    pass


def load():
    """Returns the loaded library; raises NerfHipError if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NerfHipError(
            f"{LIB_PATH} not found: build it with `python project-nerf_amd/build.py` "
            "(hipcc --offload-arch=gfx950).  There is no fallback path.")
    import torch  # noqa: F401  (brings the process-wide HIP runtime in first)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.nerf_abi_version() != ABI_VERSION:
        raise NerfHipError(f"ABI version mismatch: {LIB_PATH} reports {lib.nerf_abi_version()}, this binding was written against "
                           f"{ABI_VERSION} (rebuild with `python project-nerf_amd/build.py`)")
    _lib = lib
    return lib


NORMSQ_WS_FLOATS = 4136
COMPACT_CHAIN_WORDS = 40     # NERF_COMPACT_CHAIN_WORDS     # NERF_NORMSQ_WS_FLOATS
SUM_WS_FLOATS = 12288       # NERF_SUM_WS_FLOATS


def set_option(name: str, value: int) -> None:
    """Development switch of the library (kernel family, timing skeletons); see include/nerf_hip.h."""
    check(load().nerf_set_option(name.encode(), int(value)), f"nerf_set_option({name})")


def get_option(name: str) -> int:
    out = i32(0)
    check(load().nerf_get_option(name.encode(), ctypes.byref(out)), f"nerf_get_option({name})")
    return out.value


def check(code, what):
    if code != 0:
        msg = load().nerf_last_error().decode("utf-8", "replace")
        raise NerfHipError(f"{what} failed ({code}): {msg}")
