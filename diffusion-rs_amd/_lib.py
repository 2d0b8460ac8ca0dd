"""ctypes loader for libflux_mi355x.so (the C-ABI of include/flux_mi355x.h).

The product path has NO CPU fallback: if the HIP library is missing this module raises, and
every compute entry point raises FmiError on a non-zero status.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FMI_LIB = another build of the same library (A/B of two builds on one box: tools/ab_toggle.py, DESIGN's "alternating processes"); default: in-tree
LIB_PATH = os.environ.get("FMI_LIB") or os.path.join(_HERE, "libflux_mi355x.so")
# the TEST build of the same sources (make alt: -DFMI_ALT_KERNELS=1, the superseded kernels compiled in as well): loaded next to the product library by the
# bit-identity cross-checks of tests/ (load_alt / use_alt), never by the product
ALT_LIB_PATH = os.path.join(_HERE, "libflux_mi355x_alt.so")


class FmiError(RuntimeError):
    """A non-zero fmi_status; `code` is the status (include/flux_mi355x.h: fmi_status), None for loader errors."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


ERR_INVALID, ERR_HIP, ERR_STATE, ERR_UNSUPPORTED, ERR_NOMEM = -1, -2, -3, -4, -5


class FluxConfig(C.Structure):
    """fmi_flux_config == models::flux::Config (model.rs:21-31) + model constants."""
    _fields_ = [("in_channels", C.c_int), ("pooled_projection_dim", C.c_int), ("joint_attention_dim", C.c_int),
                ("num_attention_heads", C.c_int), ("num_layers", C.c_int), ("num_single_layers", C.c_int),
                ("guidance_embeds", C.c_int), ("axes_dim", C.c_int * 3), ("theta", C.c_int)]


# fmi_all_to_all_fn (include/flux_mi355x.h): int (*)(void* user, const void* send, void* recv, size_t bytes_per_peer, void* stream)
ALL_TO_ALL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class FluxInputs(C.Structure):
    _fields_ = [("img", C.c_void_p), ("img_dtype", C.c_int), ("img_ids", C.c_void_p), ("txt", C.c_void_p), ("txt_dtype", C.c_int),
                ("txt_ids", C.c_void_p), ("timesteps", C.c_void_p), ("y", C.c_void_p), ("y_dtype", C.c_int), ("guidance", C.c_void_p),
                ("B", C.c_int), ("S", C.c_int), ("T", C.c_int), ("ids_per_sample", C.c_int)]


class VaeConfig(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("block_out_channels", C.c_int * 4), ("n_blocks", C.c_int),
                ("layers_per_block", C.c_int), ("latent_channels", C.c_int), ("norm_num_groups", C.c_int),
                ("mid_block_add_attention", C.c_int), ("use_post_quant_conv", C.c_int), ("scaling_factor", C.c_double),
                ("shift_factor", C.c_double), ("use_quant_conv", C.c_int)]


class SchedulerConfigC(C.Structure):
    _fields_ = [("base_image_seq_len", C.c_int), ("base_shift", C.c_double), ("max_image_seq_len", C.c_int), ("max_shift", C.c_double),
                ("shift", C.c_double), ("use_dynamic_shifting", C.c_int)]


class T5Config(C.Structure):
    """fmi_t5_config == t5::T5Config (t5/mod.rs:72-91)."""
    _fields_ = [("vocab_size", C.c_int), ("d_model", C.c_int), ("d_kv", C.c_int), ("d_ff", C.c_int), ("num_layers", C.c_int), ("num_heads", C.c_int),
                ("relative_attention_num_buckets", C.c_int), ("relative_attention_max_distance", C.c_int), ("layer_norm_epsilon", C.c_float),
                ("feed_forward_act", C.c_int)]


class ClipConfig(C.Structure):
    """fmi_clip_config == clip::text::ClipTextConfig (clip/text.rs:24-33)."""
    _fields_ = [("vocab_size", C.c_int), ("projection_dim", C.c_int), ("intermediate_size", C.c_int), ("max_position_embeddings", C.c_int),
                ("num_hidden_layers", C.c_int), ("num_attention_heads", C.c_int)]


F32, F16, BF16, U8, I8 = 0, 1, 2, 3, 4
MODEL_AUTO, MODEL_BF16, MODEL_F16, MODEL_F32 = 0, 1, 2, 3

_lib = None
_product = None
_alt = None


def load():
    """Load the HIP library.  torch (if used in this process) must be imported first so that a
    single libamdhip64.so.7 serves both (same SONAME; see DESIGN.md §process model)."""
    global _lib, _product
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FmiError(f"{LIB_PATH} not found — build it with `make lib` (hipcc --offload-arch=gfx950); there is no CPU fallback")
    try:
        import torch  # noqa: F401  (ensures torch's HIP runtime is the one already mapped)
    except Exception:
        pass
    _lib = _product = _declare(C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL))
    return _lib


def load_alt():
    """The test build (libflux_mi355x_alt.so), as a second library in this process; FmiError if it was not built (`make alt`)."""
    global _alt
    if _alt is None:
        load()
        if not os.path.exists(ALT_LIB_PATH):
            raise FmiError(f"{ALT_LIB_PATH} not found — build it with `make alt`")
        _alt = _declare(C.CDLL(ALT_LIB_PATH, mode=C.RTLD_LOCAL))
        if not _alt.fmi_has_alt_kernels():
            raise FmiError(f"{ALT_LIB_PATH} was not compiled with -DFMI_ALT_KERNELS=1")
    return _alt


class use_alt:
    """`with use_alt() as lib:` — inside the block load() returns the test build, so handles created there (FluxModel, ...) and check() bind to it."""

    def __enter__(self):
        global _lib
        alt = load_alt()
        self.prev, _lib = _lib, alt
        return alt

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


def _declare(lib):
    lib.fmi_last_error.restype = C.c_char_p
    lib.fmi_device_info.restype = C.c_char_p
    lib.fmi_build_id.restype = C.c_char_p
    lib.fmi_flux_missing_name.restype = C.c_char_p
    lib.fmi_vae_missing_name.restype = C.c_char_p
    lib.fmi_flux_phase_name.restype = C.c_char_p
    lib.fmi_flux_size_in_bytes.restype = C.c_size_t
    lib.fmi_calculate_shift.restype = C.c_double
    lib.fmi_calculate_shift.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
    lib.fmi_vae_scale_factor.restype = C.c_double
    lib.fmi_vae_shift_factor.restype = C.c_double
    lib.fmi_unpack_latents.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    lib.fmi_randn.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p]
    lib.fmi_philox_u32.argtypes = lib.fmi_randn.argtypes
    lib.fmi_release_scratch.argtypes = [C.POINTER(C.c_size_t)]
    lib.fmi_timestep_embedding.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.fmi_rope_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_void_p]
    lib.fmi_rmsnorm_rope.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p]
    lib.fmi_malloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    lib.fmi_free.argtypes = [C.c_void_p]
    lib.fmi_memcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.fmi_memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    lib.fmi_flux_set_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]
    lib.fmi_vae_set_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]
    lib.fmi_flux_set_linear_bnb4.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.fmi_flux_set_linear_int8.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.fmi_flux_state_export.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.fmi_flux_state_adopt.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.fmi_flux_state_buffer.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.fmi_flux_set_quant_dense_cache.argtypes = [C.c_void_p, C.c_int]
    lib.fmi_flux_set_split_k.argtypes = [C.c_void_p, C.c_int]
    lib.fmi_flux_set_sequence_parallel.argtypes = [C.c_void_p, C.c_int, C.c_int, ALL_TO_ALL_FN, C.c_void_p]
    lib.fmi_comm_unique_id.argtypes = [C.c_void_p]
    lib.fmi_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.fmi_comm_destroy.argtypes = [C.c_void_p]
    lib.fmi_comm_destroy.restype = None
    lib.fmi_comm_rank.argtypes = [C.c_void_p]
    lib.fmi_comm_world_size.argtypes = [C.c_void_p]
    lib.fmi_comm_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    lib.fmi_comm_all_to_all.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.fmi_comm_broadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    lib.fmi_comm_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    lib.fmi_flux_set_attention_rescale_threshold.argtypes = [C.c_void_p, C.c_int]
    lib.fmi_flux_set_attention_kernel.argtypes = [C.c_void_p, C.c_int]
    lib.fmi_flux_forward.argtypes = [C.c_void_p, C.POINTER(FluxInputs), C.c_void_p, C.c_void_p]
    for enc in ("t5", "clip"):
        getattr(lib, f"fmi_{enc}_set_tensor").argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]
        getattr(lib, f"fmi_{enc}_missing_name").restype = C.c_char_p
        getattr(lib, f"fmi_{enc}_missing_name").argtypes = [C.c_void_p, C.c_int]
        getattr(lib, f"fmi_{enc}_missing_count").argtypes = [C.c_void_p]
        getattr(lib, f"fmi_{enc}_size_in_bytes").restype = C.c_size_t
        getattr(lib, f"fmi_{enc}_size_in_bytes").argtypes = [C.c_void_p]
        getattr(lib, f"fmi_{enc}_destroy").argtypes = [C.c_void_p]
        getattr(lib, f"fmi_{enc}_destroy").restype = None
    lib.fmi_t5_set_linear_bnb4.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.fmi_t5_set_linear_int8.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.fmi_t5_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    lib.fmi_clip_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.fmi_flux_denoise.argtypes = [C.c_void_p, C.POINTER(FluxInputs), C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_void_p]
    lib.fmi_vae_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.fmi_vae_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fmi_vae_mid_attention.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.fmi_pack_latents.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fmi_postprocess_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.fmi_linear_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.fmi_quantize_rows_fp8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fmi_linear_fp8.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.fmi_quantize_rows_i8.argtypes = lib.fmi_quantize_rows_fp8.argtypes
    lib.fmi_linear_i8.argtypes = lib.fmi_linear_fp8.argtypes
    lib.fmi_gemm_q8.argtypes = [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p]
    lib.fmi_quantize_rows_i8_asym.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fmi_rowsum_i8.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.fmi_quantize_rows_i8_scaled.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fmi_col_absmax.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.fmi_flux_calibrate_int8.argtypes = [C.c_void_p, C.c_int]
    lib.fmi_gemm_i8_asym.argtypes = [C.c_void_p] * 8 + [C.c_int] * 4 + [C.c_void_p]
    lib.fmi_flux_quantize_fp8.argtypes = [C.c_void_p, C.c_void_p]
    lib.fmi_flux_quantize_int8.argtypes = [C.c_void_p, C.c_uint, C.c_void_p]
    lib.fmi_flux_set_fp8_attention.argtypes = [C.c_void_p, C.c_int]
    lib.fmi_flux_set_modulation_gemm.argtypes = [C.c_void_p, C.c_int]
    lib.fmi_linear_bnb4_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_void_p]
    lib.fmi_linear_int8_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.fmi_sdpa_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                  C.c_void_p]
    lib.fmi_sdpa_fp8qk.argtypes = lib.fmi_sdpa_bf16.argtypes
    lib.fmi_sdpa_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.fmi_sdpa_workspace_bytes.restype = C.c_size_t
    lib.fmi_sdpa_bf16_ws.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_float, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.fmi_sdpa_fp8qk_ws.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.fmi_sdpa_fp8.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_float, C.c_int, C.c_float, C.c_int, C.c_void_p]
    lib.fmi_sdpa_fp8_ws.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_float, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.fmi_linear_q8_workspace_bytes.argtypes = [C.c_int, C.c_int]
    lib.fmi_linear_q8_workspace_bytes.restype = C.c_size_t
    lib.fmi_linear_fp8_ws.argtypes = list(lib.fmi_linear_fp8.argtypes[:-1]) + [C.c_void_p, C.c_size_t, C.c_void_p]
    lib.fmi_linear_i8_ws.argtypes = lib.fmi_linear_fp8_ws.argtypes
    lib.fmi_layernorm_mod.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
    lib.fmi_groupnorm_nhwc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                       C.c_void_p]
    lib.fmi_conv2d_nhwc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_void_p]
    for name in ("f32", "f16", "bf16"):
        for q in ("int8", "fp4", "nf4"):
            getattr(lib, f"dequantize_blockwise_{name}_{q}").argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
            getattr(lib, f"dequantize_blockwise_{name}_{q}").restype = None
        getattr(lib, f"dequantize_8bit_kernel_{name}").argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        getattr(lib, f"dequantize_8bit_kernel_{name}").restype = None
    lib.fmi_event_create.argtypes = [C.POINTER(C.c_void_p)]
    lib.fmi_event_record.argtypes = [C.c_void_p, C.c_void_p]
    lib.fmi_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    lib.fmi_event_destroy.argtypes = [C.c_void_p]
    lib.fmi_stream_synchronize.argtypes = [C.c_void_p]
    return lib


SDPA_NO_EXP2 = 1 << 30  # FMI_SDPA_NO_EXP2: fmi_sdpa_fp8qk_ws derives the score factor's exponent from `scale`


def tree_build_id(root=None):
    """The build id of the SOURCE TREE, computed exactly as the Makefile does (sha256 over csrc/*, include/*.h, Makefile sorted by path;
    16 hex digits).  Equal to load().fmi_build_id() iff the .so was built from these sources."""
    import glob
    import hashlib
    root = root or os.path.dirname(_HERE)
    files = sorted(glob.glob(os.path.join(root, "diffusion-rs_amd", "csrc", "*")) + glob.glob(os.path.join(root, "include", "*.h")) +
                   [os.path.join(root, "Makefile")], key=lambda p: os.path.relpath(p, root))
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def check(rc, lib=None):
    """`lib`: the library the failing call went to (a handle created under use_alt() keeps calling the test build after the block has ended:
    its error string lives in THAT library's thread-local, not in the one load() returns now)."""
    if rc != 0:
        raise FmiError(f"fmi status {rc}: {(lib or load()).fmi_last_error().decode(errors='replace')}", code=int(rc))


# every symbol include/flux_mi355x.h declares (tests/test_host_logic.py::test_library_exports_every_header_symbol checks they are all exported)
EXPORTED = [
    "fmi_last_error", "fmi_abi_version", "fmi_init", "fmi_device_info", "fmi_build_id", "fmi_has_alt_kernels", "fmi_flux_default_config", "fmi_flux_create", "fmi_flux_destroy",
    "fmi_flux_set_tensor", "fmi_flux_set_linear_bnb4", "fmi_flux_set_linear_int8", "fmi_flux_set_quant_dense_cache", "fmi_flux_set_sequence_parallel", "fmi_flux_set_split_k", "fmi_set_bnb4_onewave_min_rows", "fmi_flux_set_attention_rescale_threshold", "fmi_flux_set_attention_kernel", "fmi_flux_state_buffer_count", "fmi_flux_state_export", "fmi_flux_state_adopt", "fmi_flux_state_buffer", "fmi_flux_set_modulation_gemm", "fmi_flux_quantize_fp8", "fmi_flux_quantize_int8", "fmi_flux_calibrate_int8", "fmi_flux_set_fp8_attention", "fmi_flux_missing_count", "fmi_flux_missing_name", "fmi_flux_size_in_bytes",
    "fmi_flux_forward", "fmi_flux_denoise", "fmi_flux_set_profiling", "fmi_flux_set_fused_qkv_relayout", "fmi_flux_phase_count", "fmi_flux_phase_name", "fmi_flux_phase_ms",
    "fmi_vae_default_config", "fmi_vae_create", "fmi_vae_destroy", "fmi_vae_set_tensor", "fmi_vae_missing_count", "fmi_vae_missing_name",
    "fmi_vae_scale_factor", "fmi_vae_shift_factor", "fmi_vae_decode", "fmi_vae_encode", "fmi_vae_mid_attention",
    "fmi_t5_default_config", "fmi_t5_create", "fmi_t5_destroy", "fmi_t5_set_tensor", "fmi_t5_set_linear_bnb4", "fmi_t5_set_linear_int8", "fmi_t5_missing_count", "fmi_t5_missing_name",
    "fmi_t5_size_in_bytes", "fmi_t5_forward", "fmi_clip_default_config", "fmi_clip_create", "fmi_clip_destroy", "fmi_clip_set_tensor",
    "fmi_clip_missing_count", "fmi_clip_missing_name", "fmi_clip_size_in_bytes", "fmi_clip_forward", "fmi_pack_latents", "fmi_unpack_latents", "fmi_postprocess_u8",
    "fmi_randn", "fmi_philox_u32", "fmi_calculate_shift", "fmi_get_timesteps", "fmi_linear_bf16", "fmi_linear_bnb4_bf16", "fmi_linear_int8_bf16", "fmi_quantize_rows_fp8", "fmi_linear_fp8", "fmi_quantize_rows_i8", "fmi_linear_i8", "fmi_gemm_q8", "fmi_quantize_rows_i8_asym", "fmi_rowsum_i8", "fmi_quantize_rows_i8_scaled", "fmi_col_absmax", "fmi_gemm_i8_asym", "fmi_linear_q8_workspace_bytes", "fmi_linear_fp8_ws", "fmi_linear_i8_ws", "fmi_sdpa_bf16", "fmi_sdpa_fp8qk", "fmi_sdpa_workspace_bytes", "fmi_sdpa_bf16_ws", "fmi_sdpa_fp8qk_ws", "fmi_sdpa_fp8", "fmi_sdpa_fp8_ws", "fmi_set_attention_kernel", "fmi_layernorm_mod",
    "fmi_release_scratch", "fmi_timestep_embedding", "fmi_rope_table", "fmi_rmsnorm_rope",
    "fmi_groupnorm_nhwc", "fmi_conv2d_nhwc",
    "fmi_comm_probe", "fmi_comm_unique_id", "fmi_comm_create", "fmi_comm_destroy", "fmi_comm_rank", "fmi_comm_world_size", "fmi_comm_stats", "fmi_comm_all_to_all",
    "fmi_comm_broadcast", "fmi_comm_gather",
    "dequantize_blockwise_f32_int8", "dequantize_blockwise_f32_fp4", "dequantize_blockwise_f32_nf4", "dequantize_blockwise_f16_int8",
    "dequantize_blockwise_f16_fp4", "dequantize_blockwise_f16_nf4", "dequantize_blockwise_bf16_int8", "dequantize_blockwise_bf16_fp4",
    "dequantize_blockwise_bf16_nf4", "dequantize_8bit_kernel_f32", "dequantize_8bit_kernel_f16", "dequantize_8bit_kernel_bf16",
    "fmi_malloc", "fmi_free", "fmi_memcpy", "fmi_memset", "fmi_stream_synchronize", "fmi_event_create", "fmi_event_record",
    "fmi_event_elapsed_ms", "fmi_event_destroy",
]
