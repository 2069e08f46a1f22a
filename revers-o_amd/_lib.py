"""ctypes binding of librevo.so (C ABI: include/revo.h).

The HIP library is the product: there is no CPU fallback.  If the shared object
is missing or a call fails, this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librevo.so")


class RevoError(RuntimeError):
    pass


class VitCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "image_size", "patch_size", "width", "layers", "heads", "mlp_dim", "out_dim", "pool_heads",
        "use_cls", "use_ls")] + [("ln_eps", C.c_float), ("rope_theta", C.c_float), ("pool_mlp_dim", C.c_int32)]


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64)]


class CropJob(C.Structure):
    """revo_crop_job of include/revo.h."""
    _fields_ = [("src", C.c_void_p), ("height", C.c_int32), ("width", C.c_int32), ("row_stride", C.c_int64),
                ("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32)]


_p, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> (restype, argtypes); mirrors include/revo.h one to one
SIGNATURES = {
    "revo_last_error": (C.c_char_p, []),
    "revo_version": (_i32, []),
    "revo_sync": (_i32, [_p]),
    "revo_vit_create": (_i32, [C.POINTER(VitCfg), C.POINTER(Tensor), _i32, _i32, _i32, C.POINTER(_p)]),
    "revo_vit_destroy": (_i32, [_p]),
    "revo_vit_forward": (_i32, [_p, _p, _i32, _i32, _p, _i32, _p]),
    "revo_vit_seq_len": (_i32, [_p]),
    "revo_vit_stats": (_i32, [_p, C.POINTER(C.c_double), _i32, _p]),
    "revo_gallery_create": (_i32, [_i32, _i64, _i32, _i32, C.POINTER(_p)]),
    "revo_gallery_destroy": (_i32, [_p]),
    "revo_gallery_append": (_i32, [_p, _p, _i64, _i32, _i32, _p]),
    "revo_gallery_size": (_i64, [_p]),
    "revo_gallery_clear": (_i32, [_p]),
    "revo_gallery_read": (_i32, [_p, _i64, _i64, _p, _i32]),
    "revo_search_topk": (_i32, [_p, _p, _i32, _i32, _i32, _f32, _i64, _p, _p, _p, _p]),
    "revo_search_ksel": (_i32, [_i32]),
    "revo_search_set_total_rows": (_i32, [_p, _i64]),
    "revo_search_estimates": (_i32, [_i32]),
    "revo_search_plan": (_i32, [_p, _i32, _i32, C.POINTER(C.c_int64)]),
    "revo_search_candidates": (_i32, [_p, _p, _i32, _i32, _i32, _p, _p]),
    "revo_search_finish": (_i32, [_p, _i32, _i32, _i32, _f32, _i64, _p, _i32, _i32, _p, _p, _p, _p, _p]),
    "revo_search_exact": (_i32, [_p, _i32, _p, _p, _i32, _i32, _f32, _i64, _p, _p, _p, _p]),
    "revo_search_stats": (_i32, [_p, C.POINTER(C.c_int32), _p]),
    "revo_topk_merge": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _f32, _p, _p, _p, _p]),
    "revo_topk_packed_bytes": (_i64, [_i32, _i32]),
    "revo_topk_merge_packed": (_i32, [_p, _i32, _i32, _i32, _i32, _f32, _p, _p, _p, _p, _p, _p, _p]),
    "revo_op_gemm": (_i32, [_i32, _p, _i64, _p, _i64, _i32, _i32, _i32, _p, _i64, _p, _p, _p]),
    "revo_op_gemm_resid_ln": (_i32, [_p, _i64, _p, _i64, _i32, _i32, _i32, _p, _i64, _p, _p, _p, _i64, _p, C.POINTER(C.c_int32), _p, _i32,
                                     _i32, _p]),
    "revo_op_gemm_ln_in": (_i32, [_i32, _p, _i64, _p, _i64, _i32, _i32, _i32, _p, _i64, _p, _p, _p, _i32, _f32, _p, _p]),
    "revo_op_gemm_rope": (_i32, [_p, _i64, _p, _i64, _i32, _i32, _i32, _p, _i64, _p, _p, _i32, _i32, _i32, _p]),
    "revo_op_layernorm": (_i32, [_p, _i64, _p, _p, _f32, _i32, _i32, _p, _i64, _i32, _p]),
    "revo_op_layernorm_logits": (_i32, [_p, _i64, _p, _p, _f32, _i32, _i32, _p, _i64, _p, _p, _i32, _i32, _p, _p]),
    "revo_op_linear_f32": (_i32, [_i32, _p, _i64, _p, _i64, _p, _i32, _i32, _i32, _p, _i64, _p]),
    "revo_op_pool_rows": (_i32, [_p, _i64, _p, _i32, _i32, _i32, _i32, _p, _p]),
    "revo_op_rope": (_i32, [_p, _i64, _p, _i32, _i32, _i32, _i32, _p]),
    "revo_op_attention": (_i32, [_p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "revo_op_f32_to_bf16": (_i32, [_p, _i64, _p, _i64, _i64, _i32, _p]),
    "revo_prof_enable": (_i32, [_i32]),
    "revo_prof_reset": (_i32, []),
    "revo_prof_report": (_i32, [C.c_char_p, _i32]),
    "revo_preprocess_crop_resize": (_i32, [C.POINTER(CropJob), _i32, _i32, _p, _p]),
    "revo_probe_mfma": (_i32, [_p, _i64, _p, _i32, _i32, _p]),
    "revo_probe_mfma_flops": (_i64, [_i32, _i32]),
    "revo_probe_copy": (_i32, [_p, _p, _i64, _p]),
}

# only in librevo_exp.so (built by `make exp` with -DREVO_EXPERIMENTS; timing scripts under scripts/)
EXPERIMENT_SIGNATURES = {
    "revo_vit_set_debug_layers": (_i32, [_p, _i32]),
    "revo_vit_read_residual": (_i32, [_p, _i32, _p, _p]),
    "revo_vit_read_tap": (_i32, [_p, _i32, _i32, _p, _p]),
    "revo_search_set_mode": (_i32, [_p, _i32]),
    "revo_op_set_gemm_tile": (_i32, [_i32]),
    "revo_op_set_phase_groups": (_i32, [_i32]),
    "revo_op_set_qstores": (_i32, [_i32]),
    "revo_debug_gemm_stamps": (_i32, [_p, _i32]),
    "revo_debug_attention_clock": (_i32, [_p]),
    "revo_op_set_variant": (_i32, [_i32]),
    "revo_op_set_ln_fold": (_i32, [_i32]),
    "revo_op_set_gemm_debug": (_i32, [_i32]),
    "revo_debug_scan_stats": (_i32, [C.POINTER(C.c_int64)]),
    "revo_debug_seed_bounds": (_i32, [_p, _p]),
    "revo_debug_read_workspace": (_i64, [_p, _i64, _i64, _p]),
    "revo_debug_stream_in_planes": (_i32, [_p, _i32]),
    "revo_probe_gridbar": (_i32, [_i32, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p]),
    "revo_debug_scan_plan": (_i64, [_i32, _i64, C.POINTER(C.c_int64), _i32]),
}
BASE_SIGNATURES = dict(SIGNATURES)
if os.environ.get("REVO_LIBRARY_PATH"):            # bisecting / A-B runs of another build of the same ABI
    LIB_PATH = os.environ["REVO_LIBRARY_PATH"]
elif os.environ.get("REVO_EXPERIMENTS") == "1":
    LIB_PATH = os.path.join(_HERE, "librevo_exp.so")
if os.environ.get("REVO_EXPERIMENTS") == "1":
    SIGNATURES = dict(SIGNATURES, **EXPERIMENT_SIGNATURES)

_lib = None
_lib_exp = None


def product_is_experiment_build():
    """True when load() itself returns an experiment build (REVO_EXPERIMENTS=1: scripts/)."""
    return os.environ.get("REVO_EXPERIMENTS") == "1"


def load_exp():
    """librevo_exp.so next to the product library: the same sources built with -DREVO_EXPERIMENTS, i.e. the whole ABI plus
    the kernel-variant and timing switches (tests of forced GEMM tiles, scripts/).  A separate library with its own
    state: handles must stay with the library that created them."""
    global _lib_exp
    if _lib_exp is not None:
        return _lib_exp
    import torch  # noqa: F401
    path = os.path.join(_HERE, "librevo_exp.so")
    if not os.path.exists(path):
        raise RevoError(f"{path} not found: build it with `make -C revers-o_amd/csrc exp`")
    lib = C.CDLL(path)
    for name, (res, args) in dict(BASE_SIGNATURES, **EXPERIMENT_SIGNATURES).items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib_exp = lib
    return lib


def load():
    """Load librevo.so and attach prototypes.  Raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch owns device memory and streams: its bundled HIP runtime must be the one
    # librevo.so binds to (a stream handle is only meaningful inside one runtime), so
    # make sure it is loaded first.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RevoError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C revers-o_amd/csrc`).  There is no CPU fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        except AttributeError:
            if os.environ.get("REVO_LIBRARY_PATH"):     # an older build in an A/B run (scripts/step_regression_ab.sh): it simply lacks the newer entry points
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().revo_last_error().decode("utf-8", "replace")
        if _lib_exp is not None:       # the experiment library keeps its own (thread-local) message
            msg = (msg + " | " if msg else "") + _lib_exp.revo_last_error().decode("utf-8", "replace")
        raise RevoError(f"{what} failed (status {rc}): {msg}")


def ptr(t):
    """Device/host address of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
