"""ctypes binding of libsimvg_hip.so (the C-ABI declared in include/simvg_hip.h).

Fails loudly when the library is missing: the product path has NO CPU / PyTorch fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SIMVG_HIP_LIB") or os.path.join(_HERE, "lib", "libsimvg_hip.so")   # override: dev builds

c_void_p, c_int, c_long, c_float = C.c_void_p, C.c_int, C.c_long, C.c_float


class WeightDesc(C.Structure):
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("dst_t", c_void_p), ("rows", c_int), ("cols", c_int),
                ("tile_start", c_int), ("split_shift", c_int)]


class LnReduceDesc(C.Structure):        # simvg_ln_reduce_desc
    _fields_ = [("partial", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p), ("group_stride", c_int), ("D", c_int),
                ("blocks0", c_int), ("blocks1", c_int)]


class WgradReduceDesc(C.Structure):     # simvg_wgrad_reduce_desc
    _fields_ = [("slabs", c_void_p), ("dW", c_void_p), ("dw_group_stride", c_long), ("lddw", c_int), ("N", c_int), ("K", c_int),
                ("Q", c_int), ("lo0", c_int), ("hi0", c_int), ("lo1", c_int), ("hi1", c_int), ("assign", c_int)]


class GemmF32Problem(C.Structure):      # simvg_gemm_f32_problem
    _fields_ = [("A", c_void_p), ("sam", c_long), ("sak", c_long), ("B", c_void_p), ("sbk", c_long), ("sbn", c_long),
                ("C", c_void_p), ("ldc", c_long), ("bias", c_void_p), ("addend", c_void_p), ("ld_addend", c_long),
                ("addend_rows", c_int), ("M", c_int), ("N", c_int), ("K", c_int), ("accumulate", c_int), ("act", c_int),
                ("A2", c_void_p), ("B2", c_void_p), ("mult", c_void_p), ("ld_mult", c_long), ("gate", c_void_p),
                ("ld_gate", c_long)]


class DecAttnArgs(C.Structure):         # simvg_dec_attn_args
    _fields_ = ([(n, c_int) for n in ("B", "R", "Lk", "kv_rows", "kv_off")] +
                [(n, c_void_p) for n in ("tgt", "qpos", "Ws", "bs", "Wso", "bso", "g0", "b0", "Wc", "bc", "Wco", "bco", "g1", "b1",
                                         "src16", "src32")] +
                [("ldsrc", c_long), ("kpos", c_void_p), ("ldkp", c_long), ("kpos_rows", c_int), ("kpm", c_void_p), ("dm0", c_void_p),
                 ("dm1", c_void_p)] +
                [(n, c_void_p) for n in ("qkv", "P0", "o", "r1", "mean1", "rstd1", "t1", "qc", "qk", "P1", "ctx", "sp", "o2", "r2",
                                         "mean2", "rstd2", "t2")] +
                [("eps", c_float)])


class DecAttnBwdArgs(C.Structure):      # simvg_dec_attn_bwd_args
    _fields_ = ([(n, c_int) for n in ("B", "R", "Lk", "kv_rows", "kv_off")] +
                [(n, c_void_p) for n in ("Ws", "Wso", "g0", "Wc", "bc", "Wco", "g1", "src16", "src32")] +
                [("ldsrc", c_long), ("kpos", c_void_p), ("ldkp", c_long), ("kpos_rows", c_int), ("dm0", c_void_p), ("dm1", c_void_p)] +
                [(n, c_void_p) for n in ("qkv", "P0", "r1", "mean1", "rstd1", "qk", "P1", "r2", "mean2", "rstd2", "dt2", "dt2_slabs")] +
                [("nslab", c_int), ("slab_stride", c_long), ("d_tgt", c_void_p), ("d_qpos", c_void_p), ("dsrc", c_void_p),
                 ("lddsrc", c_long), ("dsrc_accumulate", c_int)] +
                [(n, c_void_p) for n in ("dt2sum", "gx2", "d_r2", "d_o2", "dctx", "dqk", "dqpre", "d_t1", "gx1", "d_r1", "dqkv")])


class DecAttnWgradArgs(C.Structure):    # simvg_dec_attn_wgrad_args
    _fields_ = ([("MR", c_int)] +
                [(n, c_void_p) for n in ("tgt", "qpos", "t1", "o", "o2", "ctx", "sp", "qc", "dqkv", "d_r1", "gx1", "d_t1", "dqpre",
                                         "dqk", "d_o2", "d_r2", "gx2", "dt2sum", "dWs", "dbs", "dWso", "dbso", "dg0", "db0", "dWc",
                                         "dbc", "dWco", "dbco", "dg1", "db1")])


class DecFfnArgs(C.Structure):          # simvg_dec_ffn_args
    _fields_ = [("M", c_int), ("Fd", c_int)] + [(n, c_void_p) for n in ("t2", "W1", "b1", "W2", "m1", "h1d", "slabs")]


class DecFfnFinishArgs(C.Structure):    # simvg_dec_ffn_finish_args
    _fields_ = ([("M", c_int), ("NS", c_int)] +
                [(n, c_void_p) for n in ("t2", "slabs", "b2", "m2", "g2", "b2n", "gP", "bP", "r3", "mean3", "rstd3", "t3", "hs", "meanP",
                                         "rstdP")] + [("eps", c_float)])


class DecFfnBwdArgs(C.Structure):       # simvg_dec_ffn_bwd_args
    _fields_ = ([("M", c_int), ("Fd", c_int)] +
                [(n, c_void_p) for n in ("d_t3", "d_hs", "r3", "mean3", "rstd3", "g2", "t3", "meanP", "rstdP", "gP", "m2", "W1", "W2", "h1d",
                                         "m1", "t2", "d_r3", "gx3", "dy3", "gxP", "dr3m", "slabs", "dW1", "db1", "dW2", "db2", "dg2",
                                         "db2n", "dgP", "dbP")])


_SIGS = {
    "simvg_text_filt_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "simvg_text_filt_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "simvg_query_mix_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "simvg_query_mix_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "simvg_dec_ffn_fwd": [c_void_p, c_void_p],
    "simvg_dec_ffn_finish": [c_void_p, c_void_p],
    "simvg_dec_ffn_bwd": [c_void_p, c_void_p],
    "simvg_dec_attn_fwd": [c_void_p, c_void_p],
    "simvg_dec_attn_bwd": [c_void_p, c_void_p],
    "simvg_dec_attn_wgrad": [c_void_p, c_void_p],
    "simvg_gemm_f32_grouped": [c_void_p, c_int, c_void_p],
    "simvg_gemm_f32_grouped_ws": [c_void_p, c_int, c_void_p, c_long, c_void_p],
    "simvg_postprocess": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                          c_int, c_int, c_int, c_void_p],
    "simvg_gemm_nt": [c_void_p, c_int, c_void_p, c_long, c_int, c_void_p, c_int, c_void_p, c_int, c_int,
                      c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                      c_float, c_void_p],
    "simvg_gemm_nt_split": [c_void_p, c_int, c_void_p, c_long, c_int, c_void_p, c_int, c_void_p, c_int, c_int,
                            c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p],
    "simvg_gemm_nt_plan": [c_int] * 10,
    "simvg_gemm_tn": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_long, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                      c_float, c_void_p],
    "simvg_gemm_tn_ws": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_long, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                         c_float, c_void_p, c_void_p, c_void_p],
    "simvg_wgrad_reduce_batched": [c_void_p, c_int, c_void_p],
    "simvg_colsum": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "simvg_ln_fwd": [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                     c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p],
    "simvg_ln_bwd": [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                     c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                     c_int, c_int, c_int, c_int, c_int, c_void_p, c_float, c_float, c_void_p],
    "simvg_ln_bwd_deferred": [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                              c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                              c_int, c_int, c_int, c_int, c_int, c_void_p, c_float, c_float, c_void_p, c_void_p],
    "simvg_ln_param_reduce_batched": [c_void_p, c_int, c_void_p],
    "simvg_attn_fwd": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                       c_float, c_void_p],
    "simvg_attn_qk_probe": [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p],
    "simvg_attn_bwd": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                       c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p],
    "simvg_im2col": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "simvg_embed_fwd": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                        c_int, c_int, c_int, c_int, c_void_p],
    "simvg_embed_bwd": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                        c_int, c_int, c_int, c_int, c_float, c_void_p],
    "simvg_weight_prep": [c_void_p, c_int, c_int, c_void_p],
    "simvg_cast_f32_to_lp": [c_void_p, c_void_p, c_long, c_float, c_void_p],
    "simvg_cast_lp_to_f32": [c_void_p, c_void_p, c_long, c_void_p],
    "simvg_resize_u8": [c_void_p, c_int, c_int, c_long, c_void_p, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "simvg_normalize_pad_u8": [c_void_p, c_long, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "simvg_resize_u8_batched": [c_void_p, c_int, c_void_p],
    "simvg_normalize_pad_u8_batched": [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "simvg_attn_f32_bwd": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                           c_float, c_void_p],
    "simvg_gelu_f32": [c_void_p, c_void_p, c_void_p, c_long, c_void_p],
    "simvg_sumsq": [c_void_p, c_long, c_void_p, c_void_p, c_void_p],
    "simvg_adam_step": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_float, c_float, c_float,
                        c_float, c_void_p, c_float, c_void_p],
    "simvg_gemm_f32": [c_void_p, c_long, c_long, c_void_p, c_long, c_long, c_void_p, c_long, c_void_p, c_void_p, c_long,
                       c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "simvg_attn_small_fwd": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                             c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_int, c_void_p],
    "simvg_attn_small_bwd": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                             c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_int, c_void_p],
    "simvg_match": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                    c_float, c_float, c_float, c_void_p],
    "simvg_soft_targets": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                           c_void_p, c_int, c_int, c_int, c_void_p],
    "simvg_criterion": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                        c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float,
                        c_void_p],
    "simvg_im2col_f32": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "simvg_attn_f32_fwd": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p],
    "simvg_dropout_mult": [c_void_p, c_long, c_float, c_void_p, c_long, C.c_ulonglong, C.c_ulonglong, c_void_p, c_void_p],
    "simvg_philox4x32": [c_void_p, c_void_p, c_void_p],
    "simvg_pack_targets": [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "simvg_probe_mfma": [c_void_p, c_void_p, c_void_p, c_void_p],
    "simvg_probe_tr16": [c_void_p, c_void_p, c_void_p],
    "simvg_probe_glds": [c_void_p, c_void_p, c_void_p, c_void_p],
}

_lib = None


class SimvgHipError(RuntimeError):
    pass


def _check_source_hash(lib):
    """A stale library (sources edited after the build, an old .so that travelled with a snapshot) must not run silently:
    the embedded hash has to equal the hash of csrc/ next to this file.  Development variants (SIMVG_HIP_LIB) are built with
    extra flags that are part of their hash and are not re-derived here; SIMVG_SKIP_SOURCE_CHECK=1 disables the check."""
    if os.environ.get("SIMVG_HIP_LIB") or os.environ.get("SIMVG_SKIP_SOURCE_CHECK") == "1":
        return
    if not os.path.isdir(os.path.join(_HERE, "csrc")):
        return                                              # a binary-only installation has nothing to compare with
    from .build import source_hash
    built, now = lib.simvg_source_hash().decode(), source_hash()
    if built != now:
        raise SimvgHipError(f"{LIB_PATH} was built from other sources (library {built}, csrc/ now {now}): "
                            "run `python -m simvg_amd.build`")


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SimvgHipError(
            f"{LIB_PATH} not found: build it with `python -m simvg_amd.build` (hipcc --offload-arch=gfx950). "
            "simvg_amd has no CPU or eager-PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)   # AttributeError here == header / library mismatch
        fn.argtypes = args
        fn.restype = c_int
    lib.simvg_last_error.restype = C.c_char_p
    lib.simvg_last_error.argtypes = []
    lib.simvg_version.restype = c_int
    lib.simvg_source_hash.restype = C.c_char_p
    lib.simvg_source_hash.argtypes = []
    _check_source_hash(lib)
    lib.simvg_lowp_format.restype = c_int
    lib.simvg_lowp_format.argtypes = []
    lib.simvg_ln_bwd_ws_floats.restype = c_long
    lib.simvg_ln_bwd_ws_floats.argtypes = [c_int, c_int, c_int]
    lib.simvg_gemm_tn_ws_floats.restype = c_long
    lib.simvg_gemm_tn_ws_floats.argtypes = [c_int, c_int, c_int]
    lib.simvg_dec_attn_max_keys.restype = c_int
    lib.simvg_dec_attn_max_keys.argtypes = []
    _lib = lib
    return lib


def lowp_format():
    """"fp16" or "bf16": the 16-bit operand / storage format the loaded library was built for"""
    return {1: "fp16", 2: "bf16"}[load().simvg_lowp_format()]


def exported_symbols():
    return sorted(_SIGS) + ["simvg_last_error", "simvg_version", "simvg_source_hash", "simvg_lowp_format", "simvg_ln_bwd_ws_floats", "simvg_gemm_tn_ws_floats",
                            "simvg_dec_attn_max_keys"]


def check(rc, what):
    if rc != 0:
        raise SimvgHipError(f"{what} failed (rc={rc}): {load().simvg_last_error().decode()}")
