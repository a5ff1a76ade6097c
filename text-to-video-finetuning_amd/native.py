"""ctypes binding of the C ABI in include/t2v_abi.h (libt2v_hip.so).

This module is the ONLY place that touches the shared library.  It fails loudly: if the library
is missing or a call returns non-zero a RuntimeError is raised (the reference's train loop
swallows exceptions in backward, train.py:881-883, so errors must not be silent — they are also
printed to stderr).  There is no CPU fallback anywhere in the product path.
"""
import ctypes as C
import os
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("T2V_LIB_FILE") or os.path.join(_HERE, "libt2v_hip.so")      # (override: same-box A/B runs)
TUNE_TABLE = os.environ.get("T2V_GEMM_TABLE_FILE") or os.path.join(_HERE, "gemm_tune_gfx950.txt")   # (override: A/B runs)

c_void_p, c_int, c_ll, c_float, c_ull = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_ulonglong


class ConvGeom(C.Structure):
    _fields_ = [(n, c_int) for n in "C Hv Wv Ho Wo KH KW sy sx py px tdiv up".split()]


class LoraWgrad(C.Structure):
    _fields_ = [
        ("rows", c_ll), ("rp", c_int), ("conv", c_int),
        ("t", c_void_p), ("ldt", c_ll), ("dy", c_void_p), ("lddy", c_ll), ("N", c_int),
        ("dU", c_void_p), ("lddu", c_ll), ("dt", c_void_p), ("lddt", c_ll),
        ("x", c_void_p), ("ldx", c_ll), ("C", c_int), ("dD", c_void_p), ("lddd", c_ll),
        ("geom", ConvGeom), ("alpha", c_float), ("drop_p", c_float), ("drop_seed", c_ull),
    ]


class LoraMergeJob(C.Structure):
    _fields_ = [
        ("w32", c_void_p), ("up", c_void_p), ("ldu", c_ll), ("down", c_void_p),
        ("wf", c_void_p), ("ldwf", c_ll), ("wb", c_void_p), ("ldwb", c_ll),
        ("Np", c_int), ("Cp", c_int), ("taps", c_int), ("rp", c_int), ("scale", c_float), ("tile0", c_int),
    ]


class LoraPrepJob(C.Structure):
    _fields_ = [
        ("up", c_void_p), ("ldu", c_ll), ("down", c_void_p), ("upT", c_void_p), ("dnT", c_void_p),
        ("Np", c_int), ("Cp", c_int), ("taps", c_int), ("rp", c_int), ("rk", c_int), ("scale", c_float), ("chunk0", c_ll),
        ("ldt", c_ll), ("rkd", c_int),
    ]


class Gemm(C.Structure):
    _fields_ = [
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("A", c_void_p), ("lda", c_ll), ("a_mode", c_int), ("a_trans", c_int),
        ("B", c_void_p), ("ldb", c_ll), ("b_trans", c_int), ("b_conv", c_int),
        ("geom", ConvGeom),
        ("D", c_void_p), ("ldd", c_ll), ("out_mode", c_int),
        ("bias", c_void_p),
        ("rowbias", c_void_p), ("ldrb", c_ll), ("rows_per_rb", c_int),
        ("R", c_void_p), ("ldr", c_ll),
        ("alpha", c_float), ("beta", c_float),
        ("act", c_int),
        ("batch", c_int), ("strideA", c_ll), ("strideB", c_ll), ("strideD", c_ll), ("strideR", c_ll),
        ("split_k", c_int),
        ("drop_p", c_float), ("drop_seed", c_ull),
        ("B2", c_void_p), ("ldb2", c_ll), ("n_split", c_int), ("D2", c_void_p), ("ldd2", c_ll),
        ("b_tapflip", c_int), ("b2_k0", c_int), ("b2_klen", c_int),
        ("workspace", c_void_p), ("workspace_bytes", C.c_size_t), ("ws_split", c_int), ("raster_n", c_int),
        ("drop_epoch", c_void_p),
        ("colsum", c_void_p), ("cs_mode", c_int), ("cs_domain_rows", c_int),
        ("cs_x", c_void_p), ("cs_ldx", c_ll), ("cs_sums", c_void_p), ("cs_gamma", c_void_p), ("cs_beta", c_void_p),
        ("cs_eps", c_float), ("cs_G", c_int), ("cs_silu", c_int),
        ("lr_mode", c_int), ("lr_rp", c_int), ("lr_taps", c_int),
        ("lr_a", c_void_p), ("lr_lda", c_ll), ("lr_b", c_void_p), ("lr_ldb", c_ll),
        ("lr_scale", c_float), ("lr_drop_p", c_float), ("lr_drop_seed", c_ull),
        ("lr_group_cols", c_int), ("lr_group_seed", c_ull * 2),
        ("cs_drop_p", c_float), ("cs_drop_seed", c_ull),
        ("lr_plane", c_void_p),
    ]


class SmallConv(C.Structure):
    _fields_ = [
        ("x", c_void_p), ("ldx", c_ll), ("x_nchw_f32", c_int),
        ("w", c_void_p), ("bias", c_void_p),
        ("y", c_void_p), ("ldy", c_ll), ("y_nchw_f32", c_int),
        ("nimg", c_int), ("Cin", c_int), ("Cout", c_int), ("geom", ConvGeom),
    ]


class AttnOperand(C.Structure):
    _fields_ = [("ptr", c_void_p), ("bstride_hi", c_ll), ("bstride_lo", c_ll), ("sstride", c_ll), ("bdiv", c_int)]


class Attn(C.Structure):
    _fields_ = [
        ("nbatch", c_int), ("heads", c_int), ("Sq", c_int), ("Sk", c_int), ("scale", c_float),
        ("q", AttnOperand), ("k", AttnOperand), ("v", AttnOperand), ("o", AttnOperand),
        ("lse", c_void_p),
        ("d_o", AttnOperand), ("dq", AttnOperand), ("dk", AttnOperand), ("dv", AttnOperand),
        ("delta", c_void_p), ("causal", c_int),
    ]


class TemporalFused(C.Structure):
    _fields_ = [
        ("x", c_void_p), ("ldx", c_ll), ("out", c_void_p), ("ldo", c_ll), ("wqkv", c_void_p), ("wo", c_void_p),
        ("bo", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("eps", c_float), ("scale", c_float),
        ("B", c_int), ("F", c_int), ("HW", c_int), ("C", c_int), ("ablate", c_int),
    ]


ABI_VERSION = 9          # include/t2v_abi.h T2V_ABI_VERSION
A_DENSE, A_CONV = 0, 1
OUT_BF16, OUT_F32, OUT_F32_ATOMIC = 0, 1, 2
ACT_NONE, ACT_SILU = 0, 1

# every symbol include/t2v_abi.h declares (checked by tests/test_abi.py)
SYMBOLS = {
    "t2v_abi_version": ([], c_int),
    "t2v_last_error": ([], C.c_char_p),
    "t2v_gemm": ([C.POINTER(Gemm), c_void_p], c_int),
    "t2v_temporal_fused_fwd": ([C.POINTER(TemporalFused), c_void_p], c_int),
    "t2v_temporal_fused_ok": ([c_int, c_int], c_int),
    "t2v_gemm_lr_ok": ([C.POINTER(Gemm)], c_int),
    "t2v_gemm_colsum_rows": ([C.POINTER(Gemm)], c_int),
    "t2v_gn_finish": ([c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p], c_int),
    "t2v_gemm_w8": ([C.POINTER(Gemm), c_int, c_int, c_int, c_void_p], c_int),
    "t2v_gemm_w8_configs": ([], c_int),
    "t2v_set_dropout_epoch": ([c_void_p], c_int),
    "t2v_gemm_tune_export": ([C.c_char_p, c_ll], c_ll),
    "t2v_gemm_tune_import": ([C.c_char_p], c_int),
    "t2v_gemm_pair": ([C.POINTER(Gemm), C.POINTER(Gemm), c_void_p], c_int),
    "t2v_launch_timing_events": ([c_void_p, c_void_p], c_int),
    "t2v_launch_timing_consumed": ([], c_int),
    "t2v_smallconv": ([C.POINTER(SmallConv), c_void_p], c_int),
    "t2v_gn_workspace_floats": ([c_int, c_int], c_ll),
    "t2v_gn_stats": ([c_void_p, c_ll, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p], c_int),
    "t2v_gn_apply": ([c_void_p, c_ll, c_void_p, c_ll, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float,
                      c_int, c_float, c_ull, c_void_p], c_int),
    "t2v_gn_bwd_stats": ([c_void_p, c_ll, c_void_p, c_ll, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                          c_float, c_int, c_float, c_ull, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    "t2v_gn_bwd_pg_floats": ([c_int, c_int, c_int], c_ll),
    "t2v_layernorm_bwd_pg_floats": ([c_int, c_int], c_ll),
    "t2v_gn_bwd_apply": ([c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                          c_void_p, c_void_p, c_float, c_int, c_float, c_ull, c_void_p, c_ll, c_void_p], c_int),
    "t2v_layernorm_fwd": ([c_void_p, c_ll, c_void_p, c_ll, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p],
                          c_int),
    "t2v_layernorm_bwd": ([c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_int, c_int, c_void_p, c_void_p, c_void_p,
                           c_void_p, c_void_p, c_void_p, c_ll, c_void_p], c_int),
    "t2v_gelu_fwd": ([c_void_p, c_void_p, c_ll, c_int, c_void_p], c_int),
    "t2v_gelu_bwd": ([c_void_p, c_void_p, c_void_p, c_ll, c_int, c_void_p], c_int),
    "t2v_rowgroup_sum": ([c_void_p, c_ll, c_void_p, c_ll, c_int, c_int, c_int, c_void_p, c_void_p], c_int),
    "t2v_rowgroup_splits": ([c_int, c_int, c_int], c_int),
    "t2v_attn_fwd": ([C.POINTER(Attn), c_void_p], c_int),
    "t2v_attn_bwd": ([C.POINTER(Attn), c_void_p], c_int),
    "t2v_softmax_rows": ([c_void_p, c_ll, c_void_p, c_ll, c_ll, c_int, c_void_p], c_int),
    "t2v_dropout_mask": ([c_void_p, c_ll, c_void_p, c_ll, c_ll, c_int, c_float, c_ull, c_void_p], c_int),
    "t2v_lowrank_update": ([c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_ll, c_int, c_int, c_float, c_void_p], c_int),
    "t2v_lowrank_update_drop": ([c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_ll, c_int, c_int, c_float, c_float, c_ull,
                                 c_void_p], c_int),
    "t2v_lora_prep_chunks": ([c_int, c_int, c_int, c_int, c_int], c_ll),
    "t2v_lora_prep": ([c_void_p, c_int, c_ll, c_void_p], c_int),
    "t2v_lora_wgrad": ([C.POINTER(LoraWgrad), c_void_p], c_int),
    "t2v_lora_drop_dt": ([c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_ll, c_int, c_int, c_float, c_ull, c_void_p], c_int),
    "t2v_lora_drop_dt_group": ([c_void_p, c_ll, c_void_p, c_ll, c_ll, c_void_p, c_ll, c_ll, c_int, c_int, c_int, c_float,
                                C.POINTER(c_ull), c_void_p], c_int),
    "t2v_lora_wgrad_batch_bytes": ([c_int], c_ll),
    "t2v_lora_wgrad_batch": ([C.POINTER(LoraWgrad), c_int, c_void_p, c_void_p, c_ll, c_void_p], c_int),
    "t2v_lowrank_window_update": ([c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, C.POINTER(ConvGeom), c_ll, c_int, c_int,
                                   c_float, c_void_p], c_int),
    "t2v_lora_merge_plan": ([C.POINTER(LoraMergeJob), c_int, c_void_p, c_ll], c_ll),
    "t2v_lora_merge": ([c_void_p, c_int, c_void_p, c_ll, c_void_p], c_int),
    "t2v_geglu_fwd": ([c_void_p, c_ll, c_void_p, c_ll, c_int, c_int, c_void_p], c_int),
    "t2v_geglu_bwd": ([c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_ll, c_int, c_int, c_void_p], c_int),
    "t2v_silu_fwd": ([c_void_p, c_void_p, c_ll, c_void_p], c_int),
    "t2v_silu_bwd": ([c_void_p, c_void_p, c_void_p, c_ll, c_void_p], c_int),
    "t2v_copy2d": ([c_void_p, c_ll, c_void_p, c_ll, c_int, c_int, c_int, c_void_p], c_int),
    "t2v_pool2x2_sum": ([c_void_p, c_ll, c_void_p, c_ll, c_int, c_int, c_int, c_int, c_void_p], c_int),
    "t2v_f32_planar_to_bf16_cl": ([c_void_p, c_void_p, c_ll, c_int, c_int, c_ll, c_void_p], c_int),
    "t2v_bf16_cl_to_f32_planar": ([c_void_p, c_ll, c_void_p, c_int, c_int, c_ll, c_void_p], c_int),
    "t2v_cast_f32_to_bf16": ([c_void_p, c_void_p, c_ll, c_void_p], c_int),
    "t2v_cast_bf16_to_f32": ([c_void_p, c_void_p, c_ll, c_int, c_void_p], c_int),
    "t2v_mse_fwd_bwd": ([c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_float, c_void_p], c_int),
    "t2v_sumsq": ([c_void_p, c_ll, c_void_p, c_void_p, c_void_p], c_int),
    "t2v_adamw": ([c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_float, c_float, c_float, c_float, c_float, c_void_p,
                   c_float, c_float, c_void_p, c_void_p], c_int),
}

_lib = None


def lib():
    """Load libt2v_hip.so (once).  Raises if it has not been built — no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"t2v_amd: native library {LIB_PATH} not found. Build it with "
                f"`python __graft_entry__.py` (or `python text-to-video-finetuning_amd/build_ext.py`). "
                f"There is no CPU/eager fallback for the device path.")
        l = C.CDLL(LIB_PATH)
        for name, (argtypes, restype) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = restype
        if l.t2v_abi_version() != ABI_VERSION:
            raise RuntimeError(f"t2v_amd: ABI version mismatch (library {l.t2v_abi_version()}, binding {ABI_VERSION}): rebuild "
                               f"with `python __graft_entry__.py`")
        _lib = l
        if os.path.exists(TUNE_TABLE) and os.environ.get("T2V_GEMM_TABLE", "1") != "0":
            with open(TUNE_TABLE, "rb") as f:       # shipped tile table: t2v_gemm never times candidates at run time
                l.t2v_gemm_tune_import(f.read())
    return _lib


def export_tune_table(path=None):
    """Write the library's current tile table (shipped entries + whatever a T2V_GEMM_AUTOTUNE=live run added)."""
    l = lib()
    need = l.t2v_gemm_tune_export(None, 0)
    buf = C.create_string_buffer(int(need))
    l.t2v_gemm_tune_export(buf, need)
    lines = sorted(set(buf.value.decode().splitlines()), key=lambda s: [int(x) for x in s.split()])
    with open(path or TUNE_TABLE, "w") as f:
        f.write("\n".join(lines) + "\n")
    return len(lines)


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(rc, what):
    if rc != 0:
        msg = lib().t2v_last_error().decode("utf-8", "replace")
        text = f"t2v_amd native call {what} failed (rc={rc}): {msg}"
        print(text, file=sys.stderr, flush=True)
        raise RuntimeError(text)


def call(name, *args):
    check(getattr(lib(), name)(*args), name)


def ptr(t):
    return None if t is None else t.data_ptr()


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("t2v_amd: the native path needs tensors on a ROCm device (cuda:N); got a CPU tensor. "
                               "There is no CPU fallback in the product path (the CPU oracle lives in oracle/).")
