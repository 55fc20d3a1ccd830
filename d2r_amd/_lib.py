"""ctypes binding of libd2r_hip.so (the C ABI declared in include/d2r_hip.h).

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.  The
product path never routes through PyTorch reference code or the oracle.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libd2r_hip.so")

F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_TANH, ACT_GELU, ACT_QUICK_GELU, ACT_TANH_RELU, ACT_SIGMOID = range(7)
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2

vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t


class GemmDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("c_dtype", i32), ("layout", i32), ("act", i32),
        ("M", i32), ("N", i32), ("K", i32), ("nb", i32), ("nh", i32),
        ("alpha", f32), ("beta", f32),
        ("A", vp), ("lda", i64), ("sAb", i64), ("sAh", i64),
        ("B", vp), ("ldb", i64), ("sBb", i64), ("sBh", i64),
        ("C", vp), ("ldc", i64), ("sCb", i64), ("sCh", i64),
        ("bias", vp),
        ("residual", vp), ("ldr", i64), ("sRb", i64), ("sRh", i64),
        ("preact", vp),
        ("workspace", vp), ("workspace_bytes", sz),
        ("s_bias_b", i64),
        ("dbias", vp),
        ("grad_ref", vp), ("grad_act", i32),
    ]


class EncoderLayerDesc(C.Structure):
    _fields_ = ([("dtype", i32), ("pre_ln", i32), ("act", i32), ("B", i32), ("L", i32), ("E", i32), ("H", i32), ("F", i32),
                 ("eps", f32), ("scale", f32), ("mask", vp)]
                + [(n, vp) for n in ("w_qkv", "w_o", "w_1", "w_2", "b_qkv", "b_o", "b_1", "b_2", "ln1_g", "ln1_b", "ln2_g", "ln2_b",
                                     "gw_qkv", "gw_o", "gw_1", "gw_2", "gb_qkv", "gb_o", "gb_1", "gb_2", "gln1_g", "gln1_b",
                                     "gln2_g", "gln2_b", "x", "y", "qkv", "ctx", "h1", "n1", "f_pre", "f", "h2", "lse",
                                     "mean1", "rstd1", "mean2", "rstd2", "dy", "dx", "scratch")]
                + [("scratch_bytes", sz), ("splitk_ws", vp), ("splitk_bytes", sz), ("wgrad_stream", vp),
                   ("defer_wgrad", i32), ("o_dy", vp * 4), ("defer_ln", i32), ("o_lnws", vp * 2), ("p_attn", f32), ("p_hidden", f32), ("seed_attn", C.c_uint64),
                   ("seed_hidden", C.c_uint64 * 2)])


RL_NAMES = ("R0", "R2", "IMRC_QKV", "IMRC_FC1", "IMRC_FC2", "GLAC_Q", "GLAC_KV", "GLAC_LOC", "GLAC_FC1", "GLAC_TPOOL", "GLAC_IPOOL",
            "GLAC_GLO", "GLAC_FC2", "GLAC_SAFW", "CMRC_Q", "CMRC_KV", "CMRC_SCALE", "CMRC_SHIFT", "CMRC_FC1", "CMRC_FC2", "CRCMC_Q",
            "CRCMC_KV", "CRCMC_MLP1", "CRCMC_MLP2", "CRCMC_FC1", "CRCMC_FC2", "GESC_TPOOL", "GESC_IPOOL", "GESC_MLP0", "GESC_MLP2")
RL = {n: i for i, n in enumerate(RL_NAMES)}  # D2R_RL_* of include/d2r_hip.h


class LinearParams(C.Structure):
    _fields_ = [("w", vp), ("b", vp), ("gw", vp), ("gb", vp)]


class RoutingLayerParams(C.Structure):
    _fields_ = [("lin", LinearParams * len(RL_NAMES)), ("bn_weight", vp), ("bn_bias", vp), ("bn_running_mean", vp),
                ("bn_running_var", vp), ("g_bn_weight", vp), ("g_bn_bias", vp)]


class InteractionDesc(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("Lq", i32), ("Lk", i32), ("ncell", i32), ("nlayer", i32), ("hid_router", i32),
                ("heads_imrc", i32), ("hid_imrc", i32), ("train", i32), ("layers", C.POINTER(RoutingLayerParams)),
                ("own", vp), ("other", vp), ("out", vp), ("paths", vp), ("arena", vp), ("arena_bytes", sz),
                ("splitk_ws", vp), ("splitk_bytes", sz), ("d_out", vp), ("d_paths", vp), ("d_own", vp), ("d_other", vp),
                ("scratch", vp), ("scratch_bytes", sz), ("kv_all", LinearParams),
                ("bn_sync", vp), ("bn_sync_user", vp), ("bn_sync_buf", vp), ("bn_world", i32)]


class HeadDesc(C.Structure):
    _fields_ = [("B", i32), ("E", i32), ("mm", i32), ("chunks", i32), ("rank", i32), ("classes", i32),
                ("lin0", LinearParams), ("lin1", LinearParams), ("merge0", LinearParams), ("merge1", LinearParams),
                ("lin_out", LinearParams), ("fc", LinearParams), ("x0", vp), ("x1", vp), ("labels", vp), ("js", vp),
                ("loss", vp), ("logits", vp), ("pooled", vp), ("arena", vp), ("arena_bytes", sz), ("splitk_ws", vp),
                ("splitk_bytes", sz), ("d_loss", vp), ("d_x0", vp), ("d_x1", vp), ("d_js", vp), ("scratch", vp),
                ("scratch_bytes", sz), ("d_logits", vp), ("d_pooled", vp)]


# name -> (restype, argtypes); every symbol include/d2r_hip.h declares
SIGNATURES = {
    "d2r_head_arena_bytes": (sz, [i32, i32, i32, i32, i32, i32]),
    "d2r_head_bwd_scratch": (sz, [i32, i32, i32, i32, i32, i32]),
    "d2r_head_fwd": (i32, [C.POINTER(HeadDesc), vp]),
    "d2r_head_bwd": (i32, [C.POINTER(HeadDesc), vp]),
    "d2r_version": (C.c_char_p, []),
    "d2r_last_error": (C.c_char_p, []),
    "d2r_gemm": (i32, [C.POINTER(GemmDesc), vp]),
    "d2r_gemm_group": (i32, [C.POINTER(GemmDesc), i32, vp]),
    "d2r_gemm_tn_grouped_v": (i32, [i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64),
                                    C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), f32, vp]),
    "d2r_gemm_tn_grouped": (i32, [i32, i32, i32, i32, i64, i64, i64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                  C.POINTER(vp), i32, f32, vp]),
    "d2r_softmax_fwd": (i32, [i32, i32, vp, vp, i64, i64, i32, f32, vp, i64, vp]),
    "d2r_softmax_bwd": (i32, [i32, i32, vp, vp, vp, i64, i64, i32, f32, vp]),
    "d2r_mha_supported": (i32, [i32, i32, i32, i32]),
    "d2r_mha_fwd": (i32, [i32, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp,
                          i32, i32, i32, i32, i32, f32, f32, C.c_uint64, vp]),
    "d2r_mha_bwd": (i32, [i32, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, vp,
                          vp, i64, i64, vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, i32, f32, f32, C.c_uint64, vp]),
    "d2r_xattn_supported": (i32, [i32, i32, i32, i32]),
    "d2r_xattn_fwd": (i32, [i32, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp,
                            i32, i32, i32, i32, f32, vp]),
    "d2r_xattn_bwd": (i32, [i32, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, vp, i64, i64, vp, vp,
                            i32, i32, i32, i32, i32, f32, vp]),
    "d2r_xattn_fwd_multi": (i32, [i32, i32, C.POINTER(vp), i64, i64, C.POINTER(vp), i64, i64, C.POINTER(vp), i64, i64, C.POINTER(vp), i64, i64,
                                  C.POINTER(vp), i64, i64, vp, C.POINTER(vp), i32, i32, i32, i32, f32, vp]),
    "d2r_xattn_bwd_multi": (i32, [i32, i32, C.POINTER(vp), i64, i64, C.POINTER(vp), i64, i64, C.POINTER(vp), i64, i64, C.POINTER(vp), i64, i64,
                                  C.POINTER(vp), i64, i64, C.POINTER(vp), i64, i64, vp, C.POINTER(vp), C.POINTER(vp), i64, i64, C.POINTER(vp), i64, i64, C.POINTER(vp), i64, i64,
                                  C.POINTER(vp), C.POINTER(vp), i32, i32, i32, i32, i32, f32, vp]),
    "d2r_layernorm_fwd": (i32, [i32, vp, vp, vp, f32, i64, i32, vp, vp, vp, vp]),
    "d2r_layernorm_bwd_workspace": (sz, [i64, i32]),
    "d2r_layernorm_bwd": (i32, [i32, vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, sz, vp]),
    "d2r_layernorm_bwd_sum_grouped": (i32, [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, i64, i32, i32, vp]),
    "d2r_layernorm_bwd_ex": (i32, [i32, vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, i32, vp, sz, vp]),
    "d2r_encoder_layer_bwd_scratch": (sz, [i32, i32, i32, i32]),
    "d2r_encoder_layer_fwd": (i32, [C.POINTER(EncoderLayerDesc), vp]),
    "d2r_encoder_layer_bwd": (i32, [C.POINTER(EncoderLayerDesc), vp]),
    "d2r_interaction_supported": (i32, [i32, i32, i32, i32, i32]),
    "d2r_interaction_arena_bytes": (sz, [i32, i32, i32, i32, i32, i32, i32]),
    "d2r_interaction_bwd_scratch": (sz, [i32, i32, i32, i32, i32, i32, i32]),
    "d2r_interaction_fwd": (i32, [C.POINTER(InteractionDesc), vp]),
    "d2r_interaction_bwd": (i32, [C.POINTER(InteractionDesc), vp]),
    "d2r_l2norm_fwd": (i32, [i32, vp, vp, vp, i64, i32, vp]),
    "d2r_l2norm_bwd": (i32, [i32, vp, vp, vp, vp, i64, i32, vp]),
    "d2r_act_bwd": (i32, [i32, i32, vp, vp, vp, i64, vp]),
    "d2r_act_fwd": (i32, [i32, i32, vp, vp, i64, vp]),
    "d2r_sqdiff_fwd": (i32, [i32, vp, vp, vp, i64, vp]),
    "d2r_sqdiff_bwd": (i32, [i32, vp, vp, vp, vp, vp, i64, vp]),
    "d2r_muladd_fwd": (i32, [i32, vp, vp, vp, vp, i64, vp]),
    "d2r_muladd_bwd": (i32, [i32, vp, vp, vp, vp, vp, i64, vp]),
    "d2r_lerp_fwd": (i32, [i32, vp, vp, vp, vp, i64, vp]),
    "d2r_lerp_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "d2r_add": (i32, [i32, vp, vp, vp, i64, vp]),
    "d2r_add2": (i32, [i32, vp, vp, vp, vp, vp, vp, i64, vp]),
    "d2r_act_bwd2": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, i64, vp]),
    "d2r_dropout": (i32, [i32, vp, vp, vp, i64, f32, C.c_uint64, vp]),
    "d2r_lincomb": (i32, [C.POINTER(vp), C.POINTER(f32), i32, vp, vp]),
    "d2r_axpby": (i32, [i32, f32, vp, f32, vp, i64, vp]),
    "d2r_cast": (i32, [i32, vp, i32, vp, i64, vp]),
    "d2r_colsum_workspace": (sz, [i64, i32]),
    "d2r_colsum": (i32, [i32, vp, i64, i64, i32, vp, vp, sz, vp]),
    "d2r_colsum_add": (i32, [i32, vp, i64, i64, i32, vp, vp, sz, vp]),
    "d2r_meanpool_fwd": (i32, [i32, C.POINTER(vp), i32, i32, i32, i32, vp, vp]),
    "d2r_meanpool_bwd_multi": (i32, [i32, vp, i32, i32, i32, i32, C.POINTER(vp), C.c_uint, vp]),
    "d2r_meanpool_bwd": (i32, [i32, vp, i32, i32, i32, vp, i32, vp]),
    "d2r_route_aggregate_fwd": (i32, [i32, C.POINTER(vp), C.POINTER(vp), vp, i32, i32, i32, i32, i32, C.POINTER(vp), vp, i64, vp]),
    "d2r_route_aggregate_bwd_workspace": (sz, [i32, i32, i32, i32]),
    "d2r_route_aggregate_bwd": (i32, [i32, C.POINTER(vp), C.POINTER(vp), vp, C.POINTER(vp), C.POINTER(vp), vp, i64,
                                      i32, i32, i32, i32, i32, C.POINTER(vp), C.POINTER(vp), vp, vp, sz, vp]),
    "d2r_saf_gate_fwd": (i32, [vp, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp]),
    "d2r_saf_dweights": (i32, [i32, vp, vp, i32, i32, i32, vp, vp]),
    "d2r_saf_dscores": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    "d2r_saf_gate_bwd": (i32, [vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp]),
    "d2r_saf_gate_stats": (i32, [vp, i32, i32, vp, vp]),
    "d2r_saf_gate_fwd_ex": (i32, [vp, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, C.c_double, vp, i32, vp]),
    "d2r_saf_gate_bwd_ex": (i32, [vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, vp, i32, vp, C.c_double, vp, i32, i32, vp]),
    "d2r_jsdiv_fwd": (i32, [vp, vp, i32, vp, vp]),
    "d2r_jsdiv_bwd": (i32, [vp, vp, i32, vp, vp, vp, vp]),
    "d2r_ce_fwd": (i32, [vp, vp, i32, i32, vp, vp]),
    "d2r_ce_bwd": (i32, [vp, vp, i32, i32, vp, vp, vp]),
    "d2r_block_merge_fwd": (i32, [i32, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
    "d2r_block_merge_bwd": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
    "d2r_bert_embed_fwd": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "d2r_bert_embed_bwd": (i32, [i32, vp, vp, vp, i32, i32, i32, i32, i64, vp, vp, vp, vp]),
    "d2r_patchify": (i32, [i32, vp, i32, i32, i32, i32, vp, vp]),
    "d2r_clip_embed_finish": (i32, [i32, vp, vp, vp, i32, i32, i32, vp]),
    "d2r_clip_embed_bwd": (i32, [i32, vp, i32, i32, i32, vp, vp, vp]),
    "d2r_adamw_step": (i32, [vp, vp, vp, vp, vp, i32, i64, f32, f32, f32, f32, f32, i64, f32, vp, vp]),
    "d2r_adamw_step_dev": (i32, [vp, vp, vp, vp, vp, i32, i64, vp, f32, f32, f32, f32, vp, vp]),
    "d2r_grad_nonfinite": (i32, [vp, i64, vp, vp]),
    "d2r_copy_rows": (i32, [vp, i64, vp, i64, i64, i64, vp]),
}


class D2RError(RuntimeError):
    pass


_lib = None
_FN = {}  # name -> bound foreign function (filled by load(); one dict lookup per launch instead of two getattr)


# measurement aids declared in include/d2r_hip_probes.h (not part of the drop-in surface)
PROBE_SIGNATURES = {
    "d2r_gemm_tuning": (None, [i32, i32, i32]),
    "d2r_gemm_timer": (i32, [i32]),
    "d2r_gemm_timer_read": (i32, [vp, vp, vp, vp, i32]),
    "d2r_gemm_debug_stamps": (None, [vp]),
    "d2r_gemm8_debug_stamps": (None, [vp]),
    "d2r_xattn3_debug_stamps": (None, [vp]),
    "d2r_xattn3_debug_mode": (None, [i32]),
    "d2r_adamw_probe_mode": (None, [i32, i32]),
}


def load():
    """Loads libd2r_hip.so once; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise D2RError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run `python -m d2r_amd.build` "
            "(or __graft_entry__.build()). d2r_amd has no CPU or PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in list(SIGNATURES.items()) + list(PROBE_SIGNATURES.items()):
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch; let it propagate
        fn.restype = res
        fn.argtypes = args
        _FN[name] = fn
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().d2r_last_error().decode(errors="replace")
        raise D2RError(f"{what or 'libd2r_hip'} failed with status {rc}: {msg}")


class KernelTimer:
    """Optional per-launch timing with HIP events on the launching stream (bench.py's roofline leg).
    Usage: ``with KernelTimer() as kt: step()`` then ``kt.summary()`` -> {group: (calls, total_ms, meta sums)}."""

    def __init__(self):
        self.records = []

    def __enter__(self):
        global _timer
        _timer = self
        return self

    def __exit__(self, *exc):
        global _timer
        _timer = None

    def summary(self):
        import torch
        torch.cuda.synchronize()
        out, times = {}, {}
        for group, meta, e0, e1 in self.records:
            rec = out.setdefault(group, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0, algo_bytes=0.0, outliers=0))
            rec["calls"] += 1
            times.setdefault(group, []).append(e0.elapsed_time(e1))
            rec["flops"] += float(meta.get("flops", 0.0))
            rec["bytes"] += float(meta.get("bytes", 0.0))
            rec["algo_bytes"] += float(meta.get("algo_bytes", 0.0))
        for group, ts in times.items():
            # An interval between two events also contains whatever the host did between recording them (an allocator
            # refill, a first-use code-object load): one such hiccup of milliseconds must not be booked as kernel time.
            # Intervals above 8x the group's median are replaced by the median and counted.
            med = sorted(ts)[len(ts) // 2]
            bad = [t for t in ts if t > 8.0 * med and t > 0.2]
            out[group]["outliers"] = len(bad)
            out[group]["ms"] = sum(min(t, med) if (t > 8.0 * med and t > 0.2) else t for t in ts)
        return out


_timer = None


def call(name: str, *args, meta=None):
    if _timer is None:
        fn = _FN.get(name)
        if fn is None:
            load()
            fn = _FN[name]
        rc = fn(*args)
    else:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()  # torch's current stream == the stream every d2r kernel is launched on
        rc = getattr(load(), name)(*args)
        e1.record()
        m = meta or {}
        _timer.records.append((m.get("group", name), m, e0, e1))
    if rc != 0:
        check(rc, name)
