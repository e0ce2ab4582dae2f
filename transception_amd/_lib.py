"""ctypes binding of libtransception_hip.so -- the only way the Python host reaches the GPU arithmetic.

There is deliberately NO fallback: if the shared library is missing or a call returns a non-zero
status the host raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported here.)
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TC_LIB_PATH") or os.path.join(HERE, "libtransception_hip.so")   # override: A/B kernel experiments only

TC_F32, TC_BF16, TC_F16 = 0, 1, 2
ACT_NONE, ACT_HSWISH, ACT_COORD, ACT_SIGMOID, ACT_GELU, ACT_SCALE, ACT_RELU = 0, 1, 2, 3, 4, 5, 6
ABI_VERSION = 15
ATTN_DKV_SPLITS = 8                # include/transception_hip.h: partial dK|dV buffers in tc_attn_bwd_seg's fp32 scratch

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_longlong, C.c_float


class TcGemm(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("bias", vp), ("R", vp),
                ("M", i32), ("N", i32), ("K", i32),
                ("lda", i32), ("ldb", i32), ("ldc", i32), ("ldr", i32),
                ("transA", i32), ("transB", i32), ("nb1", i32), ("nb2", i32),
                ("sA1", i64), ("sA2", i64), ("sB1", i64), ("sB2", i64),
                ("sC1", i64), ("sC2", i64), ("sR1", i64), ("sR2", i64),
                ("alpha", f32), ("accumulate", i32), ("act", i32), ("splitk", i32), ("dtype", i32), ("c_f32", i32), ("atomic", i32), ("rowsum", vp), ("sBias1", i64), ("sRow1", i64),
                ("bgap_every", i32), ("bgap", i64), ("ws", vp), ("ws_bytes", i64),
                # MixFFN fusion hooks (include/transception_hip.h): LayerNorm + GELU applied to operand tiles / the epilogue
                ("ffn_mode", i32), ("ffn_nchunk", i32), ("ffn_chunk_n", i32), ("ffn_ldd", i32), ("ffn_eps", f32),
                ("ffn_part", vp), ("ffn_stat", vp), ("ffn_gamma", vp), ("ffn_beta", vp), ("ffn_d", vp), ("ffn_part2", vp),
                ("ffn_sRow1", i64), ("ffn_sPar1", i64), ("ffn_aout", vp),
                # BatchNorm statistics of the output left by the epilogue (tc_bn_fwd(stats_chunks=...))
                ("bn_part", vp), ("bn_shift", vp)]


FFN_NONE, FFN_LN_A, FFN_LN_B, FFN_EP = 0, 1, 2, 3


class TcDwSeg(C.Structure):
    _fields_ = [("x", vp), ("w", vp), ("bias", vp), ("y", vp), ("dy", vp), ("dw", vp), ("db", vp), ("C", i32), ("k", i32),
                ("ldx", i32), ("ldy", i32), ("lddy", i32), ("B", i32), ("H", i32), ("W", i32), ("stat", vp)]


class TcDwFold(C.Structure):
    _fields_ = [("part", vp), ("dw", vp), ("db", vp), ("dgamma", vp), ("dbeta", vp), ("wstride", i64), ("C", i32), ("k", i32), ("groups", i32),
                ("ch", i32), ("chunks", i32), ("gx", i32), ("nt", i32)]


class TcLnFold(C.Structure):
    _fields_ = [("part", vp), ("dgamma", vp), ("dbeta", vp), ("pstride", i64), ("nblk", i32), ("C", i32), ("groups", i32)]


class TcFfnSeg(C.Structure):
    _fields_ = [("gp", vp), ("d", vp), ("h", vp), ("dh", vp), ("stat", vp), ("part2", vp), ("w", vp), ("gamma", vp),
                ("dw", vp), ("db", vp), ("dgamma", vp), ("dbeta", vp),
                ("C", i32), ("ldg", i32), ("ldd", i32), ("ldh", i32), ("lddh", i32), ("B", i32), ("H", i32), ("W", i32), ("nch2", i32)]


class TcFfnFused(C.Structure):
    _fields_ = [("x", vp), ("w1", vp), ("b1", vp), ("wd", vp), ("bd", vp), ("gamma", vp), ("beta", vp), ("w2", vp), ("b2", vp),
                ("res", vp), ("out", vp), ("h", vp), ("d", vp), ("a", vp), ("stat", vp),
                ("sres", i64), ("sout", i64), ("wstride", i64),
                ("ldx", i32), ("ldr", i32), ("ldo", i32), ("C", i32), ("B", i32), ("H", i32), ("W", i32), ("groups", i32), ("eps", f32),
                ("tile_h", i32), ("tile_w", i32), ("pre_gamma", vp), ("pre_beta", vp), ("pre_eps", f32)]


class TcFfnBwd(C.Structure):
    _fields_ = [("x", vp), ("dy", vp), ("d", vp), ("stat", vp),
                ("w1", vp), ("b1", vp), ("wd", vp), ("gamma", vp), ("beta", vp), ("w2", vp),
                ("dx", vp), ("gd", vp), ("part", vp), ("part_floats", i64),
                ("dw1", vp), ("db1", vp), ("dwd", vp), ("dbd", vp), ("dgamma", vp), ("dbeta", vp), ("dw2", vp), ("db2", vp),
                ("sdy", i64), ("wstride", i64),
                ("ldx", i32), ("lddy", i32), ("lddx", i32), ("C", i32), ("B", i32), ("H", i32), ("W", i32), ("groups", i32), ("acc_dx", i32),
                ("eps", f32), ("tile_h", i32), ("tile_w", i32),
                ("pre_gamma", vp), ("pre_beta", vp), ("dpre_gamma", vp), ("dpre_beta", vp), ("pre_eps", f32)]


class TcEffAtt(C.Structure):
    _fields_ = [("t", vp), ("gamma", vp), ("beta", vp),
                ("wk", vp), ("bk", vp), ("wq", vp), ("bq", vp), ("wv", vp), ("bv", vp), ("wr", vp), ("br", vp),
                ("out", vp), ("ctx", vp), ("kstat", vp), ("part", vp), ("part_floats", i64),
                ("dout", vp), ("dt", vp), ("g1", vp),
                ("dgamma", vp), ("dbeta", vp), ("dwk", vp), ("dbk", vp), ("dwq", vp), ("dbq", vp), ("dwv", vp), ("dbv", vp), ("dwr", vp), ("dbr", vp),
                ("ldt", i32), ("ldo", i32), ("lddo", i32), ("lddt", i32), ("acc_dt", i32), ("C", i32), ("B", i32), ("N", i32),
                ("eps", f32)]


class TcEwSeg(C.Structure):
    _fields_ = [("kind", i32), ("flag", i32), ("a", vp), ("b", vp), ("sa", i64), ("sb", i64),
                ("lda", i32), ("ldb", i32), ("n0", i32), ("n1", i32), ("n2", i32), ("n3", i32), ("n4", i32), ("reserved", i32)]


EW_PATCHIFY, EW_DEINTERLEAVE, EW_COPY = 0, 1, 2


class TcSliceAug(C.Structure):
    _fields_ = [("m", C.c_double * 6), ("disp", C.c_float * 32), ("alpha", f32), ("center", f32), ("noise_sigma", f32),
                ("noise_seed", C.c_uint), ("flags", i32), ("reserved", i32)]


TC_AUG_WARP, TC_AUG_LINEAR, TC_AUG_BLUR, TC_AUG_PIECEWISE = 1, 2, 4, 8
TC_AUG_SKIP, TC_AUG_FROM_RAW = 1 << 20, 1 << 21

# name -> argtypes (every function returns int status unless listed in _RET)
SIGNATURES = {
    "tc_abi_version": [],
    "tc_gemm": [C.POINTER(TcGemm), vp],
    "tc_ew_multi": [C.POINTER(TcEwSeg), i32, i32, vp],
    "tc_gemm_pair": [C.POINTER(TcGemm), C.POINTER(TcGemm), vp],
    "tc_gemm_multi": [C.POINTER(TcGemm), i32, vp],
    "tc_colsum": [vp, i32, i32, i32, i32, i64, vp, i32, i32, vp],
    "tc_layernorm_fwd": [vp, i32, vp, vp, vp, i32, vp, vp, i32, i32, f32, i32, i32, i64, i32, vp],
    "tc_layernorm_bwd": [vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i64, vp, i64, i32, vp],
    "tc_layernorm_bwd_scratch_floats": [i32, i32, i32],
    "tc_layernorm_bwd_nblk": [i32, i32],
    "tc_dwconv_bwd_plan": [i32, i32, i32, i32, i32, i32, i32, vp],
    "tc_dwconv_multi_plan": [vp, i32, i32, i32, vp, vp],
    "tc_ffn_mid_plan": [vp, i32, i32, i32, vp, vp],
    "tc_dw_fold": [vp, i32, vp],
    "tc_layernorm_bwd_defer": [vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i64, vp, i64, i32, vp],
    "tc_layernorm_fold": [vp, i32, vp],
    "tc_layernorm_ps_fwd": [vp, i32, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp],
    "tc_layernorm_ps_bwd": [vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, i64, i32, vp],
    "tc_ln_cls_supported": [i32, i32, i32],
    "tc_ln_cls_scratch_floats": [i32, i32],
    "tc_ln_cls_fwd": [vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, f32, i32, vp],
    "tc_ln_cls_bwd": [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp],
    "tc_layernorm_bwd_params": [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i32, vp],
    "tc_dwconv_fwd": [vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i64, i32, vp],
    "tc_dwconv_bwd_input": [vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i64, i32, vp],
    "tc_dwconv_bwd_weight": [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i64, vp, i64, i32, vp],
    "tc_dwconv_bwd": [vp, i32, vp, i32, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i64, vp, i64, i32, vp],
    "tc_dwconv_multi": [C.POINTER(TcDwSeg), i32, i32, i32, i32, i32, i64, vp, i64, i32, vp],
    "tc_ffn_chunk": [i32, i32],
    "tc_ffn_dw_fwd": [vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i64, i32, vp],
    "tc_ffn_mid_bwd": [C.POINTER(TcFfnSeg), i32, i32, i64, vp, i64, i32, vp],
    "tc_ffn_fused_supported": [i32, i32],
    "tc_ffn_fused_fwd": [C.POINTER(TcFfnFused), i32, vp],
    "tc_ffn_fused_bwd_supported": [i32, i32],
    "tc_ffn_fused_bwd_scratch_floats": [i32, i32],
    "tc_ffn_fused_bwd": [C.POINTER(TcFfnBwd), i32, vp],
    "tc_effatt_supported": [i32, i32],
    "tc_effatt_scratch_floats": [i32, i32, i32],
    "tc_effatt_fwd": [C.POINTER(TcEffAtt), i32, vp],
    "tc_effatt_bwd": [C.POINTER(TcEffAtt), i32, vp],
    "tc_bn_scratch_floats": [i32, i32],
    "tc_bn_fwd": [vp, i32, vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, i32, f32, f32, i32, i32, i32, i32, vp],
    "tc_bn_bwd": [vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "tc_softmax_scratch_floats": [i32, i32, i32],
    "tc_softmax_fwd": [vp, vp, vp, i32, i64, i64, i32, i32, i32, i32, i32, i32, vp],
    "tc_softmax_bwd": [vp, vp, vp, vp, i32, i64, i64, i64, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "tc_attn_fwd": [vp, i32, i64, vp, i32, vp, i32, i64, vp, i32, i64, vp, i32, i32, i32, f32, i32, vp],
    "tc_attn_bwd": [vp, i32, i64, vp, i32, vp, i32, i64, vp, i32, i64, vp, i32, i64, vp, vp, vp, i32, i64, vp, i32,
                    vp, i32, i64, i32, i32, i32, i32, f32, i32, vp],
    "tc_attn_fwd_seg": [vp, i32, vp, i32, vp, i32, i64, vp, i32, vp, i32, i32, C.POINTER(i32), i32, f32, i32, i32, vp],
    "tc_attn_bwd_seg": [vp, i32, vp, i32, vp, i32, i64, vp, i32, vp, i32, vp, vp, vp, vp, i32, vp, i32, vp, i32, i64, i32, i32,
                        C.POINTER(i32), i32, f32, i32, i32, vp],
    "tc_fma3_fwd": [vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, f32, i32, vp],
    "tc_fma3_bwd": [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, vp, i32, i32, i32, f32, i32, vp],
    "tc_add": [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp],
    "tc_sigmoid_bwd": [vp, vp, vp, i64, i32, vp],
    "tc_copy3d": [vp, i64, i32, vp, i64, i32, i32, i32, i32, i32, i32, vp],
    "tc_transpose": [vp, vp, i32, i32, i32, i32, vp],
    "tc_chan_pool_fwd": [vp, i32, vp, i32, i32, i32, i32, vp],
    "tc_chan_pool_bwd": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "tc_chan_gate_fwd": [vp, i32, vp, vp, i32, i32, i32, i32, i32, vp],
    "tc_chan_gate_bwd": [vp, i32, vp, i32, vp, vp, i32, i32, vp, i32, i32, i32, i32, vp],
    "tc_relu_fwd": [vp, vp, i64, i32, vp],
    "tc_relu_bwd": [vp, vp, vp, i64, i32, vp],
    "tc_chan_pool2_fwd": [vp, i32, vp, vp, i32, i32, i32, i32, vp],
    "tc_chan_pool2_bwd": [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "tc_pix_stats_fwd": [vp, i32, vp, vp, i32, i32, i32, vp],
    "tc_pix_stats_bwd": [vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "tc_sa_conv_fwd": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "tc_sa_conv_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "tc_pix_gate_fwd": [vp, i32, vp, vp, i32, i32, i32, i32, vp],
    "tc_pix_gate_bwd": [vp, i32, vp, i32, vp, vp, i32, i32, vp, i32, i32, i32, vp],
    "tc_cam_att_fwd": [vp, i32, vp, i32, i32, i32, i32, vp],
    "tc_cam_apply_fwd": [vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "tc_cam_bwd": [vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "tc_gamma_res_fwd": [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp],
    "tc_gamma_res_bwd": [vp, i32, vp, i32, vp, vp, i32, vp, i32, i32, vp, i32, i32, i32, vp],
    "tc_gelu_fwd": [vp, vp, i64, i32, vp],
    "tc_gelu_bwd": [vp, vp, vp, i64, i32, vp],
    "tc_coord_pool_fwd": [vp, vp, i32, i32, i32, i32, i32, vp],
    "tc_coord_pool_bwd": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "tc_coord_gate_fwd": [vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "tc_coord_gate_bwd": [vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp],
    "tc_pixel_shuffle": [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "tc_patchify": [vp, i64, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "tc_sr_deinterleave": [vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp],
    "tc_stem_im2col": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "tc_window_rows": [vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "tc_dropout": [vp, vp, i64, f32, vp, C.c_uint, i32, vp],
    "tc_im2col3s2": [vp, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp],
    "tc_col2im3s2": [vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "tc_seg_loss_fwd": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "tc_factor_att_stats_floats": [i32, i32, i32],
    "tc_factor_att_fwd": [vp, vp, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, f32, i32, vp],
    "tc_factor_att_bwd": [vp, vp, vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, f32, i32, vp],
    "tc_argmax_counts": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "tc_seg_loss_bwd": [vp, vp, vp, vp, i32, i32, i32, f32, f32, f32, f32, vp, i32, vp],
    "tc_seg_loss_fwd_tok": [vp, i32, vp, vp, vp, i32, i32, i32, i32, vp],
    "tc_seg_loss_bwd_tok": [vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, f32, f32, f32, f32, vp, i32, vp],
    "tc_seg_loss_value": [vp, i32, C.c_double, C.c_double, C.c_double, vp, vp],
    "tc_slice_augment": [vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "tc_slice_augment_chain": [vp, vp, vp, i32, i32, vp, vp, vp, vp, i32, i32, i32, vp],
    "tc_spline_prefilter": [vp, vp, i32, i32, i32, vp],
    "tc_zoom_normalize": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, f32, vp],
    "tc_sgd_step": [vp, vp, vp, i64, f32, vp, f32, f32, f32, i32, vp],
    "tc_sgd_step_multi": [vp, vp, vp, vp, i32, i64, f32, vp, f32, f32, f32, i32, vp, f32, vp, i32, vp],
    "tc_grad_sumsq": [vp, i64, vp, vp],
    "tc_fill_f32": [vp, i64, f32, vp],
    "tc_cast": [vp, vp, i64, i32, i32, vp],
    "tc_seg_marker": [i32, vp],
    "tc_linear_ln_supported": [i32, i32],
    "tc_linear_ln_fwd": [vp, i32, vp, vp, i64, vp, i32, vp, vp, i64, vp, i32, vp, i32, vp, vp, i32, i32, i32, f32, i32, vp],
    "tc_ripm_supported": [i32, i32],
    "tc_ripm_tiles": [i32, i32, i32],
    "tc_ripm_fwd": [vp, i32, i32, vp, i32, vp, vp, vp, vp, vp, vp, f32, f32, i32, vp, i32, vp, vp, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "tc_mhca_att_supported": [i32, i32, i32],
    "tc_dw_ln_supported": [i32, i32],
    "tc_dw_ln_fwd": [vp, i32, vp, vp, i64, vp, vp, i64, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp],
    "tc_mhca_att_bwd_supported": [i32, i32, i32],
    "tc_mhca_att_bwd": [vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, f32, i32, vp],
    "tc_mhca_att_fwd": [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, f32, i32, vp],
}
_RET = {"tc_ln_cls_scratch_floats": i64, "tc_ffn_fused_bwd_scratch_floats": i64, "tc_effatt_scratch_floats": i64, "tc_bn_scratch_floats": i64, "tc_softmax_scratch_floats": i64, "tc_layernorm_bwd_scratch_floats": i64, "tc_dwconv_bwd_plan": i64, "tc_dwconv_multi_plan": i64, "tc_ffn_mid_plan": i64, "tc_factor_att_stats_floats": i64}
_RAW = {"tc_abi_version", "tc_ln_cls_supported", "tc_ln_cls_scratch_floats", "tc_linear_ln_supported", "tc_ripm_supported", "tc_ripm_tiles", "tc_mhca_att_supported", "tc_dw_ln_supported", "tc_mhca_att_bwd_supported", "tc_effatt_supported", "tc_effatt_scratch_floats", "tc_ffn_chunk", "tc_ffn_fused_supported", "tc_ffn_fused_bwd_supported", "tc_ffn_fused_bwd_scratch_floats", "tc_bn_scratch_floats", "tc_softmax_scratch_floats", "tc_layernorm_bwd_scratch_floats", "tc_layernorm_bwd_nblk", "tc_dwconv_bwd_plan", "tc_dwconv_multi_plan", "tc_ffn_mid_plan", "tc_factor_att_stats_floats"}     # not status-returning


class TcError(RuntimeError):
    pass


class _Lib:
    calls = 0

    def __init__(self, path: str = LIB_PATH):
        if not os.path.exists(path):
            raise TcError(f"{path} is missing: build it with `python -m transception_amd.build` "
                          "(there is no CPU/PyTorch fallback for the TransCeption hot path)")
        self.path = path
        self.cdll = C.CDLL(path)
        for name, args in SIGNATURES.items():
            fn = getattr(self.cdll, name)          # AttributeError if a declared symbol is not exported
            fn.argtypes = args
            fn.restype = _RET.get(name, i32)
            setattr(self, name, fn if name in _RAW else self._checked(name, fn))
        v = self.cdll.tc_abi_version()
        if v != ABI_VERSION:
            raise TcError(f"ABI mismatch: library {v}, host {ABI_VERSION}")

    @staticmethod
    def _checked(name, fn):
        def call(*a):
            _Lib.calls += 1                      # C-ABI entries since import (bench.py reports the per-step count)
            rc = fn(*a)
            if rc != 0:
                raise TcError(f"{name} failed with status {rc}")
        call.__name__ = name
        return call


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
