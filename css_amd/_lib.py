"""ctypes binding of libcsship.so (C ABI in include/csship.h).

There is deliberately no fallback: if the shared library is missing ``lib()`` raises, and
every op wrapper refuses non-CUDA tensors.  Build it with ``make -C css_amd/csrc`` or
``python -c "import __graft_entry__ as g; g.build()"``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcsship.so")

_p, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t


class GemmParams(C.Structure):
    _fields_ = [("M", _i32), ("N", _i32), ("K", _i32), ("lda", _i32), ("ldw", _i32), ("ldc", _i32),
                ("ldr", _i32), ("a_mode", _i32), ("epilogue", _i32), ("rows_per_batch", _i32),
                ("alpha", _f32)]


class ConvWeights(C.Structure):
    _fields_ = [(n, _p) for n in ("ln_g", "ln_b", "pw1_w", "pw1_b", "dw_w", "dw_b", "bn_g", "bn_b",
                                  "bn_mean", "bn_var", "pw2_w", "pw2_b")] + [("causal", _i32)]


class DropoutCfg(C.Structure):  # csship_dropout_cfg
    _fields_ = [("seed", _p), ("stream", C.c_uint32), ("p", _f32)]


class FfnWeights(C.Structure):
    _fields_ = [(n, _p) for n in ("ln_g", "ln_b", "w1", "b1", "w2", "b2", "w1_scale", "w2_scale")]


class ChainTrain(C.Structure):  # csship_chain_train
    _fields_ = [("drop_a_hidden", C.POINTER(DropoutCfg)), ("drop_a_out", C.POINTER(DropoutCfg)),
                ("drop_b_hidden", C.POINTER(DropoutCfg)), ("drop_b_out", C.POINTER(DropoutCfg)),
                ("pre_ln_out", _p), ("ln_out", _p),
                ("bn_z", _p), ("bn_x2", _p), ("bn_scale_shift", _p), ("bn_pw2_w", _p), ("bn_pw2_b", _p),
                ("drop_bn", C.POINTER(DropoutCfg)), ("bn_out", _p)]


class QkvBwdArgs(C.Structure):  # csship_qkv_bwd_args
    _fields_ = [(n, _p) for n in ("dqkv", "wqkvT_packed", "x", "dy", "ln_g", "ln_b", "y_bf16", "dgamma_dbeta")]


class FfnBwdArgs(C.Structure):  # csship_ffn_bwd_args
    _fields_ = [(n, _p) for n in ("x", "ln_g", "ln_b", "w1_packed", "b1", "w2t_packed", "w1t_packed", "a_bf16", "h_bf16", "dh_bf16",
                                  "dx", "dgamma_dbeta")] + \
               [("drop_hidden", C.POINTER(DropoutCfg)), ("drop_out", C.POINTER(DropoutCfg)), ("dyd_bf16", _p),
                ("ln2_x", _p), ("ln2_g", _p), ("dgamma_dbeta2", _p)]


class ConvStatsArgs(C.Structure):  # csship_conv_stats_args
    _fields_ = [("fwd_workspace", _p), ("B", _i32), ("T", _i32), ("pw2_w", _p), ("drop", C.POINTER(DropoutCfg)), ("grads", _p)]


class MhaWeights(C.Structure):
    _fields_ = [(n, _p) for n in ("ln_g", "ln_b", "wqkv", "bqkv", "wo", "bo")]


class LayerWeights(C.Structure):
    _fields_ = [("ffn_in", FfnWeights), ("mha", MhaWeights), ("conv", ConvWeights), ("ffn_out", FfnWeights),
                ("ln_g", _p), ("ln_b", _p)]


class ConformerWeights(C.Structure):
    _fields_ = [("idim", _i32), ("idim_pad", _i32), ("d_model", _i32), ("n_head", _i32), ("d_ffn", _i32),
                ("n_layer", _i32), ("kernel_size", _i32), ("n_out", _i32), ("pe_q_min", _i32), ("pe_q_max", _i32),
                ("bn_mode", _i32), ("generic", _i32), ("fuse_conv", _i32),
                ("embed_w", _p), ("embed_b", _p), ("embed_ln_g", _p), ("embed_ln_b", _p), ("pe_k", _p),
                ("layers", C.POINTER(LayerWeights)), ("head_w", _p), ("head_b", _p), ("ffn_mode", _i32),
                ("lin_mode", _i32), ("chain_flags", _i32)]


class BgemmParams(C.Structure):
    _fields_ = [("M", _i32), ("N", _i32), ("K", _i32), ("lda", _i64), ("ldb", _i64), ("ldc", _i64), ("ldaux", _i64),
                ("a_kmajor", _i32), ("b_kmajor", _i32), ("a_f32", _i32), ("b_f32", _i32), ("epilogue", _i32),
                ("batch1", _i32), ("batch2", _i32), ("splits", _i32),
                ("sA1", _i64), ("sA2", _i64), ("sB1", _i64), ("sB2", _i64), ("sC1", _i64), ("sC2", _i64),
                ("sX1", _i64), ("sX2", _i64), ("alpha", _f32), ("band", _i32), ("band_stride", _i32)]


_dc = C.POINTER(DropoutCfg)


class GradJob(C.Structure):  # csship_grad_job (40 bytes)
    _fields_ = [("src", _p), ("dst", _p), ("n", _i64), ("ld_src", _i64), ("cols", _i32), ("pad_", _i32)]


class PackJob(C.Structure):  # csship_pack_job (72 bytes)
    _fields_ = [("w", _p), ("row_major", _p), ("frag", _p), ("ld_w", _i64), ("ld_row_major", _i64),
                ("R", _i32), ("K", _i32), ("dst_row0", _i32), ("Kp", _i32), ("scale", _f32), ("transposed", _i32),
                ("k0", _i32), ("Kp_dst", _i32)]


A_BF16, A_F32, A_F32_MVN = 0, 1, 2
BEPI_OUT_BF16, BEPI_MASK_POS, BEPI_ADD, BEPI_SCALE_BIAS, BEPI_RELU = 1, 2, 4, 8, 16
EPI_RELU, EPI_RESIDUAL, EPI_LAYERNORM, EPI_OUT_BF16 = 1, 2, 4, 8
PRO_LN, PRO_BF16, PRO_F32 = 0, 1, 2
CHAIN_QKV_IL8 = 1
FWD_Q_LOG2E = 4
MHA_Q_LOG2E = 1

# name -> (restype, argtypes); mirrors include/csship.h one to one
SIGNATURES = {
    "csship_version": (_i32, []),
    "csship_strerror": (C.c_char_p, [_i32]),
    "csship_ffn_static_max_ffn": (_i32, []),
    "csship_attn_pos_bwd_max_t": (_i32, []),
    "csship_stft": (_i32, [_p, _i64, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _p]),
    "csship_stft_ld": (_i32, [_p, _i64, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _i64, _p]),
    "csship_stft_logpower": (_i32, [_p, _i64, _i64, _i64, _i64, _i64, _f32, _p, _p]),
    "csship_mix_workspace": (_sz, [_i32]),
    "csship_mix_batch": (_i32, [_p, _i64, _i32, _p, _i64, _i32, _p, _i32, _i32, _p, _p, _p, _i64, _p, _sz, _p]),
    "csship_mvn_stats": (_i32, [_p, _i64, _i64, _i64, _i64, _f32, _f32, _p, _p, _p]),
    "csship_transpose_norm": (_i32, [_p, _i64, _i64, _i64, _i64, _f32, _p, _p, _p, _i64, _i64, _p]),
    "csship_ipd": (_i32, [_p, _p, _i64, _i64, _i64, C.POINTER(_i32), C.POINTER(_i32), _i32, _p, _i64, _i64, _p, _p]),
    "csship_ipd_tf_workspace": (_sz, [_i64, _i32]),
    "csship_ipd_tf": (_i32, [_p, _p, _i64, _i64, _i64, C.POINTER(_i32), C.POINTER(_i32), _i32, _p, _i64, _i64, _p, _sz, _p]),
    "csship_istft": (_i32, [_p, _p, _i64, _i64, _i64, _p, _p]),
    "csship_layernorm": (_i32, [_p, _i64, _i32, _p, _p, _p, _p, _p]),
    "csship_layernorm_relu": (_i32, [_p, _i64, _i32, _p, _p, _i32, _p, _p, _p]),
    "csship_lstm_bidir": (_i32, [_p, _p, _p, _i32, _i32, _i32, _p]),
    "csship_lstm_dist_workspace": (_sz, [_i32, _i32]),
    "csship_lstm_dist_workspace_error_offset": (_sz, [_i32, _i32]),
    "csship_lstm_bidir_dist": (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _p, _sz, _p]),
    "csship_gemm": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _p, C.POINTER(GemmParams), _p]),
    "csship_ffn_fused": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _f32, _p]),
    "csship_panel_linear": (_i32, [_p, _i32, _p, _p, _p, _p, _p, _i64, _f32, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "csship_panel_linear_drop": (_i32, [_p, _i32, _p, _p, _p, _p, _p, _i64, _f32, _p, _i64, _i32, _i32, _i32, _i32, _dc, _p]),
    "csship_embed": (_i32, [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "csship_embed_train": (_i32, [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _i32, _i32, _i32, _i32, _dc, _p]),
    "csship_outproj_glu": (_i32, [_p, _p, _p, _p, _p, C.POINTER(ConvWeights), _p, _i32, _i32, _p]),
    "csship_encoder_chain_train": (_i32, [_p, _p, C.POINTER(FfnWeights), _p, _p, C.POINTER(FfnWeights), _p, _p, _p, _p, _p,
                                   _i32, _i32, _i32, _i32, C.POINTER(ChainTrain), _p]),
    "csship_encoder_chain": (_i32, [_p, _p, _p, C.POINTER(ConvWeights), _i32, C.POINTER(FfnWeights), _p, _p,
                                     C.POINTER(FfnWeights), _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "csship_mha_relpos": (_i32, [_p, _p, _i32, _i32, _p, _i32, _i32, _i32, _p]),
    "csship_mha_relpos_ex": (_i32, [_p, _p, _i32, _i32, _p, _i32, _i32, _i32, _i32, _p]),
    "csship_outproj_conv": (_i32, [_p, _p, _p, _p, _p, C.POINTER(ConvWeights), _i32, _i32, _i32, _i32, _p]),
    "csship_outproj_conv_train": (_i32, [_p, _p, _p, _p, _p, _p, C.POINTER(ConvWeights), _i32, _i32, _i32, _i32,
                                         _dc, _dc, _p, _p]),
    "csship_ffn_bwd_workspace": (_sz, [_i32]),
    "csship_ffn_bwd": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _f32, _p, _sz, _p]),
    "csship_mha_relpos_train": (_i32, [_p, _p, _i32, _i32, _p, _p, _i32, _i32, _i32, _p]),
    "csship_mha_relpos_probs": (_i32, [_p, _p, _i32, _i32, _p, _p, _i64, _i32, _i32, _i32, _p]),
    "csship_convmod_workspace": (_sz, [_i32, _i32, _i32]),
    "csship_convmod": (_i32, [_p, _p, C.POINTER(ConvWeights), _i32, _i32, _i32, _i32, _i32, _p, _p]),
    "csship_conv_from_glu": (_i32, [_p, _p, _p, C.POINTER(ConvWeights), _i32, _i32, _i32, _i32, _p]),
    "csship_pit_workspace": (_sz, [_i32, _i32, _i32, _i32]),
    "csship_pit_mse": (_i32, [_p, _i64, _p, _i64, C.POINTER(_p), _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p]),
    "csship_pit_mse_bwd": (_i32, [_p, _i64, _p, _i64, C.POINTER(_p), _p, _i32, _i32, _i32, _i32, _i32, _f32, _p, _p]),
    "csship_pit_mse_bwd_dev": (_i32, [_p, _i64, _p, _i64, C.POINTER(_p), _p, _i32, _i32, _i32, _i32, _i32, _f32, _p, _p, _p]),
    "csship_apply_masks": (_i32, [_p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _p, _p]),
    "csship_stitch_similarity": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _p, _p]),
    "csship_stitch_connect": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _f32, _p, _p]),
    "csship_bgemm_workspace": (_sz, [C.POINTER(BgemmParams)]),
    "csship_bgemm": (_i32, [_p, _p, _p, _p, _p, C.POINTER(BgemmParams), _p, _sz, _p]),
    "csship_bgemm_wgrad_workspace": (_sz, [C.POINTER(BgemmParams)]),
    "csship_bgemm_wgrad": (_i32, [_p, _p, _p, _p, _p, C.POINTER(BgemmParams), _f32, _i32, _p, _sz, _p]),
    "csship_reduce_defer": (_i32, [_i32]),
    "csship_reduce_hold": (_i32, [_i32]),
    "csship_grad_accumulate": (_i32, [_p, _i32, _p]),
    "csship_ema_update": (_i32, [_p, _i32, _f32, _f32, _i32, _f32, _p]),
    "csship_wgrad_hold": (_i32, [_i32]),
    "csship_wgrad_pending": (_i32, []),
    "csship_wgrad_group_workspace": (_sz, []),
    "csship_wgrad_flush": (_i32, [_p, _sz, _p]),
    "csship_reduce_pending": (_i32, []),
    "csship_reduce_flush": (_i32, [_p]),
    "csship_colsum_workspace": (_sz, [_i64, _i32]),
    "csship_colsum": (_i32, [_p, _i32, _i64, _i32, _i64, _f32, _p, _p, _sz, _p]),
    "csship_layernorm_bwd_workspace": (_sz, [_i64, _i32]),
    "csship_layernorm_bwd": (_i32, [_p, _p, _p, _p, _p, _i32, _i64, _i32, _p, _p, _p, _sz, _p]),
    "csship_layernorm_bwd_y": (_i32, [_p, _p, _p, _p, _p, _i32, _i64, _i32, _p, _p, _p, _p, _sz, _p]),
    "csship_relu_bwd": (_i32, [_p, _i64, _p, _i64, _i64, _i32, _p, _i64, _p]),
    "csship_relu_bwd_bf16": (_i32, [_p, _i64, _p, _i64, _i64, _i32, _p, _i64, _p]),
    "csship_mvn_apply": (_i32, [_p, _i64, _i64, _i64, _i64, _f32, _p, _p, _p, _i64, _p]),
    "csship_relpos_softmax": (_i32, [_p, _i64, _p, _i64, _i32, _i32, _i32, _p, _i64, _p]),
    "csship_attn_rowdot": (_i32, [_p, _p, _i32, _i32, _i32, _p, _p]),
    "csship_attn_softmax_bwd": (_i32, [_p, _p, _p, _i64, _p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "csship_relpos_softmax_bwd": (_i32, [_p, _i64, _p, _i64, _i32, _i32, _i32, _p, _p, _i32, _p]),
    "csship_convmod_bwd_workspace": (_sz, [_i32, _i32, _i32]),
    "csship_convmod_bwd": (_i32, [_p, _p, C.POINTER(ConvWeights), _i32, _i32, _i32, _i32, _p, _p, _p, _i32, _p, _sz, _p]),
    "csship_pack_weight": (_i32, [_p, _i32, _i32, _i64, _f32, _p, _i64, _p, _i32, _p]),
    "csship_pack_weights_multi": (_i32, [_p, _i32, _i64, _p]),
    "csship_lstm_bidir_train": (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _p]),
    "csship_lstm_bidir_bwd": (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _p]),
    "csship_lstm_bidir_bwd_dist": (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _p, _sz, _p]),
    "csship_adam_clip_workspace": (_sz, [_i64]),
    "csship_adam_clip": (_i32, [_p, _p, _p, _p, _i64, _f32, _f32, _f32, _f32, _f32, _i64, _f32, _p, _p, _p, _sz, _p]),
    "csship_mvdr_scm": (_i32, [_p, _p, _p, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _p, _p, _p]),
    "csship_mvdr_scm_multi": (_i32, [_p, _p, _p, _i32, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _p, _p, _p]),
    "csship_mvdr_souden": (_i32, [_p, _p, _p, _i64, _i32, _i32, _f32, _p, _p]),
    "csship_mvdr_apply_workspace": (_sz, [_i32, _i32, _i32]),
    "csship_mvdr_apply": (_i32, [_p, _p, _p, _p, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _sz, _p]),
    "csship_mvdr_finish": (_i32, [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i64, _p]),
    "csship_doa_workspace": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "csship_doa_likelihood": (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _p, _p, _p, _sz, _p]),
    "csship_dropout_apply": (_i32, [_p, _i64, _p, _i64, _i64, _i32, _dc, _p]),
    "csship_dropout_add": (_i32, [_p, _p, _p, _f32, _i64, _i32, _dc, _p]),
    "csship_dropout_mask": (_i32, [_p, _i64, _dc, _p]),
    "csship_ffn_fused_drop": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _f32, _dc, _dc, _p]),
    "csship_ffn_fused_train": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _f32, _dc, _dc, _p]),
    "csship_ffn_bwd_drop": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _f32, _dc, _dc, _p,
                                    _p, _sz, _p]),
    "csship_ffn_bwd_ln_workspace": (_sz, [_i32]),
    "csship_ffn_bwd_ln": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _f32, _dc, _dc, _p,
                                  _p, _p, _p, _p, _sz, _p]),
    "csship_ffn_bwd_conv_workspace": (_sz, [_i32]),
    "csship_ffn_bwd_conv": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _f32, _dc, _dc, _p,
                                   _p, _i32, _i32, _p, _dc, _p, _p, _sz, _p]),
    "csship_convmod_bwd_after_stats": (_i32, [_p, _p, C.POINTER(ConvWeights), _i32, _i32, _i32, _i32, _p, _p, _p, _i32, _dc, _dc, _p, _p,
                                              _sz, _p]),
    "csship_reduce_side_stream": (_i32, [_p]),
    "csship_reduce_side_join": (_i32, [_p]),
    "csship_qkv_in_bwd_workspace": (_sz, [_i32]),
    "csship_qkv_ffn2_bwd_workspace": (_sz, [_i32]),
    "csship_qkv_ffn2_bwd": (_i32, [C.POINTER(QkvBwdArgs), C.POINTER(FfnBwdArgs), C.POINTER(FfnBwdArgs), C.POINTER(ConvStatsArgs),
                                   _i32, _i32, _i32, _f32, _p, _sz, _p]),
    "csship_qkv_ffn_bwd_workspace": (_sz, [_i32]),
    "csship_qkv_ffn_bwd": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p,                                   # dqkv .. attn dgamma | dbeta
                                  _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,                   # x .. dgamma | dbeta
                                  _i32, _i32, _i32, _f32, _dc, _dc, _p, _p, _p, _p, _p, _sz, _p]),
    "csship_qkv_in_bwd": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _p, _sz, _p]),
    "csship_mha_relpos_train_drop": (_i32, [_p, _p, _i32, _i32, _p, _p, _i32, _i32, _i32, _dc, _p]),
    "csship_mha_relpos_probs_drop": (_i32, [_p, _p, _i32, _i32, _p, _p, _p, _i64, _i32, _i32, _i32, _dc, _p]),
    "csship_attn_softmax_bwd_drop": (_i32, [_p, _p, _p, _i64, _p, _p, _p, _i32, _i32, _i32, _i32, _dc, _p]),
    "csship_mha_relpos_bwd_core": (_i32, [_p, _p, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _dc, _p]),
    "csship_convmod_drop": (_i32, [_p, _p, C.POINTER(ConvWeights), _i32, _i32, _i32, _i32, _i32, _p, _dc, _p]),
    "csship_convmod_sums_offset": (_sz, [_i32, _i32, _i32]),
    "csship_convmod_stats": (_i32, [_p, C.POINTER(ConvWeights), _i32, _i32, _i32, _i32, _p, _p]),
    "csship_convmod_finish": (_i32, [_p, _p, C.POINTER(ConvWeights), _i32, _i32, _i32, C.c_double, _p, _dc, _p]),
    "csship_convmod_bwd_sums_offset": (_sz, [_i32, _i32, _i32]),
    "csship_convmod_bwd_stats": (_i32, [_p, _p, C.POINTER(ConvWeights), _i32, _i32, _i32, _i32, _p, _p, _p, _dc, _p, _sz, _p]),
    "csship_convmod_bwd_finish": (_i32, [_p, _p, C.POINTER(ConvWeights), _i32, _i32, _i32, _i32, _p, _p, _p, C.c_double, _dc,
                                         _p, _sz, _p]),
    "csship_convmod_bwd_next": (_i32, [_p, _p, C.POINTER(ConvWeights), _i32, _i32, _i32, _i32, _p, _p, _p, _i32, _dc, _dc, _p, _p,
                                _sz, _p]),
    "csship_convmod_bwd_drop": (_i32, [_p, _p, C.POINTER(ConvWeights), _i32, _i32, _i32, _i32, _p, _p, _p, _i32, _dc, _p,
                                        _sz, _p]),
    "csship_conformer_workspace": (_sz, [C.POINTER(ConformerWeights), _i32, _i32]),
    "csship_conformer_forward": (_i32, [C.POINTER(ConformerWeights), _p, _i64, _i32, _i32, _p, _i64, _p, _p, _sz, _p]),
    "csship_ffn_wide": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _f32, _p]),
    "csship_ln_linear_wide": (_i32, [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _p]),
    "csship_linear_res_wide": (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _p]),
    "csship_ffn_mx": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _f32, _p]),
    "csship_head_pit_workspace": (_sz, [_i32, _i32]),
    "csship_head_pit": (_i32, [_p, _p, _p, _p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _sz, _p]),
    "csship_p2p_reduce": (_i32, [_p, _i32, _i64, _i64, _f32, _p, _p]),
    "csship_attn_dvdk": (_i32, [_p, _p, _i64, _p, _p, _p, _i32, _i32, _i32, _p]),
    "csship_attn_pos_bwd_workspace": (_sz, [_i32, _i32]),
    "csship_attn_pos_bwd": (_i32, [_p, _i64, _p, _i64, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _sz, _p]),
    "csship_attn_pos_bwd_ds": (_i32, [_p, _i64, _p, _i64, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _sz, _p]),
    "csship_attn_pos_bwd_ds_parts": (_i32, [_p, _i64, _p, _i64, _p, _i64, _p, _i32, _p, _p, _i32, _i32, _i32, _i32, _p, _sz, _p]),
    "csship_mha_relpos_bwd_fused_parts": (_i32, [_i32, _i32, _i32]),
    "csship_mha_relpos_bwd_fused": (_i32, [_p, _p, _i32, _i32, _p, _p, _p, _p, _i64, _p, _p, _i32, _i32, _i32, _dc, _p]),
    "csship_debug_timeline": (None, [_i32]),
    "csship_debug_timeline_read": (_i32, [_i32, C.POINTER(C.c_double), C.POINTER(_i32)]),
    "csship_debug_timeline_calibrate": (None, [_p, _i32]),
    "csship_debug_stamps": (None, [_p]),
}

_lib = None


def lib():
    """Load libcsship.so once; raise if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"css_amd: HIP library not found at {LIB_PATH}. Build it (make -C css_amd/csrc); "
                "there is no CPU fallback for the css hot path.")
        # torch FIRST: it ships its own libamdhip64, and libcsship.so (linked against the system ROCm's) must bind to the
        # runtime torch's streams and allocations belong to.  Loaded the other way round (build() then smoke() in one
        # process, before anything imported torch) the process holds two HIP runtimes and the first launch on a torch
        # stream fails with hipErrorInvalidHandle-class errors ("kernel launch failed").
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError(lib().csship_strerror(rc).decode() + f" (rc={rc})")
