"""ctypes binding of libcodetr_hip.so (C ABI: include/codetr_hip.h).

This is the only place the Python host crosses into native code.  The library is built
in-tree next to this file (``make -C co-detr-tensorrt_amd/csrc`` or ``__graft_entry__.build()``)
and, like the reference's ``codetr/__init__.py:8-12`` does for its CUDA extension, a missing
``.so`` is an ImportError -- there is no CPU or PyTorch fallback behind these calls.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcodetr_hip.so")
ABI_VERSION = 50

_i64, _i32, _vp, _cp = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p

# name -> (restype, argtypes); mirrors include/codetr_hip.h line by line
_MSDA_ARGS = [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i64, _i32, _i64, _vp]
SIGNATURES = {
    "codetr_hip_abi_version": (_i32, []),
    "codetr_hip_strerror": (_cp, [_i32]),
    "codetr_msda_forward_f16": (_i32, _MSDA_ARGS),
    "codetr_msda_forward_bf16": (_i32, _MSDA_ARGS),
    "codetr_msda_forward_f32": (_i32, _MSDA_ARGS),
    "codetr_msda_forward_f64": (_i32, _MSDA_ARGS),
    "codetr_msda_backward_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i64, _i32, _i64,
                                        _vp, _vp, _vp]),
    "codetr_msda_backward_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i64, _i32, _i64,
                                        _vp, _vp, _vp]),
    "codetr_msda_backward_f64": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i64, _i32, _i64,
                                        _vp, _vp, _vp]),
    "codetr_msda_variant": (_cp, [_i32, _i32, _i32, _i32, _i32]),
    "codetr_msda_fused_forward_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i32, _i32, _i64, _i64,
                                             _i32, _i32, _i32, _i64, _i32, _vp]),
    "codetr_msda_fused_forward_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i32, _i32, _i64, _i64,
                                              _i32, _i32, _i32, _i64, _i32, _vp]),
    "codetr_msda_fused_forward_ref32_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i32, _i32, _i64, _i64,
                                                   _i32, _i32, _i32, _i64, _i32, _vp]),
    "codetr_msda_fused_forward_ref32_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i32, _i32, _i64, _i64,
                                                    _i32, _i32, _i32, _i64, _i32, _vp]),
    "codetr_msda_encoder_forward_packed_f16": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _i32,
                                                      _vp, _i32, _i32, _i32, _i32, _vp]),
    "codetr_msda_encoder_forward_packed_bf16": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _i32,
                                                       _vp, _i32, _i32, _i32, _i32, _vp]),
    "codetr_linear_bf16_f16out": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32]),
    "codetr_encoder_projections_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i32]),
    "codetr_encoder_projections_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i32]),
    "codetr_encoder_projections_posgen_f16": (_i32, [_vp, _vp] + [_vp] * 10 + [_vp, _i32, _vp, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                                              ctypes.c_float, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i32]),
    "codetr_encoder_projections_posgen_bf16": (_i32, [_vp, _vp] + [_vp] * 10 + [_vp, _i32, _vp, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                                              ctypes.c_float, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i32]),
    "codetr_msda_encoder_packed_lds_bytes": (_i64, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32]),
    "codetr_msda_pack_projection_index": (_i32, [_i32, _i32, _i32, _vp]),
    "codetr_mx_scale_bytes": (_i64, [_i64, _i64]),
    "codetr_linear_fp8mx": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32]),
    "codetr_cast_fp8mx_f16": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64]),
    "codetr_layernorm_fp8mx_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, ctypes.c_float]),
    "codetr_window_attention_fp8mx_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32]),
    "codetr_mha_attention_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i64, _i64, _i64, _i64]),
    "codetr_mha_attention_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i64, _i64, _i64, _i64]),
    "codetr_linear_ln_f16": (_i32, [_vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _i64, _i64, _i64, _i32]),
    "codetr_linear_ln_bf16": (_i32, [_vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _i64, _i64, _i64, _i32]),
    "codetr_linear_xadd_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64]),
    "codetr_linear_xadd_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64]),
    "codetr_im2col_tokens_b16": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _vp]),
    "codetr_topk_chunks": (_i64, [_i64, _i32, _i64, _vp]),
    "codetr_topk_chunked_f16": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _i64]),
    "codetr_topk_chunked_bf16": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _i64]),
    "codetr_topk_f16": (_i32, [_vp, _vp, _i64, _i64, _i32, _vp, _vp]),
    "codetr_topk_bf16": (_i32, [_vp, _vp, _i64, _i64, _i32, _vp, _vp]),
    "codetr_patch_im2col_b16": (_i32, [_vp, _vp, _i64, _i32, _i64, _i64, _i32, _i32, _vp]),
    "codetr_linear_variant": (_cp, [_i64, _i64, _i64, _i32, _i32, _i32]),
    "codetr_add_f16": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64]),
    "codetr_add_bf16": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64]),
    "codetr_sigmoid_f16": (_i32, [_vp, _vp, _vp, _i64]),
    "codetr_sigmoid_bf16": (_i32, [_vp, _vp, _vp, _i64]),
    "codetr_gather_rows_b16": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64]),
    "codetr_decode_boxes_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, ctypes.c_float, ctypes.c_float]),
    "codetr_decode_boxes_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, ctypes.c_float, ctypes.c_float]),
    "codetr_valid_ratios_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32]),
    "codetr_valid_ratios_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32]),
    "codetr_linear_fp8": (_i32, [_vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _i32, ctypes.c_float, _i64, _i64, _i64,
                                 _i32]),
    "codetr_cast_fp8_f16": (_i32, [_vp, _vp, _vp, _i64, ctypes.c_float]),
    "codetr_layernorm_fp8_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, ctypes.c_float, ctypes.c_float]),
    "codetr_linear_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i64, _i32]),
    "codetr_linear_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i64, _i32]),
    "codetr_query_sine_embed_f16": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, ctypes.c_float, _i32, _vp, _vp,
                                           _vp]),
    "codetr_query_sine_embed_bf16": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, ctypes.c_float, _i32, _vp, _vp,
                                           _vp]),
    "codetr_encoder_geometry_f16": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "codetr_encoder_geometry_bf16": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "codetr_row_max_f16": (_i32, [_vp, _vp, _vp, _i64, _i64]),
    "codetr_row_max_bf16": (_i32, [_vp, _vp, _vp, _i64, _i64]),
    "codetr_preprocess_u8_f16": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "codetr_preprocess_u8_f32": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "codetr_batched_nms_f32": (_i32, [_vp, _vp, _vp, _i64, ctypes.c_float, _vp]),
    "codetr_patch_merge_layernorm_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, ctypes.c_float]),
    "codetr_patch_merge_layernorm_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, ctypes.c_float]),
    "codetr_mask_pyramid": (_i32, [_vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32]),
    "codetr_linear_splitk_plan": (_i32, [_i64, _i64, _i64, _vp]),
    "codetr_linear_splitk_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _i64]),
    "codetr_linear_splitk_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _i64]),
    "codetr_linear_sk_workspace_bytes": (_i64, []),
    "codetr_linear_sk_supported": (_i32, [_i64, _i64, _i64]),
    "codetr_linear_sk_preferred": (_i32, [_i64, _i64, _i64, _i32, _i32]),
    "codetr_linear_sk_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _i64, _i32]),
    "codetr_msda_op4_supported": (_i32, [_i32, _i64, _i64, _i32, _i32, _i32, _i64, _i32]),
    "codetr_msda_op4_forward_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i64, _i32, _vp]),
    "codetr_swin_mlp_supported": (_i32, [_i64, _i64, _i64]),
    "codetr_swin_mlp_f16": (_i32, [_vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _i64, _i64]),
    "codetr_swin_mlp_bf16": (_i32, [_vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _i64, _i64]),
    "codetr_linear_pp_supported": (_i32, [_i64, _i64, _i64]),
    "codetr_linear_pp_preferred": (_i32, [_i64, _i64, _i64, _i32, _i32]),
    "codetr_linear_pp_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32]),
    "codetr_linear_pp_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32]),
    "codetr_linear_sk_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _i64, _i32]),
    "codetr_decoder_layer_supported": (_i32, [_i32] * 7),
    "codetr_decoder_layer_blob_halfs": (_i64, [_i32] * 4),
    "codetr_decoder_layer_f16": (_i32, [_vp] * 18 + [_i64, _i64, _i64, _i32, _i32, _i32, ctypes.c_float, ctypes.c_float]),
    "codetr_decoder_layer_bf16": (_i32, [_vp] * 18 + [_i64, _i64, _i64, _i32, _i32, _i32, ctypes.c_float, ctypes.c_float]),
    "codetr_layernorm_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, ctypes.c_float]),
    "codetr_layernorm_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, ctypes.c_float]),
    "codetr_groupnorm_tokens_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "codetr_groupnorm_tokens_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _i32,
                                           ctypes.c_float]),
    "codetr_groupnorm_tokens_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _i32,
                                           ctypes.c_float]),
    "codetr_sine_pos_tokens_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, ctypes.c_float,
                                          ctypes.c_float, ctypes.c_float, ctypes.c_float, _i32]),
    "codetr_sine_pos_tokens_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, ctypes.c_float,
                                          ctypes.c_float, ctypes.c_float, ctypes.c_float, _i32]),
    "codetr_ffn_relu_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64]),
    "codetr_ffn_relu_ln_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, ctypes.c_float,
                                      _vp, _vp]),
    "codetr_ffn_relu_ln2_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, ctypes.c_float,
                                       _vp, _vp, ctypes.c_float, _vp, _vp]),
    "codetr_ffn_relu_ln2_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, ctypes.c_float,
                                       _vp, _vp, ctypes.c_float, _vp, _vp]),
    "codetr_ffn_pack_w2_f16": (_i32, [_vp, _vp, _vp, _i64, _i64]),
    "codetr_ffn_oproj_relu_ln2_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp,
                                             ctypes.c_float, _vp, _vp, ctypes.c_float, _vp, _vp]),
    "codetr_ffn_oproj_relu_ln2_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp,
                                              ctypes.c_float, _vp, _vp, ctypes.c_float, _vp, _vp]),
    "codetr_ffn_oproj_w1_index": (_i32, [_i64, _vp]),
    "codetr_ffn_fp8": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, ctypes.c_float, ctypes.c_float,
                              _vp, _vp, ctypes.c_float, _vp, _vp, ctypes.c_float, _vp, _vp]),
    "codetr_window_attention_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32]),
    "codetr_window_attention_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32]),
    "codetr_window_attention_ex": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _i64, _i64, _i64, _i32, _i32, _i32, _i32,
                                          _i32, _i32, _i32]),
    "codetr_window_attention_bias_index": (_i32, [_i32, _vp]),
    "codetr_window_attention_fp8out_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, ctypes.c_float, _i64, _i64, _i64, _i32, _i32,
                                                  _i32, _i32]),
}

_lib = None

# how many times each native entry point was enqueued in this process: lets tests and bench.py prove
# that the HIP kernels -- not a library path -- served a run
CALLS = {"encoder_projections_posgen": 0, "msda": 0, "msda_fused": 0, "linear": 0, "layernorm": 0, "window_attention": 0, "groupnorm_tokens": 0,
         "sine_pos_tokens": 0, "ffn_fused": 0, "ffn_oproj_fused": 0, "linear_splitk": 0, "linear_sk": 0, "mask_pyramid": 0,
         "query_sine_embed": 0, "encoder_geometry": 0, "row_max": 0, "preprocess": 0, "batched_nms": 0,
         "msda_backward": 0, "patch_merge_layernorm": 0, "msda_encoder": 0, "msda_encoder_packed": 0, "patch_im2col": 0, "mha_attention": 0, "topk": 0,
         # which kernel behind codetr_linear_* served a launch (codetr_linear_variant), and the two fused operand loads
         "linear_pp": 0, "swin_mlp": 0, "linear_tile128": 0, "linear_tile256": 0, "linear_xs": 0, "linear_ln": 0, "linear_xadd": 0, "encoder_projections": 0,
         "linear_fp8": 0, "cast_fp8": 0, "layernorm_fp8": 0, "small_ops": 0, "ffn_fp8": 0, "decoder_layer": 0}


# Launch recording (codetr/export.py): while RECORDER is a list, every launch-type entry point called through `load()`
# appends (name, args) -- the raw ctypes-level arguments: ints (device pointers, sizes), floats, None, ctypes arrays
# (host data such as level shapes).  Pure host queries (plans, variants, workspace sizes) are not launches.
RECORDER = None
_QUERIES = {"codetr_hip_abi_version", "codetr_hip_strerror", "codetr_msda_variant", "codetr_linear_variant",
            "codetr_topk_chunks", "codetr_linear_splitk_plan", "codetr_groupnorm_tokens_workspace_bytes",
            "codetr_linear_sk_workspace_bytes", "codetr_linear_sk_supported", "codetr_linear_sk_preferred",
            "codetr_linear_pp_supported", "codetr_linear_pp_preferred", "codetr_msda_op4_supported", "codetr_swin_mlp_supported",
            "codetr_msda_encoder_packed_lds_bytes", "codetr_msda_pack_projection_index", "codetr_window_attention_bias_index",
            "codetr_mx_scale_bytes", "codetr_decoder_layer_supported",
            "codetr_decoder_layer_blob_halfs"}


class _RecordingLib:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name in _QUERIES or name not in SIGNATURES:
            return fn

        def launch(*args):
            rc = fn(*args)
            # only launches that were accepted: a probe that the library turns down (E_UNSUPPORTED, then the caller
            # retries another form) enqueued nothing and must not reach an exported plan
            if RECORDER is not None and rc == 0:
                RECORDER.append((name, args))
            return rc

        return launch


_rec_lib = None


def load():
    """Load the shared library once; raise ImportError (loudly) if it is absent or stale."""
    global _lib, _rec_lib
    if _lib is not None:
        if RECORDER is None:
            return _lib
        if _rec_lib is None:
            _rec_lib = _RecordingLib(_lib)
        return _rec_lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            f"HIP extension not found at {LIB_PATH}; build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C co-detr-tensorrt_amd/csrc`"
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export {name}; rebuild the extension") from e
        fn.restype = res
        fn.argtypes = args
    ver = lib.codetr_hip_abi_version()
    if ver != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {ver}, host expects {ABI_VERSION}; rebuild the extension")
    _lib = lib
    return lib


def strerror(code: int) -> str:
    return load().codetr_hip_strerror(code).decode()


def check(code: int, what: str):
    if code != 0:
        raise RuntimeError(f"{what} failed: {strerror(code)} (code {code})")


def current_stream_ptr(device) -> int:
    """The raw hipStream_t of torch's current stream on `device` (plumbing only)."""
    return torch.cuda.current_stream(device).cuda_stream


_MSDA_BY_DTYPE = {
    torch.float16: ("codetr_msda_forward_f16", 2),
    torch.bfloat16: ("codetr_msda_forward_bf16", 2),
    torch.float32: ("codetr_msda_forward_f32", 4),
    torch.float64: ("codetr_msda_forward_f64", 8),
}


def msda_variant(dtype, M, D, L, P) -> str:
    return load().codetr_msda_variant(_MSDA_BY_DTYPE[dtype][1], M, D, L, P).decode()


def msda_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step, out):
    """Enqueue the MSDA forward on torch's current stream.  All tensors must already satisfy the
    contract checked in ops.py (contiguous, on one HIP device, one float dtype, int64 shapes)."""
    lib = load()
    CALLS["msda"] += 1
    name, _ = _MSDA_BY_DTYPE[value.dtype]
    B, S, M, D = value.shape
    Nq, L, P = sampling_loc.shape[1], sampling_loc.shape[3], sampling_loc.shape[4]
    rc = getattr(lib, name)(
        current_stream_ptr(value.device),
        value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
        sampling_loc.data_ptr(), attn_weight.data_ptr(),
        B, S, M, D, L, Nq, P, int(im2col_step), out.data_ptr(),
    )
    if rc == -2:
        # same condition and wording as the reference's AT_ASSERTM (ms_deform_attn.cu:924-926)
        raise RuntimeError(f"batch({B}) must divide im2col_step({min(B, int(im2col_step))})")
    check(rc, name)
    return out


_ACT = {None: 0, "relu": 1, "gelu": 2}
_LINEAR_BY_DTYPE = {torch.float16: "codetr_linear_f16", torch.bfloat16: "codetr_linear_bf16"}


def linear_supported(x, weight) -> bool:
    """Shapes / dtypes the native fused linear implements (K % 64 == 0, f16 / bf16)."""
    return x.dtype in _LINEAR_BY_DTYPE and weight.dtype == x.dtype and weight.shape[1] % 64 == 0


_VARIANTS = {}


def linear_variant(M, N, K, act=None, has_residual=False, hm_head_dim=0) -> str:
    """'tile128' | 'tile256' | 'xs': the kernel codetr_linear_* launches for this (aligned) problem"""
    key = (M, N, K, act, bool(has_residual), hm_head_dim)
    v = _VARIANTS.get(key)
    if v is None:
        v = _VARIANTS[key] = load().codetr_linear_variant(M, N, K, _ACT[act], 1 if has_residual else 0,
                                                          hm_head_dim).decode()
    return v


def linear(x2d, weight, bias, residual2d, act, out2d, row_mask=None, hm_rows=0, hm_head_dim=0):
    """Enqueue y = act(x @ w.T + b) (+ r) on torch's current stream.  x2d [M,K], weight [N,K] contiguous;
    row_mask [M] bool/uint8: masked rows are written as zeros; hm_rows/hm_head_dim: head-major destination
    y[b][head][position][channel] (see include/codetr_hip.h)."""
    lib = load()
    CALLS["linear"] += 1
    M, K = x2d.shape
    N = weight.shape[0]
    variant = linear_variant(M, N, K, act, residual2d is not None, hm_head_dim)
    if variant != "unsupported":   # (the launch below reports the unsupported shape with its own error)
        CALLS["linear_" + variant] += 1
    rc = getattr(lib, _LINEAR_BY_DTYPE[x2d.dtype])(
        current_stream_ptr(x2d.device), x2d.data_ptr(), weight.data_ptr(),
        bias.data_ptr() if bias is not None else None,
        residual2d.data_ptr() if residual2d is not None else None,
        row_mask.data_ptr() if row_mask is not None else None,
        out2d.data_ptr(), M, N, K, _ACT[act], hm_rows, hm_head_dim)
    check(rc, "codetr_linear")
    return out2d


_SK_PREFERRED = {}


def linear_sk_preferred(M, N, K, act=None, has_residual=False) -> bool:
    """True where the persistent GEMM (csrc/gemm_sk.hip) measured faster than codetr_linear_* (the library's own rule)"""
    key = (M, N, K, act, bool(has_residual))
    v = _SK_PREFERRED.get(key)
    if v is None:
        v = _SK_PREFERRED[key] = bool(load().codetr_linear_sk_preferred(M, N, K, _ACT[act], 1 if has_residual else 0))
    return v


_SK_WORKSPACE = {}


def linear_sk_workspace(device):
    """the zero-filled workspace of the stream-K split (flags & 0x40 only), one per device and stream"""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _SK_WORKSPACE.get(key)
    if ws is None:
        with torch.cuda.device(device):
            ws = _SK_WORKSPACE[key] = torch.zeros(load().codetr_linear_sk_workspace_bytes(), dtype=torch.uint8, device=device)
    return ws


def linear_sk(x2d, weight, bias, residual2d, act, out2d, flags=0):
    """y = act(x @ w.T + b) (+ r) by the persistent 256-tile GEMM; flags as in include/codetr_hip.h (0 = default;
    0x40 = stream-K split of the left-over tiles, which needs the workspace)"""
    lib = load()
    CALLS["linear"] += 1
    CALLS["linear_sk"] += 1
    M, K = x2d.shape
    N = weight.shape[0]
    ws = linear_sk_workspace(x2d.device) if flags & 0x40 else None
    fn = lib.codetr_linear_sk_f16 if x2d.dtype == torch.float16 else lib.codetr_linear_sk_bf16
    rc = fn(current_stream_ptr(x2d.device), x2d.data_ptr(), weight.data_ptr(),
            bias.data_ptr() if bias is not None else None,
            residual2d.data_ptr() if residual2d is not None else None, out2d.data_ptr(), M, N, K, _ACT[act],
            ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, flags)
    check(rc, "codetr_linear_sk")
    return out2d


def swin_mlp(x2d, ln_w, ln_b, eps, w1, b1, w2_packed, b2, out2d):
    """out = x + fc2(gelu(fc1(layer_norm(x)))) in one launch (csrc/swin_mlp.hip); w2_packed = ffn_pack_w2(fc2.weight)"""
    CALLS["swin_mlp"] += 1
    lib = load()
    fn = lib.codetr_swin_mlp_bf16 if x2d.dtype == torch.bfloat16 else lib.codetr_swin_mlp_f16
    rc = fn(current_stream_ptr(x2d.device), x2d.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), float(eps), w1.data_ptr(),
            b1.data_ptr(), w2_packed.data_ptr(), b2.data_ptr(), out2d.data_ptr(), x2d.shape[0], x2d.shape[1])
    check(rc, "codetr_swin_mlp")
    return out2d


_PP_PREFERRED = {}


def linear_pp_preferred(M, N, K, act=None, has_residual=False) -> bool:
    """True where the ping-pong GEMM (csrc/gemm_pp.hip) measured faster than the other two (the library's own rule)"""
    key = (M, N, K, act, bool(has_residual))
    v = _PP_PREFERRED.get(key)
    if v is None:
        v = _PP_PREFERRED[key] = bool(load().codetr_linear_pp_preferred(M, N, K, _ACT[act], 1 if has_residual else 0))
    return v


def linear_pp(x2d, weight, bias, residual2d, act, out2d, flags=0):
    """y = act(x @ w.T + b) (+ r) by the ping-pong persistent GEMM; flags as in include/codetr_hip.h"""
    lib = load()
    CALLS["linear"] += 1
    CALLS["linear_pp"] += 1
    M, K = x2d.shape
    N = weight.shape[0]
    fn = lib.codetr_linear_pp_f16 if x2d.dtype == torch.float16 else lib.codetr_linear_pp_bf16
    rc = fn(current_stream_ptr(x2d.device), x2d.data_ptr(), weight.data_ptr(),
            bias.data_ptr() if bias is not None else None,
            residual2d.data_ptr() if residual2d is not None else None, out2d.data_ptr(), M, N, K, _ACT[act], flags)
    check(rc, "codetr_linear_pp")
    return out2d


def query_sine_embed(ref, valid_ratios, pos_feat, temperature=10000.0, apply_sigmoid=True, valid_ratios32=None):
    """ref [B,Nq,2|4] f16 (unactivated), valid_ratios [B,L,2] f16 -> (ref_in [B,Nq,L,d] f16, embed [B,Nq,d*pos_feat] f16,
    ref_in32); valid_ratios32 [B,L,2] fp32: ref_in32 [B,Nq,L,d] fp32 = the same points unrounded (else None)"""
    CALLS["query_sine_embed"] += 1
    B, Nq, d = ref.shape
    L = valid_ratios.shape[1]
    ref_in = torch.empty((B, Nq, L, d), dtype=ref.dtype, device=ref.device)
    ref_in32 = torch.empty((B, Nq, L, d), dtype=torch.float32, device=ref.device) if valid_ratios32 is not None else None
    embed = torch.empty((B, Nq, d * pos_feat), dtype=ref.dtype, device=ref.device)
    fn = load().codetr_query_sine_embed_bf16 if ref.dtype == torch.bfloat16 else load().codetr_query_sine_embed_f16
    rc = fn(current_stream_ptr(ref.device), ref.data_ptr(), valid_ratios.data_ptr(),
                                            valid_ratios32.data_ptr() if valid_ratios32 is not None else None, B, Nq,
                                            d, L, pos_feat, float(temperature), 1 if apply_sigmoid else 0,
                                            ref_in.data_ptr(), ref_in32.data_ptr() if ref_in32 is not None else None,
                                            embed.data_ptr())
    check(rc, "codetr_query_sine_embed_f16")
    return ref_in, embed, ref_in32


_MSDA_BWD = {torch.float16: "codetr_msda_backward_f16", torch.float32: "codetr_msda_backward_f32",
             torch.float64: "codetr_msda_backward_f64"}


def msda_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, grad_value,
                  grad_sampling_loc, grad_attn_weight, im2col_step):
    """gradients accumulated into the three (pre-zeroed) output tensors, see include/codetr_hip.h"""
    if value.dtype not in _MSDA_BWD:
        raise RuntimeError(f"multi_scale_deformable_attention_backward: unsupported dtype {value.dtype} (f16 / f32 / f64)")
    CALLS["msda_backward"] += 1
    B, S, M, D = value.shape
    Nq, L, P = sampling_loc.shape[1], sampling_loc.shape[3], sampling_loc.shape[4]
    rc = getattr(load(), _MSDA_BWD[value.dtype])(
        current_stream_ptr(value.device), value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
        sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(), B, S, M, D, L, Nq, P, int(im2col_step),
        grad_value.data_ptr(), grad_sampling_loc.data_ptr(), grad_attn_weight.data_ptr())
    if rc == -2:
        step = min(B, int(im2col_step))
        raise RuntimeError(f"batch({B}) must divide im2col_step({step})")
    check(rc, "codetr_msda_backward")


def encoder_geometry(valid_ratios, mask_flat, shapes):
    """-> (reference_points [B,S,2], reference_by_level [B,S,L,2], proposals [B,S,4]) f16, row_state [B,S] uint8"""
    CALLS["encoder_geometry"] += 1
    B, S = mask_flat.shape
    L = len(shapes)
    dev = mask_flat.device
    dt = valid_ratios.dtype
    ref = torch.empty((B, S, 2), dtype=dt, device=dev)
    ref_lvl = torch.empty((B, S, L, 2), dtype=dt, device=dev)
    prop = torch.empty((B, S, 4), dtype=dt, device=dev)
    state = torch.empty((B, S), dtype=torch.uint8, device=dev)
    hw = (ctypes.c_int64 * (2 * L))(*[int(v) for s in shapes for v in s])
    fn = load().codetr_encoder_geometry_bf16 if valid_ratios.dtype == torch.bfloat16 else load().codetr_encoder_geometry_f16
    rc = fn(current_stream_ptr(dev), valid_ratios.data_ptr(), mask_flat.data_ptr(), B, L, hw,
                                            ref.data_ptr(), ref_lvl.data_ptr(), prop.data_ptr(), state.data_ptr())
    check(rc, "codetr_encoder_geometry_f16")
    return ref, ref_lvl, prop, state


def row_max(x2d):
    CALLS["row_max"] += 1
    rows, C = x2d.shape
    out = torch.empty((rows,), dtype=x2d.dtype, device=x2d.device)
    fn = load().codetr_row_max_bf16 if x2d.dtype == torch.bfloat16 else load().codetr_row_max_f16
    rc = fn(current_stream_ptr(x2d.device), x2d.data_ptr(), out.data_ptr(), rows, C)
    check(rc, "codetr_row_max_f16")
    return out


def preprocess_u8(src, resized_hw, pad_hw, mean, std, pad_value, dtype, with_mask=True):
    """src [H, W, 3] uint8 on the device -> (inputs [3, Hp, Wp] dtype, mask [Hp, Wp] dtype or None)"""
    CALLS["preprocess"] += 1
    if dtype not in (torch.float16, torch.float32):
        raise RuntimeError("preprocess_u8 writes f16 or f32")
    Hs, Ws, _ = src.shape
    (Hr, Wr), (Hp, Wp) = resized_hw, pad_hw
    dst = torch.empty((3, Hp, Wp), dtype=dtype, device=src.device)
    mask = torch.empty((Hp, Wp), dtype=dtype, device=src.device) if with_mask else None
    f3 = ctypes.c_float * 3
    fn = load().codetr_preprocess_u8_f16 if dtype == torch.float16 else load().codetr_preprocess_u8_f32
    rc = fn(current_stream_ptr(src.device), src.data_ptr(), Hs, Ws, Hr, Wr, Hp, Wp, f3(*[float(v) for v in mean]),
            f3(*[float(v) for v in std]), (ctypes.c_int * 3)(*[int(v) for v in pad_value]), dst.data_ptr(),
            mask.data_ptr() if mask is not None else None)
    check(rc, "codetr_preprocess_u8")
    return dst, mask


def batched_nms_sorted(boxes_sorted, labels_sorted, iou_threshold):
    """boxes [N,4] fp32 / labels [N] int64 in descending score order -> keep flags [N] bool"""
    CALLS["batched_nms"] += 1
    N = boxes_sorted.shape[0]
    keep = torch.empty((N,), dtype=torch.bool, device=boxes_sorted.device)
    rc = load().codetr_batched_nms_f32(current_stream_ptr(boxes_sorted.device), boxes_sorted.data_ptr(),
                                       labels_sorted.data_ptr(), N, float(iou_threshold), keep.data_ptr())
    check(rc, "codetr_batched_nms_f32")
    return keep


def mask_pyramid(img_masks, shapes):
    """img_masks [B,H,W] bool/uint8 -> (mask_flat [B,S] bool, ycum, xcum (flat fp32, level l = [B,H_l,W_l] at
    B*start_l), valid_counts [B,L,2] fp32); see include/codetr_hip.h"""
    CALLS["mask_pyramid"] += 1
    B, Hi, Wi = img_masks.shape
    L = len(shapes)
    S = sum(int(h) * int(w) for h, w in shapes)
    dev = img_masks.device
    mask_flat = torch.empty((B, S), dtype=torch.bool, device=dev)
    cums = torch.empty((2, B * S), dtype=torch.float32, device=dev)
    counts = torch.empty((B, L, 2), dtype=torch.float32, device=dev)
    hw = (ctypes.c_int64 * (2 * L))(*[int(v) for s in shapes for v in s])
    rc = load().codetr_mask_pyramid(current_stream_ptr(dev), img_masks.data_ptr(), B, Hi, Wi, L, hw,
                                    mask_flat.data_ptr(), cums[0].data_ptr(), cums[1].data_ptr(), counts.data_ptr(),
                                    img_masks.element_size())
    check(rc, "codetr_mask_pyramid")
    return mask_flat, cums[0], cums[1], counts


_SPLITK_PLANS = {}


def linear_splitk_plan(M, N, K):
    """(splits, workspace_bytes) the library wants for this problem; splits == 1 -> single-pass codetr_linear_*"""
    key = (M, N, K)
    if key not in _SPLITK_PLANS:
        nbytes = ctypes.c_int64(0)
        splits = load().codetr_linear_splitk_plan(M, N, K, ctypes.byref(nbytes))
        _SPLITK_PLANS[key] = (int(splits), int(nbytes.value))
    return _SPLITK_PLANS[key]


def linear_splitk(x2d, weight, bias, residual2d, act, out2d, splits, workspace, row_mask=None):
    """Two-pass split-K form of `linear` (few output tiles, long K); workspace: uint8 tensor from the plan."""
    lib = load()
    CALLS["linear"] += 1
    CALLS["linear_splitk"] += 1
    M, K = x2d.shape
    N = weight.shape[0]
    fn = lib.codetr_linear_splitk_f16 if x2d.dtype == torch.float16 else lib.codetr_linear_splitk_bf16
    rc = fn(current_stream_ptr(x2d.device), x2d.data_ptr(), weight.data_ptr(),
            bias.data_ptr() if bias is not None else None,
            residual2d.data_ptr() if residual2d is not None else None,
            row_mask.data_ptr() if row_mask is not None else None,
            out2d.data_ptr(), M, N, K, _ACT[act], splits, workspace.data_ptr(), workspace.numel())
    check(rc, "codetr_linear_splitk")
    return out2d


_LN_BY_DTYPE = {torch.float16: "codetr_layernorm_f16", torch.bfloat16: "codetr_layernorm_bf16"}


def layernorm_supported(x, weight) -> bool:
    C = x.shape[-1]
    return x.dtype in _LN_BY_DTYPE and weight is not None and weight.dtype == x.dtype and C % 8 == 0 and C <= 4096


def patch_merge_layernorm(x4d, weight_kkc, bias_kkc, eps):
    """x4d [B,H,W,C] f16 contiguous -> LayerNorm of the 2x2-merged rows [B, H2*W2, 4C] ((ky, kx, c) order)"""
    CALLS["patch_merge_layernorm"] += 1
    B, H, W, C = x4d.shape
    out = torch.empty((B, ((H + 1) // 2) * ((W + 1) // 2), 4 * C), dtype=x4d.dtype, device=x4d.device)
    lib = load()
    fn = lib.codetr_patch_merge_layernorm_bf16 if x4d.dtype == torch.bfloat16 else lib.codetr_patch_merge_layernorm_f16
    rc = fn(current_stream_ptr(x4d.device), x4d.data_ptr(), weight_kkc.data_ptr(), bias_kkc.data_ptr(), out.data_ptr(),
            B, H, W, C, float(eps))
    check(rc, "codetr_patch_merge_layernorm")
    return out


def layernorm(x2d, weight, bias, eps, out2d):
    rows, C = x2d.shape
    CALLS["layernorm"] += 1
    rc = getattr(load(), _LN_BY_DTYPE[x2d.dtype])(
        current_stream_ptr(x2d.device), x2d.data_ptr(), weight.data_ptr(), bias.data_ptr(), out2d.data_ptr(),
        rows, C, float(eps))
    check(rc, "codetr_layernorm")
    return out2d


def window_attention_supported(dtype, embed_dims, num_heads, window_size) -> bool:
    return dtype in (torch.float16, torch.bfloat16) and embed_dims == num_heads * 32 and window_size in (4, 7, 8, 12)


def window_attention_bias_index(window_size):
    """source key of every position of a lane-order bias row (codetr_window_attention_bias_index), or None where the
    window size has no lane order"""
    idx = (ctypes.c_int32 * (window_size * window_size))()
    rc = load().codetr_window_attention_bias_index(int(window_size), idx)
    if rc == E_UNSUPPORTED:
        return None
    check(rc, "codetr_window_attention_bias_index")
    return list(idx)


def window_attention(qkv, qkv_bias, rel_bias, out, B, H, W, num_heads, window_size, shift, out_scale=None,
                     out_scales=None, bias_layout=0):
    """out: 16-bit like qkv, or (with out_scale, f16 qkv) torch.float8_e4m3fn = sat(f16(o) / out_scale), or (with
    out_scales, a uint8 MX scale tensor) block-scaled e4m3.  bias_layout 1: rel_bias rows in the kernel's lane order
    (window_attention_bias_index)."""
    CALLS["window_attention"] += 1
    lib = load()
    mode = 2 if out_scales is not None else 1 if out_scale is not None else 0
    rc = lib.codetr_window_attention_ex(current_stream_ptr(qkv.device), qkv.data_ptr(), qkv_bias.data_ptr(), rel_bias.data_ptr(),
                                        out.data_ptr(), out_scales.data_ptr() if out_scales is not None else None,
                                        float(out_scale) if out_scale is not None else 1.0, B, H, W, num_heads, 32,
                                        window_size, shift, 1 if qkv.dtype == torch.bfloat16 else 0, mode, int(bias_layout))
    check(rc, "codetr_window_attention_ex")
    return out


def encoder_projections_posgen(x2d, S, cums, level_shapes, level_embed, temperature, scale, eps, offset, normalize, w_cat,
                               bias_cat, row_mask, value_out, packed_out, hm_rows=0, hm_head_dim=0) -> bool:
    """encoder_projections with the positional operand generated in the kernel (codetr_encoder_projections_posgen_*):
    cums = [(ycum_l, xcum_l)] per level, contiguous [B, H_l, W_l] fp32; level_embed [L, K] in x's type or None.
    False when the library declines the shape."""
    lib = load()
    M, K = x2d.shape
    Np = packed_out.shape[1]
    Nv = w_cat.shape[0] - Np
    L = len(level_shapes)
    if (x2d.dtype not in (torch.float16, torch.bfloat16) or value_out.dtype != torch.float16 or packed_out.dtype != x2d.dtype
            or w_cat.dtype != x2d.dtype or bias_cat.dtype != x2d.dtype or value_out.numel() != M * Nv or L > 5 or len(cums) != L
            or (level_embed is not None and (level_embed.dtype != x2d.dtype or not level_embed.is_contiguous()
                                             or tuple(level_embed.shape) != (L, K)))):
        raise ValueError("encoder_projections_posgen: operand types / shapes")
    B = M // S
    ptrs = []
    for which in (0, 1):
        for l in range(5):
            if l < L:
                t = cums[l][which]
                if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != B * level_shapes[l][0] * level_shapes[l][1]:
                    raise ValueError("encoder_projections_posgen: running sums must be contiguous [B, H_l, W_l] fp32")
                ptrs.append(t.data_ptr())
            else:
                ptrs.append(None)
    shapes = (ctypes.c_int64 * (2 * L))(*[int(v) for hw in level_shapes for v in hw])
    fn = lib.codetr_encoder_projections_posgen_bf16 if x2d.dtype == torch.bfloat16 else lib.codetr_encoder_projections_posgen_f16
    rc = fn(current_stream_ptr(x2d.device), x2d.data_ptr(), *ptrs, shapes, L,   # (the ctypes array itself: export.py records it as host bytes)
            level_embed.data_ptr() if level_embed is not None else None, float(temperature), float(scale), float(eps),
            float(offset), int(bool(normalize)), w_cat.data_ptr(), bias_cat.data_ptr(),
            row_mask.data_ptr() if row_mask is not None else None, value_out.data_ptr(), packed_out.data_ptr(), M, S, Nv, Np, K,
            hm_rows, hm_head_dim)
    if rc == E_UNSUPPORTED:
        return False
    check(rc, "codetr_encoder_projections_posgen")
    CALLS["linear"] += 1
    CALLS["encoder_projections"] += 1
    CALLS["encoder_projections_posgen"] += 1
    return True


_MSDA_FUSED_BY_DTYPE = {torch.float16: "codetr_msda_fused_forward_f16", torch.bfloat16: "codetr_msda_fused_forward_bf16"}


def msda_fused_supported(dtype, D, L, P) -> bool:
    return dtype in _MSDA_FUSED_BY_DTYPE and D in (16, 32, 64) and L * P * (256 // (D // 8)) * 32 <= 60 * 1024


_MSDA_FUSED_REF32_BY_DTYPE = {torch.float16: "codetr_msda_fused_forward_ref32_f16",
                              torch.bfloat16: "codetr_msda_fused_forward_ref32_bf16"}


def msda_fused(value, spatial_shapes, level_start_index, proj, off_col, logit_col, ref, num_levels, num_points, out,
               head_major=False):
    """value [B,S,M,D] (or [B,M,S,D] when head_major); proj [B,Nq,Ncols] holds the sampling offsets at columns [off_col, off_col+M*L*P*2) and
    the attention logits at [logit_col, logit_col+M*L*P); ref [B,Nq,L,2|4] in value's dtype or fp32; out [B,Nq,M*D]."""
    lib = load()
    CALLS["msda_fused"] += 1
    if head_major:
        B, M, S, D = value.shape
    else:
        B, S, M, D = value.shape
    Nq, ncols = proj.shape[1], proj.shape[2]
    es = proj.element_size()
    table = _MSDA_FUSED_REF32_BY_DTYPE if ref.dtype == torch.float32 else _MSDA_FUSED_BY_DTYPE
    rc = getattr(lib, table[value.dtype])(
        current_stream_ptr(value.device), value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
        proj.data_ptr() + off_col * es, ncols, proj.data_ptr() + logit_col * es, ncols,
        ref.data_ptr(), ref.shape[-1], 1 if head_major else 0, B, S, M, D, num_levels, Nq, num_points, out.data_ptr())
    check(rc, "codetr_msda_fused_forward")
    return out


E_UNSUPPORTED = -4


def _windows_array(windows, M, L):
    """[M][L][4] (x lo, x hi, y lo, y hi) -> ctypes int8 array; an int h means the symmetric halo (-h, h, -h, h)"""
    if isinstance(windows, int):
        windows = [[(-windows, windows, -windows, windows)] * L] * M
    flat = [int(v) for head in windows for lvl in head for v in lvl]
    if len(flat) != M * L * 4 or any(not -128 <= v <= 127 for v in flat):
        raise ValueError(f"msda_encoder: windows must be [M={M}][L={L}][4] int8 values")
    return (ctypes.c_int8 * len(flat))(*flat)


def linear_bf16_f16out(x2d, weight, bias, out2d, row_mask=None, hm_rows=0, hm_head_dim=0) -> bool:
    """bf16 x [M,K] @ bf16 w [N,K]^T (+ bias) -> FP16 out [M,N] (row mask / head-major destination as `linear`): the value
    projection of a bf16 model in front of the packed encoder MSDA kernel.  False when the library has no kernel for the
    shape (the X-stationary kernel's: K in {192, 256}, >= 32 768 rows)."""
    M, K = x2d.shape
    N = weight.shape[0]
    rc = load().codetr_linear_bf16_f16out(current_stream_ptr(x2d.device), x2d.data_ptr(), weight.data_ptr(),
                                          bias.data_ptr() if bias is not None else None,
                                          row_mask.data_ptr() if row_mask is not None else None, out2d.data_ptr(), M, N, K,
                                          hm_rows, hm_head_dim)
    if rc == E_UNSUPPORTED:
        return False
    check(rc, "codetr_linear_bf16_f16out")
    CALLS["linear"] += 1
    CALLS["linear_xs"] += 1
    return True


def encoder_projections(x2d, pos2d, w_cat, bias_cat, row_mask, value_out, packed_out, hm_rows=0, hm_head_dim=0) -> bool:
    """ONE launch: value_out = x @ w_cat[:Nv]^T + b[:Nv] (row mask, head-major destination as `linear`; FP16 for a bf16
    model) and packed_out [M, Np] = (x + pos) @ w_cat[Nv:]^T + b[Nv:] (include/codetr_hip.h codetr_encoder_projections_*).
    Nv = value_out.numel() / M, Np = packed_out.shape[1].  False when the library declines the shape."""
    lib = load()
    M, K = x2d.shape
    Np = packed_out.shape[1]
    Nv = w_cat.shape[0] - Np
    if x2d.dtype == torch.bfloat16:
        fn, vt = lib.codetr_encoder_projections_bf16, torch.float16
    else:
        fn, vt = lib.codetr_encoder_projections_f16, torch.float16
    if (x2d.dtype not in (torch.float16, torch.bfloat16) or value_out.dtype != vt or packed_out.dtype != x2d.dtype
            or pos2d.dtype != x2d.dtype or w_cat.dtype != x2d.dtype or bias_cat.dtype != x2d.dtype
            or value_out.numel() != M * Nv or pos2d.shape != x2d.shape):
        raise ValueError("encoder_projections: operand types / shapes")
    rc = fn(current_stream_ptr(x2d.device), x2d.data_ptr(), pos2d.data_ptr(), w_cat.data_ptr(), bias_cat.data_ptr(),
            row_mask.data_ptr() if row_mask is not None else None, value_out.data_ptr(), packed_out.data_ptr(), M, Nv, Np, K,
            hm_rows, hm_head_dim)
    if rc == E_UNSUPPORTED:
        return False
    check(rc, "codetr_encoder_projections")
    CALLS["linear"] += 1
    CALLS["encoder_projections"] += 1
    return True


def msda_encoder_packed_lds_bytes(level_shapes, M, num_points, windows, region, threads) -> int:
    """LDS bytes per workgroup of codetr_msda_encoder_forward_packed_f16 (negative: CODETR_E_* code)"""
    L = len(level_shapes)
    shapes = (ctypes.c_int64 * (2 * L))(*[int(v) for hw in level_shapes for v in hw])
    return int(load().codetr_msda_encoder_packed_lds_bytes(shapes, M, L, num_points, _windows_array(windows, M, L),
                                                           int(region[0]), int(region[1]), int(threads)))


def msda_pack_projection_index(M, L, P):
    """source row of the concatenated (sampling_offsets | attention_weights) projection for every column of the
    lane-major packed layout (list of 64 M ints, -1 = pad), or None when the library has no packed kernel for (L, P)"""
    idx = (ctypes.c_int32 * (64 * M))()
    rc = load().codetr_msda_pack_projection_index(M, L, P, idx)
    if rc == E_UNSUPPORTED:
        return None
    check(rc, "codetr_msda_pack_projection_index")
    return list(idx)


def msda_encoder_packed(value, level_shapes, packed, num_points, windows, valid_counts, region, threads, out,
                        head_major=False) -> bool:
    """Round-5 encoder kernel: value [B,S,M,32] fp16 ([B,M,S,32] with head_major); packed [B,S,>=64 M] fp16 (lane-major packed projection);
    valid_counts [B,L,2] fp32; windows [M][L][4]; region (w, h) in finest-level pixels; threads 256 | 512.  Returns
    False when the library reports the shape as unsupported, raises on any other error."""
    lib = load()
    if head_major:
        B, M, S, D = value.shape
    else:
        B, S, M, D = value.shape
    L = len(level_shapes)
    shapes = (ctypes.c_int64 * (2 * L))(*[int(v) for hw in level_shapes for v in hw])
    fn = (lib.codetr_msda_encoder_forward_packed_bf16 if packed.dtype == torch.bfloat16
          else lib.codetr_msda_encoder_forward_packed_f16)   # (bf16: packed / out bf16, the value map fp16 -- see the header)
    rc = fn(
        current_stream_ptr(value.device), value.data_ptr(), shapes, packed.data_ptr(), packed.shape[-1],
        valid_counts.data_ptr(), B, S, M, D, L, num_points, _windows_array(windows, M, L), int(region[0]), int(region[1]),
        int(threads), 1 if head_major else 0, out.data_ptr())
    if rc == E_UNSUPPORTED:
        return False
    check(rc, "codetr_msda_encoder_forward_packed_f16")
    CALLS["msda_encoder"] += 1
    CALLS["msda_encoder_packed"] += 1
    return True


def mha_attention_supported(q, k, v, num_heads) -> bool:
    C = q.shape[-1]
    return (q.dtype in (torch.float16, torch.bfloat16) and k.dtype == q.dtype and v.dtype == q.dtype
            and C == num_heads * 32 and k.shape[1] <= 1024 and q.stride(-1) == 1 and k.stride(-1) == 1
            and v.stride(-1) == 1 and all(t.stride(0) == t.shape[1] * t.stride(1) for t in (q, k, v))
            and all(t.stride(1) % 8 == 0 and t.data_ptr() % 16 == 0 for t in (q, k, v)))


def mha_attention(q, k, v, num_heads, out):
    """q [B,Nq,C], k / v [B,Nk,C] (rows may be strided views of a wider projection output), out [B,Nq,C] contiguous."""
    lib = load()
    CALLS["mha_attention"] += 1
    fn = lib.codetr_mha_attention_bf16 if q.dtype == torch.bfloat16 else lib.codetr_mha_attention_f16
    rc = fn(current_stream_ptr(q.device), q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), q.shape[0],
            q.shape[1], k.shape[1], num_heads, 32, q.stride(1), k.stride(1), v.stride(1), out.stride(1))
    check(rc, "codetr_mha_attention")
    return out


def decoder_layer_supported(embed_dims, num_heads, num_levels, num_points, hidden, ref_dim, pos_feat) -> bool:
    return bool(load().codetr_decoder_layer_supported(embed_dims, num_heads, num_levels, num_points, hidden, ref_dim, pos_feat))


def decoder_layer_blob_halfs(which, num_levels, num_points, hidden) -> int:
    """element count of a packed weight blob of codetr_decoder_layer_f16: which = 0 tail | 1 head | 2 pos | 3 final norm"""
    return int(load().codetr_decoder_layer_blob_halfs(which, num_levels, num_points, hidden))


def decoder_layer(x, attn, qpos, ref, vr32, value, shapes, starts, tail_w, pos_w, head_w, final_norm, x_out, ref_out,
                  qpos_out, qk_out, v_out, B, Nq, S, L, P, hidden, eps, temperature):
    """one launch of codetr_decoder_layer_{f16,bf16} (include/codetr_hip.h); tensors or None, see the header for the roles"""
    CALLS["decoder_layer"] += 1
    p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    fn = load().codetr_decoder_layer_bf16 if x.dtype == torch.bfloat16 else load().codetr_decoder_layer_f16
    rc = fn(current_stream_ptr(x.device), p(x), p(attn), p(qpos), p(ref), p(vr32), p(value),
            p(shapes), p(starts), p(tail_w), p(pos_w), p(head_w), p(final_norm), p(x_out),
            p(ref_out), p(qpos_out), p(qk_out), p(v_out), B, Nq, S, L, P, hidden, float(eps), float(temperature))
    check(rc, "codetr_decoder_layer")


def linear_xadd_supported(M, N, K, dtype) -> bool:
    """mirror of the library's rule for codetr_linear_xadd_* (the X-stationary kernel's shapes)"""
    return (dtype in (torch.float16, torch.bfloat16) and K in (192, 256) and N % 8 == 0 and 128 <= N <= 1536
            and M >= 128 * 256)


def linear_ln(x2d, gamma, beta, eps, w, bias, act, out2d) -> bool:
    """out = act(LayerNorm(x) @ w.T + bias); False when the library declines the shape"""
    lib = load()
    fn = lib.codetr_linear_ln_bf16 if x2d.dtype == torch.bfloat16 else lib.codetr_linear_ln_f16
    M, K = x2d.shape
    rc = fn(current_stream_ptr(x2d.device), x2d.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), w.data_ptr(),
            bias.data_ptr() if bias is not None else None, out2d.data_ptr(), M, w.shape[0], K, _ACT[act])
    if rc == E_UNSUPPORTED:
        return False
    check(rc, "codetr_linear_ln")
    CALLS["linear"] += 1
    CALLS["linear_ln"] += 1
    return True


def linear_xadd(x2d, xadd2d, w, bias, out2d) -> bool:
    """out = (x + x_add) @ w.T + bias; False when the library declines the shape (caller adds and calls linear)."""
    lib = load()
    fn = lib.codetr_linear_xadd_bf16 if x2d.dtype == torch.bfloat16 else lib.codetr_linear_xadd_f16
    M, K = x2d.shape
    rc = fn(current_stream_ptr(x2d.device), x2d.data_ptr(), xadd2d.data_ptr(), w.data_ptr(),
            bias.data_ptr() if bias is not None else None, out2d.data_ptr(), M, w.shape[0], K)
    if rc == E_UNSUPPORTED:
        return False
    check(rc, "codetr_linear_xadd")
    CALLS["linear"] += 1
    CALLS["linear_xadd"] += 1
    return True


def im2col_tokens(x4d, k, stride, pad, out):
    """x4d [B,H,W,C] 16-bit token-major -> out [B*Ho*Wo, k*k*C] with K ordered (ky, kx, c), zero padding"""
    lib = load()
    CALLS["patch_im2col"] += 1
    B, H, W, C = x4d.shape
    rc = lib.codetr_im2col_tokens_b16(current_stream_ptr(x4d.device), x4d.data_ptr(), B, H, W, C, k, stride, pad,
                                      out.data_ptr())
    check(rc, "codetr_im2col_tokens_b16")
    return out


def topk_supported(x2d, k) -> bool:
    return (x2d.dtype in (torch.float16, torch.bfloat16) and x2d.dim() == 2 and x2d.is_contiguous()
            and 0 < k <= 1024 and k <= x2d.shape[1] < (1 << 24))


def topk(x2d, k, values, indices):
    """x2d [rows, n] f16 / bf16 -> values [rows, k] (or None), indices [rows, k] int64; sorted descending, ties by
    ascending index, NaN first.  Long rows are cut over several workgroups (two passes, same result)."""
    lib = load()
    CALLS["topk"] += 1
    rows, n = x2d.shape
    bf = x2d.dtype == torch.bfloat16
    ws_bytes = ctypes.c_int64(0)
    chunks = lib.codetr_topk_chunks(n, k, rows, ctypes.cast(ctypes.pointer(ws_bytes), ctypes.c_void_p))
    vp = values.data_ptr() if values is not None else None
    if chunks > 1 and x2d.data_ptr() % 16 == 0:
        ws = torch.empty(ws_bytes.value, dtype=torch.uint8, device=x2d.device)
        fn = lib.codetr_topk_chunked_bf16 if bf else lib.codetr_topk_chunked_f16
        rc = fn(current_stream_ptr(x2d.device), x2d.data_ptr(), rows, n, k, chunks, vp, indices.data_ptr(),
                ws.data_ptr(), ws_bytes.value)
    else:
        fn = lib.codetr_topk_bf16 if bf else lib.codetr_topk_f16
        rc = fn(current_stream_ptr(x2d.device), x2d.data_ptr(), rows, n, k, vp, indices.data_ptr())
    check(rc, "codetr_topk")


def patch_im2col(x, k, kpad, out):
    """x [B,C,H,W] 16-bit -> out [B*ceil(H/k)*ceil(W/k), kpad] patch rows in (c, ky, kx) order, zero padded."""
    lib = load()
    CALLS["patch_im2col"] += 1
    B, C, H, W = x.shape
    rc = lib.codetr_patch_im2col_b16(current_stream_ptr(x.device), x.data_ptr(), B, C, H, W, k, kpad, out.data_ptr())
    check(rc, "codetr_patch_im2col_b16")
    return out


def groupnorm_tokens_supported(x, groups) -> bool:
    return x.dtype in (torch.float16, torch.bfloat16) and x.shape[-1] == groups * 8 and 256 % groups == 0


def groupnorm_tokens(x, gamma, beta, groups, eps, out_slice, out_batch_stride):
    """x [B,HW,C] contiguous -> GN written to out_slice (a view whose data_ptr is the destination of image 0,
    row 0; consecutive images are out_batch_stride elements apart)."""
    lib = load()
    CALLS["groupnorm_tokens"] += 1
    B, HW, C = x.shape
    ws = torch.empty(lib.codetr_groupnorm_tokens_workspace_bytes(B, HW, C), dtype=torch.uint8, device=x.device)
    fn = lib.codetr_groupnorm_tokens_bf16 if x.dtype == torch.bfloat16 else lib.codetr_groupnorm_tokens_f16
    rc = fn(current_stream_ptr(x.device), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out_slice.data_ptr(),
            out_batch_stride, ws.data_ptr(), B, HW, C, groups, float(eps))
    check(rc, "codetr_groupnorm_tokens")


def msda_head_major_supported(dtype, D, L, P) -> bool:
    """the head-major fused kernel: 16-bit storage, 64- or 128-byte head rows, L*P <= 4 * lanes per pair"""
    return dtype in _MSDA_FUSED_BY_DTYPE and D in (32, 64) and L * P <= 4 * 2 * (D // 8)


def sine_pos_tokens(ycum, xcum, level_embed, out_slice, out_batch_stride, num_feats, temperature, scale, eps, offset,
                    normalize):
    """ycum/xcum [B,H,W] fp32 -> encoding written at out_slice.data_ptr() (image 0), see include/codetr_hip.h."""
    CALLS["sine_pos_tokens"] += 1
    B, H, W = ycum.shape
    lib = load()
    fn = lib.codetr_sine_pos_tokens_bf16 if out_slice.dtype == torch.bfloat16 else lib.codetr_sine_pos_tokens_f16
    rc = fn(
        current_stream_ptr(ycum.device), ycum.data_ptr(), xcum.data_ptr(),
        level_embed.data_ptr() if level_embed is not None else None, out_slice.data_ptr(), out_batch_stride, B, H, W,
        num_feats, float(temperature), float(scale), float(eps), float(offset), 1 if normalize else 0)
    check(rc, "codetr_sine_pos_tokens")


def ffn_fused_supported(x, w1, w2, act) -> bool:
    return (x.dtype in (torch.float16, torch.bfloat16) and act == "relu" and x.shape[-1] == 256 and w1.shape[1] == 256
            and w2.shape[0] == 256 and w1.shape[0] % 64 == 0 and w1.dtype == x.dtype and w2.dtype == x.dtype)


def ffn_pack_w2(w2):
    """one-time re-layout of the second Linear's weight for ffn_fused (see include/codetr_hip.h)"""
    out = torch.empty_like(w2)
    rc = load().codetr_ffn_pack_w2_f16(current_stream_ptr(w2.device), w2.data_ptr(), out.data_ptr(), w2.shape[0], w2.shape[1])
    check(rc, "codetr_ffn_pack_w2_f16")
    return out


def ffn_fused(x2d, w1, b1, w2, b2, out2d, ln=None, pos2d=None, out_plus_pos2d=None, ln_in=None):
    """w2 must be the PACKED weight (ffn_pack_w2).  ln = (gamma, beta, eps): LayerNorm folded into the epilogue;
    pos2d / out_plus_pos2d: second output `out + pos`; ln_in = (gamma, beta, eps): LayerNorm of the input rows."""
    CALLS["ffn_fused"] += 1
    g, b, eps = ln if ln is not None else (None, None, 0.0)
    gi, bi, epsi = ln_in if ln_in is not None else (None, None, 0.0)
    lib = load()
    fn = lib.codetr_ffn_relu_ln2_bf16 if x2d.dtype == torch.bfloat16 else lib.codetr_ffn_relu_ln2_f16
    rc = fn(
        current_stream_ptr(x2d.device), x2d.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
        out2d.data_ptr(), x2d.shape[0], x2d.shape[1], w1.shape[0],
        gi.data_ptr() if gi is not None else None, bi.data_ptr() if bi is not None else None, float(epsi),
        g.data_ptr() if g is not None else None, b.data_ptr() if b is not None else None, float(eps),
        pos2d.data_ptr() if pos2d is not None else None,
        out_plus_pos2d.data_ptr() if out_plus_pos2d is not None else None)
    check(rc, "codetr_ffn_relu_ln2")
    return out2d


def ffn_oproj_w1_index(C):
    """column order of the first Linear's weight for ffn_oproj_fused (host list of C ints)"""
    idx = (ctypes.c_int32 * C)()
    rc = load().codetr_ffn_oproj_w1_index(C, ctypes.cast(idx, ctypes.c_void_p))
    check(rc, "codetr_ffn_oproj_w1_index")
    return list(idx)


def ffn_oproj_fused(attn2d, wo, bo, identity2d, w1_perm, b1, w2_packed, b2, out2d, ln_in, ln, pos2d=None,
                    out_plus_pos2d=None):
    """out = LN(x1 + relu(x1 W1^T + b1) W2^T + b2), x1 = LN_in(identity + (attn Wo^T + bo)) -- the attention output
    projection folded into the fused FFN (include/codetr_hip.h codetr_ffn_oproj_relu_ln2_*).  w1_perm: W1 with its columns
    in ffn_oproj_w1_index order; w2_packed: ffn_pack_w2."""
    CALLS["ffn_fused"] += 1
    CALLS["ffn_oproj_fused"] += 1
    g, b, eps = ln if ln is not None else (None, None, 0.0)
    gi, bi, epsi = ln_in if ln_in is not None else (None, None, 0.0)
    lib = load()
    fn = lib.codetr_ffn_oproj_relu_ln2_bf16 if attn2d.dtype == torch.bfloat16 else lib.codetr_ffn_oproj_relu_ln2_f16
    rc = fn(
        current_stream_ptr(attn2d.device), attn2d.data_ptr(), wo.data_ptr(), bo.data_ptr(), identity2d.data_ptr(),
        w1_perm.data_ptr(), b1.data_ptr(), w2_packed.data_ptr(), b2.data_ptr(), out2d.data_ptr(), attn2d.shape[0],
        attn2d.shape[1], w1_perm.shape[0],
        gi.data_ptr() if gi is not None else None, bi.data_ptr() if bi is not None else None, float(epsi),
        g.data_ptr() if g is not None else None, b.data_ptr() if b is not None else None, float(eps),
        pos2d.data_ptr() if pos2d is not None else None,
        out_plus_pos2d.data_ptr() if out_plus_pos2d is not None else None)
    check(rc, "codetr_ffn_oproj_relu_ln2")
    return out2d


# ---- fp8 (e4m3) path: BASELINE config 5 ----------------------------------------------------------------------
FP8 = torch.float8_e4m3fn


def linear_fp8(x8, w8, w_scale, x_scale, bias, residual2d, act, out2d, out_scale=0.0):
    """x8 [M,K] / w8 [N,K] e4m3 (torch.float8_e4m3fn), w_scale [N] fp32, x_scale python float; out2d fp16 [M,N], or
    e4m3 [M,N] with out_scale (no residual then)."""
    lib = load()
    CALLS["linear_fp8"] += 1
    M, K = x8.shape
    N = w8.shape[0]
    out8 = out2d.dtype == FP8
    rc = lib.codetr_linear_fp8(current_stream_ptr(x8.device), x8.data_ptr(), w8.data_ptr(), w_scale.data_ptr(), float(x_scale),
                               bias.data_ptr() if bias is not None else None,
                               residual2d.data_ptr() if residual2d is not None else None, out2d.data_ptr(),
                               1 if out8 else 0, float(out_scale), M, N, K, _ACT[act])
    check(rc, "codetr_linear_fp8")
    return out2d


def mx_scale_bytes(M, K) -> int:
    n = int(load().codetr_mx_scale_bytes(M, K))
    if n < 0:
        raise ValueError(f"no MX scale layout for [{M}, {K}]")
    return n


def linear_fp8mx(x8, x_scales, w8, w_scale, bias, residual2d, act, out2d, out_scales=None):
    """x8 [M,K] e4m3 + x_scales (uint8, mx_scale_bytes(M, K)); out2d fp16 [M,N], or e4m3 + out_scales (mx_scale_bytes(M, N))"""
    CALLS["linear_fp8"] += 1
    M, K = x8.shape
    N = w8.shape[0]
    rc = load().codetr_linear_fp8mx(current_stream_ptr(x8.device), x8.data_ptr(), x_scales.data_ptr(), w8.data_ptr(),
                                    w_scale.data_ptr(), bias.data_ptr() if bias is not None else None,
                                    residual2d.data_ptr() if residual2d is not None else None, out2d.data_ptr(),
                                    out_scales.data_ptr() if out_scales is not None else None, M, N, K, _ACT[act])
    check(rc, "codetr_linear_fp8mx")
    return out2d


def cast_fp8mx(x2d, out2d, out_scales):
    CALLS["cast_fp8"] += 1
    rows, C = x2d.shape
    check(load().codetr_cast_fp8mx_f16(current_stream_ptr(x2d.device), x2d.data_ptr(), out2d.data_ptr(), out_scales.data_ptr(),
                                       rows, C), "codetr_cast_fp8mx_f16")
    return out2d


def layernorm_fp8mx(x2d, weight, bias, eps, out2d, out_scales):
    CALLS["layernorm_fp8"] += 1
    rows, C = x2d.shape
    check(load().codetr_layernorm_fp8mx_f16(current_stream_ptr(x2d.device), x2d.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                            out2d.data_ptr(), out_scales.data_ptr(), rows, C, float(eps)),
          "codetr_layernorm_fp8mx_f16")
    return out2d


def ffn_fp8(x2d, w1q, w1_scale, b1, w2q_packed, w2_scale, b2, out2d, x_scale, h_scale, ln=None, pos2d=None,
            out_plus_pos2d=None, ln_in=None):
    """codetr_ffn_fp8 (include/codetr_hip.h): x2d / out2d fp16 [M, 256]; w1q [hidden, 256] / w2q_packed [256, hidden]
    e4m3, the latter in the kernel's column order (hip_ops.ffn_fp8_weights)."""
    CALLS["ffn_fp8"] += 1
    g, b, eps = ln if ln is not None else (None, None, 0.0)
    gi, bi, epsi = ln_in if ln_in is not None else (None, None, 0.0)
    rc = load().codetr_ffn_fp8(
        current_stream_ptr(x2d.device), x2d.data_ptr(), w1q.data_ptr(), w1_scale.data_ptr(), b1.data_ptr(),
        w2q_packed.data_ptr(), w2_scale.data_ptr(), b2.data_ptr(), out2d.data_ptr(), x2d.shape[0], x2d.shape[1],
        w1q.shape[0], float(x_scale), float(h_scale),
        gi.data_ptr() if gi is not None else None, bi.data_ptr() if bi is not None else None, float(epsi),
        g.data_ptr() if g is not None else None, b.data_ptr() if b is not None else None, float(eps),
        pos2d.data_ptr() if pos2d is not None else None,
        out_plus_pos2d.data_ptr() if out_plus_pos2d is not None else None)
    check(rc, "codetr_ffn_fp8")
    return out2d


def cast_fp8(x, scale, out):
    CALLS["cast_fp8"] += 1
    rc = load().codetr_cast_fp8_f16(current_stream_ptr(x.device), x.data_ptr(), out.data_ptr(), x.numel(), float(scale))
    check(rc, "codetr_cast_fp8_f16")
    return out


def layernorm_fp8(x2d, weight, bias, eps, scale, out2d):
    CALLS["layernorm_fp8"] += 1
    rows, C = x2d.shape
    rc = load().codetr_layernorm_fp8_f16(current_stream_ptr(x2d.device), x2d.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                         out2d.data_ptr(), rows, C, float(eps), float(scale))
    check(rc, "codetr_layernorm_fp8_f16")
    return out2d


# ---- small element-wise / gather kernels (csrc/small_ops.hip) -------------------------------------------------
def add_f16(a, b, out, a_period):
    CALLS["small_ops"] += 1
    fn = load().codetr_add_bf16 if b.dtype == torch.bfloat16 else load().codetr_add_f16
    check(fn(current_stream_ptr(b.device), a.data_ptr(), b.data_ptr(), out.data_ptr(), b.numel(), a_period), "codetr_add")
    return out


def sigmoid_f16(x, out):
    CALLS["small_ops"] += 1
    fn = load().codetr_sigmoid_bf16 if x.dtype == torch.bfloat16 else load().codetr_sigmoid_f16
    check(fn(current_stream_ptr(x.device), x.data_ptr(), out.data_ptr(), x.numel()), "codetr_sigmoid")
    return out


def gather_rows(src, idx, out):
    CALLS["small_ops"] += 1
    B, S, C = src.shape
    check(load().codetr_gather_rows_b16(current_stream_ptr(src.device), src.data_ptr(), idx.data_ptr(), out.data_ptr(), B, S,
                                        idx.shape[1], C), "codetr_gather_rows_b16")
    return out


def decode_boxes(coords_unact, idx, num_classes, img_w, img_h, boxes, labels):
    CALLS["small_ops"] += 1
    B, Nq, _ = coords_unact.shape
    fn = load().codetr_decode_boxes_bf16 if coords_unact.dtype == torch.bfloat16 else load().codetr_decode_boxes_f16
    check(fn(current_stream_ptr(idx.device), coords_unact.data_ptr(), idx.data_ptr(), boxes.data_ptr(),
                                         labels.data_ptr(), B, Nq, idx.shape[1], num_classes, float(img_w), float(img_h)),
          "codetr_decode_boxes_f16")


def valid_ratios(counts, level_wh, out, out32=None):
    CALLS["small_ops"] += 1
    B, L, _ = counts.shape
    fn = load().codetr_valid_ratios_bf16 if level_wh.dtype == torch.bfloat16 else load().codetr_valid_ratios_f16
    check(fn(current_stream_ptr(counts.device), counts.data_ptr(), level_wh.data_ptr(), out.data_ptr(),
             out32.data_ptr() if out32 is not None else None, B, L), "codetr_valid_ratios")
    return out
