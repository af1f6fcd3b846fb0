"""ctypes binding of libsacv_hip.so (the C ABI declared in include/sacv_hip.h).

There is NO fallback: if the shared library is missing or a symbol is absent the import
fails loudly -- build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C simpleaicv-pytorch-imagenet-coco-training_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SACV_HIP_LIB') or os.path.join(_HERE, 'lib', 'libsacv_hip.so')  # override: tuning builds

P, I, F, LL = C.c_void_p, C.c_int, C.c_float, C.c_longlong

# name -> (restype, argtypes); mirrors include/sacv_hip.h
SIGNATURES = {
    'sacv_last_error': (C.c_char_p, []),
    'sacv_abi_version': (I, []),
    'sacv_stream_wait_stream': (I, [P, P]),
    'sacv_stream_wait_stream_ticket': (I, [P, P]),
    'sacv_stream_ticket_timeouts': (LL, []),
    'sacv_launch_share': (I, [I, I, I, I, I, I, I, I]),
    'sacv_conv2d_gemm': (I, [I, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P, P, P, P]),
    'sacv_conv2d_gemm_stats_rows': (I, [I, P, P, P]),
    'sacv_conv2d_gemm_bnbwd': (I, [P, P, P, I, I, I, I, I, P, P, I, I, I, I, I, I, I, I, P, P, P, P, P, P, P, P, P]),
    'sacv_bn_finalize_stats': (I, [P, I, I, I, F, F, P, P, P, P, P, P]),
    'sacv_colsum_finalize': (I, [P, I, I, I, P, I, P]),
    'sacv_colsum_finalize_multi': (I, [I, P, P, P, P, P, P, P]),
    'sacv_layernorm_bwd_partial_rows': (I, [I]),
    'sacv_colsum_bf16_partial_rows': (I, [I]),
    'sacv_conv2d_gemm_workspace_bytes': (LL, []),
    'sacv_wino_tiles': (I, [I, I, P, P]),
    'sacv_wino_weights': (I, [P, P, I, I, I, I, P]),
    'sacv_wino_weights_batched': (I, [I, P, P, P, P, P, I, P]),
    'sacv_wino_input': (I, [I, P, I, P, P, I, P, I, I, P]),
    'sacv_wino_stats_rows': (I, [I]),
    'sacv_linear_bf16': (I, [P, P, P, P, P, LL, I, I, I, P]),
    'sacv_wino_pad_k': (I, [I]),
    'sacv_wino_pad_n': (I, [I]),
    'sacv_wino_wgrad_workspace_bytes': (LL, [I, I, I]),
    'sacv_wino_wgrad': (I, [I, P, I, P, P, I, P, I, I, I, P, I, I, P, P]),
    'sacv_wino_gemm': (I, [I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, I, P]),
    'sacv_conv2d_wgrad_chunk': (I, [I, P, I, I, I]),
    'sacv_conv2d_wgrad_workspace_bytes': (LL, [I, P, P, P, I, I, I, I, I]),
    'sacv_conv2d_wgrad': (I, [I, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    'sacv_pack_dgrad_batched': (I, [P, P, I, P]),
    'sacv_pack_weight': (I, [P, P, I, I, I, I, I, I, P]),
    'sacv_im2col_stem': (I, [P, P, I, I, I, I, LL, LL, LL, LL, I, I, I, I, I, I, I, P]),
    'sacv_stem_conv_stats_rows': (I, [I, I, I]),
    'sacv_stem_conv_fwd': (I, [P, I, I, I, LL, LL, LL, LL, P, I, P, I, I, P, P]),
    'sacv_stem_conv_wgrad_workspace_bytes': (LL, [I, I, I]),
    'sacv_stem_conv_wgrad': (I, [P, I, I, I, LL, LL, LL, LL, P, I, I, P, P, I, P]),
    'sacv_bn_workspace_floats': (LL, [I, I]),
    'sacv_bn_fold_batched': (I, [P, I, F, P]),
    'sacv_bn_train_stats': (I, [P, I, I, F, F, P, P, P, P, P, P, P]),
    'sacv_bn_apply': (I, [P, P, P, P, P, P, P, LL, I, I, I, F, P]),
    'sacv_bn_train_bwd': (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, P, P]),
    'sacv_bn_train_bwd_partial': (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, P, I, P, P]),
    'sacv_gn_workspace_floats': (LL, [I, I, I]),
    'sacv_groupnorm_fwd': (I, [P, P, P, P, P, I, I, I, I, F, I, P, P]),
    'sacv_groupnorm_bwd': (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, P, P]),
    'sacv_fcos_assign': (I, [I, P, P, P, P, P, P, P, I, I, I, F, I, P]),
    'sacv_fcos_reg_ctr_loss_fwd_bwd': (I, [I, P, P, P, P, P, P, I, I, I, F, F, P, P, P, I, I, P]),
    'sacv_fcos_split_fwd': (I, [P, P, P, P, I, P]),
    'sacv_fcos_split_bwd': (I, [P, P, P, P, P, F, P, P, I, I, I, P, P]),
    'sacv_maxpool_fwd': (I, [P, P, P, I, I, I, I, I, I, I, I, I, P]),
    'sacv_maxpool_bwd': (I, [P, P, P, I, I, I, I, I, I, I, I, I, P]),
    'sacv_upsample_bilinear_add_fwd': (I, [P, P, P, I, I, I, I, I, I, P]),
    'sacv_upsample_bilinear_bwd': (I, [P, P, I, I, I, I, I, I, I, P]),
    'sacv_eltwise': (I, [P, P, P, LL, I, F, P]),
    'sacv_colsum_workspace_floats': (LL, [I, I]),
    'sacv_colsum': (I, [P, P, I, I, I, P, P]),
    'sacv_retina_anchors': (I, [I, P, P, P, P, I, P, I, P, P, P]),
    'sacv_fcos_points': (I, [I, P, P, P, P, P]),
    'sacv_retina_assign': (I, [P, P, P, P, I, I, I, I, P]),
    'sacv_loss_workspace_floats': (LL, []),
    'sacv_focal_loss_fwd_bwd': (I, [I, P, P, P, P, I, I, I, I, F, F, F, P, P, P, I, P]),
    'sacv_retina_box_loss_fwd_bwd': (I, [I, P, P, P, P, P, I, I, I, F, F, P, P, P, I, P]),
    'sacv_decode_workspace_bytes': (LL, [I, I, I]),
    'sacv_decode_topk_nms': (I, [I, I, P, P, P, P, P, I, I, I, F, I, I, F, I, P, P, P, P, P]),
    'sacv_decode_topk_soft_nms': (I, [I, I, P, P, P, P, P, I, I, I, F, I, I, F, F, F, I, P, P, P, P, P]),
    'sacv_collate_u8': (I, [I, P, P, P, I, P, P]),
    'sacv_collate_u8_rect': (I, [I, P, P, P, I, I, P, P]),
    'sacv_eval_postprocess': (I, [P, P, P, I, I, P]),
    'sacv_avgpool_fwd': (I, [P, P, I, I, I, P]),
    'sacv_avgpool_bwd': (I, [P, P, I, I, I, P]),
    'sacv_softmax_ce_fwd_bwd': (I, [P, P, P, P, I, I, F, P, P]),
    'sacv_sgd_step': (I, [P, P, P, LL, LL, F, F, F, F, I, F, P]),
    'sacv_nonfinite_flag': (I, [P, LL, I, P, P]),
    'sacv_loss_skip_flag': (I, [P, I, P, I, P]),
    'sacv_step_gate': (I, [P, P, F, F, P]),
    'sacv_grad_clip_workspace_bytes': (LL, []),
    'sacv_grad_clip_coef': (I, [P, LL, F, F, F, P, P, P]),
    'sacv_adamw_step_gated': (I, [P, P, P, P, LL, LL, F, F, F, F, F, F, F, P, F, F, P]),
    'sacv_sgd_step_gated': (I, [P, P, P, LL, LL, F, F, F, F, F, I, P, F, F, P]),
    'sacv_ema_update': (I, [P, P, LL, F, F, P, P]),
    'sacv_adamw_step': (I, [P, P, P, P, LL, LL, F, F, F, F, F, F, I, F, P]),
    'sacv_fill': (I, [P, LL, F, P]),
    'sacv_conv2d_gemm_bf16': (I, [P, P, I, I, I, I, I, P, P, I, I, I, I, I, I, I, I, I, I, I, P, P, P]),
    'sacv_conv2d_wgrad_bf16_workspace_bytes': (LL, [I, I, I, I, I, I, I, I]),
    'sacv_conv2d_wgrad_bf16': (I, [P, P, I, I, I, I, I, P, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    'sacv_cast_bf16': (I, [P, P, LL, P]),
    'sacv_cast_f32': (I, [P, P, LL, P]),
    'sacv_pack_dgrad_batched_bf16': (I, [P, P, I, P]),
    'sacv_im2col_stem_bf16': (I, [P, P, I, I, I, I, LL, LL, LL, LL, I, I, I, I, I, I, I, P]),
    'sacv_bn_apply_bf16': (I, [P, P, P, P, P, P, P, LL, I, I, I, F, P]),
    'sacv_bn_train_bwd_bf16': (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, P, P]),
    'sacv_bn_train_bwd_partial_bf16': (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, P, I, P, P]),
    'sacv_conv2d_gemm_bf16_bnbwd': (I, [P, P, I, I, I, I, I, P, P, I, I, I, I, I, I, I, I, P, P, P, P, P, P, P, P]),
    'sacv_maxpool_fwd_bf16': (I, [P, P, P, I, I, I, I, I, I, I, I, I, P]),
    'sacv_maxpool_bwd_bf16': (I, [P, P, P, I, I, I, I, I, I, I, I, I, P]),
    'sacv_avgpool_fwd_bf16': (I, [P, P, I, I, I, P]),
    'sacv_avgpool_bwd_bf16': (I, [P, P, I, I, I, P]),
    'sacv_mae_random_masking': (I, [P, I, I, I, P, P, P, P, P]),
    'sacv_mae_patchify': (I, [P, P, P, I, I, I, I, I, I, LL, LL, LL, LL, P]),
    'sacv_mae_patch_labels': (I, [P, P, I, I, I, I, I, I, LL, LL, LL, LL, P]),
    'sacv_mae_encoder_tokens': (I, [P, P, P, P, P, I, I, I, P]),
    'sacv_mae_encoder_tokens_bwd': (I, [P, P, I, I, I, P]),
    'sacv_mae_decoder_tokens': (I, [P, P, P, P, P, I, I, I, I, P]),
    'sacv_mae_decoder_tokens_bwd': (I, [P, P, P, P, I, I, I, I, P]),
    'sacv_layernorm_fwd': (I, [P, P, P, P, P, P, I, I, F, P]),
    'sacv_layernorm_bwd_workspace_floats': (LL, [I, I]),
    'sacv_layernorm_bwd': (I, [P, P, P, P, P, P, P, P, I, I, I, I, P, P]),
    'sacv_gelu_fwd': (I, [P, P, LL, P]),
    'sacv_gelu_bwd': (I, [P, P, P, LL, P]),
    'sacv_colsum_bf16_workspace_floats': (LL, [I, I]),
    'sacv_colsum_bf16': (I, [P, P, I, I, I, P, P]),
    'sacv_attention_fwd': (I, [P, P, I, I, I, I, F, P]),
    'sacv_attention_bwd': (I, [P, P, P, I, I, I, I, F, P]),
    'sacv_attention_mfma_supported': (I, [I, I]),
    'sacv_attention_fwd_mfma': (I, [P, P, P, I, I, I, I, F, P]),
    'sacv_attention_bwd_mfma': (I, [P, P, P, P, P, I, I, I, I, F, P]),
    'sacv_mae_mse_workspace_floats': (LL, [I, I]),
    'sacv_mae_mse_loss': (I, [P, P, P, P, P, I, I, I, F, P, P]),
    'sacv_comm_unique_id_bytes': (I, []),
    'sacv_comm_unique_id': (I, [P]),
    'sacv_comm_init': (I, [P, I, I, P]),
    'sacv_comm_destroy': (I, [P]),
    'sacv_allreduce_bucket': (I, [P, P, LL, P]),
    'sacv_comm_wait': (I, [P, P]),
    'sacv_comm_emulate_load': (I, [I, I, F, P, LL, P]),
    'sacv_comm_broadcast': (I, [P, P, LL, I, P]),
}


class SacvError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} is missing: the HIP extension has not been built. There is no CPU or '
            'PyTorch fallback for this package; run __graft_entry__.build() (hipcc --offload-arch=gfx950).')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    return lib


ABI_VERSION = 5
lib = _load()
if lib.sacv_abi_version() != ABI_VERSION:
    raise ImportError(f'{LIB_PATH} has ABI version {lib.sacv_abi_version()}, this package needs {ABI_VERSION}: rebuild it '
                      '(__graft_entry__.build())')


def check(rc, what=''):
    if rc != 0:
        msg = lib.sacv_last_error()
        raise SacvError(f'{what} failed (code {rc}): {msg.decode() if msg else ""}')


def iarr(vals):
    return (C.c_int * len(vals))(*[int(v) for v in vals])


def farr(vals):
    return (C.c_float * len(vals))(*[float(v) for v in vals])


def parr(ptrs):
    return (C.c_void_p * len(ptrs))(*[C.c_void_p(int(p)) if p else None for p in ptrs])
