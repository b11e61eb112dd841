"""ctypes binding of libdcsnet_hip.so (C ABI: include/dcsnet_hip.h).

The product path has NO fallback: if the library is missing or a symbol is absent this module
raises, and every op built on it fails loudly.  PyTorch only owns device memory and streams;
device pointers are borrowed for the duration of one call.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DCS_LIB_PATH: a diagnostic build of the same library (tools/pk_hazard_probe.py); never a different implementation
LIB_PATH = os.environ.get('DCS_LIB_PATH') or os.path.join(os.path.dirname(_HERE), 'lib', 'libdcsnet_hip.so')

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_SIGMOID = 0, 1, 2, 3

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_long
_F = ctypes.c_float
_U64 = ctypes.c_ulonglong



class AttentionItem(ctypes.Structure):
    """dcs_attention_item (include/dcsnet_hip.h): one attention block of a batched launch."""
    _fields_ = [(n, _P) for n in ('x', 'w1', 'w2', 'wsa', 'sa_bias', 'ca', 'pooled', 'hidden', 'sp', 'sa', 'y', 'g_out',
                                  'wsa_bwd', 'g_pre', 'g_sp', 'g_x', 'g_fc0_r', 'g_fc0_i', 'g_fc2_r', 'g_fc2_i')] + \
               [(n, _I) for n in ('H', 'W', 'C', 'Ch')] + [('g_pooled', _P)]


# name -> (restype, argtypes); must mirror include/dcsnet_hip.h exactly
SIGNATURES = {
    'dcs_abi_version': (_I, []),
    'dcs_error_string': (ctypes.c_char_p, [_I]),
    'dcs_pack_conv_weight': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'dcs_cconv2d_fwd_workspace_bytes': (_L, [_I] * 14),
    'dcs_cconv2d_fwd': (_I, [_P, _P, _P, _P, _P, _P, _L] + [_I] * 15 + [_P]),
    'dcs_cconv2d_fwd_affine': (_I, [_P, _P, _P, _P, _P, _P, _P, _L] + [_I] * 15 + [_P]),
    'dcs_cconv2d_fwd_stats_rows': (_I, [_I] * 14),
    'dcs_cconv2d_fwd_stats': (_I, [_P] * 6 + [_I, _P, _P, _L] + [_I] * 14 + [_P]),
    'dcs_rconv2d_fwd_workspace_bytes': (_L, [_I] * 14),
    'dcs_rconv2d_fwd': (_I, [_P, _P, _P, _P, _P, _P, _L] + [_I] * 15 + [_P]),
    'dcs_rconv2d_bwd_data_workspace_bytes': (_L, [_I] * 11),
    'dcs_rconv2d_bwd_data': (_I, [_P, _P, _P, _P, _L] + [_I] * 11 + [_P]),
    'dcs_rbn_fwd': (_I, [_P] * 9 + [_L, _L, _I, _F, _F, _I, _I, _P]),
    'dcs_rbn_bwd': (_I, [_P] * 8 + [_L, _L, _I, _I, _I, _P]),
    'dcs_packed_weight_floats': (_L, [_I] * 6),
    'dcs_packed_weight_bwd_floats': (_L, [_I] * 10),
    'dcs_pack_conv_weight_bwd': (_I, [_P, _P] + [_I] * 10 + [_P]),
    'dcs_cconv2d_bwd_data_workspace_bytes': (_L, [_I] * 14),
    'dcs_cconv2d_bwd_data': (_I, [_P] * 5 + [_L] + [_I] * 14 + [_P]),
    'dcs_upsample_cat_bwd': (_I, [_P, _P, _P] + [_I] * 7 + [_P]),
    'dcs_cconv2d_bwd_weight_workspace_bytes': (_L, [_I] * 14),
    'dcs_cconv2d_bwd_weight': (_I, [_P] * 8 + [_L] + [_I] * 15 + [_P]),
    'dcs_cbn_workspace_bytes': (_L, [_L, _I]),
    'dcs_cbn_fwd': (_I, [_P] * 9 + [_L, _L, _I, _F, _F, _I, _I, _F, _U64, _P, _P]),
    'dcs_cbn_fwd_slabs': (_I, [_P] * 9 + [_I, _I, _P, _L, _I, _F, _F, _I, _F, _U64, _P, _P]),
    'dcs_ca_pool_chunks': (_I, [_L, _I]),
    'dcs_cbn_fwd_slabs_pool': (_I, [_P] * 9 + [_I, _I, _P, _P, _L, _I, _L, _I, _F, _F, _I, _P]),
    'dcs_channel_attention_fc_fwd': (_I, [_P] * 6 + [_I, _L, _I, _I, _P]),
    'dcs_cbn_bwd_workspace_bytes': (_L, [_L, _I]),
    'dcs_cbn_bwd': (_I, [_P] * 9 + [_L, _L, _I, _I, _I, _F, _U64, _P, _P]),
    'dcs_cbn_bwd_add': (_I, [_P] * 9 + [_L, _L, _I, _I, _I, _F, _U64, _P, _P, _F, _L, _P, _P]),
    'dcs_ca_workspace_bytes': (_L, [_I, _L, _I]),
    'dcs_channel_attention_fwd': (_I, [_P] * 7 + [_L, _I, _L, _I, _I, _P]),
    'dcs_spatial_pool_fwd': (_I, [_P, _P, _P, _I, _L, _I, _P]),
    'dcs_attention_apply_fwd': (_I, [_P, _P, _P, _P, _I, _L, _I, _F, _U64, _P, _P]),
    'dcs_rattention_workspace_bytes': (_L, [_I, _L, _I]),
    'dcs_rattention_fwd': (_I, [_P] * 8 + [_L] + [_I] * 6 + [_P]),
    'dcs_rattention_train_workspace_bytes': (_L, [_I, _L, _I, _I]),
    'dcs_rattention_pool_fwd': (_I, [_P] * 8 + [_L] + [_I] * 5 + [_P]),
    'dcs_rattention_apply_fwd': (_I, [_P] * 4 + [_I] * 4 + [_P]),
    'dcs_rattention_apply_bwd': (_I, [_P] * 8 + [_L] + [_I] * 4 + [_P]),
    'dcs_rattention_pool_bwd': (_I, [_P] * 11 + [_I] + [_P, _L] + [_I] * 5 + [_P]),
    'dcs_attention_bwd_sa': (_I, [_P] * 5 + [_I, _L, _I, _F, _U64, _P, _P]),
    'dcs_attention_bwd_workspace_bytes': (_L, [_I, _L, _I, _I]),
    'dcs_attention_bwd_x': (_I, [_P] * 16 + [_L, _I, _L, _I, _I, _F, _U64, _P, _P]),
    'dcs_attention_bwd_fc_weights': (_I, [_P, _L, _P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _P]),
    'dcs_attention_fwd_batched_workspace_bytes': (_L, [_I, _P, _I]),
    'dcs_attention_fwd_batched': (_I, [_I, _P, _P, _L, _I, _P]),
    'dcs_attention_bwd_batched_workspace_bytes': (_L, [_I, _P, _I]),
    'dcs_attention_bwd_batched': (_I, [_I, _P, _P, _L, _I, _P]),
    'dcs_lstm_layer_fwd': (_I, [_P] * 6 + [_I, _I, _I, _I, _L, _L, _L, _P]),
    'dcs_lstm_layer_fwd_bias': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _P]),
    'dcs_lstm_layer_bwd': (_I, [_P] * 6 + [_I] * 4 + [_P]),
    'dcs_dropout_fwd': (_I, [_P, _P, _L, _F, _U64, _P, _P]),
    'dcs_complex_act_fwd': (_I, [_P, _P, _L, _I, _P]),
    'dcs_complex_upsample_fwd': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'dcs_tapsum_fwd': (_I, [_P, _P, _P, _P] + [_I] * 10 + [_P]),
    'dcs_cconv_up2_single_fwd': (_I, [_P] * 6 + [_I] * 6 + [_P]),
    'dcs_cconv_up2_single_bwd_data': (_I, [_P] * 4 + [_I] * 6 + [_P]),
    'dcs_cconv_up2_single_bwd_weight_workspace_bytes': (_L, []),
    'dcs_cconv_up2_single_bwd_weight': (_I, [_P] * 7 + [_I, _P, _L] + [_I] * 5 + [_P]),
    'dcs_tapsum_bwd_workspace_bytes': (_L, []),
    'dcs_tapsum_bwd': (_I, [_P, _P, _P, _P, _P, _L] + [_I] * 10 + [_P]),
    'dcs_bound_crm_fwd': (_I, [_P, _P, _L, _F, _P]),
    'dcs_bound_mask_apply_fwd': (_I, [_P, _P, _P, _P, _P, _L, _F, _P]),
    'dcs_bound_mask_apply_bwd': (_I, [_P] * 6 + [_L, _F, _P]),
    'dcs_bound2_mask_apply_fwd': (_I, [_P] * 6 + [_L, _F, _F, _U64, _P, _P]),
    'dcs_bound2_mask_apply_bwd': (_I, [_P] * 7 + [_L, _F, _F, _U64, _P, _P]),
    'dcs_bound2_apply_polar_frames_fwd': (_I, [_P] * 4 + [_I] * 4 + [_F, _F, _U64, _P, _P]),
    'dcs_bound2_apply_polar_frames_bwd': (_I, [_P] * 5 + [_I] * 4 + [_F, _I, _F, _U64, _P, _P]),
    'dcs_irfft512_frames': (_I, [_P, _P, _L, _P]),
    'dcs_rfft512_frames': (_I, [_P, _P, _L, _P]),
    'dcs_rfft512_ola_frames': (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _P]),
    'dcs_irfft512_ola_frames': (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _P]),
    'dcs_polar_frames_fwd': (_I, [_P, _P, _I, _I, _I, _I, _F, _P]),
    'dcs_polar_frames_bwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _I, _P]),
    'dcs_istft_envelope': (_I, [_P, _P, _I, _I, _I, _P]),
    'dcs_istft_ola_fwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    'dcs_istft_ola_bwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    'dcs_stft_frames_fwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'dcs_stft_bins_fwd': (_I, [_P, _P, _I, _I, _I, _F, _P]),
    'dcs_sisnr_fwd': (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
    'dcs_sisnr_bwd': (_I, [_P, _P, _P, _P, _F, _P, _I, _I, _P]),
    'dcs_sisnr_losses_fwd': (_I, [_P, _P, _P, _I, _F, _P]),
    'dcs_lstm_whh_grad': (_I, [_P, _P, _P, _I, _I, _I, _P]),
    'dcs_atb_chunks': (_I, [_P, _P, _P, _L, _L, _L, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'dcs_chunk_sum_acc': (_I, [_P, _P, _L, _L, _I, _I, _I, _L, _P]),
    'dcs_atb_chunks_strided': (_I, [_P, _P, _P, _L, _L, _L, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'dcs_gemm_f32': (_I, [_P, _P, _P] + [_I] * 8 + [_L, _L, _I, _L, _L, _L, _I, _I, _P]),
    'dcs_lstm_combine_fwd': (_I, [_P, _P, _L, _P]),
    'dcs_lstm_combine_bwd': (_I, [_P, _P, _L, _P]),
    'dcs_lstm_param_grads': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'dcs_lstm_param_grads_ih': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _I, _L, _I, _P]),
    'dcs_sisnr_losses_guard_fwd': (_I, [_P, _P, _P, _I, _F, _P, _P]),
    'dcs_sisnr_pair_bwd': (_I, [_P, _P, _P, _P, _P, _P, _F, _P, _I, _I, _P]),
    'dcs_crm_fwd': (_I, [_P, _P, _P, _L, _F, _P]),
    'dcs_adam_amsgrad_step': (_I, [_P] * 6 + [_F, _F, _L, _F, _F, _F, _F, _F, _I, _P, _P, _P]),
    'dcs_adam_amsgrad_step_sumsq': (_I, [_P] * 6 + [_I, _F, _F, _L, _F, _F, _F, _F, _F, _I, _P, _P, _P]),
    'dcs_grad_sumsq_parts': (_I, [_P, _L, _P, _I, _P, _P, _P, _P, _I, _P]),
    'dcs_step_guard': (_I, [_P, _P, _P]),
    'dcs_stream_hold': (_I, [_P, _I, _P]),
    'dcs_kernel_timer_begin': (_I, [_I]),
    'dcs_kernel_timer_end': (_I, []),
    'dcs_kernel_timer_read': (_I, [_I, _P]),
    'dcs_step_advance': (_I, [_P, _P, _P, _P]),
    'dcs_step_advance_counters': (_I, [_P, _P, _P, _P, _I, _P]),
    'dcs_pack_tap_rows': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'dcs_tap_rows_wgrad_scatter': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'dcs_set_conv_precision': (_I, [_I]),
    'dcs_get_conv_precision': (_I, []),
    'dcs_wgrad_defer_begin': (_I, []),
    'dcs_wgrad_defer_suspend': (_I, [_I]),
    'dcs_wgrad_defer_flush': (_I, [_P]),
    'dcs_pack_plan_begin': (_I, []),
    'dcs_pack_plan_end': (_I, [_P]),
    'dcs_pack_plan_jobs': (_I, [_P, _P, _P]),
    'dcs_pack_plan_run': (_I, [_P, _P]),
    'dcs_pack_plan_destroy': (_I, [_P]),
}

_lib = None


class DcsHipError(RuntimeError):
    pass


# bf16-activation forms (include/dcsnet_hip.h, last section): the same argument lists as the fp32 entry points they mirror
for _n in ('dcs_cconv2d_fwd', 'dcs_cconv2d_fwd_affine', 'dcs_cconv2d_fwd_stats', 'dcs_cconv2d_bwd_data', 'dcs_cconv2d_bwd_weight',
           'dcs_cconv_up2_single_fwd', 'dcs_cconv_up2_single_bwd_data', 'dcs_cconv_up2_single_bwd_weight', 'dcs_tapsum_bwd', 'dcs_cbn_fwd', 'dcs_cbn_fwd_slabs', 'dcs_cbn_fwd_slabs_pool', 'dcs_cbn_bwd', 'dcs_cbn_bwd_add',
           'dcs_channel_attention_fwd', 'dcs_spatial_pool_fwd', 'dcs_attention_apply_fwd', 'dcs_attention_fwd_batched',
           'dcs_attention_bwd_sa', 'dcs_attention_bwd_x', 'dcs_attention_bwd_batched'):
    SIGNATURES[_n + '_h'] = SIGNATURES[_n]


def load():
    """Load the library once; raises if it has not been built (python dcs-net_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64.so.7; it must be the HIP runtime already in the
    # process when our library's DT_NEEDED entry of the same soname is resolved, otherwise two
    # runtimes coexist and launches on torch's streams fail.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise DcsHipError(f'{LIB_PATH} not found: build it with `python dcs-net_amd/build.py` '
                          '(there is no CPU or eager fallback)')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise DcsHipError(f'{LIB_PATH} does not export {name}; rebuild it') from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().dcs_error_string(code).decode()
        raise DcsHipError(f'{what}: {msg} (code {code})')


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def cur_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
