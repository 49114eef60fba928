"""ctypes binding of the C-ABI in include/ssunet_hip.h (libssunet_hip.so, built in-tree).

There is NO fallback: if the shared library is missing or a call fails, this raises.  The
product path never routes through oracle/ or through stock torch compute ops.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SSG_LIB_PATH: load another build of the library (diagnostic builds of tools/clock_probe.py); default: the in-tree .so
LIB_PATH = os.environ.get('SSG_LIB_PATH') or os.path.join(_HERE, 'libssunet_hip.so')
MAX_TAPS = 9
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_SWISH = 0, 1, 2, 3


class HipLibraryError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ('in1', C.c_void_p), ('in2', C.c_void_p),
        ('C1', C.c_int), ('C2', C.c_int), ('ld1', C.c_int), ('ld2', C.c_int),
        ('N', C.c_int), ('H', C.c_int), ('W', C.c_int),
        ('w', C.c_void_p), ('Kp', C.c_int), ('kmode', C.c_int),
        ('bias', C.c_void_p),
        ('res', C.c_void_p), ('ldr', C.c_int),
        ('out', C.c_void_p), ('Cout', C.c_int), ('ldo', C.c_int),
        ('GH', C.c_int), ('GW', C.c_int), ('OH', C.c_int), ('OW', C.c_int),
        ('in_sy', C.c_int), ('in_sx', C.c_int), ('out_sy', C.c_int), ('out_sx', C.c_int),
        ('out_oy', C.c_int), ('out_ox', C.c_int),
        ('ntaps', C.c_int), ('dy', C.c_int * MAX_TAPS), ('dx', C.c_int * MAX_TAPS),
        ('act', C.c_int), ('slope', C.c_float),
        ('bnpart', C.c_void_p),
        ('ws', C.c_void_p), ('ws_bytes', C.c_int64),
        ('w_split', C.c_void_p), ('parity_merge', C.c_int),
        ('in_scale', C.c_void_p), ('in_shift', C.c_void_p), ('in_act', C.c_int), ('in_slope', C.c_float),
        ('bwd_x', C.c_void_p), ('bwd_ldx', C.c_int), ('bwd_scale', C.c_void_p), ('bwd_shift', C.c_void_p), ('bwd_mean', C.c_void_p),
        ('bwd_act', C.c_int), ('bwd_slope', C.c_float),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ('in1', C.c_void_p), ('in2', C.c_void_p),
        ('C1', C.c_int), ('C2', C.c_int), ('ld1', C.c_int), ('ld2', C.c_int),
        ('N', C.c_int), ('H', C.c_int), ('W', C.c_int),
        ('dout', C.c_void_p), ('Cout', C.c_int), ('ldd', C.c_int), ('GH', C.c_int), ('GW', C.c_int),
        ('in_sy', C.c_int), ('in_sx', C.c_int),
        ('ntaps', C.c_int), ('dy', C.c_int * MAX_TAPS), ('dx', C.c_int * MAX_TAPS),
        ('ky', C.c_int * MAX_TAPS), ('kx', C.c_int * MAX_TAPS),
        ('KH', C.c_int), ('KW', C.c_int), ('Cin_real', C.c_int),
        ('dw_oihw', C.c_void_p),
        ('ws', C.c_void_p), ('ws_bytes', C.c_int64),
        ('flags', C.c_int),
        ('in_scale', C.c_void_p), ('in_shift', C.c_void_p), ('in_act', C.c_int), ('in_slope', C.c_float),
    ]


_P, _I, _L, _F, _D = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double


class BnFin(C.Structure):
    """include/ssunet_hip.h `ssg_bn_fin`."""
    _fields_ = [('weight', C.c_void_p), ('bias', C.c_void_p), ('eps', C.c_float), ('momentum', C.c_float), ('var_mode', C.c_int),
                ('running_mean', C.c_void_p), ('running_var', C.c_void_p),
                ('mean', C.c_void_p), ('invstd', C.c_void_p), ('scale', C.c_void_p), ('shift', C.c_void_p)]


# name -> argtypes (restype is int unless listed in _RESTYPES).  Must cover every symbol of
# include/ssunet_hip.h; tests/test_abi_symbols.py cross-checks this table against the header.
SIGNATURES = {
    'ssg_abi_version': [],
    'ssg_conv2d_igemm_f32': [C.POINTER(ConvDesc), _P],
    'ssg_conv2d_f32': [C.POINTER(ConvDesc), _P],
    'ssg_conv2d_kernel_id': [C.POINTER(ConvDesc)],
    'ssg_conv2d_bnpart_rows': [C.POINTER(ConvDesc)],
    'ssg_conv2d_workspace_bytes': [C.POINTER(ConvDesc)],
    'ssg_conv2d_thin_bf16': [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    'ssg_conv2d_thin_bf16_wgrad_workspace_bytes': [_I, _I, _I, _I, _I, _I],
    'ssg_conv2d_thin_bf16_wgrad': [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    'ssg_conv2d_thin_bf16_dgrad': [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    'ssg_conv2d_split_bn': [C.POINTER(ConvDesc)],
    'ssg_conv2d_in_affine_ok': [C.POINTER(ConvDesc)],
    'ssg_conv2d_bwd_stats_ok': [C.POINTER(ConvDesc)],
    'ssg_conv2d_wgrad_in_affine_ok': [C.POINTER(WgradDesc)],
    'ssg_pack_weights_split_bytes': [_I, _I, _I],
    'ssg_conv_set_k32_mode': [_I],
    'ssg_wgrad_set_k32_mode': [_I],
    'ssg_pack_weights_split_bf16x3': [_P, _I, _I, _I, _P, _P],
    'ssg_bn_stats_from_partials_workspace_bytes': [_I, _I],
    'ssg_bn_stats_from_partials_f32': [_P, _I, _I, _P, _D, _P, _P],
    'ssg_conv2d_wgrad_kernel_id': [C.POINTER(WgradDesc)],
    'ssg_pack_weights_f32': [_P, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_int), C.POINTER(C.c_int), _I, _I, _I, _P, _P],
    'ssg_pack_weights_scaled_f32': [_P, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_int), C.POINTER(C.c_int), _I, _I, _I, _P, _P, _P],
    'ssg_conv2d_wgrad_workspace_bytes': [C.POINTER(WgradDesc)],
    'ssg_conv2d_wgrad_f32': [C.POINTER(WgradDesc), _P],
    'ssg_nchw_to_nhwc_f32': [_P, _I, _I, _I, _I, _P, _I, _P],
    'ssg_nhwc_to_nchw_f32': [_P, _I, _I, _I, _I, _I, _P, _P],
    'ssg_bn_workspace_bytes': [_L, _I],
    'ssg_bn_stats_f32': [_P, _L, _I, _I, _P, _I, _P, _P],
    'ssg_bn_finalize_f32': [_P, _D, _I, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P],
    'ssg_bn_apply_f32': [_P, _L, _I, _I, _P, _P, _P, _I, _I, _F, _P, _I, _P],
    'ssg_bn_bwd_reduce_f32': [_P, _P, _P, _L, _I, _I, _I, _I, _P, _P, _P, _P, _I, _F, _P, _I, _P, _P],
    'ssg_bn_bwd_apply_f32': [_P, _P, _P, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _D, _I, _F, _P, _I, _P, _I, _P, _P, _P],
    'ssg_maxpool2x2_fwd_f32': [_P, _I, _I, _I, _I, _I, _P, _I, _P, _P],
    'ssg_maxpool2x2_bwd_f32': [_P, _I, _P, _I, _I, _I, _I, _P, _I, _P],
    'ssg_maxpool2x2_bwd_add_f32': [_P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P],
    'ssg_maxunpool2x2_fwd_f32': [_P, _I, _P, _I, _I, _I, _I, _P, _I, _P],
    'ssg_maxunpool2x2_bwd_f32': [_P, _I, _P, _I, _I, _I, _I, _P, _I, _P],
    'ssg_upsample2x_bilinear_fwd_f32': [_P, _I, _I, _I, _I, _I, _P, _I, _P],
    'ssg_upsample2x_bilinear_bwd_f32': [_P, _I, _I, _I, _I, _I, _P, _I, _P],
    'ssg_upsample2x_nearest_fwd_f32': [_P, _I, _I, _I, _I, _I, _P, _I, _P],
    'ssg_upsample2x_nearest_bwd_f32': [_P, _I, _I, _I, _I, _I, _P, _I, _P],
    'ssg_adaptive_avgpool_flat_fwd_f32': [_P, _I, _I, _I, _I, _I, _I, _P, _P],
    'ssg_adaptive_avgpool_flat_bwd_f32': [_P, _I, _I, _I, _I, _I, _P, _I, _P],
    'ssg_spade_conv_modulate_ok': [C.POINTER(ConvDesc)],
    'ssg_spade_conv_modulate_f32': [C.POINTER(ConvDesc), _P, _I, _P, _I, _P],
    'ssg_spade_modulate_fwd_f32': [_P, _I, _P, _I, _L, _I, _P, _I, _P],
    'ssg_spade_modulate_bwd_f32': [_P, _I, _P, _I, _P, _I, _L, _I, _P, _I, _P, _I, _P],
    'ssg_spade_modulate_bwd_sums_f32': [_P, _I, _P, _I, _P, _I, _L, _I, _P, _I, _P, _I, _P, _P, _P],
    'ssg_act_bwd_f32': [_P, _I, _P, _I, _L, _I, _I, _F, _P, _I, _P],
    'ssg_add_f32': [_P, _P, _L, _P, _P],
    'ssg_copy_channels_f32': [_P, _I, _L, _I, _P, _I, _P],
    'ssg_nan_to_zero_f32': [_P, _L, _P, _P],
    'ssg_mask_zero_f32': [_P, _P, _L, _P, _P],
    'ssg_seg_loss_workspace_bytes': [_I, _L, _I],
    'ssg_seg_loss_fwd_f32': [_P, _I, _P, _I, _I, _L, _I, _I, _P, _P, _P, _P],
    'ssg_seg_loss_bwd_f32': [_P, _I, _P, _I, _I, _L, _I, _P, _P, _P, _P, _P, _P, _I, _P],
    'ssg_bce_logits_const_fwd_f32': [_P, _I, _I, _F, _P, _P],
    'ssg_bce_logits_const_bwd_f32': [_P, _I, _I, _F, _P, _P, _I, _P],
    'ssg_clamp_adam_multi_f32': [_P, _P, _P, _P, _I, _F, _D, _D, _D, _D, _D, _D, _D, _P],
    'ssg_clamp_f32': [_P, _L, _F, _F, _P],
    'ssg_clamp_multi_f32': [_P, _P, _P, _P, _I, _I, _F, _F, _P],
    'ssg_channel_sum_f32': [_P, _L, _I, _I, _P, _P, _P],
    'ssg_dwconv2d_fwd_f32': [_P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    'ssg_dwconv2d_dgrad_f32': [_P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    'ssg_dwconv2d_wgrad_workspace_bytes': [_I, _I, _I, _I, _I, _I],
    'ssg_dwconv2d_wgrad_f32': [_P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    'ssg_unary_fwd_f32': [_P, _I, _L, _I, _I, _P, _I, _P],
    'ssg_unary_bwd_f32': [_P, _I, _P, _I, _L, _I, _I, _P, _I, _P],
    'ssg_mul_fwd_f32': [_P, _I, _P, _I, _L, _I, _P, _I, _P],
    'ssg_mul_bwd_f32': [_P, _I, _P, _I, _P, _I, _L, _I, _P, _I, _P, _I, _P],
    'ssg_channel_scale_fwd_f32': [_P, _I, _P, _I, _L, _I, _P, _I, _P],
    'ssg_sample_channel_sum_workspace_bytes': [_I, _L, _I],
    'ssg_sample_channel_sum_f32': [_P, _I, _P, _I, _I, _L, _I, _F, _P, _P, _P],
    'ssg_broadcast_rows_f32': [_P, _I, _L, _I, _F, _P, _I, _P],
    'ssg_spectral_norm_workspace_bytes': [_I, _I],
    'ssg_spectral_norm_fwd_f32': [_P, _I, _I, _P, _P, _I, _D, _P, _P, _P, _P],
    'ssg_spectral_norm_bwd_f32': [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P],
    'ssg_pixel_gate_fwd_f32': [_P, _I, _P, _I, _L, _I, _P, _I, _P],
    'ssg_pixel_gate_bwd_f32': [_P, _I, _P, _I, _P, _I, _L, _I, _P, _I, _P, _I, _P],
    'ssg_linear_fwd_workspace_bytes': [_I, _I, _I],
    'ssg_linear_fwd_f32': [_P, _I, _I, _I, _P, _I, _P, _I, _F, _P, _I, _P, _L, _P],
    'ssg_linear_wgrad_f32': [_P, _I, _I, _I, _P, _I, _I, _P, _P],
    'ssg_pack_weights_bf16': [_P, _I, _I, _I, _I, _I, _P, _P],
    'ssg_gemm_bf16': [_P, _L, _I, _I, _P, _I, _I, _P, _I, _P, _I, _P],
    'ssg_gemm_wgrad_bf16_workspace_bytes': [_L, _I, _I],
    'ssg_gemm_wgrad_bf16': [_P, _I, _P, _I, _L, _I, _I, _P, _P, _L, _P],
    'ssg_bn_stats_finalize_f32': [_P, _L, _I, _I, C.POINTER(BnFin), _P, _P],
    'ssg_bn_stats_finalize_bf16': [_P, _L, _I, _I, C.POINTER(BnFin), _P, _P],
    'ssg_bn_stats_from_partials_finalize_f32': [_P, _I, _I, _D, C.POINTER(BnFin), _P, _P],
    'ssg_se_gate_ok': [_I, _I, _I],
    'ssg_se_gate_workspace_floats': [_I, _I, _I],
    'ssg_se_gate_fwd_f32': [_P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P],
    'ssg_se_gate_bwd_f32': [_P, _I, _P, _I, _P, _P, _I, _I, _I, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P],
    'ssg_tool_mfma_peak_f32': [_P, _I, _I, _P],
    'ssg_tool_mfma_peak_bf16': [_P, _I, _I, _P],
    'ssg_tool_mfma_peak_bf16_data': [_P, _I, _I, _P, _P],
    'ssg_tool_mfma_peak_bf16_data16': [_P, _I, _I, _P, _P],
    'ssg_tool_copy_f32': [_P, _P, _L, _P],
}

# bf16 twins: same argument lists as their _f32 namesakes
for _n in ('ssg_bn_stats', 'ssg_bn_apply', 'ssg_bn_bwd_reduce', 'ssg_bn_bwd_apply', 'ssg_dwconv2d_fwd', 'ssg_dwconv2d_dgrad', 'ssg_dwconv2d_wgrad', 'ssg_channel_scale_fwd', 'ssg_sample_channel_sum', 'ssg_broadcast_rows'):
    SIGNATURES[_n + '_bf16'] = SIGNATURES[_n + '_f32']
SIGNATURES['ssg_add_bf16'] = [_P, _I, _P, _I, _L, _I, _P, _I, _P]
SIGNATURES['ssg_convert_f32_to_bf16'] = [_P, _I, _L, _I, _P, _I, _P]
SIGNATURES['ssg_convert_bf16_to_f32'] = [_P, _I, _L, _I, _P, _I, _P]
_RESTYPES = {
    'ssg_conv2d_wgrad_workspace_bytes': C.c_int64,
    'ssg_conv2d_workspace_bytes': C.c_int64,
    'ssg_pack_weights_split_bytes': C.c_int64,
    'ssg_bn_workspace_bytes': C.c_int64,
    'ssg_se_gate_workspace_floats': C.c_int64,
    'ssg_seg_loss_workspace_bytes': C.c_int64,
    'ssg_dwconv2d_wgrad_workspace_bytes': C.c_int64,
    'ssg_spectral_norm_workspace_bytes': C.c_int64,
    'ssg_linear_fwd_workspace_bytes': C.c_int64,
    'ssg_sample_channel_sum_workspace_bytes': C.c_int64,
    'ssg_gemm_wgrad_bf16_workspace_bytes': C.c_int64,
    'ssg_conv2d_thin_bf16_wgrad_workspace_bytes': C.c_int64,
    'ssg_bn_stats_from_partials_workspace_bytes': C.c_int64,
}
_NO_STATUS = set(_RESTYPES) | {'ssg_abi_version', 'ssg_conv2d_split_bn', 'ssg_conv2d_in_affine_ok', 'ssg_conv2d_wgrad_in_affine_ok', 'ssg_conv2d_bwd_stats_ok', 'ssg_conv2d_kernel_id', 'ssg_conv2d_bnpart_rows', 'ssg_conv2d_wgrad_kernel_id',
                                'ssg_spade_conv_modulate_ok', 'ssg_se_gate_ok'}

ABI_VERSION = 9          # ssg_abi_version() of the library this ctypes table (ConvDesc layout, SIGNATURES) was written against

_lib = None


def load():
    """Load libssunet_hip.so (once).  Raises HipLibraryError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            'HIP extension not built: %s is missing. Run `python -c "import __graft_entry__ as g; g.build()"` '
            '(or `make -C ssunet-gan_amd/csrc`). There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.ssg_last_error.restype = C.c_char_p
    lib.ssg_last_error.argtypes = []
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, C.c_int)
    got = lib.ssg_abi_version()
    if got != ABI_VERSION:                # a stale .so with another descriptor layout would corrupt memory, not just fail
        raise HipLibraryError('%s reports ABI version %d, this package was written against %d: rebuild it '
                              '(`make -C ssunet-gan_amd/csrc`)' % (LIB_PATH, got, ABI_VERSION))
    _lib = lib
    return lib


def call(name, *args):
    """Call an ABI entry point; non-zero status -> RuntimeError with the library's message."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if name in _NO_STATUS:
        return rc
    if rc != 0:
        msg = lib.ssg_last_error().decode('utf-8', 'replace')
        raise HipLibraryError('%s failed (status %d): %s' % (name, rc, msg))
    return rc


def stream_ptr():
    """hipStream_t of torch's current stream on the current device."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(t):
    if not t.is_cuda:
        raise HipLibraryError('ssunet-gan_amd ops run only on an MI355X (cuda/HIP) device; got a %s tensor. '
                              'There is no CPU fallback.' % t.device)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
