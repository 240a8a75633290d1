"""ctypes binding of libsenas_hip.so (the C ABI declared in include/senas_hip.h).

There is no fallback: if the shared library is missing or an entry point reports an error this
module raises.  Nothing here imports ``oracle/``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libsenas_hip.so')

OK = 0
EXPECTED_ABI = 35          # senas_abi_version() of the library these bindings were written against (include/senas_hip.h)
MAX_TERMS = 32
SKIP_MAX = 8               # SENAS_SKIP_MAX
MAX_STACK = 4


class ConvGeom(C.Structure):
    """senas_conv_geom (include/senas_hip.h)."""
    _fields_ = [(k, C.c_int32) for k in ('n', 'hi', 'wi', 'ci', 'ho', 'wo', 'co', 'kh', 'kw', 'stride', 'pad', 'dil',
                                         'transposed', 'groups')]


class ConvEpilogue(C.Structure):
    """senas_conv_epilogue (include/senas_hip.h)."""
    _fields_ = [('scale', C.c_void_p), ('bias', C.c_void_p), ('addend', C.c_void_p), ('add_scale', C.c_void_p),
                ('relu', C.c_int32)]


class BnReluItem(C.Structure):
    """senas_bnrelu_item (include/senas_hip.h)."""
    _fields_ = [('z', C.c_void_p), ('y', C.c_void_p), ('mask8', C.c_void_p), ('stats', C.c_void_p), ('gamma', C.c_void_p),
                ('beta', C.c_void_p), ('running_mean', C.c_void_p), ('running_var', C.c_void_p), ('num_batches_tracked', C.c_void_p),
                ('mean_invstd', C.c_void_p), ('dy', C.c_void_p), ('dy_pixel_stride', C.c_int64), ('dz', C.c_void_p),
                ('dgamma', C.c_void_p), ('dbeta', C.c_void_p), ('sums', C.c_void_p)]


class DsTailItem(C.Structure):
    """senas_dstail_item (include/senas_hip.h)."""
    _fields_ = [('z1', C.c_void_p), ('stats1', C.c_void_p), ('gamma1', C.c_void_p), ('beta1', C.c_void_p),
                ('running_mean1', C.c_void_p), ('running_var1', C.c_void_p), ('num_batches_tracked1', C.c_void_p),
                ('mean_invstd', C.c_void_p), ('w', C.c_void_p), ('z2', C.c_void_p), ('stats2', C.c_void_p), ('dz2', C.c_void_p),
                ('dz2_pixel_stride', C.c_int64), ('sums', C.c_void_p), ('dz1', C.c_void_p), ('dgamma1', C.c_void_p),
                ('dbeta1', C.c_void_p), ('dw', C.c_void_p), ('dw_acc', C.c_void_p)]


class ArchMix(C.Structure):
    """senas_arch_mix (include/senas_hip.h)."""
    _fields_ = [('alpha', C.c_void_p * 4), ('beta', C.c_void_p * 2), ('gamma', C.c_void_p), ('s_alpha', C.c_void_p * 4),
                ('s_beta', C.c_void_p * 2), ('s_gamma', C.c_void_p), ('M', C.c_void_p * 2), ('dM', C.c_void_p * 2), ('dG', C.c_void_p),
                ('d_alpha', C.c_void_p * 4), ('d_beta', C.c_void_p * 2), ('d_gamma', C.c_void_p),
                ('k', C.c_int32), ('ops', C.c_int32), ('nodes', C.c_int32), ('grows', C.c_int32), ('slots', C.c_int32)]


class SumItem(C.Structure):
    """senas_sum_item (include/senas_hip.h)."""
    _fields_ = [('part', C.c_void_p), ('dw', C.c_void_p), ('kind', C.c_int32), ('A', C.c_int32), ('B', C.c_int32), ('taps', C.c_int32),
                ('n_elem', C.c_int32), ('nblk', C.c_int32)]


MAX_SUMS = 64
MAX_DSTAIL = 12
MAX_BNRELU = 8
MAX_DWMULTI = 12
MAX_PWMULTI = 8
UNSUPPORTED = -3

_T = MAX_TERMS


class NodeDesc(C.Structure):
    """senas_node_desc (include/senas_hip.h)."""
    _fields_ = [('nterms', C.c_int32), ('n', C.c_int32), ('c', C.c_int32), ('training', C.c_int32), ('relu', C.c_int32),
                ('hw', C.c_int64), ('eps', C.c_float), ('momentum', C.c_float),
                ('stats', C.c_void_p * _T), ('gamma', C.c_void_p * _T), ('beta', C.c_void_p * _T),
                ('running_mean', C.c_void_p * _T), ('running_var', C.c_void_p * _T),
                ('num_batches_tracked', C.c_void_p * _T), ('se_w1', C.c_void_p * _T), ('se_w2', C.c_void_p * _T),
                ('se_mid', C.c_int32 * _T), ('stats_image_stride', C.c_int32 * _T), ('mix', C.c_void_p)]


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
_G = C.POINTER(ConvGeom)
_N = C.POINTER(NodeDesc)
_PP = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/senas_hip.h one to one
SIGNATURES = {
    'senas_conv2d_ws_bytes': (C.c_int64, [_G]),
    'senas_conv2d_fwd': (_I, [_G, _P, _P, _P, _I, _P, _P, _P, _P]),
    'senas_conv2d_fwd_bf16s': (_I, [_G, _P, _P, _P, _I, _P, _P, _P, _P]),
    'senas_conv2d_bwd_data_bf16s': (_I, [_G, _P, _P, _P, _I, _P, _P, _P, _P]),
    'senas_conv2d_bwd_weight_bf16s': (_I, [_G, _P, _I, _P, _P, _P, C.POINTER(SumItem), _P]),
    'senas_conv2d_fwd_planar': (_I, [_G, _P, _P, _P, _L, _I, _P, _P, _P, _P]),
    'senas_conv2d_fwd_epilogue': (_I, [_G, _P, _P, _P, _I, C.POINTER(ConvEpilogue), _P, _P, _P]),
    'senas_conv2d_bwd_data': (_I, [_G, _P, _P, _P, _I, _P, _P, _P, _P]),
    'senas_conv2d_pack_layout': (_I, [_G, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    'senas_pack_batched': (_I, [_P, _I, _L, _P]),
    'senas_copy_rows_batched': (_I, [_P, _I, _L, _P]),
    'senas_conv2d_bwd_weight_ws': (_I, [_G, _P, _P]),
    'senas_conv2d_bwd_weight': (_I, [_G, _P, _I, _P, _P, _P, _I, _P]),
    'senas_conv2d_bwd_weight_deferred': (_I, [_G, _P, _I, _P, _P, _P, _I, C.POINTER(SumItem), _P]),
    'senas_wgrad_sum_batched': (_I, [C.POINTER(SumItem), _I, _P]),
    'senas_avgpool3_fwd': (_I, [_I, _I, _I, _I, _I, _P, _I, _P, _P, _P]),
    'senas_avgpool3_bwd': (_I, [_I, _I, _I, _I, _I, _P, _I, _P, _P, _P]),
    'senas_maxpool3_fwd': (_I, [_I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P]),
    'senas_maxpool3_bwd': (_I, [_I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P]),
    'senas_bilinear2x_fwd': (_I, [_I, _I, _I, _I, _P, _P, _P, _P]),
    'senas_bilinear2x_bwd': (_I, [_I, _I, _I, _I, _P, _P, _P]),
    'senas_unstack_fwd': (_I, [_I, _L, _I, _I, _P, _PP, _PP, _P]),
    'senas_stamp': (_I, [_P, _P]),
    'senas_relay_marker': (_I, [_P]),
    'senas_marker': (_I, [_I, _P]),
    'senas_sched_contract': (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _I]),
    'senas_sched_plan2': (_I, [_I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    'senas_stream_create': (_I, [C.POINTER(C.c_void_p)]),
    'senas_sched_create': (_I, [_P, _I, C.POINTER(C.c_void_p)]),
    'senas_sched_plan': (_I, [_I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    'senas_sched_launch': (_I, [_P, _P]),
    'senas_sched_info': (_I, [_P, C.POINTER(C.c_int32)]),
    'senas_sched_destroy': (None, [_P]),
    'senas_relu_fwd': (_I, [_L, _P, _P, _P]),
    'senas_relu_bwd': (_I, [_L, _P, _P, _P, _P]),
    'senas_blend2_fwd': (_I, [_L, _P, _P, _P, _P, _P]),
    'senas_blend2_bwd': (_I, [_L, _P, _P, _P, _P, _P, _P, _P, _P]),
    'senas_skipcat_fwd': (_I, [_L, _I, _I, _PP, _P, _I, _P, _P, _P]),
    'senas_skipcat_bwd': (_I, [_L, _I, _I, _P, _PP, _P, _I, _P, _PP, _P, _P]),
    'senas_chan_stats': (_I, [_I, _L, _I, _P, _P, _P]),
    'senas_bn_finalize': (_I, [_I, _L, _I, _P, _P, _P, _P, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P]),
    'senas_dwconv_multi_fwd': (_I, [_G, _I, _P, _PP, _PP, _PP, _P]),
    'senas_dwconv_multi_bwd_data': (_I, [_G, _I, _PP, _PP, _P, _P]),
    'senas_dwconv_multi_ws_bytes': (C.c_int64, [_G, _I]),
    'senas_dwconv_multi_bwd_weight': (_I, [_G, _I, _P, _PP, _PP, _P, _P]),
    'senas_dwconv_multi_bwd_weight_deferred': (_I, [_G, _I, _P, _PP, _PP, _P, C.POINTER(SumItem), _P]),
    'senas_dwconv_pair_fwd': (_I, [_G, _I, _G, _I, _P, _PP, _PP, _PP, _P]),
    'senas_dwconv_pair_fwd_xs': (_I, [_G, _I, _G, _I, _P, _PP, _PP, _PP, _PP, _P]),
    'senas_dwconv_pair_bwd_weight_xs': (_I, [_G, _I, _G, _I, _P, _PP, _PP, _PP, _P, C.POINTER(SumItem), _P]),
    'senas_dwconv_pair_bwd_data': (_I, [_G, _I, _G, _I, _PP, _PP, _P, _P]),
    'senas_dwconv_pair_ws_bytes': (C.c_int64, [_G, _I, _G, _I]),
    'senas_dwconv_pair_bwd_weight': (_I, [_G, _I, _G, _I, _P, _PP, _PP, _P, C.POINTER(SumItem), _P]),
    'senas_pw_multi_fwd': (_I, [_I, _I, _L, _I, _I, _PP, _PP, _PP, _PP, _P]),
    'senas_pw_multi_bwd_data': (_I, [_I, _I, _L, _I, _I, _PP, _PP, _PP, _P]),
    'senas_pw_multi_ws_bytes': (C.c_int64, [_I, _I, _L, _I, _I]),
    'senas_pw_multi_bwd_weight': (_I, [_I, _I, _L, _I, _I, _PP, _PP, _PP, _P, _P]),
    'senas_bnrelu_multi_fwd': (_I, [C.POINTER(BnReluItem), _I, _I, _L, _I, _I, _F, _F, _P]),
    'senas_bnrelu_multi_bwd': (_I, [C.POINTER(BnReluItem), _I, _I, _L, _I, _P]),
    'senas_dstail_fwd': (_I, [C.POINTER(DsTailItem), _I, _I, _L, _I, _I, _I, _F, _F, _P]),
    'senas_dstail_ws_bytes': (C.c_int64, [_I, _I, _L, _I, _I]),
    'senas_dstail_bwd': (_I, [C.POINTER(DsTailItem), _I, _I, _L, _I, _I, _P]),
    'senas_combine_fwd': (_I, [_I, _L, _I, _I, _PP, _P, _P, _P, _I, _P, _P]),
    'senas_combine_bwd_reduce': (_I, [_I, _L, _I, _I, _PP, _P, _P, _I, _P, _P, _P]),
    'senas_combine_bwd_apply': (_I, [_I, _L, _I, _I, _PP, _P, _P, _I, _P, _P, _P, _PP, _P, _P]),
    'senas_sum_n': (_I, [_I, _L, _PP, _P, _P]),
    'senas_sum_n_strided': (_I, [_I, _L, _I, _PP, _P, _P, _P]),
    'senas_dice_ce_fwd': (_I, [_L, _I, _P, _P, _F, _F, _F, _I, _P, _P, _P, _P]),
    'senas_dice_ce_bwd': (_I, [_L, _I, _P, _P, _P, _P, _P, _P]),
    'senas_seg_metric_update': (_I, [_I, _L, _I, _P, _P, _F, _P, _P, _P, _P]),
    'senas_sgd_clip_step': (_I, [_P, _I, _L, _P, _F, _F, _F, _F, _F, _I, _I, _P, _P]),
    'senas_node_fwd': (_I, [_N, _PP, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P]),
    'senas_node_bwd': (_I, [_N, _PP, _P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _PP, _PP, _P, _I, _PP, _PP, _P, _PP, _P, _P, _P]),
    'senas_conv2d_fwd_pair': (_I, [_G, _G, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    'senas_conv2d_fwd_pair_planar': (_I, [_G, _G, _P, _P, _P, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P]),
    'senas_conv2d_bwd_data_pair': (_I, [_G, _G, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    'senas_conv2d_bwd_weight_pair': (_I, [_G, _G, _P, _I, _P, _P, _P, _P, _P, _P, C.POINTER(SumItem), C.POINTER(SumItem), _P]),
    'senas_conv2d_fwd_lp': (_I, [_G, _P, _P, _P, _I, _P, _P, _P, _I, _P]),
    'senas_conv2d_bwd_data_lp': (_I, [_G, _P, _P, _P, _I, _P, _P, _P, _I, _P]),
    'senas_conv2d_bwd_weight_ws_lp': (_I, [_G, _I, C.POINTER(C.c_int64)]),
    'senas_conv2d_bwd_weight_lp': (_I, [_G, _P, _I, _P, _P, _P, _I, C.POINTER(SumItem), _P]),
    'senas_conv2d_pack_layout_lp': (_I, [_G, _I, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    'senas_pack_batched_lp': (_I, [_P, _I, _L, _P]),
    'senas_conv2d_kernel_name_lp': (C.c_char_p, [_G, _I, _I]),
    'senas_arch_mix_fwd': (_I, [C.POINTER(ArchMix), _P]),
    'senas_arch_mix_bwd': (_I, [C.POINTER(ArchMix), _P]),
    'senas_conv2d_kernel_name': (C.c_char_p, [_G, _I]),
    'senas_last_error': (C.c_char_p, []),
    'senas_abi_version': (_I, []),
}

_lib = None


class SenasHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SenasHipError('%s is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                                'or `make -C senas_amd/csrc`; there is no CPU fallback' % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the .so does not export the symbol
            fn.restype, fn.argtypes = res, args
        abi = handle.senas_abi_version()
        if abi != EXPECTED_ABI:
            # a stale build product would read the arguments of a changed entry point as something else (a stream pointer
            # as an int, ...): a wild launch instead of an error
            raise SenasHipError('%s reports ABI %d, these bindings expect %d: rebuild it with `make -C senas_amd/csrc`'
                                % (LIB_PATH, abi, EXPECTED_ABI))
        _lib = handle
    return _lib


def check(code, what):
    if code != OK:
        raise SenasHipError('%s failed (%d): %s' % (what, code, lib().senas_last_error().decode()))


def ptr_array(ptrs):
    arr = (C.c_void_p * len(ptrs))(*ptrs)
    return arr
