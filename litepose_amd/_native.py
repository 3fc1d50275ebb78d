"""ctypes binding of liblitepose_amd.so (the C ABI in include/litepose_amd.h).

torch is imported first so that the HIP runtime already in the process (torch's
libamdhip64, SONAME libamdhip64.so.7) is the one our library binds to; tensors are
only used as device-memory containers (``data_ptr``) and for the current stream.
There is NO fallback: a missing library is a hard error.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL: shares the HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'liblitepose_amd.so')
# diagnostics only: LP_NATIVE_FLAVOUR=regstage loads lib/liblitepose_amd_regstage.so, the same sources built
# with -DLP_NO_LDS_DMA -DLP_CLAIM_CU by `python -m litepose_amd.build --flavour regstage`; missing -> the usual loud failure
if os.environ.get('LP_NATIVE_FLAVOUR'):
    LIB_PATH = os.path.join(_HERE, 'lib', 'liblitepose_amd_%s.so' % os.environ['LP_NATIVE_FLAVOUR'])

LP_MAX_STAGES, LP_MAX_BLOCKS, LP_MAX_DECONV = 8, 32, 4


class LpArch(C.Structure):
    _fields_ = [
        ('input_channel', C.c_int32), ('num_stages', C.c_int32),
        ('num_blocks', C.c_int32 * LP_MAX_STAGES), ('stride', C.c_int32 * LP_MAX_STAGES),
        ('channel', C.c_int32 * LP_MAX_STAGES),
        ('expand', (C.c_int32 * LP_MAX_BLOCKS) * LP_MAX_STAGES),
        ('kernel', (C.c_int32 * LP_MAX_BLOCKS) * LP_MAX_STAGES),
        ('num_deconv', C.c_int32), ('deconv_filters', C.c_int32 * LP_MAX_DECONV),
        ('head_channels', C.c_int32 * LP_MAX_DECONV),
    ]


class LpParseParams(C.Structure):
    _fields_ = [
        ('num_joints', C.c_int32), ('max_num_people', C.c_int32),
        ('detection_threshold', C.c_double), ('tag_threshold', C.c_double),
        ('use_detection_val', C.c_int32), ('ignore_too_much', C.c_int32),
        ('nms_kernel', C.c_int32), ('joint_order', C.c_int32 * 32), ('tag_per_joint', C.c_int32),
    ]


class LitePoseNativeError(RuntimeError):
    pass


_lib = None
vp, i32, i64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t

_SIGS = {
    'lp_last_error': (C.c_char_p, []),
    'lp_version': (C.c_char_p, []),
    'lp_net_create': (i32, [C.POINTER(vp), C.POINTER(LpArch)]),
    'lp_net_destroy': (None, [vp]),
    'lp_net_num_keys': (i32, [vp]),
    'lp_net_key': (C.c_char_p, [vp, i32, C.POINTER(i64), C.POINTER(i32)]),
    'lp_net_set_weight': (i32, [vp, C.c_char_p, vp, C.POINTER(i64), i32]),
    'lp_net_finalize': (i32, [vp, i32]),
    'lp_net_set_storage': (i32, [vp, i32]),
    'lp_net_get_storage': (i32, [vp]),
    'lp_net_get_weight': (i32, [vp, C.c_char_p, vp, i64]),
    'lp_net_workspace_bytes': (sz, [vp, i32, i32, i32]),
    'lp_net_forward': (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp, sz, vp]),
    'lp_net_tap': (i64, [vp, C.c_char_p, vp, vp]),
    'lp_net_tap_offset': (i64, [vp, C.c_char_p, i32, i32, i32, C.POINTER(i64)]),
    'lp_net_set_profiling': (i32, [vp, i32]),
    'lp_net_set_streams': (i32, [vp, i32]),
    'lp_net_set_option': (i32, [vp, C.c_char_p, i32]),
    'lp_net_get_option': (i32, [vp, C.c_char_p]),
    'lp_diag_read': (i32, [vp, i32, i32]),
    'lp_phase_trace_read': (i32, [vp, i32]),
    'lp_wg_trace_read': (i32, [vp, i32, i32]),
    'lp_net_profile': (i32, [vp, vp, vp, vp, vp, i32]),
    'lp_net_profile2': (i32, [vp, vp, vp, vp, vp, vp, i32]),
    'lp_net_profile_launches': (i32, [vp, vp, vp, vp, vp, i32]),
    'lp_tta_merge': (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, sz, vp]),
    'lp_tta_workspace_bytes': (sz, [i32, i32, i32, i32]),
    'lp_tta_merge_ex': (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp,
                              sz, vp]),
    'lp_maps_accumulate': (i32, [vp, vp, i64, vp]),
    'lp_tta_stage': (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, sz, vp]),
    'lp_tta_stage_add': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, sz,
                               vp]),
    'lp_tta_project': (i32, [vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    'lp_parse_mid': (i32, [vp, i32, i32, i32, i32, i32, C.POINTER(LpParseParams), i32, i32, i32,
                         vp, vp, vp, vp, sz, vp]),
    'lp_parse_dm': (i32, [vp, vp, i32, i32, i32, i32, i32, C.POINTER(LpParseParams), i32, i32, i32,
                        vp, vp, vp, vp, sz, vp]),
    'lp_peaks_topk': (i32, [vp, vp, i32, i32, i32, i32, i32, C.POINTER(LpParseParams), vp, vp, vp, vp]),
    'lp_group': (i32, [vp, vp, vp, i32, i32, i32, C.POINTER(LpParseParams), i32, vp, vp, vp]),
    'lp_refine_workspace_bytes': (sz, [i32, i32]),
    'lp_adjust_refine': (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, sz, vp]),
    'lp_parse_workspace_bytes': (sz, [i32, i32, i32, i32, i32]),
    'lp_parse': (i32, [vp, vp, i32, i32, i32, i32, i32, C.POINTER(LpParseParams), i32, i32, i32,
                       vp, vp, vp, vp, sz, vp]),
    'lp_preprocess': (i32, [vp, i32, i32, C.POINTER(C.c_double), i32, i32, C.POINTER(C.c_float),
                            C.POINTER(C.c_float), vp, vp, vp]),
    'lp_preprocess_batch': (i32, [vp, i32, i32, i32, C.POINTER(C.c_double), i32, i32, C.POINTER(C.c_float),
                                  C.POINTER(C.c_float), vp, vp, vp]),
    'lp_stream_abort_capture': (i32, [vp]),
    'lp_final_preds': (i32, [vp, vp, i32, i32, i32, i32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                             i32, i32, vp]),
}
EXPORTS = sorted(_SIGS)


def lib():
    """Load (once) and return the shared library; raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LitePoseNativeError(
                'HIP extension %s is missing -- build it with `python -m litepose_amd.build` '
                '(there is no CPU fallback)' % LIB_PATH)
        l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what=''):
    if rc < 0:
        raise LitePoseNativeError('%s failed (%d): %s' % (what, rc, lib().lp_last_error().decode()))
    return rc


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t):
    """Device pointer of a torch tensor after validating what the C side assumes."""
    if t is None:
        return None
    if not t.is_cuda:
        raise LitePoseNativeError('expected a GPU tensor')
    if not t.is_contiguous():
        raise LitePoseNativeError('expected a contiguous tensor')
    return C.c_void_p(t.data_ptr())
