"""ctypes binding of libveon_hip.so (the C ABI declared in include/veon_hip.h).

The product path has NO fallback: if the library is missing or a tensor is not
on a ROCm device, the call raises.  Pointers are passed as raw device addresses
(``tensor.data_ptr()``) and the launch goes to PyTorch's current HIP stream, so
the ops are hipGraph-capturable and race-free against surrounding torch ops
(the reference launches on the legacy default stream, bev_pool_cuda.cu:127,136).
"""
import ctypes
import os

import torch

from . import half as _half

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libveon_hip.so')
LIB_PATHS = {'bf16': LIB_PATH, 'fp16': os.path.join(_HERE, 'libveon_hip_f16.so')}
# tools/lib_ab.py only: time another BUILD of the bf16 library in the same process tree
# (same-box A/B of kernel changes); never set in production
if os.environ.get('VEON_HIP_LIB'):
    LIB_PATHS['bf16'] = os.environ['VEON_HIP_LIB']
_libs = {}  # flavour -> loaded library
_lib = None

_vp = ctypes.c_void_p
_ci = ctypes.c_int
_i64 = ctypes.c_int64
_cf = ctypes.c_float



class VitBlockWeights(ctypes.Structure):
    """``veon_vit_block_weights`` of include/veon_hip.h."""
    _fields_ = [('ln1_w', ctypes.c_void_p), ('ln1_b', ctypes.c_void_p),
                ('w_qkv', ctypes.c_void_p), ('b_qkv', ctypes.c_void_p),
                ('w_proj', ctypes.c_void_p), ('b_proj', ctypes.c_void_p),
                ('gamma1', ctypes.c_void_p),
                ('ln2_w', ctypes.c_void_p), ('ln2_b', ctypes.c_void_p),
                ('w_fc1', ctypes.c_void_p), ('b_fc1', ctypes.c_void_p),
                ('w_fc2', ctypes.c_void_p), ('b_fc2', ctypes.c_void_p),
                ('gamma2', ctypes.c_void_p),
                ('ln1_eps', ctypes.c_float), ('ln2_eps', ctypes.c_float),
                ('mlp_dim', ctypes.c_int), ('act', ctypes.c_int),
                ('q_log2', ctypes.c_int)]


_SIGNATURES = {
    'veon_abi_version': (_ci, []),
    'veon_half_mode': (_ci, []),
    'veon_status_string': (ctypes.c_char_p, [_ci]),
    'veon_bev_pool_v2_fwd': (_ci, [_ci, _ci] + [_vp] * 8 + [_vp]),
    'veon_bev_pool_v2_bwd': (_ci, [_ci, _ci] + [_vp] * 10 + [_vp]),
    'veon_bev_pool_v2_fwd_fused': (_ci, [_ci, _ci, _ci, _i64] + [_vp] * 9 + [_ci, _vp]),
    'veon_feat_nchw_to_nhwc': (_ci, [_vp, _vp] + [_ci] * 4 + [_vp]),
    'veon_bev_pool_v2_fwd_fused_strided': (_ci, [_ci, _ci, _ci, _i64, _vp, _vp, _ci] + [_vp] * 7 + [_i64, _vp]),
    'veon_bev_pool_v2_fwd_fused_ex': (_ci, [_ci, _ci, _ci, _i64, _vp, _vp, _ci] + [_vp] * 7 + [_ci, _vp]),
    'veon_bev_pool_v2_fwd_maxpool_ex': (_ci, [_ci] * 9 + [_vp, _vp, _ci] + [_vp] * 7 + [_vp]),
    'veon_bev_pool_v2_fwd_maxpool_padded': (_ci, [_ci] * 9 + [_vp, _vp, _ci] + [_vp] * 7 + [_vp]),
    'veon_bev_pool_tile_voxels': (_ci, []),
    'veon_volume_maxpool2_f32': (_ci, [_vp, _vp, _i64] + [_ci] * 3 + [_vp]),
    'veon_pool_debug_set': (None, [_ci]),
    'veon_pool_tune_set': (None, [_ci, _ci, _ci]),
    'veon_bev_pool_voxel_table_ints': (_i64, [_ci, _i64]),
    'veon_bev_pool_voxel_table': (_ci, [_ci, _ci, _ci, _i64, _vp, _vp, _vp, _vp, _vp]),
    'veon_bev_pool_v2_fwd_rows': (_ci, [_ci, _ci, _i64, _vp, _vp, _ci, _vp, _vp, _vp, _vp,
                                        _i64, _i64, _ci, _vp]),
    'veon_bev_pool_v2_fwd_rows_maxpool': (_ci, [_ci] * 8 + [_vp, _vp, _ci, _vp, _vp, _vp, _vp,
                                                _ci, _i64, _vp]),
    'veon_bev_pool_rows_maxpool_chunk': (_ci, []),
    'veon_bev_pool_v2_fwd_rows_maxpool_ordered': (_ci, [_ci] * 8 + [_vp, _vp, _ci, _vp, _vp, _vp,
                                                        _vp, _ci, _i64, _vp, _vp]),
    'veon_bev_pool_plan_ints': (_i64, [_ci, _i64]),
    'veon_bev_pool_plan': (_ci, [_ci, _ci, _ci, _i64, _vp, _vp, _vp, _vp, _vp]),
    'veon_bev_pool_row_table': (_ci, [_ci, _ci, _ci, _i64, _ci, _vp, _vp, _vp, _vp, _vp, _vp]),
    'veon_bev_pool_v2_fwd_maxpool': (_ci, [_ci] * 9 + [_vp] * 9 + [_vp]),
    'veon_downsample_depth': (_ci, [_ci] * 4 + [_vp, _vp, _vp]),
    'veon_two_hot_depth': (_ci, [_ci] * 5 + [_cf] * 3 + [_vp, _vp, _vp]),
    'veon_two_hot_window_slots': (_ci, [_ci, _cf, _cf]),
    'veon_two_hot_window': (_ci, [_ci] * 5 + [_cf] * 4 + [_ci, _vp, _vp, _vp, _vp]),
    'veon_gemm_ring_set': (None, [_ci]),
    'veon_vit_cast_bf16': (_ci, [_vp, _vp, _i64, _vp]),
    'veon_vit_layernorm': (_ci, [_vp, _vp, _vp, _vp, _ci, _ci, _cf, _vp]),
    'veon_vit_layernorm_padded': (_ci, [_vp, _vp, _vp, _vp, _ci, _ci, _ci, _cf, _vp]),
    'veon_vit_patchify': (_ci, [_vp, _vp] + [_ci] * 7 + [_vp]),
    'veon_vit_gemm': (_ci, [_vp] * 6 + [_ci] * 4 + [_vp]),
    'veon_vit_attention': (_ci, [_vp, _vp, _i64, _i64, _vp, _ci, _ci, _ci, _ci, _vp]),
    'veon_vit_attention_log2': (_ci, [_vp, _vp, _i64, _i64, _vp, _ci, _ci, _ci, _ci, _vp]),
    'veon_vit_block_workspace_bytes': (_i64, [_ci] * 4),
    'veon_vit_gemm_splitk_plan': (_i64, [_ci, _ci, _ci, _vp]),
    'veon_vit_gemm_splitk': (_ci, [_vp] * 5 + [_ci] * 3 + [_vp, _i64, _vp, _i64, _vp]),
    'veon_vit_block': (_ci, [_vp, _vp, _vp, _i64, _i64, _vp, _i64] + [_ci] * 4 + [_vp]),
    'veon_conv3d_guard_rows': (_i64, [_ci, _ci]),
    'veon_conv_debug_set': (None, [_ci]),
    'veon_conv_tile_choice': (_ci, [_ci] * 8),
    'veon_conv3d_k3_bf16': (_ci, [_vp] * 6 + [_ci] * 7 + [_vp]),
    'veon_conv2d_k3_bf16': (_ci, [_vp] * 6 + [_ci] * 6 + [_vp]),
    'veon_conv2d_k3_bf16_ex': (_ci, [_vp] * 8 + [_ci] * 6 + [_vp]),
    'veon_conv2d_k3s2_bf16': (_ci, [_vp] * 6 + [_ci] * 6 + [_vp]),
    'veon_image_resize_bilinear': (_ci, [_vp, _vp] + [_ci] * 6 + [_vp]),
    'veon_image_dot': (_ci, [_vp, _vp, _cf, _vp] + [_ci] * 5 + [_vp]),
    'veon_tokens_to_image': (_ci, [_vp, _i64] + [_ci] * 6 + [_vp, _ci, _vp]),
    'veon_image_subsample': (_ci, [_vp, _vp] + [_ci] * 5 + [_vp]),
    'veon_occ_classify': (_ci, [_vp, _vp, _ci, _vp, _vp] + [_ci] * 7 + [_vp, _vp, _vp, _vp]),
    'veon_image_pack_bf16': (_ci, [_vp, _ci, _vp] + [_ci] * 4 + [_vp]),
    'veon_image_unpack': (_ci, [_vp, _vp, _ci] + [_ci] * 4 + [_vp]),
    'veon_alloc_contiguous': (_ci, [_vp, _i64]),
    'veon_alloc_device_flags': (_ci, [_vp, _i64, ctypes.c_uint]),
    'veon_free_device': (_ci, [_vp]),
    'veon_layernorm_f32': (_ci, [_vp] * 4 + [_ci, _ci, _cf, _vp]),
    'veon_layernorm_f32_add_nearest': (_ci, [_vp] * 5 + [_ci] * 7 + [_cf, _vp]),
    'veon_layernorm_f32_to_padded': (_ci, [_vp] * 4 + [_ci] * 4 + [_cf, _vp]),
    'veon_image_layernorm_bf16': (_ci, [_vp] * 4 + [_ci] * 5 + [_cf, _vp, _vp]),
    'veon_deform_attention_bf16': (_ci, [_vp] * 4 + [_ci] * 8 + [_vp]),
    'veon_volume_warp_bf16': (_ci, [_vp] * 3 + [_ci] * 5 + [_vp]),
    'veon_warp_affine': (_ci, [_vp, _vp, _ci, _vp, _vp, _vp, _ci, _vp]),
    'veon_volume_zero_halo_bf16': (_ci, [_vp] + [_ci] * 5 + [_vp]),
    'veon_volume_pack_bf16': (_ci, [_vp, _vp] + [_ci] * 5 + [_vp]),
    'veon_volume_unpack_f32': (_ci, [_vp, _vp] + [_ci] * 5 + [_vp]),
    'veon_camera_matrices': (_ci, [_ci] + [_vp] * 6 + [_vp]),
    'veon_sensor2keyego': (_ci, [_ci, _ci, _vp, _vp, _vp, _vp]),
    'veon_lidar_coor': (_ci, [_ci] * 5 + [_vp] * 9 + [_vp]),
    'veon_lss_prepare_workspace_bytes': (_i64, [_i64, _i64]),
    'veon_lss_prepare_cameras': (_ci, [_ci] * 5 + [_vp] * 8 + [_vp] * 3 + [_i64, _vp, _i64, _ci]
                                 + [_vp] * 8 + [_vp]),
    'veon_lss_prepare_cameras_sparse': (_ci, [_ci] * 5 + [_vp] * 8 + [_vp] * 3
                                        + [_i64, _vp, _i64, _ci] + [_vp] * 8 + [_vp, _cf, _vp]),
    'veon_lss_prepare_cameras_twohot': (_ci, [_ci] * 5 + [_vp] * 8 + [_vp] * 3
                                        + [_i64, _vp, _i64, _ci] + [_vp] * 8 + [_vp, _ci, _vp]),
    'veon_lss_prepare': (_ci, [_ci] * 5 + [_vp] * 9 + [_vp] * 3 + [_i64, _vp, _i64]
                         + [_vp] * 7 + [_vp]),
}

LAYOUT_BZYXC = 0
LAYOUT_BCZYX = 1
FEAT_F32, FEAT_F16, FEAT_BF16 = 0, 1, 2


class VeonHipError(RuntimeError):
    pass


def declared_symbols():
    """Every entry point include/veon_hip.h declares (checked by the CPU tests)."""
    return sorted(_SIGNATURES)


def register(name, restype, argtypes):
    _SIGNATURES[name] = (restype, argtypes)
    for loaded in _libs.values():
        fn = getattr(loaded, name)
        fn.restype, fn.argtypes = restype, argtypes


def lib():
    """The native library of the process's half flavour (veon_amd/half.py):
    libveon_hip.so (bf16 operands) or libveon_hip_f16.so (fp16), the same entry
    points in both; raise (never fall back) when it is absent."""
    global _lib
    flavour = _half.name()
    loaded = _libs.get(flavour)
    if loaded is None:
        path = LIB_PATHS[flavour]
        if not os.path.exists(path):
            raise VeonHipError(
                'veon_amd: %s not found -- build it with `python -m veon_amd.build` '
                '(hipcc --offload-arch=gfx950). There is no CPU fallback.' % path)
        loaded = ctypes.CDLL(path)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(loaded, name)
            fn.restype, fn.argtypes = restype, argtypes
        if loaded.veon_half_mode() != (1 if flavour == 'fp16' else 0):
            raise VeonHipError('%s was not built for %s operands' % (path, flavour))
        _libs[flavour] = loaded
    _lib = loaded
    return loaded


# number of native calls per entry point (every call ends in check()); the GPU
# tests assert on it so that a silent non-native path would be noticed
CALLS = {}


def check(status, what):
    CALLS[what] = CALLS.get(what, 0) + 1
    if status != 0:
        msg = lib().veon_status_string(status).decode()
        raise VeonHipError('%s failed: %s (status %d)' % (what, msg, status))


def require_device(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise VeonHipError(
                'veon_amd HIP op called with a %s tensor: the op runs only on a '
                'ROCm device (no CPU path).' % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise VeonHipError('tensors on different devices: %s vs %s' % (dev, t.device))
    return dev


def require_half(*tensors):
    """Operands of the MFMA kernels must have THE half dtype of the process
    (veon_amd/half.py) -- a module built under the other flavour holds packed
    weights of the other type: raise, never reinterpret the bits."""
    want = _half.dtype()
    for t in tensors:
        if t is not None and t.dtype != want:
            raise VeonHipError(
                'half-precision operand is %s but this process runs the %s flavour of the '
                'library (VEON_HALF / veon_amd.half.set_half_dtype): build the module under '
                'the flavour it is used with' % (t.dtype, _half.name()))


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def stream_ptr(device):
    """PyTorch's current stream on ``device`` as a raw hipStream_t (no Stream
    object is built: this sits on every launch)."""
    idx = device.index
    if idx is None:
        idx = torch.cuda.current_device()
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(idx))


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(device):
    """``torch.cuda.device(device)`` unless it is the current device already
    (then a no-op context: the guard costs more than a small launch)."""
    if device.index is None or device.index == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)
