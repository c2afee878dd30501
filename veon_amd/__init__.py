"""veon_amd -- MI355X-native (gfx950 / CDNA4) implementation of VEON's
multi-camera 2D->3D lift hot path: the LSS view transformers and bev_pool_v2,
behind the reference's own op / NECKS plugin names.

Reference-path mirror:
    mmdet3d.ops.bev_pool_v2.bev_pool          -> veon_amd.ops.bev_pool_v2.bev_pool
    mmdet3d.models.builder (NECKS, build_neck) -> veon_amd.models.builder
    mmdet3d.models.necks.view_transformer      -> veon_amd.models.necks.view_transformer
    mmdet3d.models.necks.view_transformer_raw  -> veon_amd.models.necks.view_transformer_raw
Native code: veon_amd/csrc/*.hip -> veon_amd/libveon_hip.so (C ABI in
include/veon_hip.h), loaded with ctypes by veon_amd._lib.
"""
__version__ = '0.1.0'
