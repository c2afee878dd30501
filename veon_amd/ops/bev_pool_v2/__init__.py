from .bev_pool import QuickCumsumCuda, TRTBEVPoolv2, bev_pool_v2

__all__ = ['bev_pool_v2', 'QuickCumsumCuda', 'TRTBEVPoolv2']
