// bev_pool_v2 for MI355X (gfx950 / CDNA4): hand-written HIP kernels + C ABI.
//
// Replaces the reference's only native code,
//   mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu  (bev_pool_v2_kernel :21-48,
//   bev_pool_grad_kernel :67-121, launchers :125-140),
// designed for CDNA4 rather than translated: the op is HBM-write-bound (a
// 205-655 MB volume against a few MB of inputs), so the fused kernels write
// the volume exactly once -- zero-fill, pooled sums and the output layout in
// one pass -- with 64-lane wavefront-wide contiguous stores.
//
// Numerics: every pooled value is the serial fmaf chain over its interval in
// storage order, i.e. exactly what one thread of the reference kernel computes
// (:38-43, nvcc contracts `psum += f*d` to FFMA).  The CPU oracle uses the
// same chain, so forward parity is bit-exact.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/veon_hip.h"

namespace {

constexpr int kBlock = 256;
constexpr int kWave = 64;

struct PoolArgs {
  const float* __restrict__ depth;
  const float* __restrict__ feat;
  const int* __restrict__ ranks_depth;
  const int* __restrict__ ranks_feat;
  const int* __restrict__ ranks_bev;
  const int* __restrict__ interval_starts;
  const int* __restrict__ interval_lengths;
};

template <int VEC>
struct Vec;
template <>
struct Vec<1> {
  using T = float;
};
template <>
struct Vec<2> {
  using T = float2;
};
template <>
struct Vec<4> {
  using T = float4;
};

template <int VEC>
__device__ __forceinline__ void vfma(typename Vec<VEC>::T& acc,
                                     const typename Vec<VEC>::T& f, float d);
template <>
__device__ __forceinline__ void vfma<1>(float& acc, const float& f, float d) {
  acc = fmaf(f, d, acc);
}
template <>
__device__ __forceinline__ void vfma<2>(float2& acc, const float2& f, float d) {
  acc.x = fmaf(f.x, d, acc.x);
  acc.y = fmaf(f.y, d, acc.y);
}
template <>
__device__ __forceinline__ void vfma<4>(float4& acc, const float4& f, float d) {
  acc.x = fmaf(f.x, d, acc.x);
  acc.y = fmaf(f.y, d, acc.y);
  acc.z = fmaf(f.z, d, acc.z);
  acc.w = fmaf(f.w, d, acc.w);
}

template <int VEC>
__device__ __forceinline__ typename Vec<VEC>::T vzero();
template <>
__device__ __forceinline__ float vzero<1>() {
  return 0.f;
}
template <>
__device__ __forceinline__ float2 vzero<2>() {
  return make_float2(0.f, 0.f);
}
template <>
__device__ __forceinline__ float4 vzero<4>() {
  return make_float4(0.f, 0.f, 0.f, 0.f);
}

// Serial pooled sum of one interval for VEC consecutive channels starting at
// channel `ch`.  The index / depth loads are identical across the lanes that
// share an interval (hardware broadcast); the feat loads are channel-contiguous.
template <int VEC>
__device__ __forceinline__ typename Vec<VEC>::T interval_sum(
    const PoolArgs& a, int c, int start, int len, int ch) {
  using V = typename Vec<VEC>::T;
  V acc = vzero<VEC>();
  int i = 0;
  // 4 points per trip: all index loads first, then the dependent gathers, so
  // four feat rows are in flight per lane.
  for (; i + 4 <= len; i += 4) {
    int rd[4], rf[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      rd[k] = a.ranks_depth[start + i + k];
      rf[k] = a.ranks_feat[start + i + k];
    }
    float d[4];
    V f[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      d[k] = a.depth[rd[k]];
      f[k] = *reinterpret_cast<const V*>(a.feat + (int64_t)rf[k] * c + ch);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) vfma<VEC>(acc, f[k], d[k]);
  }
  for (; i < len; ++i) {
    const int rd = a.ranks_depth[start + i];
    const int rf = a.ranks_feat[start + i];
    const float d = a.depth[rd];
    const V f = *reinterpret_cast<const V*>(a.feat + (int64_t)rf * c + ch);
    vfma<VEC>(acc, f, d);
  }
  return acc;
}

// ---------------------------------------------------------------------------
// (1) Reference-semantics scatter: one lane per (interval, VEC channels).
//     `out` pre-zeroed by the caller; any interval order.
// ---------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_pool_scatter(
    PoolArgs a, int c, int cq, int n_intervals, float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t interval = idx / cq;
  if (interval >= n_intervals) return;
  const int ch = (int)(idx - interval * cq) * VEC;
  const int start = a.interval_starts[interval];
  const int len = a.interval_lengths[interval];
  using V = typename Vec<VEC>::T;
  const V acc = interval_sum<VEC>(a, c, start, len, ch);
  *reinterpret_cast<V*>(out + (int64_t)a.ranks_bev[start] * c + ch) = acc;
}

// ---------------------------------------------------------------------------
// Wave-wide 64-ary lower bound over the interval keys
//   key(i) = ranks_bev[interval_starts[i]]   (ascending, unique)
// Returns the first i in [0, n) with key(i) >= target (n if none).  All 64
// lanes of the calling wave must be active; <= 4 rounds for n < 16.7 M.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int wave_lower_bound(const PoolArgs& a, int n,
                                                int64_t target) {
  const int lane = threadIdx.x & (kWave - 1);
  int lo = 0, hi = n;
  while (hi > lo) {
    const int span = hi - lo;
    const int step = (span + kWave - 1) / kWave;
    const int64_t p = (int64_t)lo + (int64_t)lane * step;
    bool below = false;
    if (p < hi) below = (int64_t)a.ranks_bev[a.interval_starts[p]] < target;
    const int cnt = __popcll(__ballot(below));
    if (cnt == 0) {
      hi = lo;
    } else {
      const int64_t last = (int64_t)lo + (int64_t)(cnt - 1) * step;
      const int64_t nhi = last + step;
      lo = (int)(last + 1);
      hi = (int)(nhi < hi ? nhi : hi);
    }
  }
  return lo;
}

struct TileInfo {
  int b;              // batch element
  int64_t vox0;       // first voxel of the tile inside its batch element
  int nvox;           // voxels in this tile (<= V)
  int64_t rank0;      // b * voxels_per_batch + vox0
};

__device__ __forceinline__ TileInfo tile_info(int64_t t, int64_t tiles_per_batch,
                                              int64_t vpb, int V) {
  TileInfo ti;
  ti.b = (int)(t / tiles_per_batch);
  ti.vox0 = (t - (int64_t)ti.b * tiles_per_batch) * V;
  const int64_t rem = vpb - ti.vox0;
  ti.nvox = (int)(rem < V ? rem : V);
  ti.rank0 = (int64_t)ti.b * vpb + ti.vox0;
  return ti;
}

// Tile prologue shared by the fused kernels: find the first interval of the
// tile (table or in-kernel search by wave 0) and fill slot[v] = interval index
// of voxel v of the tile, or -1.  A tile holds <= V intervals (unique voxels),
// and they are contiguous from i0, so one probe per thread suffices (V <= 256).
template <int V>
__device__ __forceinline__ void tile_slots(const PoolArgs& a, int n_intervals,
                                           const int* __restrict__ tile_first,
                                           int64_t t, const TileInfo& ti,
                                           int* slot, int* s_i0) {
  const int tid = threadIdx.x;
  if (tile_first != nullptr) {
    if (tid == 0) *s_i0 = tile_first[t];
  } else if (tid < kWave) {
    const int i0 = wave_lower_bound(a, n_intervals, ti.rank0);
    if (tid == 0) *s_i0 = i0;
  }
  if (tid < V) slot[tid] = -1;
  __syncthreads();
  const int i0 = *s_i0;
  if (tid < V) {
    const int64_t i = (int64_t)i0 + tid;
    if (i < n_intervals) {
      const int64_t k = (int64_t)a.ranks_bev[a.interval_starts[i]] - ti.rank0;
      if (k < ti.nvox) slot[k] = (int)i;
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------
// (2) Fused forward, channels-last (B,Z,Y,X,C): a tile of V voxels is one
//     contiguous V*C-float region; lanes map to (voxel, VEC channels) in memory
//     order, so every store instruction is 64 x VEC*4 contiguous bytes.
// ---------------------------------------------------------------------------
template <int VEC, int V>
__global__ __launch_bounds__(kBlock) void k_pool_fused_cl(
    PoolArgs a, int c, int cq, int n_intervals, int64_t vpb,
    int64_t tiles_per_batch, const int* __restrict__ tile_first,
    float* __restrict__ out) {
  __shared__ int slot[V];
  __shared__ int s_i0;
  const int64_t t = blockIdx.x;
  const TileInfo ti = tile_info(t, tiles_per_batch, vpb, V);
  tile_slots<V>(a, n_intervals, tile_first, t, ti, slot, &s_i0);
  using VT = typename Vec<VEC>::T;
  float* obase = out + ti.rank0 * c;
  const int items = ti.nvox * cq;
  for (int item = threadIdx.x; item < items; item += kBlock) {
    const int v = item / cq;
    const int ch = (item - v * cq) * VEC;
    const int ii = slot[v];
    VT acc = vzero<VEC>();
    if (ii >= 0)
      acc = interval_sum<VEC>(a, c, a.interval_starts[ii],
                              a.interval_lengths[ii], ch);
    *reinterpret_cast<VT*>(obase + (int64_t)v * c + ch) = acc;
  }
}

// ---------------------------------------------------------------------------
// (3) Fused forward, channels-first (B,C,Z,Y,X): the layout bev_pool_v2()
//     returns.  Per tile of V = 64 voxels and slab of CS channels:
//       gather phase : lanes = (interval of the tile, channel) -> feat rows are
//                      read channel-contiguous; sums land in an LDS tile
//                      [CS][V+1] (odd row stride: conflict-free both ways);
//       store phase  : lanes = voxels; each wave stores 64 consecutive voxels
//                      of one channel (256 contiguous bytes) per instruction;
//                      empty voxels store 0 without touching LDS data.
// ---------------------------------------------------------------------------
constexpr int kVcf = 64;

__global__ __launch_bounds__(kBlock) void k_pool_fused_cf(
    PoolArgs a, int c, int cs, int n_intervals, int64_t vpb,
    int64_t tiles_per_batch, const int* __restrict__ tile_first,
    float* __restrict__ out) {
  extern __shared__ float lds[];  // [cs][kVcf+1] floats, then ints
  constexpr int V = kVcf;
  constexpr int LD = V + 1;
  float* tile = lds;
  int* slot = reinterpret_cast<int*>(lds + (size_t)cs * LD);  // [V]
  int* vloc = slot + V;                                       // [V]
  int* s_misc = vloc + V;                                     // [2]: i0, cnt
  const int tid = threadIdx.x;
  const int64_t t = blockIdx.x;
  const int c0 = blockIdx.y * cs;
  const int nch = (c - c0) < cs ? (c - c0) : cs;
  const TileInfo ti = tile_info(t, tiles_per_batch, vpb, V);

  // prologue: i0, slots, and the compact list vloc[j] = voxel of interval i0+j
  if (tile_first != nullptr) {
    if (tid == 0) s_misc[0] = tile_first[t];
  } else if (tid < kWave) {
    const int i0 = wave_lower_bound(a, n_intervals, ti.rank0);
    if (tid == 0) s_misc[0] = i0;
  }
  if (tid < V) slot[tid] = -1;
  __syncthreads();
  const int i0 = s_misc[0];
  if (tid < kWave) {
    const int64_t i = (int64_t)i0 + tid;
    bool in = false;
    int k = 0;
    if (i < n_intervals) {
      const int64_t kk = (int64_t)a.ranks_bev[a.interval_starts[i]] - ti.rank0;
      in = kk < ti.nvox;
      k = (int)kk;
    }
    if (in) {
      slot[k] = (int)i;
      vloc[tid] = k;
    }
    const int cnt = __popcll(__ballot(in));
    if (tid == 0) s_misc[1] = cnt;
  }
  __syncthreads();
  const int cnt = s_misc[1];

  // gather phase
  const int items = cnt * nch;
  for (int item = tid; item < items; item += kBlock) {
    const int j = item / nch;
    const int cc = item - j * nch;
    const int ii = i0 + j;
    const float acc = interval_sum<1>(a, c, a.interval_starts[ii],
                                      a.interval_lengths[ii], c0 + cc);
    tile[cc * LD + vloc[j]] = acc;
  }
  __syncthreads();

  // store phase: wave w handles channels w, w+4, ...
  const int v = tid & (V - 1);
  const int w = tid >> 6;
  const bool occupied = slot[v] >= 0;
  if (v < ti.nvox) {
    float* obase = out + ((int64_t)ti.b * c + c0) * vpb + ti.vox0 + v;
    for (int cc = w; cc < nch; cc += kBlock / kWave) {
      const float val = occupied ? tile[cc * LD + v] : 0.f;
      obase[(int64_t)cc * vpb] = val;
    }
  }
}

// ---------------------------------------------------------------------------
// (4) Tile table: tile_first[t] = lower_bound(key, first rank of tile t).
//     One wave per tile boundary (64-ary search).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_tile_table(
    PoolArgs a, int n_intervals, int64_t vpb, int64_t tiles_per_batch, int V,
    int64_t n_tiles, int* __restrict__ tile_first) {
  const int64_t t = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  if (t > n_tiles) return;  // wave-uniform
  int r;
  if (t == n_tiles) {
    r = n_intervals;
  } else {
    const TileInfo ti = tile_info(t, tiles_per_batch, vpb, V);
    r = wave_lower_bound(a, n_intervals, ti.rank0);
  }
  if ((threadIdx.x & (kWave - 1)) == 0) tile_first[t] = r;
}

// ---------------------------------------------------------------------------
// (5) Backward.  One 128-thread block per interval of the feat-sorted list.
//     phase A (bev_pool_cuda.cu:91-105): lanes = points of the interval, each
//       runs the serial channel chain for depth_grad;
//     phase B (:107-120): lanes = channels, serial chain over the points for
//       feat_grad (all points of an interval share one feat row).
// ---------------------------------------------------------------------------
constexpr int kBwdBlock = 128;

template <int VEC>
__global__ __launch_bounds__(kBwdBlock) void k_pool_bwd(
    PoolArgs a, int c, int n_intervals, const float* __restrict__ out_grad,
    float* __restrict__ depth_grad, float* __restrict__ feat_grad) {
  const int interval = blockIdx.x;
  const int start = a.interval_starts[interval];
  const int len = a.interval_lengths[interval];
  using VT = typename Vec<VEC>::T;
  for (int i = threadIdx.x; i < len; i += kBwdBlock) {
    const float* og = out_grad + (int64_t)a.ranks_bev[start + i] * c;
    const float* ft = a.feat + (int64_t)a.ranks_feat[start + i] * c;
    float s = 0.f;
    for (int cc = 0; cc < c; cc += VEC) {
      const VT o = *reinterpret_cast<const VT*>(og + cc);
      const VT f = *reinterpret_cast<const VT*>(ft + cc);
      if constexpr (VEC == 4) {
        s = fmaf(o.x, f.x, s);
        s = fmaf(o.y, f.y, s);
        s = fmaf(o.z, f.z, s);
        s = fmaf(o.w, f.w, s);
      } else {
        s = fmaf(o, f, s);
      }
    }
    depth_grad[a.ranks_depth[start + i]] = s;
  }
  float* fg = feat_grad + (int64_t)a.ranks_feat[start] * c;
  for (int cc = threadIdx.x; cc < c; cc += kBwdBlock) {
    float s = 0.f;
    for (int i = 0; i < len; ++i) {
      const float og = out_grad[(int64_t)a.ranks_bev[start + i] * c + cc];
      s = fmaf(og, a.depth[a.ranks_depth[start + i]], s);
    }
    fg[cc] = s;
  }
}

inline int launch_status() {
  return hipGetLastError() == hipSuccess ? VEON_OK : VEON_ERR_LAUNCH;
}

inline bool aligned16(const void* p) {
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

}  // namespace

extern "C" {

int veon_abi_version(void) { return VEON_ABI_VERSION; }

const char* veon_status_string(int status) {
  switch (status) {
    case VEON_OK:
      return "ok";
    case VEON_ERR_BAD_ARG:
      return "bad argument (null pointer, negative size or unsupported shape)";
    case VEON_ERR_LAUNCH:
      return "kernel launch failed";
    case VEON_ERR_WORKSPACE:
      return "workspace too small";
    default:
      return "unknown status";
  }
}

int veon_bev_pool_v2_fwd(int c, int n_intervals, const float* depth,
                         const float* feat, const int* ranks_depth,
                         const int* ranks_feat, const int* ranks_bev,
                         const int* interval_starts,
                         const int* interval_lengths, float* out,
                         void* stream) {
  if (c <= 0 || n_intervals < 0) return VEON_ERR_BAD_ARG;
  if (n_intervals == 0) return VEON_OK;
  if (!depth || !feat || !ranks_depth || !ranks_feat || !ranks_bev ||
      !interval_starts || !interval_lengths || !out)
    return VEON_ERR_BAD_ARG;
  PoolArgs a{depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
             interval_lengths};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool v4 = (c % 4 == 0) && aligned16(feat) && aligned16(out);
  const int cq = v4 ? c / 4 : c;
  const int64_t threads = (int64_t)n_intervals * cq;
  const int64_t blocks = (threads + kBlock - 1) / kBlock;
  if (blocks > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  if (v4)
    hipLaunchKernelGGL(k_pool_scatter<4>, dim3((unsigned)blocks), dim3(kBlock),
                       0, s, a, c, cq, n_intervals, out);
  else
    hipLaunchKernelGGL(k_pool_scatter<1>, dim3((unsigned)blocks), dim3(kBlock),
                       0, s, a, c, cq, n_intervals, out);
  return launch_status();
}

int veon_bev_pool_v2_bwd(int c, int n_intervals, const float* out_grad,
                         const float* depth, const float* feat,
                         const int* ranks_depth, const int* ranks_feat,
                         const int* ranks_bev, const int* interval_starts,
                         const int* interval_lengths, float* depth_grad,
                         float* feat_grad, void* stream) {
  if (c <= 0 || n_intervals < 0) return VEON_ERR_BAD_ARG;
  if (n_intervals == 0) return VEON_OK;
  if (!out_grad || !depth || !feat || !ranks_depth || !ranks_feat ||
      !ranks_bev || !interval_starts || !interval_lengths || !depth_grad ||
      !feat_grad)
    return VEON_ERR_BAD_ARG;
  PoolArgs a{depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
             interval_lengths};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool v4 = (c % 4 == 0) && aligned16(feat) && aligned16(out_grad);
  if (v4)
    hipLaunchKernelGGL(k_pool_bwd<4>, dim3((unsigned)n_intervals),
                       dim3(kBwdBlock), 0, s, a, c, n_intervals, out_grad,
                       depth_grad, feat_grad);
  else
    hipLaunchKernelGGL(k_pool_bwd<1>, dim3((unsigned)n_intervals),
                       dim3(kBwdBlock), 0, s, a, c, n_intervals, out_grad,
                       depth_grad, feat_grad);
  return launch_status();
}

static int cf_slab(int c) { return c <= 128 ? c : 64; }

int veon_bev_pool_tile_voxels(int c, int out_layout) {
  (void)c;
  return out_layout == VEON_LAYOUT_BCZYX ? kVcf : 64;
}

int veon_bev_pool_v2_fwd_fused(int c, int n_intervals, int batch,
                               int64_t voxels_per_batch, const float* depth,
                               const float* feat, const int* ranks_depth,
                               const int* ranks_feat, const int* ranks_bev,
                               const int* interval_starts,
                               const int* interval_lengths,
                               const int* tile_first, float* out,
                               int out_layout, void* stream) {
  if (c <= 0 || n_intervals < 0 || batch <= 0 || voxels_per_batch <= 0 || !out)
    return VEON_ERR_BAD_ARG;
  if (out_layout != VEON_LAYOUT_BZYXC && out_layout != VEON_LAYOUT_BCZYX)
    return VEON_ERR_BAD_ARG;
  if (n_intervals > 0 &&
      (!depth || !feat || !ranks_depth || !ranks_feat || !ranks_bev ||
       !interval_starts || !interval_lengths))
    return VEON_ERR_BAD_ARG;
  if ((int64_t)batch * voxels_per_batch > 0x7fffffffLL)
    return VEON_ERR_BAD_ARG;  // ranks_bev is int32 (the reference ABI)
  PoolArgs a{depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
             interval_lengths};
  hipStream_t s = static_cast<hipStream_t>(stream);
  constexpr int V = 64;
  const int64_t tiles_per_batch = (voxels_per_batch + V - 1) / V;
  const int64_t n_tiles = tiles_per_batch * batch;
  if (n_tiles > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  if (out_layout == VEON_LAYOUT_BZYXC) {
    const bool v4 = (c % 4 == 0) && aligned16(feat) && aligned16(out);
    if (v4)
      hipLaunchKernelGGL((k_pool_fused_cl<4, V>), dim3((unsigned)n_tiles),
                         dim3(kBlock), 0, s, a, c, c / 4, n_intervals,
                         voxels_per_batch, tiles_per_batch, tile_first, out);
    else
      hipLaunchKernelGGL((k_pool_fused_cl<1, V>), dim3((unsigned)n_tiles),
                         dim3(kBlock), 0, s, a, c, c, n_intervals,
                         voxels_per_batch, tiles_per_batch, tile_first, out);
  } else {
    const int cs = cf_slab(c);
    const int slabs = (c + cs - 1) / cs;
    const size_t lds = (size_t)cs * (kVcf + 1) * sizeof(float) +
                       (2 * kVcf + 2) * sizeof(int);
    hipLaunchKernelGGL(k_pool_fused_cf, dim3((unsigned)n_tiles, (unsigned)slabs),
                       dim3(kBlock), lds, s, a, c, cs, n_intervals,
                       voxels_per_batch, tiles_per_batch, tile_first, out);
  }
  return launch_status();
}

int veon_bev_pool_tile_table(int n_intervals, int batch,
                             int64_t voxels_per_batch, int tile_voxels,
                             const int* ranks_bev, const int* interval_starts,
                             int* tile_first, void* stream) {
  if (n_intervals < 0 || batch <= 0 || voxels_per_batch <= 0 ||
      tile_voxels <= 0 || !tile_first)
    return VEON_ERR_BAD_ARG;
  if (n_intervals > 0 && (!ranks_bev || !interval_starts))
    return VEON_ERR_BAD_ARG;
  PoolArgs a{nullptr, nullptr, nullptr, nullptr, ranks_bev, interval_starts,
             nullptr};
  const int64_t tiles_per_batch =
      (voxels_per_batch + tile_voxels - 1) / tile_voxels;
  const int64_t n_tiles = tiles_per_batch * batch;
  const int waves_per_block = kBlock / kWave;
  const int64_t blocks = (n_tiles + 1 + waves_per_block - 1) / waves_per_block;
  if (blocks > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_tile_table, dim3((unsigned)blocks), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), a, n_intervals,
                     voxels_per_batch, tiles_per_batch, tile_voxels, n_tiles,
                     tile_first);
  return launch_status();
}

}  // extern "C"
