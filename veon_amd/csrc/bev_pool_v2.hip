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
#include "half_mode.h"

extern "C" int veon_pool_debug_flags;  // bev_pool_rows.hip

namespace {

constexpr int kBlock = 256;
constexpr int kWave = 64;

struct PoolArgs {
  const float* __restrict__ depth;
  const void* __restrict__ feat;  // float / fp16 / bf16 rows, see FeatF32...
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

struct float8 {
  float4 lo, hi;
};
template <>
struct Vec<8> {
  using T = float8;
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

template <>
__device__ __forceinline__ void vfma<8>(float8& acc, const float8& f, float d) {
  vfma<4>(acc.lo, f.lo, d);
  vfma<4>(acc.hi, f.hi, d);
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

template <>
__device__ __forceinline__ float8 vzero<8>() {
  return float8{make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
}

// Feature storage types.  The reference casts feat to fp32 before the kernel
// (bev_pool.py:21); reading half-precision rows and widening in registers is
// the same arithmetic (the widening is exact) at half the gather bytes.
struct FeatF32 {
  using S = float;
};
struct FeatF16 {
  using S = _Float16;
};
struct FeatBF16 {
  using S = unsigned short;
};
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) {
  return __uint_as_float(u & 0xffff0000u);
}

// VEC consecutive channels of one feat row starting at element `e`
template <typename FT, int VEC>
__device__ __forceinline__ typename Vec<VEC>::T load_feat(const void* feat,
                                                          int64_t e) {
  using V = typename Vec<VEC>::T;
  if constexpr (__is_same(FT, FeatF32)) {
    static_assert(VEC <= 4, "fp32 rows are read 16 bytes at a time");
    return *reinterpret_cast<const V*>(static_cast<const float*>(feat) + e);
  } else if constexpr (__is_same(FT, FeatF16)) {
    const _Float16* p = static_cast<const _Float16*>(feat) + e;
    if constexpr (VEC == 1) {
      return (float)p[0];
    } else if constexpr (VEC == 4) {
      const half4_t h = *reinterpret_cast<const half4_t*>(p);
      return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    } else {
      const half8_t h = *reinterpret_cast<const half8_t*>(p);
      return float8{make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]),
                    make_float4((float)h[4], (float)h[5], (float)h[6], (float)h[7])};
    }
  } else {
    const unsigned short* p = static_cast<const unsigned short*>(feat) + e;
    if constexpr (VEC == 1) {
      return __uint_as_float((unsigned)p[0] << 16);
    } else if constexpr (VEC == 4) {
      const uint2 u = *reinterpret_cast<const uint2*>(p);
      return make_float4(bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y));
    } else {
      const uint4 u = *reinterpret_cast<const uint4*>(p);
      return float8{make_float4(bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y)),
                    make_float4(bf_lo(u.z), bf_hi(u.z), bf_lo(u.w), bf_hi(u.w))};
    }
  }
}

// Serial pooled sum of one interval for VEC consecutive channels starting at
// channel `ch`.  The index / depth loads are identical across the lanes that
// share an interval (hardware broadcast); the feat loads are channel-contiguous.
template <int VEC, typename FT = FeatF32>
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
      f[k] = load_feat<FT, VEC>(a.feat, (int64_t)rf[k] * c + ch);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) vfma<VEC>(acc, f[k], d[k]);
  }
  for (; i < len; ++i) {
    const int rd = a.ranks_depth[start + i];
    const int rf = a.ranks_feat[start + i];
    const float d = a.depth[rd];
    const V f = load_feat<FT, VEC>(a.feat, (int64_t)rf * c + ch);
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
// Pool plan.  The fused kernels walk the output in tiles of kTileV = 64
// consecutive voxel ranks of one batch element.  plan[t] = {i0, cnt, p0, npts}:
// first interval / number of intervals / first point / number of points of
// tile t (intervals are ascending in voxel rank, so a tile's intervals and
// points are contiguous).  Built by two tiny kernels; cache it with the ranks
// (accelerate=True) or let the prepare kernel emit it.
// ---------------------------------------------------------------------------
constexpr int kTileV = 64;

// one thread per interval (+1 sentinel): tiles (tile(i-1), tile(i)] start at i
__global__ __launch_bounds__(kBlock) void k_plan_bounds(
    const int* __restrict__ ranks_bev, const int* __restrict__ interval_starts,
    int n_intervals, int n_points, int64_t vpb, int64_t tiles_per_batch,
    int64_t n_tiles, int tile_voxels, int* __restrict__ tile_first,
    int* __restrict__ tile_point, const int* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (counts != nullptr) {  // sizes known only on the device (sync-free path)
    n_points = counts[0];
    n_intervals = counts[1];
  }
  if (i > n_intervals) return;
  auto tile_of = [&](int rank) -> int64_t {
    const int64_t b = rank / vpb;
    return b * tiles_per_batch + (rank - b * vpb) / tile_voxels;
  };
  int64_t hi = n_tiles;
  int st = n_points;
  if (i < n_intervals) {
    st = interval_starts[i];
    hi = tile_of(ranks_bev[st]);
  }
  const int64_t lo =
      (i == 0) ? -1 : tile_of(ranks_bev[interval_starts[i - 1]]);
  for (int64_t tt = lo + 1; tt <= hi; ++tt) {
    tile_first[tt] = (int)i;
    tile_point[tt] = st;
  }
}

__global__ __launch_bounds__(kBlock) void k_plan_pack(
    const int* __restrict__ tile_first, const int* __restrict__ tile_point,
    int64_t n_tiles, int4* __restrict__ plan) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= n_tiles) return;
  plan[t] = make_int4(tile_first[t], tile_first[t + 1] - tile_first[t],
                      tile_point[t], tile_point[t + 1] - tile_point[t]);
}

struct TileInfo {
  int b;          // batch element
  int64_t vox0;   // first voxel of the tile inside its batch element
  int nvox;       // voxels in this tile (<= kTileV)
  int64_t rank0;  // b * voxels_per_batch + vox0
};

__device__ __forceinline__ TileInfo tile_info(int64_t t, int64_t tiles_per_batch,
                                              int64_t vpb) {
  TileInfo ti;
  ti.b = (int)(t / tiles_per_batch);
  ti.vox0 = (t - (int64_t)ti.b * tiles_per_batch) * kTileV;
  const int64_t rem = vpb - ti.vox0;
  ti.nvox = (int)(rem < kTileV ? rem : kTileV);
  ti.rank0 = (int64_t)ti.b * vpb + ti.vox0;
  return ti;
}

// Occupancy mask of a tile: bit v set iff voxel v of the tile has an interval.
// Column of an occupied voxel = number of occupied voxels below it = index of
// its interval inside the tile (intervals are ascending in voxel rank).
__device__ __forceinline__ unsigned long long wave_or(unsigned long long bit) {
  for (int off = 32; off > 0; off >>= 1) bit |= __shfl_xor(bit, off);
  return bit;
}

// ---------------------------------------------------------------------------
// (2) Fused forward, channels-last (B,Z,Y,X,C): a tile is one contiguous
//     64*C-float region; lanes map to (voxel, VEC channels) in memory order, so
//     every store instruction is 64 x VEC*4 contiguous bytes.  No LDS data.
// ---------------------------------------------------------------------------
template <int VEC, typename FT>
__global__ __launch_bounds__(kBlock) void k_pool_fused_cl(
    PoolArgs a, const int4* __restrict__ plan, int c, int cq, int64_t vpb,
    int64_t tiles_per_batch, float* __restrict__ out) {
  const int64_t t = blockIdx.x;
  const TileInfo ti = tile_info(t, tiles_per_batch, vpb);
  const int4 pl = plan[t];
  const int i0 = pl.x, cnt = pl.y;
  using VT = typename Vec<VEC>::T;
  float* obase = out + ti.rank0 * c;
  const int items = ti.nvox * cq;
  if (cnt == 0) {
    for (int item = threadIdx.x; item < items; item += kBlock)
      reinterpret_cast<VT*>(obase)[item] = vzero<VEC>();
    return;
  }
  const int lane = threadIdx.x & (kWave - 1);
  unsigned long long bit = 0;
  if (lane < cnt)
    bit = 1ull << (int)((int64_t)a.ranks_bev[a.interval_starts[i0 + lane]] -
                        ti.rank0);
  const unsigned long long mask = wave_or(bit);
  for (int item = threadIdx.x; item < items; item += kBlock) {
    const int v = item / cq;
    const int ch = (item - v * cq) * VEC;
    VT acc = vzero<VEC>();
    if ((mask >> v) & 1ull) {
      const int ii = i0 + __popcll(mask & ((1ull << v) - 1ull));
      acc = interval_sum<VEC, FT>(a, c, a.interval_starts[ii],
                                  a.interval_lengths[ii], ch);
    }
    reinterpret_cast<VT*>(obase)[item] = acc;
  }
}

// ---------------------------------------------------------------------------
// (3) Fused forward, channels-first (B,C,Z,Y,X): the layout bev_pool_v2()
//     returns.  Per tile of 64 voxels and slab of `cs` channels, three load
//     levels and then one wave-wide row store per channel:
//       plan entry -> {interval starts, voxel of each interval, ranks_feat /
//       ranks_depth of the tile's points (staged in LDS)} -> {feat rows, depth}
//     gather: lanes = (interval, VEC channels): feat rows are read
//       channel-contiguous, UNROLL loads in flight, serial fmaf chain;
//     transpose: sums land in a compact LDS tile [cs][CAP+1], column = index of
//       the interval inside the tile (odd row stride: conflict-free both ways);
//     store: lanes = voxels; a wave stores 64 consecutive voxels of one
//       channel (256 contiguous bytes) per instruction, non-temporal so the
//       streamed volume does not evict the gathered rows from L2; empty voxels
//       store 0.  Tiles with more than CAP intervals run as two 32-voxel halves.
// ---------------------------------------------------------------------------
constexpr int kPmax = 1024;  // points staged per window (8 B each)

template <int VEC, int CAP, typename FT>
__global__ __launch_bounds__(kBlock) void k_pool_fused_cf(
    PoolArgs a, const int4* __restrict__ plan, int c, int cs, int64_t vpb,
    int64_t tiles_per_batch, float* __restrict__ out, int64_t ostride, int chunked) {
  // ostride: distance between channel planes of `out` in floats (>= vpb; vpb for
  // the contiguous (B,C,Z,Y,X) tensor the reference returns)
  extern __shared__ float lds[];
  constexpr int LDC = CAP + 1;
  constexpr int UNROLL = 8;  // rows in flight per lane
  constexpr int SB = 5;  // LDS reads batched ahead of the stores
  constexpr int NW = kBlock / kWave;
  float* tile = lds;                                             // [cs][LDC]
  int* istart = reinterpret_cast<int*>(lds + (size_t)cs * LDC);  // [kTileV+2]
  int* ivox = istart + kTileV + 2;                               // [kTileV]
  int* s_rf = ivox + kTileV;                                     // [kPmax]
  int* s_rd = s_rf + kPmax;                                      // [kPmax]

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int w = tid >> 6;
  // XCD k (workgroups k, k + 8, ...) takes the k-th contiguous eighth of the tiles:
  // plan entries, interval starts and ranks of neighbouring tiles share cache
  // lines, which then live in ONE L2 instead of being fetched by all eight
  int64_t t = blockIdx.x;
  if (chunked >= 0) {  // runs of 2^chunked tiles per XCD (bev_pool_rows.hip: xcd_grouped)
    const int64_t sg = t >> (3 + chunked);
    if (((sg + 1) << (3 + chunked)) <= (int64_t)gridDim.x)
      t = (((sg << 3) + (t & 7)) << chunked) + ((t >> 3) & ((1 << chunked) - 1));
  }
  const int c0 = blockIdx.y * cs;
  const int nch = (c - c0) < cs ? (c - c0) : cs;
  const TileInfo ti = tile_info(t, tiles_per_batch, vpb);
  float* obase = out + ((int64_t)ti.b * c + c0) * ostride + ti.vox0;

  const int4 pl = plan[t];
  const int i0 = pl.x, cnt = pl.y, p0 = pl.z, npts = pl.w;

  if (cnt == 0) {  // empty tile: pure streaming zero fill
    if (lane < ti.nvox)
      for (int cc = w; cc < nch; cc += NW)
        __builtin_nontemporal_store(0.f, obase + (int64_t)cc * ostride + lane);
    return;
  }
  // level 2: everything that depends only on the plan entry
  if (tid < cnt) {
    const int st = a.interval_starts[i0 + tid];
    istart[tid] = st - p0;
    ivox[tid] = (int)((int64_t)a.ranks_bev[st] - ti.rank0);
  }
  if (tid == 0) istart[cnt] = npts;
  {
    const int n = npts < kPmax ? npts : kPmax;
    for (int p = tid; p < n; p += kBlock) {
      s_rf[p] = a.ranks_feat[p0 + p];
      s_rd[p] = a.ranks_depth[p0 + p];
    }
  }
  __syncthreads();

  const unsigned long long mask =
      wave_or(lane < cnt ? (1ull << ivox[lane]) : 0ull);
  using VT = typename Vec<VEC>::T;
  const int nq = nch / VEC;
  const int npass = cnt > CAP ? 2 : 1;
  const int jsplit = __popcll(mask & 0xffffffffull);  // intervals in voxels 0..31
  int base = 0;  // first staged point (relative to p0)
  for (int pass = 0; pass < npass; ++pass) {
    const int ja = (npass == 2 && pass == 1) ? jsplit : 0;
    const int jb = (npass == 2 && pass == 0) ? jsplit : cnt;
    int j0 = ja;
    while (j0 < jb) {
      // group [j0, j1): consecutive intervals whose points are staged
      int j1 = j0;
      while (j1 < jb && istart[j1 + 1] - base <= kPmax) ++j1;
      if (j1 == j0) {
        const int st = istart[j0];
        const int len = istart[j0 + 1] - st;
        if (len > kPmax) {
          // one interval longer than the window: straight from global memory
          for (int q = tid; q < nq; q += kBlock) {
            const VT acc =
                interval_sum<VEC, FT>(a, c, p0 + st, len, c0 + q * VEC);
            const float* ap = reinterpret_cast<const float*>(&acc);
#pragma unroll
            for (int k = 0; k < VEC; ++k)
              tile[(q * VEC + k) * LDC + (j0 - ja)] = ap[k];
          }
          ++j0;
          continue;
        }
        // restage the window from interval j0
        __syncthreads();
        base = st;
        const int n = (npts - base) < kPmax ? (npts - base) : kPmax;
        for (int p = tid; p < n; p += kBlock) {
          s_rf[p] = a.ranks_feat[p0 + base + p];
          s_rd[p] = a.ranks_depth[p0 + base + p];
        }
        __syncthreads();
        continue;
      }
      const int items = (j1 - j0) * nq;
      for (int item = tid; item < items; item += kBlock) {
        const int j = j0 + item / nq;
        const int q = item - (j - j0) * nq;
        const int st = istart[j] - base;
        const int len = istart[j + 1] - istart[j];
        const int64_t fcol = c0 + q * VEC;
        VT acc = vzero<VEC>();
        int i = 0;
        for (; i + UNROLL <= len; i += UNROLL) {
          VT f[UNROLL];
          float d[UNROLL];
#pragma unroll
          for (int u = 0; u < UNROLL; ++u) {
            f[u] = load_feat<FT, VEC>(a.feat,
                                      fcol + (int64_t)s_rf[st + i + u] * c);
            d[u] = a.depth[s_rd[st + i + u]];
          }
#pragma unroll
          for (int u = 0; u < UNROLL; ++u) vfma<VEC>(acc, f[u], d[u]);
        }
        for (; i < len; ++i) {
          const VT f =
              load_feat<FT, VEC>(a.feat, fcol + (int64_t)s_rf[st + i] * c);
          vfma<VEC>(acc, f, a.depth[s_rd[st + i]]);
        }
        const float* ap = reinterpret_cast<const float*>(&acc);
#pragma unroll
        for (int k = 0; k < VEC; ++k)
          tile[(q * VEC + k) * LDC + (j - ja)] = ap[k];
      }
      j0 = j1;
    }
    __syncthreads();
    if (npass == 1) {
      const bool occupied = (mask >> lane) & 1ull;
      const int col = __popcll(mask & ((1ull << lane) - 1ull));
      if (lane < ti.nvox) {
        float* op = obase + lane;
        int cc = w;
        for (; cc + (SB - 1) * NW < nch; cc += SB * NW) {
          float vals[SB];
#pragma unroll
          for (int u = 0; u < SB; ++u)
            vals[u] = occupied ? tile[(cc + NW * u) * LDC + col] : 0.f;
#pragma unroll
          for (int u = 0; u < SB; ++u)
            __builtin_nontemporal_store(vals[u],
                                        op + (int64_t)(cc + NW * u) * ostride);
        }
        for (; cc < nch; cc += NW)
          __builtin_nontemporal_store(occupied ? tile[cc * LDC + col] : 0.f,
                                      op + (int64_t)cc * ostride);
      }
    } else {
      // half tile (32 voxels): one wave instruction = 2 channels x 128 B
      const int v = pass * 32 + (lane & 31);
      const bool occupied = (mask >> v) & 1ull;
      const int col = __popcll(mask & ((1ull << v) - 1ull)) - ja;
      if (v < ti.nvox)
        for (int cc = 2 * w + (lane >> 5); cc < nch; cc += 2 * NW)
          __builtin_nontemporal_store(occupied ? tile[cc * LDC + col] : 0.f,
                                      obase + (int64_t)cc * ostride + v);
      if (pass == 0) __syncthreads();  // the tile is reused by the second half
    }
  }
}

// ---------------------------------------------------------------------------
// (4) Fused forward + (dz,dy,dx) max-pool, channels-first: what
//     LSSViewTransformerRaw.forward computes (view_transformer_raw.py:537-555:
//     bev_pool_v2 then rearrange + max over 2x2x2 blocks) without ever writing
//     the full-resolution volume.  One workgroup = one pooled row (b, zo, yo)
//     x one channel slab: the dz*dy input rows' intervals are summed exactly as
//     above and folded into an LDS tile [cs][Xo] by integer atomic max on an
//     order-preserving key (max is order independent, so the result is
//     deterministic and bit-equal to max-pooling the full volume).  A pooled
//     voxel with fewer than dz*dy*dx occupied inputs also competes against the
//     zeros of its empty inputs.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int float_key(float f) {
  const int b = __float_as_int(f);
  return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float key_float(int k) {
  return __int_as_float(k ^ ((k >> 31) & 0x7fffffff));
}

// PADDED: the result goes, rounded to bf16, into the interior of the zero-padded
// channels-last grid [B][Zo+2][Yo+2][Xo+2][C] that the Conv3d body consumes
// (csrc/conv3d.hip) instead of the (B,C,Zo,Yo,Xo) fp32 tensor.
template <int VEC, typename FT, bool PADDED = false>
__global__ __launch_bounds__(kBlock) void k_pool_maxpool_cf(
    PoolArgs a, const int* __restrict__ row_first, int c, int cs, int Z, int Y,
    int X, int dz, int dy, int dx, float* __restrict__ out) {
  extern __shared__ int ldsi[];
  const int Zo = Z / dz, Yo = Y / dy, Xo = X / dx;
  int* tile = ldsi;            // [cs][Xo] keys
  int* occ = ldsi + cs * Xo;   // [Xo] occupied input voxels per pooled voxel
  const int tid = threadIdx.x;
  const int64_t orow = blockIdx.x;  // (b, zo, yo)
  const int yo = (int)(orow % Yo);
  const int zo = (int)((orow / Yo) % Zo);
  const int b = (int)(orow / ((int64_t)Yo * Zo));
  const int c0 = blockIdx.y * cs;
  const int nch = (c - c0) < cs ? (c - c0) : cs;
  const int nq = nch / VEC;
  const int kmin = float_key(-__builtin_inff());
  for (int i = tid; i < cs * Xo; i += kBlock) tile[i] = kmin;
  for (int i = tid; i < Xo; i += kBlock) occ[i] = 0;
  __syncthreads();
  using VT = typename Vec<VEC>::T;
  for (int rz = 0; rz < dz; ++rz)
    for (int ry = 0; ry < dy; ++ry) {
      const int64_t row = ((int64_t)b * Z + (zo * dz + rz)) * Y + (yo * dy + ry);
      const int64_t rank0 = row * X;
      const int i0 = row_first[row];
      const int cnt = row_first[row + 1] - i0;
      const int items = cnt * nq;
      for (int item = tid; item < items; item += kBlock) {
        const int j = item / nq;
        const int q = item - j * nq;
        const int start = a.interval_starts[i0 + j];
        const int len = a.interval_lengths[i0 + j];
        const int xo = (int)((int64_t)a.ranks_bev[start] - rank0) / dx;
        const VT acc = interval_sum<VEC, FT>(a, c, start, len, c0 + q * VEC);
        const float* ap = reinterpret_cast<const float*>(&acc);
#pragma unroll
        for (int k = 0; k < VEC; ++k)
          atomicMax(&tile[(q * VEC + k) * Xo + xo], float_key(ap[k]));
        if (q == 0) atomicAdd(&occ[xo], 1);
      }
    }
  __syncthreads();
  const int full = dz * dy * dx;
  if constexpr (PADDED) {
    unsigned short* ob =
        reinterpret_cast<unsigned short*>(out) +
        ((((int64_t)b * (Zo + 2) + zo + 1) * (Yo + 2) + yo + 1) * (Xo + 2) + 1) *
            (int64_t)c + c0;
    for (int i = tid; i < nch * Xo; i += kBlock) {
      const int xo = i / nch;       // lanes run over channels: 2-byte stores,
      const int cc = i - xo * nch;  // channel-contiguous per voxel
      const int n = occ[xo];
      float v = 0.f;
      if (n > 0) {
        v = key_float(tile[cc * Xo + xo]);
        if (n < full && !(v > 0.f)) v = 0.f;
      }
      ob[(int64_t)xo * c + cc] = __builtin_bit_cast(unsigned short, (veon_half_native)v);
    }
    return;
  }
  float* obase =
      out + ((((int64_t)b * c + c0) * Zo + zo) * Yo + yo) * (int64_t)Xo;
  const int64_t cstride = (int64_t)Zo * Yo * Xo;
  for (int i = tid; i < nch * Xo; i += kBlock) {
    const int cc = i / Xo;
    const int xo = i - cc * Xo;
    const int n = occ[xo];
    float v = 0.f;
    if (n > 0) {
      v = key_float(tile[cc * Xo + xo]);
      if (n < full && !(v > 0.f)) v = 0.f;
    }
    obase[(int64_t)cc * cstride + xo] = v;
  }
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
    const float* ft =
        static_cast<const float*>(a.feat) + (int64_t)a.ranks_feat[start + i] * c;
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

// (B*N, C, H*W) -> (B*N, H*W, C): the `feat.permute(0,1,3,4,2).contiguous()` in
// front of the reference op (view_transformer.py:273-275 + bev_pool.py:21) as one
// LDS-tiled transpose (64 pixels x 64 channels per workgroup; coalesced 256-byte
// rows on both sides).  PyTorch's strided copy of this 1.35 MB tensor takes 5.5 us,
// a sixth of the pool kernel it feeds.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void k_feat_nchw_to_nhwc(const T* __restrict__ in,
                                                           T* __restrict__ out, int C,
                                                           int HW) {
  __shared__ T t[64][65];
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const T* ib = in + (int64_t)blockIdx.z * C * HW;
  T* ob = out + (int64_t)blockIdx.z * HW * C;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  // all sixteen loads of a thread in flight before the first LDS store: the kernel has
  // 132 workgroups at S2 and is pure latency (8.6 us with a load -> store loop, a fifth
  // of the pool kernel it feeds)
  T v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int c = c0 + ty + 4 * k, p = p0 + tx;
    v[k] = (c < C && p < HW) ? ib[(int64_t)c * HW + p] : T(0);
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) t[ty + 4 * k][tx] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = t[tx][ty + 4 * k];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int p = p0 + ty + 4 * k, c = c0 + tx;
    if (p < HW && c < C) ob[(int64_t)p * C + c] = v[k];
  }
}
// ---------------------------------------------------------------------------
// (dz,dy,dx) = 2x2x2 block max of a (B,C,Z,Y,X) fp32 volume -> (B,C,Z/2,Y/2,X/2):
// the ds_feat step of LSSViewTransformerRaw.forward (view_transformer_raw.py:549-553)
// for volumes that could not go through the fused pool + max-pool kernel -- the
// camera-sharded path, where the block max must follow the cross-rank sum (torch's
// view + amax takes 1.2 ms on the 655 MB VEON volume; this is one streaming pass).
// A lane owns one output voxel: four 8-byte loads (x pairs of two y rows of two z
// planes), consecutive lanes consecutive x.  NaN propagates as in torch.amax.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float max_nan(float m, float v) {
  return (v > m || v != v) ? v : m;
}
__global__ __launch_bounds__(256) void k_volume_maxpool2(const float* __restrict__ in,
                                                         float* __restrict__ out,
                                                         int64_t planes, int Z, int Y, int X) {
  const int Zo = Z / 2, Yo = Y / 2, Xo = X / 2;
  const int64_t total = planes * Zo * Yo * Xo;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int xo = (int)(i % Xo);
  const int64_t r1 = i / Xo;
  const int yo = (int)(r1 % Yo);
  const int64_t r2 = r1 / Yo;
  const int zo = (int)(r2 % Zo);
  const int64_t p = r2 / Zo;
  const float* base = in + ((p * Z + 2 * zo) * Y + 2 * yo) * (int64_t)X + 2 * xo;
  float m = -__builtin_inff();
#pragma unroll
  for (int dz = 0; dz < 2; ++dz)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const float* q = base + ((int64_t)dz * Y + dy) * X;
      // X even: 2*xo keeps the 8-byte alignment of the row when the row start has it
      const float a = __builtin_nontemporal_load(q), b = __builtin_nontemporal_load(q + 1);
      m = max_nan(max_nan(m, a), b);
    }
  out[i] = m;
}

}  // namespace

extern "C" {

int veon_abi_version(void) { return VEON_ABI_VERSION; }
int veon_half_mode(void) { return VEON_HALF_MODE; }

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

int veon_bev_pool_tile_voxels(void) { return kTileV; }

int64_t veon_bev_pool_plan_ints(int batch, int64_t voxels_per_batch) {
  if (batch <= 0 || voxels_per_batch <= 0) return 0;
  const int64_t n_tiles =
      ((voxels_per_batch + kTileV - 1) / kTileV) * (int64_t)batch;
  return 4 * n_tiles + 2 * (n_tiles + 1);  // plan entries + build scratch
}

int veon_bev_pool_plan(int n_intervals, int n_points, int batch,
                       int64_t voxels_per_batch, const int* ranks_bev,
                       const int* interval_starts, const int* counts, int* plan,
                       void* stream) {
  if (n_intervals < 0 || n_points < 0 || batch <= 0 || voxels_per_batch <= 0 ||
      !plan)
    return VEON_ERR_BAD_ARG;
  if (n_intervals > 0 && (!ranks_bev || !interval_starts))
    return VEON_ERR_BAD_ARG;
  if ((reinterpret_cast<uintptr_t>(plan) & 15u) != 0) return VEON_ERR_BAD_ARG;
  const int64_t tiles_per_batch = (voxels_per_batch + kTileV - 1) / kTileV;
  const int64_t n_tiles = tiles_per_batch * batch;
  if (n_tiles > 0x3fffffffLL) return VEON_ERR_BAD_ARG;
  int* tile_first = plan + 4 * n_tiles;
  int* tile_point = tile_first + (n_tiles + 1);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned b1 = (unsigned)(((int64_t)n_intervals + 1 + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_plan_bounds, dim3(b1), dim3(kBlock), 0, s, ranks_bev,
                     interval_starts, n_intervals, n_points, voxels_per_batch,
                     tiles_per_batch, n_tiles, kTileV, tile_first, tile_point, counts);
  const unsigned b2 = (unsigned)((n_tiles + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_plan_pack, dim3(b2), dim3(kBlock), 0, s, tile_first,
                     tile_point, n_tiles, reinterpret_cast<int4*>(plan));
  return launch_status();
}

int veon_bev_pool_row_table(int n_intervals, int n_points, int batch,
                            int64_t voxels_per_batch, int row_voxels,
                            const int* ranks_bev, const int* interval_starts,
                            const int* counts, int* row_first, int* row_point,
                            void* stream) {
  if (n_intervals < 0 || n_points < 0 || batch <= 0 || voxels_per_batch <= 0 ||
      row_voxels <= 0 || voxels_per_batch % row_voxels != 0 || !row_first ||
      !row_point)
    return VEON_ERR_BAD_ARG;
  if (n_intervals > 0 && (!ranks_bev || !interval_starts))
    return VEON_ERR_BAD_ARG;
  const int64_t rows_per_batch = voxels_per_batch / row_voxels;
  const int64_t n_rows = rows_per_batch * batch;
  if (n_rows > 0x3fffffffLL) return VEON_ERR_BAD_ARG;
  const unsigned b1 = (unsigned)(((int64_t)n_intervals + 1 + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_plan_bounds, dim3(b1), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), ranks_bev,
                     interval_starts, n_intervals, n_points, voxels_per_batch,
                     rows_per_batch, n_rows, row_voxels, row_first, row_point, counts);
  return launch_status();
}

static int maxpool_impl(bool padded, int c, int n_intervals, int batch, int Z,
                                    int Y, int X, int dz, int dy, int dx,
                                    const float* depth, const void* feat,
                                    int feat_dtype, const int* ranks_depth,
                                    const int* ranks_feat, const int* ranks_bev,
                                    const int* interval_starts,
                                    const int* interval_lengths,
                                    const int* row_first, float* out,
                                    void* stream) {
  if (c <= 0 || n_intervals < 0 || batch <= 0 || Z <= 0 || Y <= 0 || X <= 0 ||
      dz <= 0 || dy <= 0 || dx <= 0 || Z % dz || Y % dy || X % dx || !out ||
      !row_first)
    return VEON_ERR_BAD_ARG;
  if (feat_dtype != VEON_FEAT_F32 && feat_dtype != VEON_FEAT_F16 &&
      feat_dtype != VEON_FEAT_BF16)
    return VEON_ERR_BAD_ARG;
  if (n_intervals > 0 &&
      (!depth || !feat || !ranks_depth || !ranks_feat || !ranks_bev ||
       !interval_starts || !interval_lengths))
    return VEON_ERR_BAD_ARG;
  if ((int64_t)batch * Z * Y * X > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  PoolArgs a{depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
             interval_lengths};
  const int Xo = X / dx;
  // channel slab so that the [cs][Xo] key tile stays <= 16 KB (occupancy first)
  int cs = c;
  while ((int64_t)cs * Xo * 4 > 16384 && cs > 8) cs = (cs / 2 + 7) / 8 * 8;
  const int slabs = (c + cs - 1) / cs;
  const size_t lds = ((size_t)cs * Xo + Xo) * sizeof(int);
  const int64_t orows = (int64_t)batch * (Z / dz) * (Y / dy);
  if (orows > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  const dim3 grid((unsigned)orows, (unsigned)slabs);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define VEON_LAUNCH_MP(VEC, FT)                                               \
  do {                                                                        \
    if (padded)                                                               \
      hipLaunchKernelGGL((k_pool_maxpool_cf<VEC, FT, true>), grid,            \
                         dim3(kBlock), lds, s, a, row_first, c, cs, Z, Y, X,  \
                         dz, dy, dx, out);                                    \
    else                                                                      \
      hipLaunchKernelGGL((k_pool_maxpool_cf<VEC, FT, false>), grid,           \
                         dim3(kBlock), lds, s, a, row_first, c, cs, Z, Y, X,  \
                         dz, dy, dx, out);                                    \
  } while (0)
  if (feat_dtype == VEON_FEAT_F32) {
    if ((c % 4 == 0) && (cs % 4 == 0) && aligned16(feat))
      VEON_LAUNCH_MP(4, FeatF32);
    else
      VEON_LAUNCH_MP(1, FeatF32);
  } else {
    // 16-byte gathers (8 channels per lane) pay at VEON's C = 256; at C = 80
    // they leave too few lanes busy and 8-byte gathers are faster (measured)
    const bool v4 = (c % 4 == 0) && (cs % 4 == 0) && aligned16(feat);
    const bool v8 = v4 && (c % 8 == 0) && (cs % 8 == 0) && c >= 128;
    if (feat_dtype == VEON_FEAT_F16) {
      if (v8) VEON_LAUNCH_MP(8, FeatF16);
      else if (v4) VEON_LAUNCH_MP(4, FeatF16);
      else VEON_LAUNCH_MP(1, FeatF16);
    } else {
      if (v8) VEON_LAUNCH_MP(8, FeatBF16);
      else if (v4) VEON_LAUNCH_MP(4, FeatBF16);
      else VEON_LAUNCH_MP(1, FeatBF16);
    }
  }
#undef VEON_LAUNCH_MP
  return launch_status();
}

int veon_bev_pool_v2_fwd_maxpool_ex(int c, int n_intervals, int batch, int Z,
                                    int Y, int X, int dz, int dy, int dx,
                                    const float* depth, const void* feat,
                                    int feat_dtype, const int* ranks_depth,
                                    const int* ranks_feat, const int* ranks_bev,
                                    const int* interval_starts,
                                    const int* interval_lengths,
                                    const int* row_first, float* out,
                                    void* stream) {
  return maxpool_impl(false, c, n_intervals, batch, Z, Y, X, dz, dy, dx, depth,
                      feat, feat_dtype, ranks_depth, ranks_feat, ranks_bev,
                      interval_starts, interval_lengths, row_first, out, stream);
}

int veon_bev_pool_v2_fwd_maxpool_padded(int c, int n_intervals, int batch, int Z,
                                        int Y, int X, int dz, int dy, int dx,
                                        const float* depth, const void* feat,
                                        int feat_dtype, const int* ranks_depth,
                                        const int* ranks_feat,
                                        const int* ranks_bev,
                                        const int* interval_starts,
                                        const int* interval_lengths,
                                        const int* row_first,
                                        void* out_padded_bf16, void* stream) {
  return maxpool_impl(true, c, n_intervals, batch, Z, Y, X, dz, dy, dx, depth,
                      feat, feat_dtype, ranks_depth, ranks_feat, ranks_bev,
                      interval_starts, interval_lengths, row_first,
                      static_cast<float*>(out_padded_bf16), stream);
}

int veon_bev_pool_v2_fwd_maxpool(int c, int n_intervals, int batch, int Z, int Y,
                                 int X, int dz, int dy, int dx,
                                 const float* depth, const float* feat,
                                 const int* ranks_depth, const int* ranks_feat,
                                 const int* ranks_bev,
                                 const int* interval_starts,
                                 const int* interval_lengths,
                                 const int* row_first, float* out,
                                 void* stream) {
  return veon_bev_pool_v2_fwd_maxpool_ex(
      c, n_intervals, batch, Z, Y, X, dz, dy, dx, depth, feat, VEON_FEAT_F32,
      ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths,
      row_first, out, stream);
}

static int fused_impl(int c, int n_intervals, int batch, int64_t voxels_per_batch,
                      const float* depth, const void* feat, int feat_dtype,
                      const int* ranks_depth, const int* ranks_feat,
                      const int* ranks_bev, const int* interval_starts,
                      const int* interval_lengths, const int* plan, float* out,
                      int out_layout, int64_t plane_stride, void* stream) {
  if (c <= 0 || n_intervals < 0 || batch <= 0 || voxels_per_batch <= 0 ||
      !out || !plan)
    return VEON_ERR_BAD_ARG;
  if (plane_stride == 0) plane_stride = voxels_per_batch;
  if (plane_stride < voxels_per_batch ||
      (plane_stride != voxels_per_batch && out_layout != VEON_LAYOUT_BCZYX))
    return VEON_ERR_BAD_ARG;
  if (out_layout != VEON_LAYOUT_BZYXC && out_layout != VEON_LAYOUT_BCZYX)
    return VEON_ERR_BAD_ARG;
  if (feat_dtype != VEON_FEAT_F32 && feat_dtype != VEON_FEAT_F16 &&
      feat_dtype != VEON_FEAT_BF16)
    return VEON_ERR_BAD_ARG;
  if (n_intervals > 0 &&
      (!depth || !feat || !ranks_depth || !ranks_feat || !ranks_bev ||
       !interval_starts || !interval_lengths))
    return VEON_ERR_BAD_ARG;
  if ((int64_t)batch * voxels_per_batch > 0x7fffffffLL)
    return VEON_ERR_BAD_ARG;  // ranks_bev is int32 (the reference ABI)
  if ((reinterpret_cast<uintptr_t>(plan) & 15u) != 0) return VEON_ERR_BAD_ARG;
  PoolArgs a{depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
             interval_lengths};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t tiles_per_batch = (voxels_per_batch + kTileV - 1) / kTileV;
  const int64_t n_tiles = tiles_per_batch * batch;
  if (n_tiles > 0x3fffffffLL) return VEON_ERR_BAD_ARG;
  const int4* plan4 = reinterpret_cast<const int4*>(plan);
  if (out_layout == VEON_LAYOUT_BZYXC) {
    // lanes = (voxel, 4 channels) so that stores stay 16 bytes per lane
    const bool v4 = (c % 4 == 0) && aligned16(feat) && aligned16(out);
#define VEON_LAUNCH_CL(VEC, FT)                                                \
  hipLaunchKernelGGL((k_pool_fused_cl<VEC, FT>), dim3((unsigned)n_tiles),      \
                     dim3(kBlock), 0, s, a, plan4, c, c / VEC,                 \
                     voxels_per_batch, tiles_per_batch, out)
    if (feat_dtype == VEON_FEAT_F32) {
      if (v4) VEON_LAUNCH_CL(4, FeatF32); else VEON_LAUNCH_CL(1, FeatF32);
    } else if (feat_dtype == VEON_FEAT_F16) {
      if (v4) VEON_LAUNCH_CL(4, FeatF16); else VEON_LAUNCH_CL(1, FeatF16);
    } else {
      if (v4) VEON_LAUNCH_CL(4, FeatBF16); else VEON_LAUNCH_CL(1, FeatBF16);
    }
#undef VEON_LAUNCH_CL
    return launch_status();
  }
  // channels-first: channel slab and LDS tile width by problem shape
  const int cs = c <= 128 ? c : 64;
  const int slabs = (c + cs - 1) / cs;
  const bool dense = (int64_t)n_intervals > 12 * n_tiles;  // mean intervals/tile
  const int cap = dense ? 64 : 32;
  const size_t lds = (size_t)cs * (cap + 1) * sizeof(float) +
                     (size_t)(2 * kTileV + 2 + 2 * kPmax) * sizeof(int);
  const dim3 grid((unsigned)n_tiles, (unsigned)slabs);
  // (with several channel slabs the linear workgroup id is not the tile id)
  const int ov = (veon_pool_debug_flags >> 8) & 15;
  const int chunked = (slabs != 1 || (veon_pool_debug_flags & 16)) ? -1 : (ov ? ov - 1 : -1);
#define VEON_LAUNCH_CF(VEC, CAP, FT)                                          \
  hipLaunchKernelGGL((k_pool_fused_cf<VEC, CAP, FT>), grid, dim3(kBlock), lds, \
                     s, a, plan4, c, cs, voxels_per_batch, tiles_per_batch, out,   \
                     plane_stride, chunked)
#define VEON_LAUNCH_CF2(VEC, FT) \
  do { if (dense) VEON_LAUNCH_CF(VEC, 64, FT); else VEON_LAUNCH_CF(VEC, 32, FT); } while (0)
  if (feat_dtype == VEON_FEAT_F32) {
    if ((c % 4 == 0) && aligned16(feat)) VEON_LAUNCH_CF2(4, FeatF32);
    else VEON_LAUNCH_CF2(1, FeatF32);
  } else {
    // 4 channels (8 bytes) per lane: the kernel is latency-, not byte-bound,
    // and 8 channels per lane halve the lanes in flight (measured slower)
    const bool v4 = (c % 4 == 0) && (cs % 4 == 0) && aligned16(feat);
    if (feat_dtype == VEON_FEAT_F16) {
      if (v4) VEON_LAUNCH_CF2(4, FeatF16); else VEON_LAUNCH_CF2(1, FeatF16);
    } else {
      if (v4) VEON_LAUNCH_CF2(4, FeatBF16); else VEON_LAUNCH_CF2(1, FeatBF16);
    }
  }
#undef VEON_LAUNCH_CF2
#undef VEON_LAUNCH_CF
  return launch_status();
}

int veon_bev_pool_v2_fwd_fused_ex(int c, int n_intervals, int batch,
                                  int64_t voxels_per_batch, const float* depth,
                                  const void* feat, int feat_dtype,
                                  const int* ranks_depth, const int* ranks_feat,
                                  const int* ranks_bev,
                                  const int* interval_starts,
                                  const int* interval_lengths, const int* plan,
                                  float* out, int out_layout, void* stream) {
  return fused_impl(c, n_intervals, batch, voxels_per_batch, depth, feat, feat_dtype,
                    ranks_depth, ranks_feat, ranks_bev, interval_starts,
                    interval_lengths, plan, out, out_layout, 0, stream);
}

int veon_bev_pool_v2_fwd_fused_strided(int c, int n_intervals, int batch,
                                       int64_t voxels_per_batch, const float* depth,
                                       const void* feat, int feat_dtype,
                                       const int* ranks_depth, const int* ranks_feat,
                                       const int* ranks_bev,
                                       const int* interval_starts,
                                       const int* interval_lengths, const int* plan,
                                       float* out, int64_t plane_stride,
                                       void* stream) {
  return fused_impl(c, n_intervals, batch, voxels_per_batch, depth, feat, feat_dtype,
                    ranks_depth, ranks_feat, ranks_bev, interval_starts,
                    interval_lengths, plan, out, VEON_LAYOUT_BCZYX, plane_stride,
                    stream);
}

int veon_bev_pool_v2_fwd_fused(int c, int n_intervals, int batch,
                               int64_t voxels_per_batch, const float* depth,
                               const float* feat, const int* ranks_depth,
                               const int* ranks_feat, const int* ranks_bev,
                               const int* interval_starts,
                               const int* interval_lengths, const int* plan,
                               float* out, int out_layout, void* stream) {
  return veon_bev_pool_v2_fwd_fused_ex(
      c, n_intervals, batch, voxels_per_batch, depth, feat, VEON_FEAT_F32,
      ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths,
      plan, out, out_layout, stream);
}

int veon_feat_nchw_to_nhwc(const void* in, void* out, int elem_bytes, int images,
                           int C, int HW, void* stream) {
  if (!in || !out || in == out || images <= 0 || C <= 0 || HW <= 0 ||
      (elem_bytes != 4 && elem_bytes != 2) || images > 65535)
    return VEON_ERR_BAD_ARG;
  const dim3 grid((unsigned)((HW + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)images);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (elem_bytes == 4)
    hipLaunchKernelGGL(k_feat_nchw_to_nhwc<float>, grid, dim3(256), 0, s,
                       static_cast<const float*>(in), static_cast<float*>(out), C, HW);
  else
    hipLaunchKernelGGL(k_feat_nchw_to_nhwc<unsigned short>, grid, dim3(256), 0, s,
                       static_cast<const unsigned short*>(in),
                       static_cast<unsigned short*>(out), C, HW);
  return launch_status();
}

int veon_volume_maxpool2_f32(const float* in, float* out, int64_t planes, int Z, int Y,
                             int X, void* stream) {
  // planes = B * C (every (b, c) volume is pooled on its own)
  if (!in || !out || planes <= 0 || Z <= 0 || Y <= 0 || X <= 0 || (Z & 1) || (Y & 1) || (X & 1))
    return VEON_ERR_BAD_ARG;
  const int64_t total = planes * (Z / 2) * (Y / 2) * (X / 2);
  const int64_t blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_volume_maxpool2, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, out, planes, Z, Y, X);
  return launch_status();
}

}  // extern "C"
