// bev_pool_v2 "row" kernels for MI355X (gfx950): the wide-channel shapes
// (VEON: C = 256 into 200x200x16) where one feature row is >= 512 bytes.
//
// Same arithmetic as csrc/bev_pool_v2.hip (and as one thread of the reference
// kernel, mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu:21-48): every pooled
// value is the serial fmaf chain over its voxel's points in storage order.
// What differs is the decomposition:
//
//   * index side: ONE dense table vstart[B*Z*Y*X + 1] (exclusive scan of the
//     voxel histogram: voxel v owns points [vstart[v], vstart[v+1]) of the
//     rank-sorted point arrays).  It is what the counting sort of
//     csrc/lss_prepare.hip produces anyway, replaces interval_starts /
//     interval_lengths / ranks_bev / tile plan / row table with O(1) lookups,
//     and is built from the reference's five arrays by veon_bev_pool_voxel_table
//     when those are what the caller holds (accelerate=True cache);
//   * gather side: lanes = channels.  A wave owns a few voxels, flattens their
//     points into one list (<= 64 per batch: lane i loads ranks_feat / depth of
//     point i), then walks the list with wave-uniform row addresses: every
//     feature row is read ONCE per workgroup as one full-width coalesced wave
//     load (1 KiB for fp32 C = 256), 8 rows in flight per wave.  The old
//     kernels split C into 32-64 channel slabs, i.e. re-staged the indices 4-8x
//     and gathered 128-256-byte row pieces;
//   * max-pool variant: the <= dz*dy*dx sums of a pooled voxel are reduced in
//     registers (no LDS tile, no atomics), result stored as one contiguous
//     channels-last row (bf16, the Conv3d body's input) or transposed through a
//     small LDS tile for the (B,C,Zo,Yo,Xo) fp32 layout.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/veon_hip.h"
#include "half_mode.h"

namespace {

constexpr int kWave = 64;

// Experiment knob of tools/poolbench.py (ablations: bit 0 = workers exit at
// once, bit 1 = cold workgroups exit at once, bit 2 / 3 = skip warm / hot lists;
// bit 4 = tile = blockIdx instead of the XCD-contiguous order, results unchanged).
// 0 in production; results of bits 0-3 are only meaningful with 0.
int g_pool_debug = 0;
int g_pool_workers = 0, g_pool_cold = 0, g_pool_warm = 0;  // 0 = built-in defaults
// default tile order (xcd_grouped lg; -1 = blockIdx), from tools/xcd_order_ab.py at SV:
// fused 165 -> 157 us fp32 rows, 135 -> 128.5 us bf16 rows with runs of 32 tiles; the
// max-pool kernel (and the S2 slab kernel) gain nothing measurable
constexpr int kOrderCf = 5, kOrderMp = -1;

inline int launch_status() {
  return hipGetLastError() == hipSuccess ? VEON_OK : VEON_ERR_LAUNCH;
}

__device__ __forceinline__ int rl(int v, int lane) {
  return __builtin_amdgcn_readlane(v, lane);
}
__device__ __forceinline__ float rlf(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// Workgroups are dealt to the 8 XCDs round-robin by their linear id, each XCD with
// its own 4 MiB L2.  The feature row of a pixel is shared by all depth bins of its
// ray, i.e. by voxels that are neighbours in y (and x).  tile = blockIdx spreads
// neighbouring tiles over eight L2s; this order hands runs of 2^lg consecutive
// tiles to ONE XCD (super-groups of 8 * 2^lg tiles permuted among themselves: runs
// stay short against a z-plane, so the XCDs remain evenly loaded -- one contiguous
// eighth of the volume per XCD was measured 17 % SLOWER: whole z-planes are empty).
// Pure scheduling: every tile is still processed once, by one workgroup.
__device__ __forceinline__ int64_t xcd_grouped(int64_t b, int64_t n, int lg) {
  if (lg < 0) return b;
  const int64_t sg = b >> (3 + lg);
  if (((sg + 1) << (3 + lg)) > n) return b;  // ragged last super-group
  const int64_t g = (b >> 3) & ((1 << lg) - 1);
  return (((sg << 3) + (b & 7)) << lg) + g;
}
// order knob from the debug flags: bit 4 = identity, bits 8..11 = lg + 1 (0: default)
inline int tile_order_lg(int flags, int dflt) {
  if (flags & 16) return -1;
  const int v = (flags >> 8) & 15;
  return v ? v - 1 : dflt;
}

__device__ __forceinline__ int uni(int v) {
  return __builtin_amdgcn_readfirstlane(v);
}

// running block maximum: `if (v > m) m = v` from -inf, the oracle's / torch's
// comparison (a NaN sum is ignored by both; +0 and -0 compare equal)
__device__ __forceinline__ float vmax(float m, float v) { return v > m ? v : m; }

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) {
  return __uint_as_float(u & 0xffff0000u);
}
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

// ---- feature-row fragments ------------------------------------------------
// FT: 0 fp32, 1 fp16, 2 bf16 (VEON_FEAT_*).  Loads return the RAW bits and are
// unconditional; widening to fp32 (exact) happens where the value is consumed,
// so that no load sits in a branch with its own use (hipcc then waits
// vmcnt(0) right behind every load and the gather serialises).
template <int FT> struct Raw2 { using T = unsigned; };
template <> struct Raw2<VEON_FEAT_F32> { using T = float2; };
template <int FT> struct Raw4 { using T = uint2; };
template <> struct Raw4<VEON_FEAT_F32> { using T = float4; };

template <int FT>
__device__ __forceinline__ typename Raw2<FT>::T load2(const void* feat, int64_t e) {
  if constexpr (FT == VEON_FEAT_F32)
    return *reinterpret_cast<const float2*>(static_cast<const float*>(feat) + e);
  else
    return *reinterpret_cast<const unsigned*>(static_cast<const unsigned short*>(feat) + e);
}
template <int FT>
__device__ __forceinline__ typename Raw4<FT>::T load4(const void* feat, int64_t e) {
  if constexpr (FT == VEON_FEAT_F32)
    return *reinterpret_cast<const float4*>(static_cast<const float*>(feat) + e);
  else
    return *reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(feat) + e);
}
template <int FT>
__device__ __forceinline__ float2 cvt2(const typename Raw2<FT>::T& r) {
  if constexpr (FT == VEON_FEAT_F32) {
    return r;
  } else if constexpr (FT == VEON_FEAT_F16) {
    const half2_t h = __builtin_bit_cast(half2_t, r);
    return make_float2((float)h[0], (float)h[1]);
  } else {
    return make_float2(bf_lo(r), bf_hi(r));
  }
}
template <int FT>
__device__ __forceinline__ float4 cvt4(const typename Raw4<FT>::T& r) {
  if constexpr (FT == VEON_FEAT_F32) {
    return r;
  } else if constexpr (FT == VEON_FEAT_F16) {
    const half4_t h = __builtin_bit_cast(half4_t, r);
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
  } else {
    return make_float4(bf_lo(r.x), bf_hi(r.x), bf_lo(r.y), bf_hi(r.y));
  }
}

// ---------------------------------------------------------------------------
// vstart from the reference's arrays: vstart[v] = first point of the first
// interval whose voxel is >= v (binary search over the ascending interval keys).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_voxel_table(
    const int* __restrict__ ranks_bev, const int* __restrict__ interval_starts,
    int n_intervals, int n_points, const int* __restrict__ counts, int64_t n_bins,
    int* __restrict__ vstart) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v > n_bins) return;
  if (counts != nullptr) {
    n_points = counts[0];
    n_intervals = counts[1];
  }
  int lo = 0, hi = n_intervals;  // first interval with key >= v
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)ranks_bev[interval_starts[mid]] < v) lo = mid + 1;
    else hi = mid;
  }
  vstart[v] = lo < n_intervals ? interval_starts[lo] : n_points;
}

// ---------------------------------------------------------------------------
// Per-wave point list.  A wave's points are NSEG contiguous runs of the sorted
// point arrays; run g is cut into GL sub-intervals (voxels) by the boundaries
// held in lanes [lb + g*(GL+1), lb + g*(GL+1) + GL] of `vs`.
// The dependent chain  boundaries -> {ranks_feat, ranks_depth} -> depth -> rows
// is split into stages so that a wave can issue one stage for several lists
// before it waits for any of them.
// ---------------------------------------------------------------------------
template <int NSEG, int GL>
__device__ __forceinline__ int total_points(int vs, int lb) {
  int n = 0;
#pragma unroll
  for (int g = 0; g < NSEG; ++g)
    n += rl(vs, lb + g * (GL + 1) + GL) - rl(vs, lb + g * (GL + 1));
  return n;
}

struct Stage1 {  // after the boundary load
  int p;         // this lane's point (index into the sorted arrays)
  int slot;      // g*GL + sub-interval index
  bool act;
  unsigned long long last;  // bit k: point k closes its sub-interval
  int nb;
};

// inclusive scan over the lanes of one 16-lane row (DPP row shifts)
__device__ __forceinline__ int row_scan16(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
  return x;
}

// Run table of a list, once per list: lane g (< NSEG <= 16) holds the first point
// of run g (a) and the number of points in runs 0..g (ce).
struct RunTab {
  int a, ce;
};
template <int NSEG, int GL>
__device__ __forceinline__ RunTab run_table(int vs, int lb, int lane) {
  static_assert(NSEG <= 16, "one DPP row");
  RunTab t;
  const int g = lane < NSEG ? lane : NSEG - 1;
  t.a = __shfl(vs, lb + g * (GL + 1));
  const int e = __shfl(vs, lb + g * (GL + 1) + GL);
  t.ce = row_scan16(lane < NSEG ? e - t.a : 0);
  return t;
}

// Lane i -> point base + i of the list: binary search over the cumulative counts
// (log2 NSEG lane permutes instead of a scan over all runs), then the voxel of
// the point inside its run from the run's own boundaries.
template <int NSEG, int GL>
__device__ __forceinline__ Stage1 stage1(const RunTab& t, int vs, int lb, int base, int n,
                                         int lane) {
  Stage1 s;
  const int nb = (n - base) < kWave ? (n - base) : kWave;
  s.nb = nb;
  s.act = lane < nb;
  const int idx = s.act ? base + lane : 0;
  constexpr int P2 = NSEG <= 1 ? 1 : NSEG <= 2 ? 2 : NSEG <= 4 ? 4 : NSEG <= 8 ? 8 : 16;
  int g = 0;  // number of runs that end at or before idx
#pragma unroll
  for (int step = P2 / 2; step >= 1; step >>= 1) {
    const int cand = g + step;
    const int val = __shfl(t.ce, cand - 1 < NSEG ? cand - 1 : NSEG - 1);
    if (cand <= NSEG - 1 && val <= idx) g = cand;
  }
  if constexpr (P2 == 1) g = 0;
  // (P2 / 2 .. 1 reach at most P2 - 1 >= NSEG - 1 runs skipped, which is the maximum
  //  for a valid idx < n)
  const int a_g = __shfl(t.a, g);
  // (lane permutes stay outside conditionals: an inactive SOURCE lane reads as 0)
  const int ce_prev = __shfl(t.ce, g > 0 ? g - 1 : 0);
  const int cs_g = g > 0 ? ce_prev : 0;
  const int p = s.act ? a_g + (idx - cs_g) : 0;
  int sub = 0;
  bool last = false;
#pragma unroll
  for (int k = 1; k <= GL; ++k) {
    const int bk = __shfl(vs, lb + g * (GL + 1) + k);
    sub += (bk <= p) ? 1 : 0;
    last = last || (bk == p + 1);
  }
  s.p = p;
  s.slot = g * GL + sub;
  s.last = __ballot(s.act && last);
  return s;
}

// ---------------------------------------------------------------------------
// Rolling row pipeline over ONE batch of nb <= 64 points whose row offsets
// (ranks_feat * c, in elements; < 2^31 by contract) sit in lanes 0..nb-1 of `rfv`: RING row loads stay in flight, slot u is refilled
// right after it has been consumed (in-order vmcnt makes the wait for the
// oldest load a counted one).  All loads are unconditional -- the tail re-reads
// the last row -- because hipcc counts conservatively across branches: a
// conditional refill turns every wait into vmcnt(0).  consume(k, f): point k of
// the batch, f[NCH] = this lane's NCH consecutive channels widened to fp32.
// ---------------------------------------------------------------------------
constexpr int kRing = 8;

template <int FT, int NCH> struct RawN;
template <> struct RawN<VEON_FEAT_F32, 4> { using T = float4; };
template <> struct RawN<VEON_FEAT_F32, 2> { using T = float2; };
template <> struct RawN<VEON_FEAT_F32, 1> { using T = float; };
template <int FT> struct RawN<FT, 4> { using T = uint2; };
template <int FT> struct RawN<FT, 2> { using T = unsigned; };
template <int FT> struct RawN<FT, 1> { using T = unsigned short; };

template <int FT, int NCH>
__device__ __forceinline__ typename RawN<FT, NCH>::T loadn(const void* feat, int64_t e) {
  using T = typename RawN<FT, NCH>::T;
  if constexpr (FT == VEON_FEAT_F32)
    return *reinterpret_cast<const T*>(static_cast<const float*>(feat) + e);
  else
    return *reinterpret_cast<const T*>(static_cast<const unsigned short*>(feat) + e);
}
template <int FT, int NCH>
__device__ __forceinline__ void cvtn(const typename RawN<FT, NCH>::T& r, float* f) {
  if constexpr (FT == VEON_FEAT_F32) {
    if constexpr (NCH == 4) { f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w; }
    else if constexpr (NCH == 2) { f[0] = r.x; f[1] = r.y; }
    else f[0] = r;
  } else if constexpr (FT == VEON_FEAT_F16) {
    if constexpr (NCH == 4) {
      const half4_t h = __builtin_bit_cast(half4_t, r);
      f[0] = (float)h[0]; f[1] = (float)h[1]; f[2] = (float)h[2]; f[3] = (float)h[3];
    } else if constexpr (NCH == 2) {
      const half2_t h = __builtin_bit_cast(half2_t, r);
      f[0] = (float)h[0]; f[1] = (float)h[1];
    } else {
      f[0] = (float)__builtin_bit_cast(_Float16, r);
    }
  } else {
    if constexpr (NCH == 4) {
      f[0] = bf_lo(r.x); f[1] = bf_hi(r.x); f[2] = bf_lo(r.y); f[3] = bf_hi(r.y);
    } else if constexpr (NCH == 2) {
      f[0] = bf_lo(r); f[1] = bf_hi(r);
    } else {
      f[0] = __uint_as_float((unsigned)r << 16);
    }
  }
}

template <int FT, int NCH, int RING, typename F>
__device__ __forceinline__ void gather_batch(const void* feat, int c, int chl, int rfv,
                                             int nb, F&& consume) {
  typename RawN<FT, NCH>::T raw[RING];
#pragma unroll
  for (int u = 0; u < RING; ++u) {
    const int kk = u < nb ? u : nb - 1;
    raw[u] = loadn<FT, NCH>(feat, (int64_t)rl(rfv, kk) + chl);
  }
  int k0 = 0;
  for (; k0 + RING < nb; k0 += RING) {
#pragma unroll
    for (int u = 0; u < RING; ++u) {
      float f[NCH];
      cvtn<FT, NCH>(raw[u], f);
      consume(k0 + u, f);
      const int kn = k0 + u + RING;
      const int kk = kn < nb ? kn : nb - 1;
      raw[u] = loadn<FT, NCH>(feat, (int64_t)rl(rfv, kk) + chl);
      // keep consume / refill in program order: the scheduler otherwise batches
      // the refills and the counted waits degrade towards vmcnt(0)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int u = 0; u < RING; ++u)
    if (k0 + u < nb) {
      float f[NCH];
      cvtn<FT, NCH>(raw[u], f);
      consume(k0 + u, f);
    }
}

// ---------------------------------------------------------------------------
// (A) pool + (dz,dy,dx) block max, 8 waves per workgroup.
//
// Point counts per pooled voxel are heavy-tailed (VEON shape: median 6, 95 %
// <= 32, maximum 1116; one input voxel holds up to 609 points and its sum is a
// SERIAL chain by contract) and the long lists cluster around the cameras, so
// the work is split by list length:
//   cold (<= kCold points, 95 % of the pooled voxels, half the rows): cold
//     workgroups (blockIdx >= kWorkers) take 32 consecutive pooled voxels, wave
//     w four of them, lanes = 4 consecutive channels.  Their boundary entries
//     sit in 16-lane groups of one register (one load), their point lists are
//     fetched stage by stage for all four before the first row gather starts;
//   the rest belongs to the worker workgroups (blockIdx < kWorkers), worker j
//     owning the pooled voxels j + k*kWorkers (interleaved, so that the cluster
//     of long lists spreads over all of them).  A worker finds its long lists
//     with one lane-parallel look at the table, then
//     warm (<= kWarm points): one wave per list, pulled from an LDS queue;
//     hot: the whole workgroup on one pooled voxel -- every (input voxel,
//       64-channel quarter) is a task, lanes = 1 channel, 32 rows in flight per
//       wave; the <= 8 sums meet in LDS.
// OUT: 0 = (B,C,Zo,Yo,Xo) fp32 (cold: via an LDS transpose), 1 = interior of the
// padded channels-last bf16 grid (csrc/conv3d.hip).
// ---------------------------------------------------------------------------
constexpr int kMW = 8;                 // waves per workgroup
constexpr int kNP = 4;                 // pooled voxels per wave (cold)
constexpr int kPV = kMW * kNP;         // pooled voxels per cold workgroup
constexpr int kColdDef = 64;   // measured on MI355X (tools/pool_tune.py)
constexpr int kWarmDef = 256;
constexpr int kWorkersDef = 512;
constexpr int kCand = kMW * 64;        // candidates a worker looks at per pass

template <int DZ, int DY, int DX>
__device__ __forceinline__ int64_t seg_entry(int b, int zo, int yo, int xo, int g, int Z,
                                             int Y, int X) {
  const int rz = g / DY, ry = g - rz * DY;
  const int64_t row = ((int64_t)b * Z + (zo * DZ + rz)) * Y + (yo * DY + ry);
  return row * X + xo * DX;
}

__device__ __forceinline__ uint2 pack_bf16x4(const float* v) {
  uint2 pk;
  pk.x = (unsigned)__builtin_bit_cast(unsigned short, (veon_half_native)v[0]) |
         ((unsigned)__builtin_bit_cast(unsigned short, (veon_half_native)v[1]) << 16);
  pk.y = (unsigned)__builtin_bit_cast(unsigned short, (veon_half_native)v[2]) |
         ((unsigned)__builtin_bit_cast(unsigned short, (veon_half_native)v[3]) << 16);
  return pk;
}

// max over the occupied inputs; with fewer than `full` of them occupied the
// block also holds zeros (view_transformer_raw.py:549-553 on a zero-filled volume)
__device__ __forceinline__ float pooled_value(float m, int n_occ, int full) {
  return n_occ < full ? vmax(0.f, m) : m;  // m = -inf when nothing is occupied
}

template <int FT, int DZ, int DY, int DX, int OUT>
__global__ __launch_bounds__(kMW * 64) void k_rows_maxpool(
    const float* __restrict__ depth, const void* __restrict__ feat,
    const int* __restrict__ ranks_depth, const int* __restrict__ ranks_feat,
    const int* __restrict__ vstart, int c, int batch, int Z, int Y, int X,
    void* __restrict__ outp, int dbg, int kWorkers, int kCold, int kWarm, int order,
    const int* __restrict__ chunk_order) {
  constexpr int NSEG = DZ * DY, GL = DX, NE = NSEG * (GL + 1), FULL = DZ * DY * DX;
  static_assert(NE <= 16, "boundary entries of one pooled voxel must fit 16 lanes");
  if (dbg && (((dbg & 1) && blockIdx.x < kWorkers) || ((dbg & 2) && blockIdx.x >= kWorkers)))
    return;
  extern __shared__ int lds_i[];
  const int Zo = Z / DZ, Yo = Y / DY, Xo = X / DX;
  const int plane = Zo * Yo * Xo;                     // pooled voxels per batch element
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
  const float kmin = -__builtin_inff();
  auto out_row = [&](int b, int lin) {  // OUT == 1: first channel of a pooled voxel
    const int xo = lin % Xo, yo = (lin / Xo) % Yo, zo = lin / (Xo * Yo);
    return reinterpret_cast<unsigned short*>(outp) +
           ((((int64_t)b * (Zo + 2) + zo + 1) * (Yo + 2) + yo + 1) * (Xo + 2) + 1 + xo) *
               (int64_t)c;
  };

  if (blockIdx.x < kWorkers) {
    // ================= workers: warm and hot lists =================
    int* wlist = lds_i;                  // [kCand] warm pooled ids (b*plane + lin)
    int* hlist = wlist + kCand;          // [kCand] hot pooled ids
    int* ctr = hlist + kCand;            // [4] n_warm, n_hot, pulled, pad
    int* ctab = ctr + 4;                 // [FULL][2] chain start, length
    float* hkey = reinterpret_cast<float*>(ctab + 2 * FULL);  // [FULL][256] sums
    const int64_t total = (int64_t)batch * plane;
    for (int64_t pass0 = 0; pass0 < total; pass0 += (int64_t)kWorkers * kCand) {
      if (threadIdx.x < 4) ctr[threadIdx.x] = 0;
      __syncthreads();
      {
        const int64_t pid = pass0 + blockIdx.x + (int64_t)threadIdx.x * kWorkers;
        if (pid < total) {
          const int b = (int)(pid / plane);
          const int lin = (int)(pid - (int64_t)b * plane);
          const int xo = lin % Xo, yo = (lin / Xo) % Yo, zo = lin / (Xo * Yo);
          int n = 0;
#pragma unroll
          for (int g = 0; g < NSEG; ++g) {
            const int64_t e = seg_entry<DZ, DY, DX>(b, zo, yo, xo, g, Z, Y, X);
            n += vstart[e + GL] - vstart[e];
          }
          if (n > kWarm && !(dbg & 8)) hlist[atomicAdd(&ctr[1], 1)] = (int)pid;
          else if (n > kCold && !(dbg & 4)) wlist[atomicAdd(&ctr[0], 1)] = (int)pid;
        }
      }
      __syncthreads();
      const int nwarm = ctr[0], nhot = ctr[1];
      // ---- hot: the workgroup on one pooled voxel, tasks = (input voxel, quarter)
      for (int hi = 0; hi < nhot; ++hi) {
        const int pid = hlist[hi];
        const int b = pid / plane;
        const int lin = pid - b * plane;
        const int xo = lin % Xo, yo = (lin / Xo) % Yo, zo = lin / (Xo * Yo);
        if (threadIdx.x < FULL) {
          const int g = threadIdx.x / GL, k = threadIdx.x - g * GL;
          const int64_t e = seg_entry<DZ, DY, DX>(b, zo, yo, xo, g, Z, Y, X) + k;
          const int pa = vstart[e];
          ctab[2 * threadIdx.x] = pa;
          ctab[2 * threadIdx.x + 1] = vstart[e + 1] - pa;
        }
        for (int c0 = 0; c0 < c; c0 += 256) {
          if (threadIdx.x == 0) ctr[2] = 0;
          __syncthreads();
          for (;;) {
            int ti = 0;
            if (lane == 0) ti = atomicAdd(&ctr[2], 1);
            ti = uni(ti);
            if (ti >= FULL * 4) break;
            const int q = ti >> 2, h = ti & 3;
            const int pa = ctab[2 * q], n = ctab[2 * q + 1];
            const int ch = c0 + h * 64 + lane;
            if (n == 0 || c0 + h * 64 >= c) continue;  // uniform
            const int chl = ch < c ? ch : 0;
            float acc = 0.f;
            int rfn = 0;
            float dn = 0.f;
            if (lane < n) {
              rfn = ranks_feat[pa + lane] * c;
              dn = depth[ranks_depth[pa + lane]];
            }
            for (int base = 0; base < n; base += kWave) {
              const int rfv = rfn;
              const float dj = dn;
              const int nb = (n - base) < kWave ? (n - base) : kWave;
              const int qn = base + kWave + lane;
              rfn = 0;
              dn = 0.f;
              if (qn < n) {
                rfn = ranks_feat[pa + qn] * c;
                dn = depth[ranks_depth[pa + qn]];
              }
              gather_batch<FT, 1, 32>(feat, c, chl, rfv, nb, [&](int kk, const float* f) {
                acc = fmaf(f[0], rlf(dj, kk), acc);
              });
            }
            hkey[q * 256 + h * 64 + lane] = acc;
          }
          __syncthreads();
          if (threadIdx.x < 256 && c0 + threadIdx.x < c) {
            const int cc = threadIdx.x;
            float m = kmin;
            int n_occ = 0;
#pragma unroll
            for (int q = 0; q < FULL; ++q)
              if (ctab[2 * q + 1] > 0) {
                m = vmax(m, hkey[q * 256 + cc]);
                ++n_occ;
              }
            const float v = pooled_value(m, n_occ, FULL);
            if constexpr (OUT == 1)
              out_row(b, lin)[c0 + cc] = __builtin_bit_cast(unsigned short, (veon_half_native)v);
            else
              static_cast<float*>(outp)[((int64_t)b * c + c0 + cc) * plane + lin] = v;
          }
          __syncthreads();
        }
      }
      // ---- warm: one wave per list, pulled from the queue
      if (threadIdx.x == 0) ctr[2] = 0;
      __syncthreads();
      for (;;) {
        int wi = 0;
        if (lane == 0) wi = atomicAdd(&ctr[2], 1);
        wi = uni(wi);
        if (wi >= nwarm) break;
        const int pid = wlist[wi];
        const int b = pid / plane;
        const int lin = pid - b * plane;
        const int xo = lin % Xo, yo = (lin / Xo) % Yo, zo = lin / (Xo * Yo);
        int vs = 0;
        if (lane < NE) {
          const int g = lane / (GL + 1), k = lane - g * (GL + 1);
          vs = vstart[seg_entry<DZ, DY, DX>(b, zo, yo, xo, g, Z, Y, X) + k];
        }
        const int vnext = __shfl_down(vs, 1);
        const int n_occ = __popcll(__ballot(lane < NE && (lane % (GL + 1)) < GL && vnext > vs));
        const RunTab rt = run_table<NSEG, GL>(vs, 0, lane);
        const int n = rl(rt.ce, NSEG - 1);
        for (int c0 = 0; c0 < c; c0 += 256) {
          const int ch = c0 + lane * 4;
          const bool chact = ch < c;
          const int chl = chact ? ch : 0;
          float acc[4] = {0.f, 0.f, 0.f, 0.f};
          float m[4] = {kmin, kmin, kmin, kmin};
          for (int base = 0; base < n; base += kWave) {
            const Stage1 st = stage1<NSEG, GL>(rt, vs, 0, base, n, lane);
            int rfj = 0;
            float dj = 0.f;
            if (st.act) {
              rfj = ranks_feat[st.p] * c;
              dj = depth[ranks_depth[st.p]];
            }
            gather_batch<FT, 4, kRing>(feat, c, chl, rfj, st.nb, [&](int kk, const float* f) {
              const float d = rlf(dj, kk);
#pragma unroll
              for (int k = 0; k < 4; ++k) acc[k] = fmaf(f[k], d, acc[k]);
              if ((st.last >> kk) & 1ull) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  m[k] = vmax(m[k], acc[k]);
                  acc[k] = 0.f;
                }
              }
            });
          }
          float v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = pooled_value(m[k], n_occ, FULL);
          if (chact) {
            if constexpr (OUT == 1) {
              *reinterpret_cast<uint2*>(out_row(b, lin) + ch) = pack_bf16x4(v);
            } else {
#pragma unroll
              for (int k = 0; k < 4; ++k)
                static_cast<float*>(outp)[((int64_t)b * c + ch + k) * plane + lin] = v[k];
            }
          }
        }
      }
      __syncthreads();
    }
    return;
  }

  // ================= cold workgroups =================
  float* tile = reinterpret_cast<float*>(lds_i);  // OUT == 0: [256][kPV + 1]
  int* hotf = lds_i + 256 * (kPV + 1);            // OUT == 0: [kPV] column is not cold
  const int chunks = (plane + kPV - 1) / kPV;
  // cold workgroup -> chunk of pooled voxels in XCD-grouped order (the linear id
  // of cold workgroup i is kWorkers + i, kWorkers a multiple of 8)
  // (or the caller's order, a permutation of the chunks:
  //  veon_bev_pool_v2_fwd_rows_maxpool_ordered)
  const int64_t cw = chunk_order != nullptr
                         ? (int64_t)chunk_order[blockIdx.x - kWorkers]
                         : xcd_grouped((int64_t)blockIdx.x - kWorkers,
                                       (int64_t)gridDim.x - kWorkers, order);
  const int b = (int)(cw / chunks);
  const int lin0 = (int)(cw - (int64_t)b * chunks) * kPV;

  // ---- the wave's four pooled voxels as ONE point list: kNP * NSEG runs of GL
  //      voxels; boundary entries in lanes 0 .. kNP*NE-1 (one load)
  constexpr int TSEG = kNP * NSEG;
  static_assert(kNP * NE <= kWave, "boundary entries of one wave");
  const int jl = lane / NE, il = lane - jl * NE;
  const int g_of = il / (GL + 1), k_of = il - g_of * (GL + 1);
  int vs = 0;
  int64_t orow = 0;  // OUT == 1: padded row (in rows of c) of this lane group's voxel
  {
    const int lin = lin0 + w + kMW * jl;
    if (jl < kNP && lin < plane) {
      const int xo = lin % Xo, yo = (lin / Xo) % Yo, zo = lin / (Xo * Yo);
      vs = vstart[seg_entry<DZ, DY, DX>(b, zo, yo, xo, g_of, Z, Y, X) + k_of];
      orow = (((int64_t)b * (Zo + 2) + zo + 1) * (Yo + 2) + yo + 1) * (Xo + 2) + 1 + xo;
    }
  }
  const int orow_lo = (int)orow, orow_hi = (int)(orow >> 32);
  const int vnext = __shfl_down(vs, 1);
  const unsigned long long occm = __ballot(jl < kNP && k_of < GL && vnext > vs);
  int nocc[kNP], cum[kNP + 1];
  bool mine[kNP];  // cold and inside the volume: this wave writes it
  unsigned skipm = 0;
  cum[0] = 0;
#pragma unroll
  for (int j = 0; j < kNP; ++j) {
    const int nj = total_points<NSEG, GL>(vs, NE * j);
    nocc[j] = __popcll((occm >> (NE * j)) & ((1ull << NE) - 1ull));
    mine[j] = (lin0 + w + kMW * j < plane) && nj <= kCold;
    if (nj > kCold) skipm |= 1u << j;
    cum[j + 1] = cum[j] + (nj > kCold ? 0 : nj);
  }
  static_assert(kNP <= 4 && FULL < 256, "n_occ of the wave's pooled voxels in one int");
  unsigned noccp = 0;
#pragma unroll
  for (int j = 0; j < kNP; ++j) noccp |= (unsigned)nocc[j] << (8 * j);
  // long lists belong to the workers: their runs become empty
  {
    const int first = __shfl(vs, NE * (jl < kNP ? jl : 0));
    if (jl < kNP && ((skipm >> jl) & 1u)) vs = first;
  }
  const RunTab rt = run_table<TSEG, GL>(vs, 0, lane);
  const int n = cum[kNP];

  for (int c0 = 0; c0 < c; c0 += 256) {
    const int ch = c0 + lane * 4;
    const bool chact = ch < c;
    const int chl = chact ? ch : 0;  // idle lanes re-read channel 0 (never stored)
    auto emit = [&](int j, float v0, float v1, float v2, float v3) {
      const int xl = w + kMW * j;
      if constexpr (OUT == 1) {
        if (chact) {
          const float v[4] = {v0, v1, v2, v3};
          // j is wave-uniform: two v_readlane (an indexed array would be scratch)
          const int64_t r = ((int64_t)rl(orow_hi, NE * j) << 32) | (unsigned)rl(orow_lo, NE * j);
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(outp) + r * c + ch) =
              pack_bf16x4(v);
        }
      } else {
        if (chact) {
          float* tp = tile + (lane * 4) * (kPV + 1) + xl;
          tp[0] = v0;
          tp[kPV + 1] = v1;
          tp[2 * (kPV + 1)] = v2;
          tp[3 * (kPV + 1)] = v3;
        }
      }
    };
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float m[4] = {kmin, kmin, kmin, kmin};
    for (int base = 0; base < n; base += kWave) {
      const Stage1 st = stage1<TSEG, GL>(rt, vs, 0, base, n, lane);
      int rfj = 0;
      float dj = 0.f;
      if (st.act) {
        rfj = ranks_feat[st.p] * c;
        dj = depth[ranks_depth[st.p]];
      }
      // point closes its pooled voxel: it is the last of that voxel's share of
      // the flattened list
      const int pj = st.slot / (NSEG * GL);
      int ce = cum[1];
#pragma unroll
      for (int j = 1; j < kNP; ++j) ce = (pj == j) ? cum[j + 1] : ce;
      const unsigned long long plast = __ballot(st.act && (base + lane + 1 == ce));
      // (a pooled voxel that continues in the next batch keeps acc and m)
      gather_batch<FT, 4, kRing>(feat, c, chl, rfj, st.nb, [&](int kk, const float* f) {
        const float d = rlf(dj, kk);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = fmaf(f[k], d, acc[k]);
        if ((st.last >> kk) & 1ull) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            m[k] = vmax(m[k], acc[k]);
            acc[k] = 0.f;
          }
          if ((plast >> kk) & 1ull) {
            const int jj = rl(pj, kk);
            const int no = (noccp >> (8 * jj)) & 0xff;  // (an indexed array would
                                                        // live in scratch)
            emit(jj, pooled_value(m[0], no, FULL), pooled_value(m[1], no, FULL),
                 pooled_value(m[2], no, FULL), pooled_value(m[3], no, FULL));
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = kmin;
          }
        }
      });
    }
    // pooled voxels without points: zeros
#pragma unroll
    for (int j = 0; j < kNP; ++j)
      if (mine[j] && nocc[j] == 0) emit(j, 0.f, 0.f, 0.f, 0.f);
    if constexpr (OUT == 0) {
      if (lane < kNP) hotf[w + kMW * lane] = (skipm >> lane) & 1u;
      __syncthreads();
      const int nch = (c - c0) < 256 ? (c - c0) : 256;
      float* obase = static_cast<float*>(outp) + ((int64_t)b * c + c0) * plane + lin0;
      for (int i = threadIdx.x; i < nch * kPV; i += kMW * 64) {
        const int cc = i / kPV, xl = i - cc * kPV;
        if (lin0 + xl < plane && !hotf[xl])
          obase[(int64_t)cc * plane + xl] = tile[cc * (kPV + 1) + xl];
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------
// 2+2-channel flavour of the rolling row pipeline (the fused kernel's lane map:
// channel pairs {2l,2l+1} and, for C > 128, {128+2l,129+2l}).
// ---------------------------------------------------------------------------
template <int FT, bool HI, typename F>
__device__ __forceinline__ void gather_batch22(const void* feat, int c, int cl0, int cl1,
                                               int rfv, int nb, F&& consume) {
  typename Raw2<FT>::T r0[kRing], r1[kRing];
#pragma unroll
  for (int u = 0; u < kRing; ++u) {
    const int kk = u < nb ? u : nb - 1;
    const int64_t ro = (int64_t)rl(rfv, kk);
    r0[u] = load2<FT>(feat, ro + cl0);
    if constexpr (HI) r1[u] = load2<FT>(feat, ro + cl1);
  }
  int k0 = 0;
  for (; k0 + kRing < nb; k0 += kRing) {
#pragma unroll
    for (int u = 0; u < kRing; ++u) {
      float2 f1 = make_float2(0.f, 0.f);
      if constexpr (HI) f1 = cvt2<FT>(r1[u]);
      consume(k0 + u, cvt2<FT>(r0[u]), f1);
      const int kn = k0 + u + kRing;
      const int kk = kn < nb ? kn : nb - 1;
      const int64_t ro = (int64_t)rl(rfv, kk);
      r0[u] = load2<FT>(feat, ro + cl0);
      if constexpr (HI) r1[u] = load2<FT>(feat, ro + cl1);
    }
  }
#pragma unroll
  for (int u = 0; u < kRing; ++u)
    if (k0 + u < nb) {
      float2 f1 = make_float2(0.f, 0.f);
      if constexpr (HI) f1 = cvt2<FT>(r1[u]);
      consume(k0 + u, cvt2<FT>(r0[u]), f1);
    }
}

// ---------------------------------------------------------------------------
// (B) fused zero-fill + pool + (B,C,Z,Y,X) layout, all channels of a tile of
//     TILE consecutive voxel ranks in one workgroup of NW waves.  Sums land in
//     an LDS tile [256][TILE+1] (lanes = channel pairs: at most 2-way LDS banks
//     on the transposing writes); then each wave stores whole channel rows: TILE
//     consecutive voxels = TILE*4 contiguous bytes per instruction,
//     non-temporal.  Tiles without points stream zeros and touch no LDS.
//     Every table entry the workgroup needs (tile bounds, the wave's voxel
//     boundaries, per-voxel occupancy for the store phase) is loaded in ONE
//     level at the top.
//     Point counts per voxel are heavy-tailed (VEON shape: median 1, 99th
//     percentile 20, maximum 609), so the gather runs in two phases:
//       short voxels (<= 64 / (TILE/NW) points): wave w flattens those among its
//         TILE/NW voxels into one list of at most 64 points (one batch);
//       long voxels go to a list in LDS that all waves of the workgroup then
//         drain one chain at a time (the sum of one voxel is a serial chain by
//         contract, so a chain is never split).
// ---------------------------------------------------------------------------

template <int FT, int TILE, int NW, bool HI>
__global__ __launch_bounds__(NW * 64) void k_rows_fused_cf(
    const float* __restrict__ depth, const void* __restrict__ feat,
    const int* __restrict__ ranks_depth, const int* __restrict__ ranks_feat,
    const int* __restrict__ vstart, int c, int64_t vpb, int64_t tiles_per_batch,
    float* __restrict__ out, int64_t ostride, int chunked) {
  constexpr int VW = TILE / NW;  // voxels per wave
  constexpr int LDC = TILE + 1;
  constexpr int kShort = kWave / VW;  // the wave's short voxels fit one batch
  static_assert(2 * VW <= kWave && TILE <= kWave, "tile shape");
  extern __shared__ float tile[];                              // [256][LDC]
  int* hlist = reinterpret_cast<int*>(tile + 256 * LDC);       // [TILE][3] col, start, len
  int* hctr = hlist + 3 * TILE;                                // [2] pushed, pulled
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
  const int64_t t = xcd_grouped(blockIdx.x, gridDim.x, chunked);
  const int b = (int)(t / tiles_per_batch);
  const int64_t vox0 = (t - (int64_t)b * tiles_per_batch) * TILE;
  const int64_t rem = vpb - vox0;
  const int nvox = (int)(rem < TILE ? rem : TILE);
  const int64_t rank0 = (int64_t)b * vpb + vox0;
  constexpr int spv = kWave / TILE;  // channel rows per store instruction (1 or 2)
  const int v = lane % TILE, sub = lane / TILE;

  // ---- one load level: tile bounds, this wave's boundaries, occupancy
  int e = w * VW + lane;  // boundary index inside the tile
  if (e > nvox) e = nvox;
  const int vs = vstart[rank0 + (lane <= VW ? e : 0)];
  const int vlo = vstart[rank0 + (v < nvox ? v : 0)];
  const int vhi = vstart[rank0 + (v < nvox ? v + 1 : 0)];
  const int p_first = vstart[rank0], p_end = vstart[rank0 + nvox];  // uniform
  const bool occ = (v < nvox) && (vhi > vlo);
  if (threadIdx.x < 2) hctr[threadIdx.x] = 0;

  for (int c0 = 0; c0 < c; c0 += 256) {
    const int nch = (c - c0) < 256 ? (c - c0) : 256;
    float* obase = out + ((int64_t)b * c + c0) * ostride + vox0;
    if (p_end == p_first) {  // empty tile: streaming zero fill
      if (v < nvox)
        for (int cc = w * spv + sub; cc < nch; cc += NW * spv)
          __builtin_nontemporal_store(0.f, obase + (int64_t)cc * ostride + v);
      continue;
    }
    __syncthreads();  // counters zeroed (first chunk) / tile and list free again
    const int ch0 = c0 + 2 * lane, ch1 = c0 + 128 + 2 * lane;
    const bool a0 = ch0 < c, a1 = HI && (ch1 < c);
    const int cl0 = a0 ? ch0 : 0, cl1 = a1 ? ch1 : 0;  // idle lanes: channel 0
    auto put = [&](int col, const float2& s0, const float2& s1) {
      if (a0) {
        tile[(2 * lane) * LDC + col] = s0.x;
        tile[(2 * lane + 1) * LDC + col] = s0.y;
      }
      if (HI && a1) {
        tile[(128 + 2 * lane) * LDC + col] = s1.x;
        tile[(129 + 2 * lane) * LDC + col] = s1.y;
      }
    };
    // ---- phase 1: the wave's short voxels as one list; long ones to the LDS list
    {
      const int vn = __shfl_down(vs, 1);
      const int len = vn - vs;                      // lanes 0..VW-1: voxel w*VW+lane
      const bool isv = lane < VW;
      const bool lng = isv && len > kShort;
      if (lng) {
        const int i = atomicAdd(&hctr[0], 1);
        hlist[3 * i] = w * VW + lane;
        hlist[3 * i + 1] = vs;
        hlist[3 * i + 2] = len;
      }
      // runs: lane 2j = start_j, lane 2j+1 = end_j (empty for long voxels)
      const int j2 = lane >> 1;
      const int sj = __shfl(vs, j2), ej = __shfl(vn, j2), lj = ej - sj;
      const int vs2 = (lane & 1) ? ((lj > kShort) ? sj : ej) : sj;
      const RunTab rt = run_table<VW, 1>(vs2, 0, lane);
      const int n = rl(rt.ce, VW - 1);
      if (n > 0) {
        const Stage1 st = stage1<VW, 1>(rt, vs2, 0, 0, n, lane);
        int rfj = 0;
        float dj = 0.f;
        if (st.act) {
          rfj = ranks_feat[st.p] * c;
          dj = depth[ranks_depth[st.p]];
        }
        float2 acc0 = make_float2(0.f, 0.f), acc1 = make_float2(0.f, 0.f);
        gather_batch22<FT, HI>(feat, c, cl0, cl1, rfj, st.nb,
                               [&](int kk, const float2& f0, const float2& f1) {
          const float d = rlf(dj, kk);
          acc0.x = fmaf(f0.x, d, acc0.x);
          acc0.y = fmaf(f0.y, d, acc0.y);
          if constexpr (HI) {
            acc1.x = fmaf(f1.x, d, acc1.x);
            acc1.y = fmaf(f1.y, d, acc1.y);
          }
          if ((st.last >> kk) & 1ull) {
            put(w * VW + rl(st.slot, kk), acc0, acc1);
            acc0 = make_float2(0.f, 0.f);
            acc1 = make_float2(0.f, 0.f);
          }
        });
      }
    }
    __syncthreads();
    // ---- phase 2: drain the long voxels, one chain per wave at a time
    {
      const int nh = hctr[0];
      for (;;) {
        int i = 0;
        if (lane == 0) i = atomicAdd(&hctr[1], 1);
        i = uni(i);
        if (i >= nh) break;
        const int col = hlist[3 * i], pa = hlist[3 * i + 1], n = hlist[3 * i + 2];
        float2 acc0 = make_float2(0.f, 0.f), acc1 = make_float2(0.f, 0.f);
        int rfn = 0;
        float dn = 0.f;
        if (lane < n) {
          rfn = ranks_feat[pa + lane] * c;
          dn = depth[ranks_depth[pa + lane]];
        }
        for (int base = 0; base < n; base += kWave) {
          const int rfv = rfn;
          const float dj = dn;
          const int nb = (n - base) < kWave ? (n - base) : kWave;
          const int q = base + kWave + lane;
          rfn = 0;
          dn = 0.f;
          if (q < n) {
            rfn = ranks_feat[pa + q] * c;
            dn = depth[ranks_depth[pa + q]];
          }
          gather_batch22<FT, HI>(feat, c, cl0, cl1, rfv, nb,
                                 [&](int kk, const float2& f0, const float2& f1) {
            const float d = rlf(dj, kk);
            acc0.x = fmaf(f0.x, d, acc0.x);
            acc0.y = fmaf(f0.y, d, acc0.y);
            if constexpr (HI) {
              acc1.x = fmaf(f1.x, d, acc1.x);
              acc1.y = fmaf(f1.y, d, acc1.y);
            }
          });
        }
        put(col, acc0, acc1);
      }
    }
    __syncthreads();
    // ---- store: lanes = voxels (x spv channel rows)
    if (v < nvox) {
      constexpr int SB = 8;
      float* op = obase + v;
      int cc = w * spv + sub;
      const int stepc = NW * spv;
      for (; cc + (SB - 1) * stepc < nch; cc += SB * stepc) {
        float vals[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u)
          vals[u] = occ ? tile[(cc + stepc * u) * LDC + v] : 0.f;
#pragma unroll
        for (int u = 0; u < SB; ++u)
          __builtin_nontemporal_store(vals[u], op + (int64_t)(cc + stepc * u) * ostride);
      }
      for (; cc < nch; cc += stepc)
        __builtin_nontemporal_store(occ ? tile[cc * LDC + v] : 0.f,
                                    op + (int64_t)cc * ostride);
    }
    if (c0 + 256 < c) {
      __syncthreads();
      if (threadIdx.x < 2) hctr[threadIdx.x] = 0;
    }
  }
}

inline bool aligned16(const void* p) {
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

}  // namespace

extern "C" {

int veon_pool_debug_flags = 0;  // the same flags for bev_pool_v2.hip (bit 4: tile = blockIdx)
void veon_pool_debug_set(int flags) { g_pool_debug = flags; veon_pool_debug_flags = flags; }
void veon_pool_tune_set(int workers, int cold_max, int warm_max) {
  g_pool_workers = workers;
  g_pool_cold = cold_max;
  g_pool_warm = warm_max;
}

int64_t veon_bev_pool_voxel_table_ints(int batch, int64_t voxels_per_batch) {
  if (batch <= 0 || voxels_per_batch <= 0) return 0;
  return (int64_t)batch * voxels_per_batch + 1;
}

int veon_bev_pool_voxel_table(int n_intervals, int n_points, int batch,
                              int64_t voxels_per_batch, const int* ranks_bev,
                              const int* interval_starts, const int* counts,
                              int* vstart, void* stream) {
  if (n_intervals < 0 || n_points < 0 || batch <= 0 || voxels_per_batch <= 0 ||
      !vstart)
    return VEON_ERR_BAD_ARG;
  if ((n_intervals > 0 || counts) && (!ranks_bev || !interval_starts))
    return VEON_ERR_BAD_ARG;
  const int64_t n_bins = (int64_t)batch * voxels_per_batch;
  if (n_bins > 0x7ffffffeLL) return VEON_ERR_BAD_ARG;
  const unsigned blocks = (unsigned)((n_bins + 1 + 255) / 256);
  hipLaunchKernelGGL(k_voxel_table, dim3(blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), ranks_bev, interval_starts,
                     n_intervals, n_points, counts, n_bins, vstart);
  return launch_status();
}

int veon_bev_pool_v2_fwd_rows(int c, int batch, int64_t voxels_per_batch,
                              const float* depth, const void* feat, int feat_dtype,
                              const int* ranks_depth, const int* ranks_feat,
                              const int* vstart, float* out, int64_t plane_stride,
                              int64_t feat_elems, int variant, void* stream) {
  // ranks_depth / ranks_feat may be NULL for an empty point list (vstart all 0)
  if (c <= 0 || (c & 1) || batch <= 0 || voxels_per_batch <= 0 || !depth || !feat ||
      !vstart || !out)
    return VEON_ERR_BAD_ARG;
  if (feat_dtype != VEON_FEAT_F32 && feat_dtype != VEON_FEAT_F16 &&
      feat_dtype != VEON_FEAT_BF16)
    return VEON_ERR_BAD_ARG;
  if (plane_stride == 0) plane_stride = voxels_per_batch;
  if (plane_stride < voxels_per_batch) return VEON_ERR_BAD_ARG;
  if ((int64_t)batch * voxels_per_batch > 0x7ffffffeLL) return VEON_ERR_BAD_ARG;
  if (feat_elems <= 0 || feat_elems > 0x7fffffffLL) return VEON_ERR_BAD_ARG;  // 32-bit rows
  // fp32 rows are read 8 bytes per lane, half rows 4 bytes per lane
  if ((reinterpret_cast<uintptr_t>(feat) & 7u) != 0) return VEON_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define VEON_ROWS_CF(FT, TILE, NW, HIF)                                              \
  do {                                                                               \
    const int64_t tpb = (voxels_per_batch + TILE - 1) / TILE;                        \
    const int64_t n_tiles = tpb * batch;                                             \
    if (n_tiles > 0x7fffffffLL) return VEON_ERR_BAD_ARG;                             \
    constexpr int lds = (256 * (TILE + 1) + 3 * TILE + 4) * (int)sizeof(float);                       \
    static const hipError_t attr = hipFuncSetAttribute(                              \
        reinterpret_cast<const void*>(&k_rows_fused_cf<FT, TILE, NW, HIF>),     \
        hipFuncAttributeMaxDynamicSharedMemorySize, lds);                            \
    if (attr != hipSuccess) return VEON_ERR_LAUNCH;                                  \
    hipLaunchKernelGGL((k_rows_fused_cf<FT, TILE, NW, HIF>),                    \
                       dim3((unsigned)n_tiles),                                      \
                       dim3(NW * 64), lds, s, depth, feat, ranks_depth, ranks_feat,  \
                       vstart, c, voxels_per_batch, tpb, out, plane_stride,          \
                       tile_order_lg(g_pool_debug, kOrderCf));                                 \
  } while (0)
#define VEON_ROWS_CF_D(FT, TILE, NW)                                  \
  do {                                                                \
    if (c > 128) VEON_ROWS_CF(FT, TILE, NW, true);                    \
    else VEON_ROWS_CF(FT, TILE, NW, false);                           \
  } while (0)
#define VEON_ROWS_CF_V(FT)                              \
  do {                                                  \
    if (variant == 1) VEON_ROWS_CF_D(FT, 32, 8);        \
    else if (variant == 2) VEON_ROWS_CF_D(FT, 64, 4);   \
    else if (variant == 3) VEON_ROWS_CF_D(FT, 32, 4);   \
    else VEON_ROWS_CF_D(FT, 64, 8);                     \
  } while (0)
  if (feat_dtype == VEON_FEAT_F32) VEON_ROWS_CF_V(VEON_FEAT_F32);
  else if (feat_dtype == VEON_FEAT_F16) VEON_ROWS_CF_V(VEON_FEAT_F16);
  else VEON_ROWS_CF_V(VEON_FEAT_BF16);
#undef VEON_ROWS_CF_V
#undef VEON_ROWS_CF_D
#undef VEON_ROWS_CF
  return launch_status();
}

int veon_bev_pool_rows_maxpool_chunk(void) { return kPV; }

int veon_bev_pool_v2_fwd_rows_maxpool_ordered(
    int c, int batch, int Z, int Y, int X, int dz, int dy, int dx, const float* depth,
    const void* feat, int feat_dtype, const int* ranks_depth, const int* ranks_feat,
    const int* vstart, void* out, int out_padded_bf16, int64_t feat_elems,
    const int* chunk_order, void* stream) {
  if (c <= 0 || (c & 3) || batch <= 0 || Z <= 0 || Y <= 0 || X <= 0 || !depth ||
      !feat || !vstart || !out)
    return VEON_ERR_BAD_ARG;
  if (dz != 2 || dy != 2 || dx != 2) return VEON_ERR_BAD_ARG;  // VEON's ds_feat
  if (Z % dz || Y % dy || X % dx) return VEON_ERR_BAD_ARG;
  if (feat_dtype != VEON_FEAT_F32 && feat_dtype != VEON_FEAT_F16 &&
      feat_dtype != VEON_FEAT_BF16)
    return VEON_ERR_BAD_ARG;
  if ((int64_t)batch * Z * Y * X > 0x7ffffffeLL) return VEON_ERR_BAD_ARG;
  if (feat_elems <= 0 || feat_elems > 0x7fffffffLL) return VEON_ERR_BAD_ARG;  // 32-bit rows
  if (!aligned16(feat) || (reinterpret_cast<uintptr_t>(out) & 7u))
    return VEON_ERR_BAD_ARG;
  const int64_t plane = (int64_t)(Z / dz) * (Y / dy) * (X / dx);
  if (plane * batch > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  const int kWorkers = g_pool_workers > 0 ? g_pool_workers : kWorkersDef;
  const int kCold = g_pool_cold > 0 ? g_pool_cold : kColdDef;
  const int kWarm = g_pool_warm > 0 ? g_pool_warm : kWarmDef;
  const int64_t wgs = kWorkers + (int64_t)batch * ((plane + kPV - 1) / kPV);
  if (wgs > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // hot workers: candidate list + 8 x 256 keys; cold (fp32 planar): transpose tile
  const size_t lds_hot = (size_t)(2 * kCand + 4 + 2 * 8 + 8 * 256) * sizeof(int);
  const size_t lds_cold = (size_t)(256 * (kPV + 1) + kPV) * sizeof(float);
  const size_t lds_cf = lds_cold > lds_hot ? lds_cold : lds_hot;
#define VEON_ROWS_MP(FT)                                                            \
  do {                                                                              \
    if (out_padded_bf16)                                                            \
      hipLaunchKernelGGL((k_rows_maxpool<FT, 2, 2, 2, 1>), dim3((unsigned)wgs), \
                         dim3(kMW * 64), lds_hot, s, depth, feat, ranks_depth,       \
                         ranks_feat, vstart, c, batch, Z, Y, X, out, g_pool_debug,  \
                         kWorkers, kCold, kWarm, tile_order_lg(g_pool_debug, kOrderMp), \
                         chunk_order);                                              \
    else                                                                            \
      hipLaunchKernelGGL((k_rows_maxpool<FT, 2, 2, 2, 0>), dim3((unsigned)wgs), \
                         dim3(kMW * 64), lds_cf, s, depth, feat, ranks_depth,        \
                         ranks_feat, vstart, c, batch, Z, Y, X, out, g_pool_debug,  \
                         kWorkers, kCold, kWarm, tile_order_lg(g_pool_debug, kOrderMp), \
                         chunk_order);                                              \
  } while (0)
  if (feat_dtype == VEON_FEAT_F32) VEON_ROWS_MP(VEON_FEAT_F32);
  else if (feat_dtype == VEON_FEAT_F16) VEON_ROWS_MP(VEON_FEAT_F16);
  else VEON_ROWS_MP(VEON_FEAT_BF16);
#undef VEON_ROWS_MP
  return launch_status();
}

int veon_bev_pool_v2_fwd_rows_maxpool(int c, int batch, int Z, int Y, int X, int dz,
                                      int dy, int dx, const float* depth,
                                      const void* feat, int feat_dtype,
                                      const int* ranks_depth, const int* ranks_feat,
                                      const int* vstart, void* out, int out_padded_bf16,
                                      int64_t feat_elems, void* stream) {
  return veon_bev_pool_v2_fwd_rows_maxpool_ordered(
      c, batch, Z, Y, X, dz, dy, dx, depth, feat, feat_dtype, ranks_depth, ranks_feat,
      vstart, out, out_padded_bf16, feat_elems, nullptr, stream);
}

}  // extern "C"
