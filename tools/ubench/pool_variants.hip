// Experimental variants of the fused channels-first pool kernel (not product
// code; winners are folded back into veon_amd/csrc/bev_pool_v2.hip).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
//        pool_variants.hip -o libpoolvar.so
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
typedef float __attribute__((ext_vector_type(4))) nt_f4;
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

// plan: tile_first[T+1] (first interval of tile), tile_point[T+1] (first point)
struct Plan {
  const int* __restrict__ tile_first;
  const int* __restrict__ tile_point;
  unsigned long long* stamps;  // diagnostic builds only: [tiles][2] wall clock
};

// ---------------------------------------------------------------------------
// v2: empty-tile fast path + interval meta and point data staged through LDS.
//   V = 64 voxels per tile, all `cs` channels of the slab.
//   PMAX = points staged per pass.
// ---------------------------------------------------------------------------
template <int PMAX, bool EMPTY_FAST, bool STAGE, int UNROLL>
__global__ __launch_bounds__(kBlock) void k_cf_v2(
    PoolArgs a, Plan plan, int c, int cs, int n_intervals, int64_t vpb,
    int64_t tiles_per_batch, float* __restrict__ out) {
  extern __shared__ float lds[];
  constexpr int V = 64;
  constexpr int LD = V + 1;
  float* tile = lds;                                       // [cs][LD]
  int* vloc = reinterpret_cast<int*>(lds + (size_t)cs * LD);  // [V]
  int* istart = vloc + V;                                  // [V] (relative to p0)
  int* ilen = istart + V;                                  // [V]
  int* s_rf = ilen + V;                                    // [PMAX]
  float* s_d = reinterpret_cast<float*>(s_rf + PMAX);      // [PMAX]
  unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(s_d + PMAX);

  const int tid = threadIdx.x;
  const int64_t t = blockIdx.x;
  const int c0 = blockIdx.y * cs;
  const int nch = (c - c0) < cs ? (c - c0) : cs;
  const int b = (int)(t / tiles_per_batch);
  const int64_t vox0 = (t - (int64_t)b * tiles_per_batch) * V;
  const int64_t remv = vpb - vox0;
  const int nvox = (int)(remv < V ? remv : V);
  const int64_t rank0 = (int64_t)b * vpb + vox0;

  const int i0 = plan.tile_first[t];
  const int cnt = plan.tile_first[t + 1] - i0;
  float* obase = out + ((int64_t)b * c + c0) * vpb + vox0;
  const int v = tid & (V - 1);
  const int w = tid >> 6;

  if (EMPTY_FAST && cnt == 0) {
    if (v < nvox)
      for (int cc = w; cc < nch; cc += kBlock / kWave) obase[(int64_t)cc * vpb + v] = 0.f;
    return;
  }
  const int p0 = plan.tile_point[t];
  const int npts = plan.tile_point[t + 1] - p0;

  // interval meta (wave 0) + occupancy mask
  if (tid < kWave) {
    unsigned long long bit = 0;
    if (tid < cnt) {
      const int st = a.interval_starts[i0 + tid];
      const int ln = a.interval_lengths[i0 + tid];
      const int k = (int)((int64_t)a.ranks_bev[st] - rank0);
      vloc[tid] = k;
      istart[tid] = st - p0;
      ilen[tid] = ln;
      bit = 1ull << k;
    }
    // wave OR-reduce of disjoint bits == sum
    for (int off = 32; off > 0; off >>= 1) bit |= __shfl_xor(bit, off);
    if (tid == 0) *s_mask = bit;
  }
  if (STAGE) {
    for (int p = tid; p < npts && p < PMAX; p += kBlock) {
      s_rf[p] = a.ranks_feat[p0 + p];
      s_d[p] = a.depth[a.ranks_depth[p0 + p]];
    }
  }
  __syncthreads();

  const int items = cnt * nch;
  for (int item = tid; item < items; item += kBlock) {
    const int j = item / nch;
    const int cc = item - j * nch;
    const int st = istart[j];
    const int len = ilen[j];
    const float* fcol = a.feat + c0 + cc;
    float acc = 0.f;
    int i = 0;
    if (STAGE) {
      for (; i + UNROLL <= len && st + i + UNROLL <= PMAX; i += UNROLL) {
        float f[UNROLL], d[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
          f[k] = fcol[(int64_t)s_rf[st + i + k] * c];
          d[k] = s_d[st + i + k];
        }
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) acc = fmaf(f[k], d[k], acc);
      }
      for (; i < len && st + i < PMAX; ++i)
        acc = fmaf(fcol[(int64_t)s_rf[st + i] * c], s_d[st + i], acc);
    }
    for (; i < len; ++i) {  // beyond the staged window (or STAGE off)
      const int rd = a.ranks_depth[p0 + st + i];
      const int rf = a.ranks_feat[p0 + st + i];
      acc = fmaf(fcol[(int64_t)rf * c], a.depth[rd], acc);
    }
    tile[cc * LD + vloc[j]] = acc;
  }
  __syncthreads();

  const bool occupied = ((*s_mask) >> v) & 1ull;
  if (v < nvox) {
    for (int cc = w; cc < nch; cc += kBlock / kWave) {
      const float val = occupied ? tile[cc * LD + v] : 0.f;
      obase[(int64_t)cc * vpb + v] = val;
    }
  }
}


// ---------------------------------------------------------------------------
// v3: plan-driven; interval groups staged through LDS (points: rf + depth
// value); gather lanes = (interval, VEC channels) with UNROLL loads in flight;
// LDS tile transposes to voxel-major rows for 256-B wave stores.
// ---------------------------------------------------------------------------
template <int VEC> struct VT;
template <> struct VT<1> { using T = float; };
template <> struct VT<4> { using T = float4; };

template <int VEC, int PMAX, int UNROLL, bool STAMP = false>
__global__ __launch_bounds__(kBlock) void k_cf_v3(
    PoolArgs a, Plan plan, int c, int cs, int n_intervals, int64_t vpb,
    int64_t tiles_per_batch, float* __restrict__ out) {
  extern __shared__ float lds[];
  constexpr int V = 64;
  constexpr int LD = V + 1;
  float* tile = lds;                                          // [cs][LD]
  int* vloc = reinterpret_cast<int*>(lds + (size_t)cs * LD);  // [V]
  int* istart = vloc + V;                                     // [V+1] rel. to p0
  int* s_rf = istart + V + 2;                                 // [PMAX]
  float* s_d = reinterpret_cast<float*>(s_rf + PMAX);         // [PMAX]
  unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(s_d + PMAX);

  const int tid = threadIdx.x;
  const int64_t t = blockIdx.x;
  const int c0 = blockIdx.y * cs;
  const int nch = (c - c0) < cs ? (c - c0) : cs;
  const int b = (int)(t / tiles_per_batch);
  const int64_t vox0 = (t - (int64_t)b * tiles_per_batch) * V;
  const int64_t remv = vpb - vox0;
  const int nvox = (int)(remv < V ? remv : V);
  const int64_t rank0 = (int64_t)b * vpb + vox0;

  if (STAMP && tid == 0) plan.stamps[(t * gridDim.y + blockIdx.y) * 8] = wall_clock64();
  const int i0 = plan.tile_first[t];
  const int cnt = plan.tile_first[t + 1] - i0;
  float* obase = out + ((int64_t)b * c + c0) * vpb + vox0;
  const int v = tid & (V - 1);
  const int w = tid >> 6;

  if (cnt == 0) {
    if (v < nvox)
      for (int cc = w; cc < nch; cc += kBlock / kWave) obase[(int64_t)cc * vpb + v] = 0.f;
    if (STAMP && tid == 0) plan.stamps[(t * gridDim.y + blockIdx.y) * 8 + 5] = wall_clock64();
    return;
  }
  const int p0 = plan.tile_point[t];
  const int npts = plan.tile_point[t + 1] - p0;

  if (tid < kWave) {
    unsigned long long bit = 0;
    if (tid < cnt) {
      const int st = a.interval_starts[i0 + tid];
      const int k = (int)((int64_t)a.ranks_bev[st] - rank0);
      vloc[tid] = k;
      istart[tid] = st - p0;
      bit = 1ull << k;
    }
    if (tid == 0) istart[cnt] = npts;
    for (int off = 32; off > 0; off >>= 1) bit |= __shfl_xor(bit, off);
    if (tid == 0) *s_mask = bit;
  }
  if (STAMP && tid == 0) plan.stamps[(t * gridDim.y + blockIdx.y) * 8 + 1] = wall_clock64() + (npts & 0);
  // first window of points can be staged before the meta is visible
  const int first_n = npts < PMAX ? npts : PMAX;
  for (int p = tid; p < first_n; p += kBlock) {
    s_rf[p] = a.ranks_feat[p0 + p];
    s_d[p] = a.depth[a.ranks_depth[p0 + p]];
  }
  __syncthreads();
  if (STAMP && tid == 0) plan.stamps[(t * gridDim.y + blockIdx.y) * 8 + 2] = wall_clock64();

  constexpr int CQ_DIV = VEC;
  const int nq = nch / CQ_DIV;  // caller guarantees nch % VEC == 0 for VEC=4
  int j0 = 0;
  int base = 0;  // first staged point (relative to p0)
  while (j0 < cnt) {
    // group [j0, j1): consecutive intervals whose points fit the staged window
    int j1 = j0;
    while (j1 < cnt && istart[j1 + 1] - base <= PMAX) ++j1;
    if (j1 == j0) {
      // a single interval longer than the window: straight from global memory
      const int st = istart[j0], len = istart[j0 + 1] - st;
      for (int q = tid; q < nq; q += kBlock) {
        using T = typename VT<VEC>::T;
        const float* fcol = a.feat + c0 + q * VEC;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        for (int i = 0; i < len; ++i) {
          const float d = a.depth[a.ranks_depth[p0 + st + i]];
          const T f = *reinterpret_cast<const T*>(fcol + (int64_t)a.ranks_feat[p0 + st + i] * c);
          const float* fp = reinterpret_cast<const float*>(&f);
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc[k] = fmaf(fp[k], d, acc[k]);
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) tile[(q * VEC + k) * LD + vloc[j0]] = acc[k];
      }
      j1 = j0 + 1;
    } else {
      const int items = (j1 - j0) * nq;
      for (int item = tid; item < items; item += kBlock) {
        using T = typename VT<VEC>::T;
        const int j = j0 + item / nq;
        const int q = item - (j - j0) * nq;
        const int st = istart[j] - base;
        const int len = istart[j + 1] - istart[j];
        const float* fcol = a.feat + c0 + q * VEC;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        int i = 0;
        for (; i + UNROLL <= len; i += UNROLL) {
          T f[UNROLL];
          float d[UNROLL];
#pragma unroll
          for (int u = 0; u < UNROLL; ++u) {
            f[u] = *reinterpret_cast<const T*>(fcol + (int64_t)s_rf[st + i + u] * c);
            d[u] = s_d[st + i + u];
          }
#pragma unroll
          for (int u = 0; u < UNROLL; ++u) {
            const float* fp = reinterpret_cast<const float*>(&f[u]);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = fmaf(fp[k], d[u], acc[k]);
          }
        }
        for (; i < len; ++i) {
          const T f = *reinterpret_cast<const T*>(fcol + (int64_t)s_rf[st + i] * c);
          const float d = s_d[st + i];
          const float* fp = reinterpret_cast<const float*>(&f);
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc[k] = fmaf(fp[k], d, acc[k]);
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) tile[(q * VEC + k) * LD + vloc[j]] = acc[k];
      }
    }
    j0 = j1;
    if (j0 < cnt) {
      // restage the next window starting at interval j0
      __syncthreads();
      base = istart[j0];
      const int n = (npts - base) < PMAX ? (npts - base) : PMAX;
      for (int p = tid; p < n; p += kBlock) {
        s_rf[p] = a.ranks_feat[p0 + base + p];
        s_d[p] = a.depth[a.ranks_depth[p0 + base + p]];
      }
      __syncthreads();
    }
  }
  __syncthreads();
  if (STAMP && tid == 0) plan.stamps[(t * gridDim.y + blockIdx.y) * 8 + 3] = wall_clock64();

  const bool occupied = ((*s_mask) >> v) & 1ull;
  if (v < nvox) {
    for (int cc = w; cc < nch; cc += kBlock / kWave) {
      const float val = occupied ? tile[cc * LD + v] : 0.f;
      obase[(int64_t)cc * vpb + v] = val;
    }
  }
  if (STAMP) {
    if (tid == 0) plan.stamps[(t * gridDim.y + blockIdx.y) * 8 + 4] = wall_clock64();
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) plan.stamps[(t * gridDim.y + blockIdx.y) * 8 + 5] = wall_clock64();
  }
}


// ---------------------------------------------------------------------------
// v4: 3 load levels (packed plan -> {starts, rf, rd, rb} -> {feat, depth}),
// compact LDS tile [cs][CAP+1] indexed by interval (column = rank of the voxel
// among the tile's occupied voxels = popcount of the occupancy mask below it),
// tiles with more than CAP intervals are processed as two 32-voxel halves.
// ---------------------------------------------------------------------------
template <int VEC, int PMAX, int UNROLL, int CAP, bool NT, int ABL = 0>
__global__ __launch_bounds__(kBlock) void k_cf_v4(
    PoolArgs a, const int4* __restrict__ plan4, int c, int cs, int n_intervals,
    int64_t vpb, int64_t tiles_per_batch, float* __restrict__ out) {
  extern __shared__ float lds[];
  constexpr int V = 64;
  constexpr int LDC = CAP + 1;
  float* tile = lds;                                           // [cs][LDC]
  int* istart = reinterpret_cast<int*>(lds + (size_t)cs * LDC);  // [V+2]
  int* s_rf = istart + V + 2;                                  // [PMAX]
  int* s_rd = s_rf + PMAX;                                     // [PMAX]
  int* s_rb = s_rd + PMAX;                                     // [PMAX]

  const int tid = threadIdx.x;
  const int64_t t = blockIdx.x;
  const int c0 = blockIdx.y * cs;
  const int nch = (c - c0) < cs ? (c - c0) : cs;
  const int b = (int)(t / tiles_per_batch);
  const int64_t vox0 = (t - (int64_t)b * tiles_per_batch) * V;
  const int64_t remv = vpb - vox0;
  const int nvox = (int)(remv < V ? remv : V);
  const int64_t rank0 = (int64_t)b * vpb + vox0;
  float* obase = out + ((int64_t)b * c + c0) * vpb + vox0;
  const int lane = tid & (kWave - 1);
  const int w = tid >> 6;

  unsigned long long* stamps = nullptr;
  if (ABL == 9) {
    stamps = *reinterpret_cast<unsigned long long* const*>(plan4 - 1);  // stashed in front of the plan
    if (tid == 0) stamps[(t * gridDim.y + blockIdx.y) * 8] = wall_clock64();
  }
  int4 pl = make_int4(0, 0, 0, 0);
  if (ABL < 4 || ABL == 9) pl = plan4[t];  // {i0, cnt, p0, npts}
  const int i0 = pl.x, p0 = pl.z, npts = pl.w;
  int cnt = pl.y;
  if (ABL == 3) { if (cnt > 100000) return; cnt = 0; }  // plan load kept live, all tiles "empty"

  if (cnt == 0) {
    if (lane < nvox)
      for (int cc = w; cc < nch; cc += kBlock / kWave) {
        if (NT) __builtin_nontemporal_store(0.f, obase + (int64_t)cc * vpb + lane);
        else obase[(int64_t)cc * vpb + lane] = 0.f;
      }
    if (ABL == 9) { __builtin_amdgcn_s_waitcnt(0); if (tid == 0) stamps[(t * gridDim.y + blockIdx.y) * 8 + 5] = wall_clock64(); }
    return;
  }
  // level 2: everything that depends only on the plan, in parallel
  if (tid < cnt) istart[tid] = a.interval_starts[i0 + tid] - p0;
  if (tid == 0) istart[cnt] = npts;
  const int first_n = npts < PMAX ? npts : PMAX;
  for (int p = tid; p < first_n; p += kBlock) {
    s_rf[p] = a.ranks_feat[p0 + p];
    s_rd[p] = a.ranks_depth[p0 + p];
    s_rb[p] = a.ranks_bev[p0 + p];
  }
  __syncthreads();

  // occupancy mask of the tile (every wave computes its own copy)
  unsigned long long mask = 0;
  {
    unsigned long long bit = 0;
    if (lane < cnt) {
      const int st = istart[lane];
      const int rb = st < PMAX ? s_rb[st] : a.ranks_bev[p0 + st];
      bit = 1ull << (int)((int64_t)rb - rank0);
    }
    for (int off = 32; off > 0; off >>= 1) bit |= __shfl_xor(bit, off);
    mask = bit;
  }
  if (ABL == 2) {
    // prologue only: keep the staged data live, then zero stores
    const int keep = s_rf[lane < first_n ? lane : 0] + s_rd[0] + (int)(mask & 1);
    if (lane < nvox)
      for (int cc = w; cc < nch; cc += kBlock / kWave)
        obase[(int64_t)cc * vpb + lane] = keep == -12345 ? 1.f : 0.f;
    return;
  }
  const int nq = nch / VEC;
  const int npass = cnt > CAP ? 2 : 1;
  const int jsplit = __popcll(mask & 0xffffffffull);  // intervals in voxels [0,32)
  int base = 0;  // first staged point
  for (int pass = 0; pass < npass; ++pass) {
    const int ja = (npass == 2 && pass == 1) ? jsplit : 0;
    const int jb = (npass == 2 && pass == 0) ? jsplit : cnt;
    int j0 = ja;
    while (j0 < jb) {
      int j1 = j0;
      while (j1 < jb && istart[j1 + 1] - base <= PMAX) ++j1;
      if (j1 == j0) {
        if (istart[j0] != base || true) {
          // interval does not fit the current window: restage from it, or if it
          // alone exceeds the window run it from global memory
          const int len = istart[j0 + 1] - istart[j0];
          if (len > PMAX) {
            const int st = istart[j0];
            for (int q = tid; q < nq; q += kBlock) {
              using T = typename VT<VEC>::T;
              const float* fcol = a.feat + c0 + q * VEC;
              float acc[VEC];
#pragma unroll
              for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
              for (int i = 0; i < len; ++i) {
                const float d = a.depth[a.ranks_depth[p0 + st + i]];
                const T f = *reinterpret_cast<const T*>(fcol + (int64_t)a.ranks_feat[p0 + st + i] * c);
                const float* fp = reinterpret_cast<const float*>(&f);
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = fmaf(fp[k], d, acc[k]);
              }
#pragma unroll
              for (int k = 0; k < VEC; ++k) tile[(q * VEC + k) * LDC + (j0 - ja)] = acc[k];
            }
            j0 = j0 + 1;
            continue;
          }
          __syncthreads();
          base = istart[j0];
          const int n = (npts - base) < PMAX ? (npts - base) : PMAX;
          for (int p = tid; p < n; p += kBlock) {
            s_rf[p] = a.ranks_feat[p0 + base + p];
            s_rd[p] = a.ranks_depth[p0 + base + p];
          }
          __syncthreads();
          continue;
        }
      }
      const int items = (j1 - j0) * nq;
      for (int item = tid; item < items; item += kBlock) {
        using T = typename VT<VEC>::T;
        const int j = j0 + item / nq;
        const int q = item - (j - j0) * nq;
        const int st = istart[j] - base;
        const int len = istart[j + 1] - istart[j];
        const float* fcol = a.feat + c0 + q * VEC;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        int i = 0;
        for (; i + UNROLL <= len; i += UNROLL) {
          T f[UNROLL];
          float d[UNROLL];
#pragma unroll
          for (int u = 0; u < UNROLL; ++u) {
            if (ABL == 1) { float* fw = reinterpret_cast<float*>(&f[u]); for (int k = 0; k < VEC; ++k) fw[k] = (float)s_rf[st + i + u]; d[u] = (float)s_rd[st + i + u]; continue; }
            f[u] = *reinterpret_cast<const T*>(fcol + (int64_t)s_rf[st + i + u] * c);
            d[u] = a.depth[s_rd[st + i + u]];
          }
#pragma unroll
          for (int u = 0; u < UNROLL; ++u) {
            const float* fp = reinterpret_cast<const float*>(&f[u]);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = fmaf(fp[k], d[u], acc[k]);
          }
        }
        for (; i < len; ++i) {
          T f; float d;
          if (ABL == 1) { float* fw = reinterpret_cast<float*>(&f); for (int k = 0; k < VEC; ++k) fw[k] = (float)s_rf[st + i]; d = (float)s_rd[st + i]; }
          else { f = *reinterpret_cast<const T*>(fcol + (int64_t)s_rf[st + i] * c); d = a.depth[s_rd[st + i]]; }
          const float* fp = reinterpret_cast<const float*>(&f);
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc[k] = fmaf(fp[k], d, acc[k]);
          continue;
          const float dd = 0.f; (void)dd;
          {
          }
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) tile[(q * VEC + k) * LDC + (j - ja)] = acc[k];
      }
      j0 = j1;
    }
    __syncthreads();
    // store this pass
    if (npass == 1) {
      const bool occupied = (mask >> lane) & 1ull;
      const int col = __popcll(mask & ((1ull << lane) - 1ull));
      if (lane < nvox)
        for (int cc = w; cc < nch; cc += kBlock / kWave) {
          const float val = occupied ? tile[cc * LDC + col] : 0.f;
          if (NT) __builtin_nontemporal_store(val, obase + (int64_t)cc * vpb + lane);
          else obase[(int64_t)cc * vpb + lane] = val;
        }
    } else {
      // half tile: 32 voxels; one wave instruction = 2 channels x 128 B
      const int v = pass * 32 + (lane & 31);
      const int sub = lane >> 5;
      const bool occupied = (mask >> v) & 1ull;
      const int col = __popcll(mask & ((1ull << v) - 1ull)) - ja;
      if (v < nvox)
        for (int cc = 2 * w + sub; cc < nch; cc += 2 * (kBlock / kWave)) {
          const float val = occupied ? tile[cc * LDC + col] : 0.f;
          if (NT) __builtin_nontemporal_store(val, obase + (int64_t)cc * vpb + v);
          else obase[(int64_t)cc * vpb + v] = val;
        }
      if (pass == 0) __syncthreads();  // tile is reused by the second half
    }
  }
  if (ABL == 9) { __builtin_amdgcn_s_waitcnt(0); __syncthreads(); if (tid == 0) stamps[(t * gridDim.y + blockIdx.y) * 8 + 5] = wall_clock64(); }
}


// ---------------------------------------------------------------------------
// v5: persistent workgroups pulling tiles from an atomic queue (next tile and
// its plan entry prefetched under the current tile's stores); otherwise v4.
// ---------------------------------------------------------------------------
template <int VEC, int PMAX, int UNROLL, int CAP, bool NT, int SB>
__global__ __launch_bounds__(kBlock) void k_cf_v5(
    PoolArgs a, const int4* __restrict__ plan4, int* __restrict__ queue, int c,
    int cs, int n_slabs, int n_intervals, int64_t vpb, int64_t tiles_per_batch,
    int64_t n_work, float* __restrict__ out) {
  extern __shared__ float lds[];
  constexpr int V = 64;
  constexpr int LDC = CAP + 1;
  float* tile = lds;                                           // [cs][LDC]
  int* istart = reinterpret_cast<int*>(lds + (size_t)cs * LDC);  // [V+2]
  int* s_rf = istart + V + 2;                                  // [PMAX]
  int* s_rd = s_rf + PMAX;                                     // [PMAX]
  int* s_rb = s_rd + PMAX;                                     // [PMAX]
  int* s_next = s_rb + PMAX;                                   // [1]

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int w = tid >> 6;

  int64_t work = blockIdx.x;  // work item = tile * n_slabs + slab
  int4 pl = make_int4(0, 0, 0, 0);
  if (work < n_work) pl = plan4[work / n_slabs];

  while (work < n_work) {
    const int64_t t = work / n_slabs;
    const int slab = (int)(work - t * n_slabs);
    const int c0 = slab * cs;
    const int nch = (c - c0) < cs ? (c - c0) : cs;
    const int b = (int)(t / tiles_per_batch);
    const int64_t vox0 = (t - (int64_t)b * tiles_per_batch) * V;
    const int64_t remv = vpb - vox0;
    const int nvox = (int)(remv < V ? remv : V);
    const int64_t rank0 = (int64_t)b * vpb + vox0;
    float* obase = out + ((int64_t)b * c + c0) * vpb + vox0;
    const int i0 = pl.x, cnt = pl.y, p0 = pl.z, npts = pl.w;

    int nxt = 0;
    if (tid == 0) nxt = queue ? atomicAdd(queue, 1) + (int)gridDim.x : (int)(work + gridDim.x);

    unsigned long long mask = 0;
    if (cnt > 0) {
      if (tid < cnt) istart[tid] = a.interval_starts[i0 + tid] - p0;
      if (tid == 0) istart[cnt] = npts;
      const int first_n = npts < PMAX ? npts : PMAX;
      for (int p = tid; p < first_n; p += kBlock) {
        s_rf[p] = a.ranks_feat[p0 + p];
        s_rd[p] = a.ranks_depth[p0 + p];
        s_rb[p] = a.ranks_bev[p0 + p];
      }
    }
    __syncthreads();  // #1
    const int nq = nch / VEC;
    const int npass = cnt > CAP ? 2 : 1;
    int jsplit = 0;
    if (cnt > 0) {
      unsigned long long bit = 0;
      if (lane < cnt) {
        const int st = istart[lane];
        const int rb = st < PMAX ? s_rb[st] : a.ranks_bev[p0 + st];
        bit = 1ull << (int)((int64_t)rb - rank0);
      }
      for (int off = 32; off > 0; off >>= 1) bit |= __shfl_xor(bit, off);
      mask = bit;
      jsplit = __popcll(mask & 0xffffffffull);
    }
    int base = 0;
    int64_t work_next = n_work;
    int4 pl_next = make_int4(0, 0, 0, 0);
    for (int pass = 0; pass < npass; ++pass) {
      const int ja = (npass == 2 && pass == 1) ? jsplit : 0;
      const int jb = (npass == 2 && pass == 0) ? jsplit : cnt;
      int j0 = ja;
      while (j0 < jb) {
        int j1 = j0;
        while (j1 < jb && istart[j1 + 1] - base <= PMAX) ++j1;
        if (j1 == j0) {
          const int len = istart[j0 + 1] - istart[j0];
          if (len > PMAX) {
            const int st = istart[j0];
            for (int q = tid; q < nq; q += kBlock) {
              using T = typename VT<VEC>::T;
              const float* fcol = a.feat + c0 + q * VEC;
              float acc[VEC];
#pragma unroll
              for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
              for (int i = 0; i < len; ++i) {
                const float d = a.depth[a.ranks_depth[p0 + st + i]];
                const T f = *reinterpret_cast<const T*>(fcol + (int64_t)a.ranks_feat[p0 + st + i] * c);
                const float* fp = reinterpret_cast<const float*>(&f);
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = fmaf(fp[k], d, acc[k]);
              }
#pragma unroll
              for (int k = 0; k < VEC; ++k) tile[(q * VEC + k) * LDC + (j0 - ja)] = acc[k];
            }
            j0 = j0 + 1;
            continue;
          }
          __syncthreads();
          base = istart[j0];
          const int n = (npts - base) < PMAX ? (npts - base) : PMAX;
          for (int p = tid; p < n; p += kBlock) {
            s_rf[p] = a.ranks_feat[p0 + base + p];
            s_rd[p] = a.ranks_depth[p0 + base + p];
          }
          __syncthreads();
          continue;
        }
        const int items = (j1 - j0) * nq;
        for (int item = tid; item < items; item += kBlock) {
          using T = typename VT<VEC>::T;
          const int j = j0 + item / nq;
          const int q = item - (j - j0) * nq;
          const int st = istart[j] - base;
          const int len = istart[j + 1] - istart[j];
          const float* fcol = a.feat + c0 + q * VEC;
          float acc[VEC];
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
          int i = 0;
          for (; i + UNROLL <= len; i += UNROLL) {
            T f[UNROLL];
            float d[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
              f[u] = *reinterpret_cast<const T*>(fcol + (int64_t)s_rf[st + i + u] * c);
              d[u] = a.depth[s_rd[st + i + u]];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
              const float* fp = reinterpret_cast<const float*>(&f[u]);
#pragma unroll
              for (int k = 0; k < VEC; ++k) acc[k] = fmaf(fp[k], d[u], acc[k]);
            }
          }
          for (; i < len; ++i) {
            const T f = *reinterpret_cast<const T*>(fcol + (int64_t)s_rf[st + i] * c);
            const float d = a.depth[s_rd[st + i]];
            const float* fp = reinterpret_cast<const float*>(&f);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = fmaf(fp[k], d, acc[k]);
          }
#pragma unroll
          for (int k = 0; k < VEC; ++k) tile[(q * VEC + k) * LDC + (j - ja)] = acc[k];
        }
        j0 = j1;
      }
      if (pass == npass - 1 && tid == 0) *s_next = nxt;
      __syncthreads();  // #2 (per pass)
      if (pass == npass - 1) {
        // next work item and its plan entry, fetched under this tile's stores
        work_next = *s_next;
        if (work_next < n_work) pl_next = plan4[work_next / n_slabs];
      }
      if (npass == 1) {
        const bool occupied = (mask >> lane) & 1ull;
        const int col = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane < nvox) {
          int cc = w;
          for (; cc + (SB - 1) * 4 < nch; cc += SB * 4) {
            float vals[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) vals[u] = occupied ? tile[(cc + 4 * u) * LDC + col] : 0.f;
#pragma unroll
            for (int u = 0; u < SB; ++u) {
              if (NT) __builtin_nontemporal_store(vals[u], obase + (int64_t)(cc + 4 * u) * vpb + lane);
              else obase[(int64_t)(cc + 4 * u) * vpb + lane] = vals[u];
            }
          }
          for (; cc < nch; cc += 4) {
            const float val = occupied ? tile[cc * LDC + col] : 0.f;
            if (NT) __builtin_nontemporal_store(val, obase + (int64_t)cc * vpb + lane);
            else obase[(int64_t)cc * vpb + lane] = val;
          }
        }
      } else {
        const int v = pass * 32 + (lane & 31);
        const int sub = lane >> 5;
        const bool occupied = (mask >> v) & 1ull;
        const int col = __popcll(mask & ((1ull << v) - 1ull)) - ja;
        if (v < nvox)
          for (int cc = 2 * w + sub; cc < nch; cc += 8) {
            const float val = occupied ? tile[cc * LDC + col] : 0.f;
            if (NT) __builtin_nontemporal_store(val, obase + (int64_t)cc * vpb + v);
            else obase[(int64_t)cc * vpb + v] = val;
          }
        if (pass == 0) __syncthreads();
      }
    }
    work = work_next;
    pl = pl_next;
  }
}


// ---------------------------------------------------------------------------
// v6 = v4 with: no s_rb staging (8 B/point), batched LDS reads before the
// stores (SB), long intervals (len >= LONG) on channel-per-lane chains with
// ULONG loads in flight.
// ---------------------------------------------------------------------------
template <int VEC, int PMAX, int UNROLL, int CAP, bool NT, int SB, int LONG, int ULONG, int MINW = 8>
__global__ __launch_bounds__(kBlock, MINW) void k_cf_v6(
    PoolArgs a, const int4* __restrict__ plan4, int c, int cs, int n_intervals,
    int64_t vpb, int64_t tiles_per_batch, float* __restrict__ out) {
  extern __shared__ float lds[];
  constexpr int V = 64;
  constexpr int LDC = CAP + 1;
  float* tile = lds;                                           // [cs][LDC]
  int* istart = reinterpret_cast<int*>(lds + (size_t)cs * LDC);  // [V+2]
  int* ivox = istart + V + 2;                                  // [V]
  int* s_rf = ivox + V;                                        // [PMAX]
  int* s_rd = s_rf + PMAX;                                     // [PMAX]

  const int tid = threadIdx.x;
  const int64_t t = blockIdx.x;
  const int c0 = blockIdx.y * cs;
  const int nch = (c - c0) < cs ? (c - c0) : cs;
  const int b = (int)(t / tiles_per_batch);
  const int64_t vox0 = (t - (int64_t)b * tiles_per_batch) * V;
  const int64_t remv = vpb - vox0;
  const int nvox = (int)(remv < V ? remv : V);
  const int64_t rank0 = (int64_t)b * vpb + vox0;
  float* obase = out + ((int64_t)b * c + c0) * vpb + vox0;
  const int lane = tid & (kWave - 1);
  const int w = tid >> 6;

  const int4 pl = plan4[t];  // {i0, cnt, p0, npts}
  const int i0 = pl.x, cnt = pl.y, p0 = pl.z, npts = pl.w;

  if (cnt == 0) {
    if (lane < nvox)
      for (int cc = w; cc < nch; cc += kBlock / kWave) {
        if (NT) __builtin_nontemporal_store(0.f, obase + (int64_t)cc * vpb + lane);
        else obase[(int64_t)cc * vpb + lane] = 0.f;
      }
    return;
  }
  if (tid < cnt) {
    const int st = a.interval_starts[i0 + tid];
    istart[tid] = st - p0;
    ivox[tid] = (int)((int64_t)a.ranks_bev[st] - rank0);
  }
  if (tid == 0) istart[cnt] = npts;
  const int first_n = npts < PMAX ? npts : PMAX;
  for (int p = tid; p < first_n; p += kBlock) {
    s_rf[p] = a.ranks_feat[p0 + p];
    s_rd[p] = a.ranks_depth[p0 + p];
  }
  __syncthreads();

  unsigned long long mask = 0, lmask = 0;
  {
    unsigned long long bit = 0;
    bool is_long = false;
    if (lane < cnt) {
      bit = 1ull << ivox[lane];
      is_long = (istart[lane + 1] - istart[lane]) >= LONG;
    }
    lmask = __ballot(is_long);
    for (int off = 32; off > 0; off >>= 1) bit |= __shfl_xor(bit, off);
    mask = bit;
  }
  const int nq = nch / VEC;
  const int npass = cnt > CAP ? 2 : 1;
  const int jsplit = __popcll(mask & 0xffffffffull);
  int base = 0;
  for (int pass = 0; pass < npass; ++pass) {
    const int ja = (npass == 2 && pass == 1) ? jsplit : 0;
    const int jb = (npass == 2 && pass == 0) ? jsplit : cnt;
    int j0 = ja;
    while (j0 < jb) {
      int j1 = j0;
      while (j1 < jb && istart[j1 + 1] - base <= PMAX) ++j1;
      if (j1 == j0) {
        const int len = istart[j0 + 1] - istart[j0];
        if (len > PMAX) {
          const int st = istart[j0];
          for (int q = tid; q < nq; q += kBlock) {
            using T = typename VT<VEC>::T;
            const float* fcol = a.feat + c0 + q * VEC;
            float acc[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
            for (int i = 0; i < len; ++i) {
              const float d = a.depth[a.ranks_depth[p0 + st + i]];
              const T f = *reinterpret_cast<const T*>(fcol + (int64_t)a.ranks_feat[p0 + st + i] * c);
              const float* fp = reinterpret_cast<const float*>(&f);
#pragma unroll
              for (int k = 0; k < VEC; ++k) acc[k] = fmaf(fp[k], d, acc[k]);
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) tile[(q * VEC + k) * LDC + (j0 - ja)] = acc[k];
          }
          j0 = j0 + 1;
          continue;
        }
        __syncthreads();
        base = istart[j0];
        const int n = (npts - base) < PMAX ? (npts - base) : PMAX;
        for (int p = tid; p < n; p += kBlock) {
          s_rf[p] = a.ranks_feat[p0 + base + p];
          s_rd[p] = a.ranks_depth[p0 + base + p];
        }
        __syncthreads();
        continue;
      }
      // short intervals: lanes = (interval, VEC channels)
      const int items = (j1 - j0) * nq;
      for (int item = tid; item < items; item += kBlock) {
        using T = typename VT<VEC>::T;
        const int j = j0 + item / nq;
        const int q = item - (j - j0) * nq;
        const int st = istart[j] - base;
        const int len = istart[j + 1] - istart[j];
        if (len >= LONG) continue;
        const float* fcol = a.feat + c0 + q * VEC;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        int i = 0;
        for (; i + UNROLL <= len; i += UNROLL) {
          T f[UNROLL];
          float d[UNROLL];
#pragma unroll
          for (int u = 0; u < UNROLL; ++u) {
            f[u] = *reinterpret_cast<const T*>(fcol + (int64_t)s_rf[st + i + u] * c);
            d[u] = a.depth[s_rd[st + i + u]];
          }
#pragma unroll
          for (int u = 0; u < UNROLL; ++u) {
            const float* fp = reinterpret_cast<const float*>(&f[u]);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = fmaf(fp[k], d[u], acc[k]);
          }
        }
        for (; i < len; ++i) {
          const T f = *reinterpret_cast<const T*>(fcol + (int64_t)s_rf[st + i] * c);
          const float d = a.depth[s_rd[st + i]];
          const float* fp = reinterpret_cast<const float*>(&f);
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc[k] = fmaf(fp[k], d, acc[k]);
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) tile[(q * VEC + k) * LDC + (j - ja)] = acc[k];
      }
      // long intervals of this group: lanes = channels, ULONG loads in flight
      {
        unsigned long long lm = lmask;
        if (j0 > 0) lm &= ~((1ull << j0) - 1ull);
        if (j1 < 64) lm &= ((1ull << j1) - 1ull);
        const int n_long = __popcll(lm);
        const int slots = kBlock / nch > 0 ? kBlock / nch : 1;
        const int slot = tid / nch;
        const int ch = tid - slot * nch;
        if (n_long > 0 && slot < slots) {
          // advance to this slot's first long interval
          unsigned long long rem = lm;
          for (int k = 0; k < slot && rem; ++k) rem &= rem - 1;
          while (rem) {
            const int j = __ffsll((long long)rem) - 1;
            const int st = istart[j] - base;
            const int len = istart[j + 1] - istart[j];
            const float* fcol = a.feat + c0 + ch;
            float acc = 0.f;
            int i = 0;
            for (; i + ULONG <= len; i += ULONG) {
              float f[ULONG], d[ULONG];
#pragma unroll
              for (int u = 0; u < ULONG; ++u) {
                f[u] = fcol[(int64_t)s_rf[st + i + u] * c];
                d[u] = a.depth[s_rd[st + i + u]];
              }
#pragma unroll
              for (int u = 0; u < ULONG; ++u) acc = fmaf(f[u], d[u], acc);
            }
            for (; i < len; ++i) acc = fmaf(fcol[(int64_t)s_rf[st + i] * c], a.depth[s_rd[st + i]], acc);
            tile[ch * LDC + (j - ja)] = acc;
            for (int k = 0; k < slots && rem; ++k) rem &= rem - 1;
          }
        }
      }
      j0 = j1;
    }
    __syncthreads();
    if (npass == 1) {
      const bool occupied = (mask >> lane) & 1ull;
      const int col = __popcll(mask & ((1ull << lane) - 1ull));
      if (lane < nvox) {
        int cc = w;
        for (; cc + (SB - 1) * 4 < nch; cc += SB * 4) {
          float vals[SB];
#pragma unroll
          for (int u = 0; u < SB; ++u) vals[u] = occupied ? tile[(cc + 4 * u) * LDC + col] : 0.f;
#pragma unroll
          for (int u = 0; u < SB; ++u) {
            if (NT) __builtin_nontemporal_store(vals[u], obase + (int64_t)(cc + 4 * u) * vpb + lane);
            else obase[(int64_t)(cc + 4 * u) * vpb + lane] = vals[u];
          }
        }
        for (; cc < nch; cc += 4) {
          const float val = occupied ? tile[cc * LDC + col] : 0.f;
          if (NT) __builtin_nontemporal_store(val, obase + (int64_t)cc * vpb + lane);
          else obase[(int64_t)cc * vpb + lane] = val;
        }
      }
    } else {
      const int v = pass * 32 + (lane & 31);
      const int sub = lane >> 5;
      const bool occupied = (mask >> v) & 1ull;
      const int col = __popcll(mask & ((1ull << v) - 1ull)) - ja;
      if (v < nvox)
        for (int cc = 2 * w + sub; cc < nch; cc += 8) {
          const float val = occupied ? tile[cc * LDC + col] : 0.f;
          if (NT) __builtin_nontemporal_store(val, obase + (int64_t)cc * vpb + v);
          else obase[(int64_t)cc * vpb + v] = val;
        }
      if (pass == 0) __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------
// v7 = v6 with BLK threads per tile;  v6 = v4 with: no s_rb staging (8 B/point), batched LDS reads before the
// stores (SB), long intervals (len >= LONG) on channel-per-lane chains with
// ULONG loads in flight.
// ---------------------------------------------------------------------------
template <int BLK, int VEC, int PMAX, int UNROLL, int CAP, bool NT, int SB, int LONG, int ULONG, int MINW>
__global__ __launch_bounds__(BLK, MINW) void k_cf_v7(
    PoolArgs a, const int4* __restrict__ plan4, int c, int cs, int n_intervals,
    int64_t vpb, int64_t tiles_per_batch, float* __restrict__ out) {
  extern __shared__ float lds[];
  constexpr int V = 64;
  constexpr int LDC = CAP + 1;
  float* tile = lds;                                           // [cs][LDC]
  int* istart = reinterpret_cast<int*>(lds + (size_t)cs * LDC);  // [V+2]
  int* ivox = istart + V + 2;                                  // [V]
  int* s_rf = ivox + V;                                        // [PMAX]
  int* s_rd = s_rf + PMAX;                                     // [PMAX]

  const int tid = threadIdx.x;
  const int64_t t = blockIdx.x;
  const int c0 = blockIdx.y * cs;
  const int nch = (c - c0) < cs ? (c - c0) : cs;
  const int b = (int)(t / tiles_per_batch);
  const int64_t vox0 = (t - (int64_t)b * tiles_per_batch) * V;
  const int64_t remv = vpb - vox0;
  const int nvox = (int)(remv < V ? remv : V);
  const int64_t rank0 = (int64_t)b * vpb + vox0;
  float* obase = out + ((int64_t)b * c + c0) * vpb + vox0;
  const int lane = tid & (kWave - 1);
  const int w = tid >> 6;

  const int4 pl = plan4[t];  // {i0, cnt, p0, npts}
  const int i0 = pl.x, cnt = pl.y, p0 = pl.z, npts = pl.w;

  if (cnt == 0) {
    if (lane < nvox)
      for (int cc = w; cc < nch; cc += (BLK / kWave)) {
        if (NT) __builtin_nontemporal_store(0.f, obase + (int64_t)cc * vpb + lane);
        else obase[(int64_t)cc * vpb + lane] = 0.f;
      }
    return;
  }
  if (tid < cnt) {
    const int st = a.interval_starts[i0 + tid];
    istart[tid] = st - p0;
    ivox[tid] = (int)((int64_t)a.ranks_bev[st] - rank0);
  }
  if (tid == 0) istart[cnt] = npts;
  const int first_n = npts < PMAX ? npts : PMAX;
  for (int p = tid; p < first_n; p += BLK) {
    s_rf[p] = a.ranks_feat[p0 + p];
    s_rd[p] = a.ranks_depth[p0 + p];
  }
  __syncthreads();

  unsigned long long mask = 0, lmask = 0;
  {
    unsigned long long bit = 0;
    bool is_long = false;
    if (lane < cnt) {
      bit = 1ull << ivox[lane];
      is_long = (istart[lane + 1] - istart[lane]) >= LONG;
    }
    lmask = __ballot(is_long);
    for (int off = 32; off > 0; off >>= 1) bit |= __shfl_xor(bit, off);
    mask = bit;
  }
  const int nq = nch / VEC;
  const int npass = cnt > CAP ? 2 : 1;
  const int jsplit = __popcll(mask & 0xffffffffull);
  int base = 0;
  for (int pass = 0; pass < npass; ++pass) {
    const int ja = (npass == 2 && pass == 1) ? jsplit : 0;
    const int jb = (npass == 2 && pass == 0) ? jsplit : cnt;
    int j0 = ja;
    while (j0 < jb) {
      int j1 = j0;
      while (j1 < jb && istart[j1 + 1] - base <= PMAX) ++j1;
      if (j1 == j0) {
        const int len = istart[j0 + 1] - istart[j0];
        if (len > PMAX) {
          const int st = istart[j0];
          for (int q = tid; q < nq; q += BLK) {
            using T = typename VT<VEC>::T;
            const float* fcol = a.feat + c0 + q * VEC;
            float acc[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
            for (int i = 0; i < len; ++i) {
              const float d = a.depth[a.ranks_depth[p0 + st + i]];
              const T f = *reinterpret_cast<const T*>(fcol + (int64_t)a.ranks_feat[p0 + st + i] * c);
              const float* fp = reinterpret_cast<const float*>(&f);
#pragma unroll
              for (int k = 0; k < VEC; ++k) acc[k] = fmaf(fp[k], d, acc[k]);
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) tile[(q * VEC + k) * LDC + (j0 - ja)] = acc[k];
          }
          j0 = j0 + 1;
          continue;
        }
        __syncthreads();
        base = istart[j0];
        const int n = (npts - base) < PMAX ? (npts - base) : PMAX;
        for (int p = tid; p < n; p += BLK) {
          s_rf[p] = a.ranks_feat[p0 + base + p];
          s_rd[p] = a.ranks_depth[p0 + base + p];
        }
        __syncthreads();
        continue;
      }
      // short intervals: lanes = (interval, VEC channels)
      const int items = (j1 - j0) * nq;
      for (int item = tid; item < items; item += BLK) {
        using T = typename VT<VEC>::T;
        const int j = j0 + item / nq;
        const int q = item - (j - j0) * nq;
        const int st = istart[j] - base;
        const int len = istart[j + 1] - istart[j];
        if (len >= LONG) continue;
        const float* fcol = a.feat + c0 + q * VEC;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        int i = 0;
        for (; i + UNROLL <= len; i += UNROLL) {
          T f[UNROLL];
          float d[UNROLL];
#pragma unroll
          for (int u = 0; u < UNROLL; ++u) {
            f[u] = *reinterpret_cast<const T*>(fcol + (int64_t)s_rf[st + i + u] * c);
            d[u] = a.depth[s_rd[st + i + u]];
          }
#pragma unroll
          for (int u = 0; u < UNROLL; ++u) {
            const float* fp = reinterpret_cast<const float*>(&f[u]);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = fmaf(fp[k], d[u], acc[k]);
          }
        }
        for (; i < len; ++i) {
          const T f = *reinterpret_cast<const T*>(fcol + (int64_t)s_rf[st + i] * c);
          const float d = a.depth[s_rd[st + i]];
          const float* fp = reinterpret_cast<const float*>(&f);
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc[k] = fmaf(fp[k], d, acc[k]);
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) tile[(q * VEC + k) * LDC + (j - ja)] = acc[k];
      }
      // long intervals of this group: lanes = channels, ULONG loads in flight
      {
        unsigned long long lm = lmask;
        if (j0 > 0) lm &= ~((1ull << j0) - 1ull);
        if (j1 < 64) lm &= ((1ull << j1) - 1ull);
        const int n_long = __popcll(lm);
        const int slots = BLK / nch > 0 ? BLK / nch : 1;
        const int slot = tid / nch;
        const int ch = tid - slot * nch;
        if (n_long > 0 && slot < slots) {
          // advance to this slot's first long interval
          unsigned long long rem = lm;
          for (int k = 0; k < slot && rem; ++k) rem &= rem - 1;
          while (rem) {
            const int j = __ffsll((long long)rem) - 1;
            const int st = istart[j] - base;
            const int len = istart[j + 1] - istart[j];
            const float* fcol = a.feat + c0 + ch;
            float acc = 0.f;
            int i = 0;
            for (; i + ULONG <= len; i += ULONG) {
              float f[ULONG], d[ULONG];
#pragma unroll
              for (int u = 0; u < ULONG; ++u) {
                f[u] = fcol[(int64_t)s_rf[st + i + u] * c];
                d[u] = a.depth[s_rd[st + i + u]];
              }
#pragma unroll
              for (int u = 0; u < ULONG; ++u) acc = fmaf(f[u], d[u], acc);
            }
            for (; i < len; ++i) acc = fmaf(fcol[(int64_t)s_rf[st + i] * c], a.depth[s_rd[st + i]], acc);
            tile[ch * LDC + (j - ja)] = acc;
            for (int k = 0; k < slots && rem; ++k) rem &= rem - 1;
          }
        }
      }
      j0 = j1;
    }
    __syncthreads();
    if (npass == 1) {
      const bool occupied = (mask >> lane) & 1ull;
      const int col = __popcll(mask & ((1ull << lane) - 1ull));
      if (lane < nvox) {
        int cc = w;
        for (; cc + (SB - 1) * (BLK / kWave) < nch; cc += SB * (BLK / kWave)) {
          float vals[SB];
#pragma unroll
          for (int u = 0; u < SB; ++u) vals[u] = occupied ? tile[(cc + (BLK / kWave) * u) * LDC + col] : 0.f;
#pragma unroll
          for (int u = 0; u < SB; ++u) {
            if (NT) __builtin_nontemporal_store(vals[u], obase + (int64_t)(cc + (BLK / kWave) * u) * vpb + lane);
            else obase[(int64_t)(cc + (BLK / kWave) * u) * vpb + lane] = vals[u];
          }
        }
        for (; cc < nch; cc += (BLK / kWave)) {
          const float val = occupied ? tile[cc * LDC + col] : 0.f;
          if (NT) __builtin_nontemporal_store(val, obase + (int64_t)cc * vpb + lane);
          else obase[(int64_t)cc * vpb + lane] = val;
        }
      }
    } else {
      const int v = pass * 32 + (lane & 31);
      const int sub = lane >> 5;
      const bool occupied = (mask >> v) & 1ull;
      const int col = __popcll(mask & ((1ull << v) - 1ull)) - ja;
      if (v < nvox)
        for (int cc = 2 * w + sub; cc < nch; cc += 2 * (BLK / kWave)) {
          const float val = occupied ? tile[cc * LDC + col] : 0.f;
          if (NT) __builtin_nontemporal_store(val, obase + (int64_t)cc * vpb + v);
          else obase[(int64_t)cc * vpb + v] = val;
        }
      if (pass == 0) __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------
// v8 = v7 with tile order taken from the plan (heavy first); v7 = v6 with BLK threads per tile;  v6 = v4 with: no s_rb staging (8 B/point), batched LDS reads before the
// stores (SB), long intervals (len >= LONG) on channel-per-lane chains with
// ULONG loads in flight.
// ---------------------------------------------------------------------------
template <int BLK, int VEC, int PMAX, int UNROLL, int CAP, bool NT, int SB, int LONG, int ULONG, int MINW>
__global__ __launch_bounds__(BLK, MINW) void k_cf_v8(
    PoolArgs a, const int4* __restrict__ plan4, int c, int cs, int n_intervals,
    int64_t vpb, int64_t tiles_per_batch, float* __restrict__ out) {
  extern __shared__ float lds[];
  constexpr int V = 64;
  constexpr int LDC = CAP + 1;
  float* tile = lds;                                           // [cs][LDC]
  int* istart = reinterpret_cast<int*>(lds + (size_t)cs * LDC);  // [V+2]
  int* ivox = istart + V + 2;                                  // [V]
  int* s_rf = ivox + V;                                        // [PMAX]
  int* s_rd = s_rf + PMAX;                                     // [PMAX]

  const int tid = threadIdx.x;
  const int4 pl = plan4[blockIdx.x];  // {tile, i0, p0, cnt << 24 | npts}
  const int64_t t = pl.x;
  const int c0 = blockIdx.y * cs;
  const int nch = (c - c0) < cs ? (c - c0) : cs;
  const int b = (int)(t / tiles_per_batch);
  const int64_t vox0 = (t - (int64_t)b * tiles_per_batch) * V;
  const int64_t remv = vpb - vox0;
  const int nvox = (int)(remv < V ? remv : V);
  const int64_t rank0 = (int64_t)b * vpb + vox0;
  float* obase = out + ((int64_t)b * c + c0) * vpb + vox0;
  const int lane = tid & (kWave - 1);
  const int w = tid >> 6;

  const int i0 = pl.y, cnt = (int)((unsigned)pl.w >> 24), p0 = pl.z, npts = pl.w & 0xffffff;

  if (cnt == 0) {
    if (lane < nvox)
      for (int cc = w; cc < nch; cc += (BLK / kWave)) {
        if (NT) __builtin_nontemporal_store(0.f, obase + (int64_t)cc * vpb + lane);
        else obase[(int64_t)cc * vpb + lane] = 0.f;
      }
    return;
  }
  if (tid < cnt) {
    const int st = a.interval_starts[i0 + tid];
    istart[tid] = st - p0;
    ivox[tid] = (int)((int64_t)a.ranks_bev[st] - rank0);
  }
  if (tid == 0) istart[cnt] = npts;
  const int first_n = npts < PMAX ? npts : PMAX;
  for (int p = tid; p < first_n; p += BLK) {
    s_rf[p] = a.ranks_feat[p0 + p];
    s_rd[p] = a.ranks_depth[p0 + p];
  }
  __syncthreads();

  unsigned long long mask = 0, lmask = 0;
  {
    unsigned long long bit = 0;
    bool is_long = false;
    if (lane < cnt) {
      bit = 1ull << ivox[lane];
      is_long = (istart[lane + 1] - istart[lane]) >= LONG;
    }
    lmask = __ballot(is_long);
    for (int off = 32; off > 0; off >>= 1) bit |= __shfl_xor(bit, off);
    mask = bit;
  }
  const int nq = nch / VEC;
  const int npass = cnt > CAP ? 2 : 1;
  const int jsplit = __popcll(mask & 0xffffffffull);
  int base = 0;
  for (int pass = 0; pass < npass; ++pass) {
    const int ja = (npass == 2 && pass == 1) ? jsplit : 0;
    const int jb = (npass == 2 && pass == 0) ? jsplit : cnt;
    int j0 = ja;
    while (j0 < jb) {
      int j1 = j0;
      while (j1 < jb && istart[j1 + 1] - base <= PMAX) ++j1;
      if (j1 == j0) {
        const int len = istart[j0 + 1] - istart[j0];
        if (len > PMAX) {
          const int st = istart[j0];
          for (int q = tid; q < nq; q += BLK) {
            using T = typename VT<VEC>::T;
            const float* fcol = a.feat + c0 + q * VEC;
            float acc[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
            for (int i = 0; i < len; ++i) {
              const float d = a.depth[a.ranks_depth[p0 + st + i]];
              const T f = *reinterpret_cast<const T*>(fcol + (int64_t)a.ranks_feat[p0 + st + i] * c);
              const float* fp = reinterpret_cast<const float*>(&f);
#pragma unroll
              for (int k = 0; k < VEC; ++k) acc[k] = fmaf(fp[k], d, acc[k]);
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) tile[(q * VEC + k) * LDC + (j0 - ja)] = acc[k];
          }
          j0 = j0 + 1;
          continue;
        }
        __syncthreads();
        base = istart[j0];
        const int n = (npts - base) < PMAX ? (npts - base) : PMAX;
        for (int p = tid; p < n; p += BLK) {
          s_rf[p] = a.ranks_feat[p0 + base + p];
          s_rd[p] = a.ranks_depth[p0 + base + p];
        }
        __syncthreads();
        continue;
      }
      // short intervals: lanes = (interval, VEC channels)
      const int items = (j1 - j0) * nq;
      for (int item = tid; item < items; item += BLK) {
        using T = typename VT<VEC>::T;
        const int j = j0 + item / nq;
        const int q = item - (j - j0) * nq;
        const int st = istart[j] - base;
        const int len = istart[j + 1] - istart[j];
        if (len >= LONG) continue;
        const float* fcol = a.feat + c0 + q * VEC;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        int i = 0;
        for (; i + UNROLL <= len; i += UNROLL) {
          T f[UNROLL];
          float d[UNROLL];
#pragma unroll
          for (int u = 0; u < UNROLL; ++u) {
            f[u] = *reinterpret_cast<const T*>(fcol + (int64_t)s_rf[st + i + u] * c);
            d[u] = a.depth[s_rd[st + i + u]];
          }
#pragma unroll
          for (int u = 0; u < UNROLL; ++u) {
            const float* fp = reinterpret_cast<const float*>(&f[u]);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = fmaf(fp[k], d[u], acc[k]);
          }
        }
        for (; i < len; ++i) {
          const T f = *reinterpret_cast<const T*>(fcol + (int64_t)s_rf[st + i] * c);
          const float d = a.depth[s_rd[st + i]];
          const float* fp = reinterpret_cast<const float*>(&f);
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc[k] = fmaf(fp[k], d, acc[k]);
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) tile[(q * VEC + k) * LDC + (j - ja)] = acc[k];
      }
      // long intervals of this group: lanes = channels, ULONG loads in flight
      {
        unsigned long long lm = lmask;
        if (j0 > 0) lm &= ~((1ull << j0) - 1ull);
        if (j1 < 64) lm &= ((1ull << j1) - 1ull);
        const int n_long = __popcll(lm);
        const int slots = BLK / nch > 0 ? BLK / nch : 1;
        const int slot = tid / nch;
        const int ch = tid - slot * nch;
        if (n_long > 0 && slot < slots) {
          // advance to this slot's first long interval
          unsigned long long rem = lm;
          for (int k = 0; k < slot && rem; ++k) rem &= rem - 1;
          while (rem) {
            const int j = __ffsll((long long)rem) - 1;
            const int st = istart[j] - base;
            const int len = istart[j + 1] - istart[j];
            const float* fcol = a.feat + c0 + ch;
            float acc = 0.f;
            int i = 0;
            for (; i + ULONG <= len; i += ULONG) {
              float f[ULONG], d[ULONG];
#pragma unroll
              for (int u = 0; u < ULONG; ++u) {
                f[u] = fcol[(int64_t)s_rf[st + i + u] * c];
                d[u] = a.depth[s_rd[st + i + u]];
              }
#pragma unroll
              for (int u = 0; u < ULONG; ++u) acc = fmaf(f[u], d[u], acc);
            }
            for (; i < len; ++i) acc = fmaf(fcol[(int64_t)s_rf[st + i] * c], a.depth[s_rd[st + i]], acc);
            tile[ch * LDC + (j - ja)] = acc;
            for (int k = 0; k < slots && rem; ++k) rem &= rem - 1;
          }
        }
      }
      j0 = j1;
    }
    __syncthreads();
    if (npass == 1) {
      const bool occupied = (mask >> lane) & 1ull;
      const int col = __popcll(mask & ((1ull << lane) - 1ull));
      if (lane < nvox) {
        int cc = w;
        for (; cc + (SB - 1) * (BLK / kWave) < nch; cc += SB * (BLK / kWave)) {
          float vals[SB];
#pragma unroll
          for (int u = 0; u < SB; ++u) vals[u] = occupied ? tile[(cc + (BLK / kWave) * u) * LDC + col] : 0.f;
#pragma unroll
          for (int u = 0; u < SB; ++u) {
            if (NT) __builtin_nontemporal_store(vals[u], obase + (int64_t)(cc + (BLK / kWave) * u) * vpb + lane);
            else obase[(int64_t)(cc + (BLK / kWave) * u) * vpb + lane] = vals[u];
          }
        }
        for (; cc < nch; cc += (BLK / kWave)) {
          const float val = occupied ? tile[cc * LDC + col] : 0.f;
          if (NT) __builtin_nontemporal_store(val, obase + (int64_t)cc * vpb + lane);
          else obase[(int64_t)cc * vpb + lane] = val;
        }
      }
    } else {
      const int v = pass * 32 + (lane & 31);
      const int sub = lane >> 5;
      const bool occupied = (mask >> v) & 1ull;
      const int col = __popcll(mask & ((1ull << v) - 1ull)) - ja;
      if (v < nvox)
        for (int cc = 2 * w + sub; cc < nch; cc += 2 * (BLK / kWave)) {
          const float val = occupied ? tile[cc * LDC + col] : 0.f;
          if (NT) __builtin_nontemporal_store(val, obase + (int64_t)cc * vpb + v);
          else obase[(int64_t)cc * vpb + v] = val;
        }
      if (pass == 0) __syncthreads();
    }
  }
}


// ---------------------------------------------------------------------------
// v9: product kernel structure, but every wave owns a private block of
// channels end to end (gather -> LDS -> stores), so there is no workgroup
// barrier between gather and store; only the staging barrier remains.
// ---------------------------------------------------------------------------
template <int CAP, int UNROLL, int SB>
__global__ __launch_bounds__(256) void k_cf_v9(
    PoolArgs a, const int4* __restrict__ plan, int c, int cs, int64_t vpb,
    int64_t tiles_per_batch, float* __restrict__ out) {
  extern __shared__ float lds[];
  constexpr int V = 64, LDC = CAP + 1, PM = 1024, NW = 4;
  float* tile = lds;
  int* istart = reinterpret_cast<int*>(lds + (size_t)cs * LDC);
  int* ivox = istart + V + 2;
  int* s_rf = ivox + V;
  int* s_rd = s_rf + PM;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t t = blockIdx.x;
  const int c0 = blockIdx.y * cs;
  const int nch = (c - c0) < cs ? (c - c0) : cs;
  const int b = (int)(t / tiles_per_batch);
  const int64_t vox0 = (t - (int64_t)b * tiles_per_batch) * V;
  const int64_t remv = vpb - vox0;
  const int nvox = (int)(remv < V ? remv : V);
  const int64_t rank0 = (int64_t)b * vpb + vox0;
  float* obase = out + ((int64_t)b * c + c0) * vpb + vox0;
  const int4 pl = plan[t];
  const int i0 = pl.x, cnt = pl.y, p0 = pl.z, npts = pl.w;
  if (cnt == 0) {
    if (lane < nvox)
      for (int cc = w; cc < nch; cc += NW)
        __builtin_nontemporal_store(0.f, obase + (int64_t)cc * vpb + lane);
    return;
  }
  if (tid < cnt) {
    const int st = a.interval_starts[i0 + tid];
    istart[tid] = st - p0;
    ivox[tid] = (int)((int64_t)a.ranks_bev[st] - rank0);
  }
  if (tid == 0) istart[cnt] = npts;
  {
    const int n = npts < PM ? npts : PM;
    for (int p = tid; p < n; p += 256) {
      s_rf[p] = a.ranks_feat[p0 + p];
      s_rd[p] = a.ranks_depth[p0 + p];
    }
  }
  __syncthreads();
  unsigned long long bit = lane < cnt ? (1ull << ivox[lane]) : 0ull;
  for (int off = 32; off > 0; off >>= 1) bit |= __shfl_xor(bit, off);
  const unsigned long long mask = bit;
  // this wave's channel block: quads [qlo, qhi)
  const int nq = nch / 4;
  const int qlo = (nq * w) / NW, qhi = (nq * (w + 1)) / NW;
  const int nqw = qhi - qlo;
  // (tiles whose points exceed the staged window or cnt > CAP are not handled
  //  by this experimental variant: host must guarantee npts <= PM, cnt <= CAP)
  const int items = cnt * nqw;
  for (int item = lane; item < items; item += 64) {
    const int j = item / nqw;
    const int q = qlo + (item - j * nqw);
    const int st = istart[j];
    const int len = istart[j + 1] - st;
    const float* fcol = a.feat + c0 + q * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int i = 0;
    for (; i + UNROLL <= len; i += UNROLL) {
      float4 f[UNROLL];
      float d[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        f[u] = *reinterpret_cast<const float4*>(fcol + (int64_t)s_rf[st + i + u] * c);
        d[u] = a.depth[s_rd[st + i + u]];
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        acc.x = fmaf(f[u].x, d[u], acc.x); acc.y = fmaf(f[u].y, d[u], acc.y);
        acc.z = fmaf(f[u].z, d[u], acc.z); acc.w = fmaf(f[u].w, d[u], acc.w);
      }
    }
    for (; i < len; ++i) {
      const float4 f = *reinterpret_cast<const float4*>(fcol + (int64_t)s_rf[st + i] * c);
      const float d = a.depth[s_rd[st + i]];
      acc.x = fmaf(f.x, d, acc.x); acc.y = fmaf(f.y, d, acc.y);
      acc.z = fmaf(f.z, d, acc.z); acc.w = fmaf(f.w, d, acc.w);
    }
    tile[(q * 4 + 0) * LDC + j] = acc.x;
    tile[(q * 4 + 1) * LDC + j] = acc.y;
    tile[(q * 4 + 2) * LDC + j] = acc.z;
    tile[(q * 4 + 3) * LDC + j] = acc.w;
  }
  // wave-local hand-off: LDS operations of one wave complete in order
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const bool occupied = (mask >> lane) & 1ull;
  const int col = __popcll(mask & ((1ull << lane) - 1ull));
  if (lane < nvox) {
    float* op = obase + lane;
    int cc = qlo * 4;
    const int cend = qhi * 4;
    for (; cc + SB <= cend; cc += SB) {
      float vals[SB];
#pragma unroll
      for (int u = 0; u < SB; ++u) vals[u] = occupied ? tile[(cc + u) * LDC + col] : 0.f;
#pragma unroll
      for (int u = 0; u < SB; ++u)
        __builtin_nontemporal_store(vals[u], op + (int64_t)(cc + u) * vpb);
    }
    for (; cc < cend; ++cc)
      __builtin_nontemporal_store(occupied ? tile[cc * LDC + col] : 0.f, op + (int64_t)cc * vpb);
  }
}

// ---------------------------------------------------------------------------
// v10: two 64-voxel tiles per 512-thread workgroup (amortises dispatch);
// v9: product kernel structure, but every wave owns a private block of
// channels end to end (gather -> LDS -> stores), so there is no workgroup
// barrier between gather and store; only the staging barrier remains.
// ---------------------------------------------------------------------------
template <int CAP, int UNROLL, int SB>
__global__ __launch_bounds__(512) void k_cf_v10(
    PoolArgs a, const int4* __restrict__ plan, int c, int cs, int64_t vpb,
    int64_t tiles_per_batch, float* __restrict__ out) {
  extern __shared__ float lds_all[];
  constexpr int V = 64, LDC = CAP + 1, PM = 1024, NW = 4;
  const int half = threadIdx.x >> 8;
  float* lds = lds_all + (size_t)half * ((size_t)cs * LDC + (V + 2 + V) + 2 * PM);
  float* tile = lds;
  int* istart = reinterpret_cast<int*>(lds + (size_t)cs * LDC);
  int* ivox = istart + V + 2;
  int* s_rf = ivox + V;
  int* s_rd = s_rf + PM;
  const int tid = threadIdx.x & 255, lane = tid & 63, w = tid >> 6;
  const int64_t t = (int64_t)blockIdx.x * 2 + half;
  const int64_t n_tiles_total = (int64_t)gridDim.x * 2;  // caller pads to even
  (void)n_tiles_total;
  const int c0 = blockIdx.y * cs;
  const int nch = (c - c0) < cs ? (c - c0) : cs;
  const int b = (int)(t / tiles_per_batch);
  const int64_t vox0 = (t - (int64_t)b * tiles_per_batch) * V;
  const int64_t remv = vpb - vox0;
  const int nvox = (int)(remv < V ? remv : V);
  const int64_t rank0 = (int64_t)b * vpb + vox0;
  float* obase = out + ((int64_t)b * c + c0) * vpb + vox0;
  const int4 pl = plan[t];
  const int i0 = pl.x, cnt = pl.y, p0 = pl.z, npts = pl.w;
  if (cnt == 0) {
    if (lane < nvox)
      for (int cc = w; cc < nch; cc += NW)
        __builtin_nontemporal_store(0.f, obase + (int64_t)cc * vpb + lane);
  }
  if (tid < cnt) {
    const int st = a.interval_starts[i0 + tid];
    istart[tid] = st - p0;
    ivox[tid] = (int)((int64_t)a.ranks_bev[st] - rank0);
  }
  if (tid == 0 && cnt > 0) istart[cnt] = npts;
  {
    const int n = npts < PM ? npts : PM;
    for (int p = tid; p < n; p += 256) {
      s_rf[p] = a.ranks_feat[p0 + p];
      s_rd[p] = a.ranks_depth[p0 + p];
    }
  }
  __syncthreads();
  if (cnt == 0) return;
  unsigned long long bit = lane < cnt ? (1ull << ivox[lane]) : 0ull;
  for (int off = 32; off > 0; off >>= 1) bit |= __shfl_xor(bit, off);
  const unsigned long long mask = bit;
  // this wave's channel block: quads [qlo, qhi)
  const int nq = nch / 4;
  const int qlo = (nq * w) / NW, qhi = (nq * (w + 1)) / NW;
  const int nqw = qhi - qlo;
  // (tiles whose points exceed the staged window or cnt > CAP are not handled
  //  by this experimental variant: host must guarantee npts <= PM, cnt <= CAP)
  const int items = cnt * nqw;
  for (int item = lane; item < items; item += 64) {
    const int j = item / nqw;
    const int q = qlo + (item - j * nqw);
    const int st = istart[j];
    const int len = istart[j + 1] - st;
    const float* fcol = a.feat + c0 + q * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int i = 0;
    for (; i + UNROLL <= len; i += UNROLL) {
      float4 f[UNROLL];
      float d[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        f[u] = *reinterpret_cast<const float4*>(fcol + (int64_t)s_rf[st + i + u] * c);
        d[u] = a.depth[s_rd[st + i + u]];
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        acc.x = fmaf(f[u].x, d[u], acc.x); acc.y = fmaf(f[u].y, d[u], acc.y);
        acc.z = fmaf(f[u].z, d[u], acc.z); acc.w = fmaf(f[u].w, d[u], acc.w);
      }
    }
    for (; i < len; ++i) {
      const float4 f = *reinterpret_cast<const float4*>(fcol + (int64_t)s_rf[st + i] * c);
      const float d = a.depth[s_rd[st + i]];
      acc.x = fmaf(f.x, d, acc.x); acc.y = fmaf(f.y, d, acc.y);
      acc.z = fmaf(f.z, d, acc.z); acc.w = fmaf(f.w, d, acc.w);
    }
    tile[(q * 4 + 0) * LDC + j] = acc.x;
    tile[(q * 4 + 1) * LDC + j] = acc.y;
    tile[(q * 4 + 2) * LDC + j] = acc.z;
    tile[(q * 4 + 3) * LDC + j] = acc.w;
  }
  // wave-local hand-off: LDS operations of one wave complete in order
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const bool occupied = (mask >> lane) & 1ull;
  const int col = __popcll(mask & ((1ull << lane) - 1ull));
  if (lane < nvox) {
    float* op = obase + lane;
    int cc = qlo * 4;
    const int cend = qhi * 4;
    for (; cc + SB <= cend; cc += SB) {
      float vals[SB];
#pragma unroll
      for (int u = 0; u < SB; ++u) vals[u] = occupied ? tile[(cc + u) * LDC + col] : 0.f;
#pragma unroll
      for (int u = 0; u < SB; ++u)
        __builtin_nontemporal_store(vals[u], op + (int64_t)(cc + u) * vpb);
    }
    for (; cc < cend; ++cc)
      __builtin_nontemporal_store(occupied ? tile[cc * LDC + col] : 0.f, op + (int64_t)cc * vpb);
  }
}


// ---------------------------------------------------------------------------
// v11: product structure + fixed-size 256-B tile RECORDS (cached plan v2).
// record (64 ints): [0] = cnt | npts << 8  (cnt == 255: tile too big, general
// path through plan4);  bytes 4..28 ivox[24];  bytes 28..53 istart[25];
// ints 16..39 ranks_feat[24];  ints 40..63 ranks_depth[24].
// One load level delivers everything static about a small tile.
// ---------------------------------------------------------------------------
template <int CAP, int UNROLL, int SB>
__global__ __launch_bounds__(256) void k_cf_v11(
    PoolArgs a, const int4* __restrict__ plan, const int* __restrict__ recs,
    int c, int cs, int64_t vpb, int64_t tiles_per_batch, float* __restrict__ out) {
  extern __shared__ float lds[];
  constexpr int V = 64, LDC = CAP + 1, PM = 1024, NW = 4;
  float* tile = lds;
  int* istart = reinterpret_cast<int*>(lds + (size_t)cs * LDC);
  int* ivox = istart + V + 2;
  int* s_rf = ivox + V;
  int* s_rd = s_rf + PM;
  int* s_rec = s_rd + PM;  // [64]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t t = blockIdx.x;
  const int c0 = blockIdx.y * cs;
  const int nch = (c - c0) < cs ? (c - c0) : cs;
  const int b = (int)(t / tiles_per_batch);
  const int64_t vox0 = (t - (int64_t)b * tiles_per_batch) * V;
  const int64_t remv = vpb - vox0;
  const int nvox = (int)(remv < V ? remv : V);
  const int64_t rank0 = (int64_t)b * vpb + vox0;
  float* obase = out + ((int64_t)b * c + c0) * vpb + vox0;
  // level 1: the whole record (wave 0) -- and nothing else for small tiles
  if (tid < 64) s_rec[tid] = recs[t * 64 + tid];
  __syncthreads();
  const int hdr = s_rec[0];
  int cnt = hdr & 255, npts = hdr >> 8;
  int i0 = 0, p0 = 0;
  const bool big = cnt == 255;
  if (cnt == 0) {
    if (lane < nvox)
      for (int cc = w; cc < nch; cc += NW)
        __builtin_nontemporal_store(0.f, obase + (int64_t)cc * vpb + lane);
    return;
  }
  if (!big) {
    const unsigned char* rb8 = reinterpret_cast<const unsigned char*>(s_rec);
    if (tid < cnt) ivox[tid] = rb8[4 + tid];
    if (tid <= cnt) istart[tid] = rb8[28 + tid];
    if (tid < npts) { s_rf[tid] = s_rec[16 + tid]; s_rd[tid] = s_rec[40 + tid]; }
  } else {
    const int4 pl = plan[t];
    i0 = pl.x; cnt = pl.y; p0 = pl.z; npts = pl.w;
    if (tid < cnt) {
      const int st = a.interval_starts[i0 + tid];
      istart[tid] = st - p0;
      ivox[tid] = (int)((int64_t)a.ranks_bev[st] - rank0);
    }
    if (tid == 0) istart[cnt] = npts;
    const int n = npts < PM ? npts : PM;
    for (int p = tid; p < n; p += 256) {
      s_rf[p] = a.ranks_feat[p0 + p];
      s_rd[p] = a.ranks_depth[p0 + p];
    }
  }
  __syncthreads();
  unsigned long long bit = lane < cnt ? (1ull << ivox[lane]) : 0ull;
  for (int off = 32; off > 0; off >>= 1) bit |= __shfl_xor(bit, off);
  const unsigned long long mask = bit;
  const int nq = nch / 4;
  // (experimental variant: host guarantees npts <= PM and cnt <= CAP)
  const int items = cnt * nq;
  for (int item = tid; item < items; item += 256) {
    const int j = item / nq;
    const int q = item - j * nq;
    const int st = istart[j];
    const int len = istart[j + 1] - st;
    const float* fcol = a.feat + c0 + q * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int i = 0;
    for (; i + UNROLL <= len; i += UNROLL) {
      float4 f[UNROLL];
      float d[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        f[u] = *reinterpret_cast<const float4*>(fcol + (int64_t)s_rf[st + i + u] * c);
        d[u] = a.depth[s_rd[st + i + u]];
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        acc.x = fmaf(f[u].x, d[u], acc.x); acc.y = fmaf(f[u].y, d[u], acc.y);
        acc.z = fmaf(f[u].z, d[u], acc.z); acc.w = fmaf(f[u].w, d[u], acc.w);
      }
    }
    for (; i < len; ++i) {
      const float4 f = *reinterpret_cast<const float4*>(fcol + (int64_t)s_rf[st + i] * c);
      const float d = a.depth[s_rd[st + i]];
      acc.x = fmaf(f.x, d, acc.x); acc.y = fmaf(f.y, d, acc.y);
      acc.z = fmaf(f.z, d, acc.z); acc.w = fmaf(f.w, d, acc.w);
    }
    tile[(q * 4 + 0) * LDC + j] = acc.x;
    tile[(q * 4 + 1) * LDC + j] = acc.y;
    tile[(q * 4 + 2) * LDC + j] = acc.z;
    tile[(q * 4 + 3) * LDC + j] = acc.w;
  }
  __syncthreads();
  const bool occupied = (mask >> lane) & 1ull;
  const int col = __popcll(mask & ((1ull << lane) - 1ull));
  if (lane < nvox) {
    float* op = obase + lane;
    int cc = w;
    for (; cc + (SB - 1) * NW < nch; cc += SB * NW) {
      float vals[SB];
#pragma unroll
      for (int u = 0; u < SB; ++u) vals[u] = occupied ? tile[(cc + NW * u) * LDC + col] : 0.f;
#pragma unroll
      for (int u = 0; u < SB; ++u)
        __builtin_nontemporal_store(vals[u], op + (int64_t)(cc + NW * u) * vpb);
    }
    for (; cc < nch; cc += NW)
      __builtin_nontemporal_store(occupied ? tile[cc * LDC + col] : 0.f, op + (int64_t)cc * vpb);
  }
}

// ---------------------------------------------------------------------------
// v12: v11 with 16-byte stores (a wave instruction = 4 channel rows x 256 B);
// v11: product structure + fixed-size 256-B tile RECORDS (cached plan v2).
// record (64 ints): [0] = cnt | npts << 8  (cnt == 255: tile too big, general
// path through plan4);  bytes 4..28 ivox[24];  bytes 28..53 istart[25];
// ints 16..39 ranks_feat[24];  ints 40..63 ranks_depth[24].
// One load level delivers everything static about a small tile.
// ---------------------------------------------------------------------------
template <int CAP, int UNROLL, int SB>
__global__ __launch_bounds__(256) void k_cf_v12(
    PoolArgs a, const int4* __restrict__ plan, const int* __restrict__ recs,
    int c, int cs, int64_t vpb, int64_t tiles_per_batch, float* __restrict__ out) {
  extern __shared__ float lds[];
  constexpr int V = 64, LDC = CAP + 1, PM = 1024, NW = 4;
  float* tile = lds;
  int* istart = reinterpret_cast<int*>(lds + (size_t)cs * LDC);
  int* ivox = istart + V + 2;
  int* s_rf = ivox + V;
  int* s_rd = s_rf + PM;
  int* s_rec = s_rd + PM;  // [64]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t t = blockIdx.x;
  const int c0 = blockIdx.y * cs;
  const int nch = (c - c0) < cs ? (c - c0) : cs;
  const int b = (int)(t / tiles_per_batch);
  const int64_t vox0 = (t - (int64_t)b * tiles_per_batch) * V;
  const int64_t remv = vpb - vox0;
  const int nvox = (int)(remv < V ? remv : V);
  const int64_t rank0 = (int64_t)b * vpb + vox0;
  float* obase = out + ((int64_t)b * c + c0) * vpb + vox0;
  // level 1: the whole record (wave 0) -- and nothing else for small tiles
  if (tid < 64) s_rec[tid] = recs[t * 64 + tid];
  __syncthreads();
  const int hdr = s_rec[0];
  int cnt = hdr & 255, npts = hdr >> 8;
  int i0 = 0, p0 = 0;
  const bool big = cnt == 255;
  if (cnt == 0) {
    const int vq0 = (lane & 15) * 4;
    if (vq0 + 3 < nvox) {
      for (int cc = w * 4 + (lane >> 4); cc < nch; cc += NW * 4)
        __builtin_nontemporal_store(nt_f4{0.f, 0.f, 0.f, 0.f},
                                    reinterpret_cast<nt_f4*>(obase + (int64_t)cc * vpb + vq0));
    } else {
      for (int cc = w * 4 + (lane >> 4); cc < nch; cc += NW * 4)
        for (int k = 0; k < 4; ++k)
          if (vq0 + k < nvox) obase[(int64_t)cc * vpb + vq0 + k] = 0.f;
    }
    return;
  }
  if (!big) {
    const unsigned char* rb8 = reinterpret_cast<const unsigned char*>(s_rec);
    if (tid < cnt) ivox[tid] = rb8[4 + tid];
    if (tid <= cnt) istart[tid] = rb8[28 + tid];
    if (tid < npts) { s_rf[tid] = s_rec[16 + tid]; s_rd[tid] = s_rec[40 + tid]; }
  } else {
    const int4 pl = plan[t];
    i0 = pl.x; cnt = pl.y; p0 = pl.z; npts = pl.w;
    if (tid < cnt) {
      const int st = a.interval_starts[i0 + tid];
      istart[tid] = st - p0;
      ivox[tid] = (int)((int64_t)a.ranks_bev[st] - rank0);
    }
    if (tid == 0) istart[cnt] = npts;
    const int n = npts < PM ? npts : PM;
    for (int p = tid; p < n; p += 256) {
      s_rf[p] = a.ranks_feat[p0 + p];
      s_rd[p] = a.ranks_depth[p0 + p];
    }
  }
  __syncthreads();
  unsigned long long bit = lane < cnt ? (1ull << ivox[lane]) : 0ull;
  for (int off = 32; off > 0; off >>= 1) bit |= __shfl_xor(bit, off);
  const unsigned long long mask = bit;
  const int nq = nch / 4;
  // (experimental variant: host guarantees npts <= PM and cnt <= CAP)
  const int items = cnt * nq;
  for (int item = tid; item < items; item += 256) {
    const int j = item / nq;
    const int q = item - j * nq;
    const int st = istart[j];
    const int len = istart[j + 1] - st;
    const float* fcol = a.feat + c0 + q * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int i = 0;
    for (; i + UNROLL <= len; i += UNROLL) {
      float4 f[UNROLL];
      float d[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        f[u] = *reinterpret_cast<const float4*>(fcol + (int64_t)s_rf[st + i + u] * c);
        d[u] = a.depth[s_rd[st + i + u]];
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        acc.x = fmaf(f[u].x, d[u], acc.x); acc.y = fmaf(f[u].y, d[u], acc.y);
        acc.z = fmaf(f[u].z, d[u], acc.z); acc.w = fmaf(f[u].w, d[u], acc.w);
      }
    }
    for (; i < len; ++i) {
      const float4 f = *reinterpret_cast<const float4*>(fcol + (int64_t)s_rf[st + i] * c);
      const float d = a.depth[s_rd[st + i]];
      acc.x = fmaf(f.x, d, acc.x); acc.y = fmaf(f.y, d, acc.y);
      acc.z = fmaf(f.z, d, acc.z); acc.w = fmaf(f.w, d, acc.w);
    }
    tile[(q * 4 + 0) * LDC + j] = acc.x;
    tile[(q * 4 + 1) * LDC + j] = acc.y;
    tile[(q * 4 + 2) * LDC + j] = acc.z;
    tile[(q * 4 + 3) * LDC + j] = acc.w;
  }
  __syncthreads();
  // lane l: channel row (l >> 4) of a group of 4, voxels 4*(l & 15) .. +3
  const int vq = (lane & 15) * 4;
  const int sub = lane >> 4;
  float4 m4;
  int col4[4];
  bool occ4[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    occ4[k] = (mask >> (vq + k)) & 1ull;
    col4[k] = __popcll(mask & ((1ull << (vq + k)) - 1ull));
  }
  const bool full_quad = vq + 3 < nvox;
  for (int cc = w * 4 + sub; cc < nch; cc += NW * 4) {
    m4.x = occ4[0] ? tile[cc * LDC + col4[0]] : 0.f;
    m4.y = occ4[1] ? tile[cc * LDC + col4[1]] : 0.f;
    m4.z = occ4[2] ? tile[cc * LDC + col4[2]] : 0.f;
    m4.w = occ4[3] ? tile[cc * LDC + col4[3]] : 0.f;
    float* op = obase + (int64_t)cc * vpb + vq;
    if (full_quad) {
      __builtin_nontemporal_store(nt_f4{m4.x, m4.y, m4.z, m4.w}, reinterpret_cast<nt_f4*>(op));
    } else {
      if (vq + 0 < nvox) op[0] = m4.x;
      if (vq + 1 < nvox) op[1] = m4.y;
      if (vq + 2 < nvox) op[2] = m4.z;
    }
  }
}


// ---------------------------------------------------------------------------
// v13: "zeros first, patch later".  Right after the plan entry is known every
// wave streams the zero rows of its channels (full 256-B stores that depend on
// nothing), the gather proceeds underneath, and only the occupied dwords are
// stored again (masked row stores from the LDS tile) once the sums exist.
// Same wave writes zero row and patch of a channel, with a vmcnt(0) in between.
// ---------------------------------------------------------------------------
template <int CAP, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_cf_v13(
    PoolArgs a, const int4* __restrict__ plan, int c, int cs, int64_t vpb,
    int64_t tiles_per_batch, float* __restrict__ out) {
  extern __shared__ float lds[];
  constexpr int V = 64, LDC = CAP + 1, PM = 1024, NW = 4;
  float* tile = lds;
  int* istart = reinterpret_cast<int*>(lds + (size_t)cs * LDC);
  int* ivox = istart + V + 2;
  int* s_rf = ivox + V;
  int* s_rd = s_rf + PM;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t t = blockIdx.x;
  const int c0 = blockIdx.y * cs;
  const int nch = (c - c0) < cs ? (c - c0) : cs;
  const int b = (int)(t / tiles_per_batch);
  const int64_t vox0 = (t - (int64_t)b * tiles_per_batch) * V;
  const int64_t remv = vpb - vox0;
  const int nvox = (int)(remv < V ? remv : V);
  const int64_t rank0 = (int64_t)b * vpb + vox0;
  float* obase = out + ((int64_t)b * c + c0) * vpb + vox0;
  const int4 pl = plan[t];
  const int i0 = pl.x, cnt = pl.y, p0 = pl.z, npts = pl.w;
  // prologue loads first (they are short), then the zero rows
  int my_st = 0, my_vox = 0;
  if (cnt > 0 && tid < cnt) {
    my_st = a.interval_starts[i0 + tid];
    my_vox = a.ranks_bev[my_st];
  }
  int pre_rf[4], pre_rd[4];
  const int n0 = npts < PM ? npts : PM;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = tid + k * 256;
    pre_rf[k] = p < n0 ? a.ranks_feat[p0 + p] : 0;
    pre_rd[k] = p < n0 ? a.ranks_depth[p0 + p] : 0;
  }
  if (lane < nvox)
    for (int cc = w; cc < nch; cc += NW) {
      if (NT) __builtin_nontemporal_store(0.f, obase + (int64_t)cc * vpb + lane);
      else obase[(int64_t)cc * vpb + lane] = 0.f;
    }
  if (cnt == 0) return;
  if (tid < cnt) {
    istart[tid] = my_st - p0;
    ivox[tid] = (int)((int64_t)my_vox - rank0);
  }
  if (tid == 0) istart[cnt] = npts;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = tid + k * 256;
    if (p < n0) { s_rf[p] = pre_rf[k]; s_rd[p] = pre_rd[k]; }
  }
  __syncthreads();
  unsigned long long bit = lane < cnt ? (1ull << ivox[lane]) : 0ull;
  for (int off = 32; off > 0; off >>= 1) bit |= __shfl_xor(bit, off);
  const unsigned long long mask = bit;
  const int nq = nch / 4;
  const int items = cnt * nq;   // (experimental: host guarantees npts <= PM, cnt <= CAP)
  for (int item = tid; item < items; item += 256) {
    const int j = item / nq;
    const int q = item - j * nq;
    const int st = istart[j];
    const int len = istart[j + 1] - st;
    const float* fcol = a.feat + c0 + q * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int i = 0;
    for (; i + UNROLL <= len; i += UNROLL) {
      float4 f[UNROLL];
      float d[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        f[u] = *reinterpret_cast<const float4*>(fcol + (int64_t)s_rf[st + i + u] * c);
        d[u] = a.depth[s_rd[st + i + u]];
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        acc.x = fmaf(f[u].x, d[u], acc.x); acc.y = fmaf(f[u].y, d[u], acc.y);
        acc.z = fmaf(f[u].z, d[u], acc.z); acc.w = fmaf(f[u].w, d[u], acc.w);
      }
    }
    for (; i < len; ++i) {
      const float4 f = *reinterpret_cast<const float4*>(fcol + (int64_t)s_rf[st + i] * c);
      const float d = a.depth[s_rd[st + i]];
      acc.x = fmaf(f.x, d, acc.x); acc.y = fmaf(f.y, d, acc.y);
      acc.z = fmaf(f.z, d, acc.z); acc.w = fmaf(f.w, d, acc.w);
    }
    tile[(q * 4 + 0) * LDC + j] = acc.x;
    tile[(q * 4 + 1) * LDC + j] = acc.y;
    tile[(q * 4 + 2) * LDC + j] = acc.z;
    tile[(q * 4 + 3) * LDC + j] = acc.w;
  }
  __syncthreads();
  // zero rows of this wave's channels must have left before the patches
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const bool occupied = (mask >> lane) & 1ull;
  if (occupied && lane < nvox) {
    const int col = __popcll(mask & ((1ull << lane) - 1ull));
    float* op = obase + lane;
    for (int cc = w; cc < nch; cc += NW) {
      const float v = tile[cc * LDC + col];
      if (NT) __builtin_nontemporal_store(v, op + (int64_t)cc * vpb);
      else op[(int64_t)cc * vpb] = v;
    }
  }
}

__global__ void k_plan4(PoolArgs a, int n_intervals, int n_points, int V,
                        int64_t n_tiles, int* __restrict__ tile_first,
                        int* __restrict__ tile_point, int4* __restrict__ plan4) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tiles) return;
  plan4[t] = make_int4(tile_first[t], tile_first[t + 1] - tile_first[t], tile_point[t],
                       tile_point[t + 1] - tile_point[t]);
}

// plan by scatter: one thread per interval (+1), fills the tiles since the
// previous interval's tile.
__global__ void k_plan2(PoolArgs a, int n_intervals, int n_points, int V,
                        int64_t n_tiles, int* __restrict__ tile_first,
                        int* __restrict__ tile_point) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_intervals) return;
  int64_t lo, hi;  // tiles (lo, hi] get first interval = i
  int st = n_points;
  if (i < n_intervals) {
    st = a.interval_starts[i];
    hi = a.ranks_bev[st] / V;
  } else {
    hi = n_tiles;
  }
  lo = (i == 0) ? -1 : (int64_t)(a.ranks_bev[a.interval_starts[i - 1]] / V);
  for (int64_t tt = lo + 1; tt <= hi; ++tt) {
    tile_first[tt] = (int)i;
    tile_point[tt] = st;
  }
}

// plan builder: one thread per tile boundary, binary search over interval keys
__global__ void k_plan(PoolArgs a, int n_intervals, int n_points, int64_t vpb,
                       int64_t tiles_per_batch, int V, int64_t n_tiles,
                       int* __restrict__ tile_first, int* __restrict__ tile_point) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t > n_tiles) return;
  int r = n_intervals;
  if (t < n_tiles) {
    const int b = (int)(t / tiles_per_batch);
    const int64_t rank0 = (int64_t)b * vpb + (t - (int64_t)b * tiles_per_batch) * V;
    int lo = 0, hi = n_intervals;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((int64_t)a.ranks_bev[a.interval_starts[mid]] < rank0) lo = mid + 1; else hi = mid;
    }
    r = lo;
  }
  tile_first[t] = r;
  tile_point[t] = r < n_intervals ? a.interval_starts[r] : n_points;
}

}  // namespace

extern "C" int poolvar_plan(int n_intervals, int n_points, int batch, int64_t vpb,
                            const int* ranks_bev, const int* interval_starts,
                            int* tile_first, int* tile_point, void* stream) {
  PoolArgs a{nullptr, nullptr, nullptr, nullptr, ranks_bev, interval_starts, nullptr};
  const int V = 64;
  const int64_t tpb = (vpb + V - 1) / V, n_tiles = tpb * batch;
  hipLaunchKernelGGL(k_plan, dim3((unsigned)((n_tiles + 1 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, a, n_intervals, n_points, vpb, tpb, V, n_tiles,
                     tile_first, tile_point);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

static unsigned long long* g_stamps = nullptr;
static const int4* g_plan4 = nullptr;
static int* g_queue = nullptr;
static const int* g_recs = nullptr;
extern "C" void poolvar_set_recs(const int* p) { g_recs = p; }

static const int4* g_plan8 = nullptr;
extern "C" void poolvar_set_plan8(const int4* p) { g_plan8 = p; }

static int g_wg_per_cu = 6;
extern "C" void poolvar_set_queue(int* q, int wg_per_cu) { g_queue = q; g_wg_per_cu = wg_per_cu; }

extern "C" int poolvar_plan4(int batch, int64_t vpb, const int* tile_first, const int* tile_point, int4* plan4, void* stream) {
  const int64_t n_tiles = ((vpb + 63) / 64) * batch;
  PoolArgs a{};
  hipLaunchKernelGGL(k_plan4, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, 0, 0, 64, n_tiles, const_cast<int*>(tile_first), const_cast<int*>(tile_point), plan4);
  g_plan4 = plan4;
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" void poolvar_set_stamps(unsigned long long* p) { g_stamps = p; }

extern "C" int poolvar_run(int variant, int c, int cs, int n_intervals, int batch,
                           int64_t vpb, const float* depth, const float* feat,
                           const int* ranks_depth, const int* ranks_feat,
                           const int* ranks_bev, const int* interval_starts,
                           const int* interval_lengths, const int* tile_first,
                           const int* tile_point, float* out, void* stream) {
  PoolArgs a{depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths};
  Plan plan{tile_first, tile_point, g_stamps};
  const int V = 64;
  const int64_t tpb = (vpb + V - 1) / V, n_tiles = tpb * batch;
  const int slabs = (c + cs - 1) / cs;
  dim3 grid((unsigned)n_tiles, (unsigned)slabs);
  hipStream_t s = (hipStream_t)stream;
#define LDS(PM) ((size_t)cs * 65 * 4 + 3 * 64 * 4 + (size_t)(PM) * 8 + 16)
  switch (variant) {
    case 0: hipLaunchKernelGGL((k_cf_v2<256, false, false, 4>), grid, dim3(kBlock), LDS(256), s, a, plan, c, cs, n_intervals, vpb, tpb, out); break;
    case 1: hipLaunchKernelGGL((k_cf_v2<256, true, false, 4>), grid, dim3(kBlock), LDS(256), s, a, plan, c, cs, n_intervals, vpb, tpb, out); break;
    case 2: hipLaunchKernelGGL((k_cf_v2<256, true, true, 4>), grid, dim3(kBlock), LDS(256), s, a, plan, c, cs, n_intervals, vpb, tpb, out); break;
    case 3: hipLaunchKernelGGL((k_cf_v2<256, true, true, 8>), grid, dim3(kBlock), LDS(256), s, a, plan, c, cs, n_intervals, vpb, tpb, out); break;
    case 4: hipLaunchKernelGGL((k_cf_v2<512, true, true, 8>), grid, dim3(kBlock), LDS(512), s, a, plan, c, cs, n_intervals, vpb, tpb, out); break;
    case 5: hipLaunchKernelGGL((k_cf_v2<1024, true, true, 8>), grid, dim3(kBlock), LDS(1024), s, a, plan, c, cs, n_intervals, vpb, tpb, out); break;
#define LDS3(PM) ((size_t)cs * 65 * 4 + (2 * 64 + 2) * 4 + (size_t)(PM) * 8 + 16)
    case 10: hipLaunchKernelGGL((k_cf_v3<4, 1024, 8>), grid, dim3(kBlock), LDS3(1024), s, a, plan, c, cs, n_intervals, vpb, tpb, out); break;
    case 11: hipLaunchKernelGGL((k_cf_v3<4, 1024, 4>), grid, dim3(kBlock), LDS3(1024), s, a, plan, c, cs, n_intervals, vpb, tpb, out); break;
    case 12: hipLaunchKernelGGL((k_cf_v3<4, 512, 8>), grid, dim3(kBlock), LDS3(512), s, a, plan, c, cs, n_intervals, vpb, tpb, out); break;
    case 13: hipLaunchKernelGGL((k_cf_v3<1, 1024, 8>), grid, dim3(kBlock), LDS3(1024), s, a, plan, c, cs, n_intervals, vpb, tpb, out); break;
    case 14: hipLaunchKernelGGL((k_cf_v3<4, 256, 8>), grid, dim3(kBlock), LDS3(256), s, a, plan, c, cs, n_intervals, vpb, tpb, out); break;
    case 15: hipLaunchKernelGGL((k_cf_v3<4, 1024, 16>), grid, dim3(kBlock), LDS3(1024), s, a, plan, c, cs, n_intervals, vpb, tpb, out); break;
#define LDS4(PM, CAP) ((size_t)cs * ((CAP) + 1) * 4 + (64 + 2) * 4 + (size_t)(PM) * 12)
    case 40: hipLaunchKernelGGL((k_cf_v4<4, 512, 8, 32, false>), grid, dim3(kBlock), LDS4(512, 32), s, a, g_plan4, c, cs, n_intervals, vpb, tpb, out); break;
    case 41: hipLaunchKernelGGL((k_cf_v4<4, 512, 8, 32, true>), grid, dim3(kBlock), LDS4(512, 32), s, a, g_plan4, c, cs, n_intervals, vpb, tpb, out); break;
    case 42: hipLaunchKernelGGL((k_cf_v4<4, 512, 8, 64, false>), grid, dim3(kBlock), LDS4(512, 64), s, a, g_plan4, c, cs, n_intervals, vpb, tpb, out); break;
    case 43: hipLaunchKernelGGL((k_cf_v4<4, 1024, 8, 32, false>), grid, dim3(kBlock), LDS4(1024, 32), s, a, g_plan4, c, cs, n_intervals, vpb, tpb, out); break;
    case 44: hipLaunchKernelGGL((k_cf_v4<4, 256, 4, 32, false>), grid, dim3(kBlock), LDS4(256, 32), s, a, g_plan4, c, cs, n_intervals, vpb, tpb, out); break;
    case 45: hipLaunchKernelGGL((k_cf_v4<4, 512, 8, 64, true>), grid, dim3(kBlock), LDS4(512, 64), s, a, g_plan4, c, cs, n_intervals, vpb, tpb, out); break;
    case 50: hipLaunchKernelGGL((k_cf_v4<4, 1024, 8, 32, false, 1>), grid, dim3(kBlock), LDS4(1024, 32), s, a, g_plan4, c, cs, n_intervals, vpb, tpb, out); break;
    case 51: hipLaunchKernelGGL((k_cf_v4<4, 1024, 8, 32, false, 2>), grid, dim3(kBlock), LDS4(1024, 32), s, a, g_plan4, c, cs, n_intervals, vpb, tpb, out); break;
    case 52: hipLaunchKernelGGL((k_cf_v4<4, 1024, 8, 32, false, 3>), grid, dim3(kBlock), LDS4(1024, 32), s, a, g_plan4, c, cs, n_intervals, vpb, tpb, out); break;
    case 53: hipLaunchKernelGGL((k_cf_v4<4, 1024, 8, 32, false, 4>), grid, dim3(kBlock), LDS4(1024, 32), s, a, g_plan4, c, cs, n_intervals, vpb, tpb, out); break;
    case 54: hipLaunchKernelGGL((k_cf_v4<4, 1024, 8, 32, false, 4>), grid, dim3(kBlock), 0, s, a, g_plan4, c, cs, n_intervals, vpb, tpb, out); break;
#define LDS5(PM, CAP) ((size_t)cs * ((CAP) + 1) * 4 + (64 + 2) * 4 + (size_t)(PM) * 12 + 16)
#define V5(ID, VEC, PM, UN, CAP, NT, SB) case ID: { const int64_t n_work = n_tiles * slabs; int g = g_wg_per_cu * 256; if (g > n_work) g = (int)n_work; if (g_queue) hipMemsetAsync(g_queue, 0, 4, s); hipLaunchKernelGGL((k_cf_v5<VEC, PM, UN, CAP, NT, SB>), dim3(g), dim3(kBlock), LDS5(PM, CAP), s, a, g_plan4, g_queue, c, cs, slabs, n_intervals, vpb, tpb, n_work, out); } break;
    V5(60, 4, 1024, 8, 32, false, 5)
    V5(61, 4, 1024, 8, 32, true, 5)
    V5(62, 4, 512, 8, 32, false, 5)
    V5(63, 4, 1024, 8, 64, false, 5)
    V5(64, 4, 1024, 8, 32, false, 1)
    V5(65, 4, 1024, 4, 32, false, 5)
    V5(66, 4, 768, 8, 32, true, 5)
#define LDS6(PM, CAP) ((size_t)cs * ((CAP) + 1) * 4 + (64 + 2 + 64) * 4 + (size_t)(PM) * 8)
#define V6(ID, VEC, PM, UN, CAP, NT, SB, LONG, ULONG, MINW) case ID: hipLaunchKernelGGL((k_cf_v6<VEC, PM, UN, CAP, NT, SB, LONG, ULONG, MINW>), grid, dim3(kBlock), LDS6(PM, CAP), s, a, g_plan4, c, cs, n_intervals, vpb, tpb, out); break;
    V6(70, 4, 1024, 8, 32, false, 5, 100000, 16, 8)
    V6(71, 4, 1024, 8, 32, true, 5, 100000, 16, 8)
    V6(72, 4, 1024, 8, 32, false, 5, 24, 16, 8)
    V6(73, 4, 1024, 8, 32, true, 5, 24, 16, 8)
    V6(74, 4, 1024, 4, 32, true, 5, 16, 16, 8)
    V6(75, 4, 1024, 8, 32, true, 5, 24, 16, 6)
    V6(76, 4, 1024, 8, 32, true, 5, 24, 16, 5)
    V6(77, 4, 1024, 8, 32, true, 5, 24, 32, 4)
#define V7(ID, BLK, VEC, PM, UN, CAP, NT, SB, LONG, ULONG, MINW) case ID: hipLaunchKernelGGL((k_cf_v7<BLK, VEC, PM, UN, CAP, NT, SB, LONG, ULONG, MINW>), grid, dim3(BLK), LDS6(PM, CAP), s, a, g_plan4, c, cs, n_intervals, vpb, tpb, out); break;
    V7(80, 128, 4, 1024, 8, 32, false, 5, 100000, 8, 4)
    V7(81, 128, 4, 1024, 8, 32, true, 5, 100000, 8, 4)
    V7(82, 128, 4, 1024, 8, 32, true, 5, 24, 16, 4)
    V7(83, 128, 4, 1024, 4, 32, true, 5, 100000, 8, 4)
    V7(84, 64, 4, 1024, 8, 32, true, 5, 100000, 8, 2)
    V7(85, 128, 4, 768, 8, 32, true, 4, 100000, 8, 4)
    V7(86, 256, 4, 1024, 4, 32, true, 5, 100000, 8, 8)
    V7(87, 128, 4, 1024, 8, 64, true, 5, 100000, 8, 4)
#define V8(ID, BLK, VEC, PM, UN, CAP, NT, SB, LONG, ULONG, MINW) case ID: hipLaunchKernelGGL((k_cf_v8<BLK, VEC, PM, UN, CAP, NT, SB, LONG, ULONG, MINW>), grid, dim3(BLK), LDS6(PM, CAP), s, a, g_plan8, c, cs, n_intervals, vpb, tpb, out); break;
    V8(90, 128, 4, 1024, 8, 32, true, 5, 100000, 8, 4)
    V8(91, 256, 4, 1024, 4, 32, true, 5, 100000, 8, 8)
    V8(92, 128, 4, 1024, 8, 64, true, 5, 100000, 8, 4)
    V8(93, 256, 4, 1024, 8, 32, true, 5, 100000, 8, 5)
    V8(94, 256, 4, 1024, 8, 32, false, 5, 100000, 8, 5)
#define LDS9(CAP) ((size_t)cs * ((CAP) + 1) * 4 + (64 + 2 + 64) * 4 + (size_t)1024 * 8)
    case 95: hipLaunchKernelGGL((k_cf_v9<64, 8, 5>), grid, dim3(256), LDS9(64), s, a, g_plan4, c, cs, vpb, tpb, out); break;
    case 96: hipLaunchKernelGGL((k_cf_v9<64, 4, 5>), grid, dim3(256), LDS9(64), s, a, g_plan4, c, cs, vpb, tpb, out); break;
    case 97: hipLaunchKernelGGL((k_cf_v9<64, 8, 10>), grid, dim3(256), LDS9(64), s, a, g_plan4, c, cs, vpb, tpb, out); break;
    case 98: hipLaunchKernelGGL((k_cf_v10<64, 8, 5>), dim3((unsigned)(n_tiles / 2), (unsigned)slabs), dim3(512), 2 * LDS9(64), s, a, g_plan4, c, cs, vpb, tpb, out); break;
    case 99: hipLaunchKernelGGL((k_cf_v10<32, 8, 5>), dim3((unsigned)(n_tiles / 2), (unsigned)slabs), dim3(512), 2 * LDS9(32), s, a, g_plan4, c, cs, vpb, tpb, out); break;
#define LDS11(CAP) (LDS9(CAP) + 256)
    case 110: hipLaunchKernelGGL((k_cf_v11<64, 8, 5>), grid, dim3(256), LDS11(64), s, a, g_plan4, g_recs, c, cs, vpb, tpb, out); break;
    case 111: hipLaunchKernelGGL((k_cf_v11<64, 4, 5>), grid, dim3(256), LDS11(64), s, a, g_plan4, g_recs, c, cs, vpb, tpb, out); break;
    case 120: hipLaunchKernelGGL((k_cf_v12<64, 8, 5>), grid, dim3(256), LDS11(64), s, a, g_plan4, g_recs, c, cs, vpb, tpb, out); break;
    case 130: hipLaunchKernelGGL((k_cf_v13<64, 8, false>), grid, dim3(256), LDS9(64), s, a, g_plan4, c, cs, vpb, tpb, out); break;
    case 131: hipLaunchKernelGGL((k_cf_v13<64, 8, true>), grid, dim3(256), LDS9(64), s, a, g_plan4, c, cs, vpb, tpb, out); break;
    case 59: hipLaunchKernelGGL((k_cf_v4<4, 1024, 8, 32, false, 9>), grid, dim3(kBlock), LDS4(1024, 32), s, a, g_plan4, c, cs, n_intervals, vpb, tpb, out); break;
    case 20: hipLaunchKernelGGL((k_cf_v3<4, 1024, 8, true>), grid, dim3(kBlock), LDS3(1024), s, a, plan, c, cs, n_intervals, vpb, tpb, out); break;
    default: return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int poolvar_plan2(int n_intervals, int n_points, int batch, int64_t vpb,
                             const int* ranks_bev, const int* interval_starts,
                             int* tile_first, int* tile_point, void* stream) {
  PoolArgs a{nullptr, nullptr, nullptr, nullptr, ranks_bev, interval_starts, nullptr};
  const int V = 64;
  const int64_t tpb = (vpb + V - 1) / V, n_tiles = tpb * batch;
  hipLaunchKernelGGL(k_plan2, dim3((unsigned)((n_intervals + 1 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, a, n_intervals, n_points, V, n_tiles,
                     tile_first, tile_point);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
