// Index half of the lift on MI355X: frustum geometry, voxelisation, sort by
// voxel rank, run-length intervals and the pool plan -- hand-written HIP.
//
// Replaces the torch-op sequence of get_lidar_coor + voxel_pooling_prepare_v2
// (mmdet3d/models/necks/view_transformer_raw.py:121-158, 244-302), which is
// sort-bound and host-syncing.  Keys are dense small integers (voxel ranks), so
// the sort is a counting sort: the voxel histogram IS the run-length encoding.
//
//   k_voxel_keys   one lane per frustum point: coordinates (optional output),
//                  voxel, in-grid test, float32 rank -> key; histogram by atomics
//                  (one returning atomic per run of equal keys in a wave)
//   k_scan_*       exclusive scan of the histogram over all voxels (2 kernels:
//                  per-block sums, then every block adds up the sums before it
//                  and emits): the dense voxel table vstart (= the row pool
//                  kernels' index), interval_starts / interval_lengths, counts,
//                  the 64-voxel tile plan of the slab pool kernels; leaves the
//                  histogram zeroed for the next call (no memset node).  (A
//                  one-pass version -- blocks publishing totals through sc1 words
//                  and waiting for the blocks before them -- measured 19 us
//                  against 6 + 5 us for the two launches: cross-XCD round trips.)
//   k_scatter      points -> their voxel's slot range (slot = the arrival index
//                  the histogram atomic returned; no second atomic pass)
//   k_rank_in_bin  deterministic stable order inside every interval
//                  (ascending point index) by rank-counting
//
// Arithmetic follows the C oracle operation for operation (compiled with
// -ffp-contract=off): ((0 + m0*x) + m1*y) + m2*z for the 3x3 products,
// (coor - lower) / interval, truncation toward zero, float32 rank.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/veon_hip.h"

namespace {

constexpr int kBlock = 256;
constexpr int kTileV = 64;        // must match bev_pool_v2.hip
constexpr int kScanItems = 4;     // bins per thread in the scan kernels
constexpr int kScanBlock = kBlock * kScanItems;  // 1024 bins = 16 tiles

struct Geometry {
  const float* xs;             // [W] frustum pixel x
  const float* ys;             // [H] frustum pixel y
  const float* ds;             // [D] frustum depth
  const float* post_rots_inv;  // [B,N,3,3]
  const float* post_trans;     // [B,N,3]
  const float* combine;        // [B,N,3,3]
  const float* trans;          // [B,N,3]
  const float* bda;            // [B,3,3]
};

struct GridF {
  float lo[3];
  float step[3];
  float size[3];
};

__device__ __forceinline__ void mat3_vec(const float* __restrict__ m, float x,
                                         float y, float z, float* o) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float acc = 0.f;
    acc = acc + m[r * 3 + 0] * x;
    acc = acc + m[r * 3 + 1] * y;
    acc = acc + m[r * 3 + 2] * z;
    o[r] = acc;
  }
}

// view_transformer_raw.py:144-155
__device__ __forceinline__ void point_coor(const Geometry& g, int bn, int b,
                                           int d, int h, int w, float* o) {
  const float* pt = g.post_trans + bn * 3;
  const float fx = g.xs[w] - pt[0];
  const float fy = g.ys[h] - pt[1];
  const float fz = g.ds[d] - pt[2];
  float p[3], q[3];
  mat3_vec(g.post_rots_inv + bn * 9, fx, fy, fz, p);
  const float cx = p[0] * p[2], cy = p[1] * p[2], cz = p[2];
  mat3_vec(g.combine + bn * 9, cx, cy, cz, q);
  const float* tr = g.trans + bn * 3;
  q[0] += tr[0];
  q[1] += tr[1];
  q[2] += tr[2];
  mat3_vec(g.bda + b * 9, q[0], q[1], q[2], o);
}

// The (B,N) camera algebra of get_lidar_coor (:145, :151) without
// torch.inverse (rocSOLVER is not stream-capturable): inverses by the adjugate
// in double precision, rounded to float; the 3x3 product in float with the
// k-ascending accumulation torch's small bmm uses.
__device__ __forceinline__ void inv3_f64(const float* m, float* o) {
  const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5],
               g = m[6], h = m[7], i = m[8];
  const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const double det = a * A + b * B + c * C;
  const double r = 1.0 / det;
  o[0] = (float)(A * r);
  o[1] = (float)(-(b * i - c * h) * r);
  o[2] = (float)((b * f - c * e) * r);
  o[3] = (float)(B * r);
  o[4] = (float)((a * i - c * g) * r);
  o[5] = (float)(-(a * f - c * d) * r);
  o[6] = (float)(C * r);
  o[7] = (float)(-(a * h - b * g) * r);
  o[8] = (float)((a * e - b * d) * r);
}

__global__ __launch_bounds__(kBlock) void k_camera_matrices(
    int BN, const float* __restrict__ sensor2ego, const float* __restrict__ cam2imgs,
    const float* __restrict__ post_rots, float* __restrict__ post_rots_inv,
    float* __restrict__ combine, float* __restrict__ trans) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= BN) return;
  float pri[9], kin[9];
  inv3_f64(post_rots + i * 9, pri);
  inv3_f64(cam2imgs + i * 9, kin);
  const float* s = sensor2ego + i * 16;
  for (int r = 0; r < 3; ++r) {
    for (int cidx = 0; cidx < 3; ++cidx) {
      float acc = 0.f;
      acc = acc + s[r * 4 + 0] * kin[0 * 3 + cidx];
      acc = acc + s[r * 4 + 1] * kin[1 * 3 + cidx];
      acc = acc + s[r * 4 + 2] * kin[2 * 3 + cidx];
      combine[i * 9 + r * 3 + cidx] = acc;
    }
    trans[i * 3 + r] = s[r * 4 + 3];
  }
  for (int k = 0; k < 9; ++k) post_rots_inv[i * 9 + k] = pri[k];
}

// AlignNetOcc3D.prepare_meta (align_net_occ3d.py:328-352): every camera's
// sensor -> KEY-ego transform, global2keyego @ ego2global @ sensor2ego with
// global2keyego = inverse(ego2global of the first camera of the first frame), in double
// precision as the reference computes it, rounded to float once.  One lane per camera:
// fourteen tiny torch launches (LU factorisation, solves, batched matmuls, copies) in one.
__device__ __forceinline__ void inv4_f64(const double* m, double* o) {
  // Gauss-Jordan with partial pivoting on [m | I]
  double a[4][8];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      a[r][c] = m[r * 4 + c];
      a[r][4 + c] = r == c ? 1.0 : 0.0;
    }
  for (int col = 0; col < 4; ++col) {
    int piv = col;
    double best = fabs(a[col][col]);
    for (int r = col + 1; r < 4; ++r)
      if (fabs(a[r][col]) > best) {
        best = fabs(a[r][col]);
        piv = r;
      }
    if (piv != col)
      for (int c = 0; c < 8; ++c) {
        const double t = a[col][c];
        a[col][c] = a[piv][c];
        a[piv][c] = t;
      }
    const double d = 1.0 / a[col][col];
    for (int c = 0; c < 8; ++c) a[col][c] *= d;
    for (int r = 0; r < 4; ++r)
      if (r != col) {
        const double f = a[r][col];
        for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c];
      }
  }
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) o[r * 4 + c] = a[r][4 + c];
}

__global__ __launch_bounds__(64) void k_sensor2keyego(
    int B, int N, const float* __restrict__ sensor2ego, const float* __restrict__ ego2global,
    float* __restrict__ out) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= B * N) return;
  const int b = i / N;
  double key[16], inv[16], e2g[16], s2e[16], t[16];
  for (int k = 0; k < 16; ++k) {
    key[k] = ego2global[(int64_t)b * N * 16 + k];   // camera 0 of this sample
    e2g[k] = ego2global[(int64_t)i * 16 + k];
    s2e[k] = sensor2ego[(int64_t)i * 16 + k];
  }
  inv4_f64(key, inv);
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 4; ++k) acc += inv[r * 4 + k] * e2g[k * 4 + c];
      t[r * 4 + c] = acc;
    }
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 4; ++k) acc += t[r * 4 + k] * s2e[k * 4 + c];
      out[(int64_t)i * 16 + r * 4 + c] = (float)acc;
    }
}

__global__ __launch_bounds__(kBlock) void k_lidar_coor(
    Geometry g, int N, int D, int H, int W, float* __restrict__ coor) {
  const int bn = blockIdx.y;
  const int dhw = D * H * W;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= dhw) return;
  const int64_t p = (int64_t)bn * dhw + i;
  const int w = i % W;
  const int h = (i / W) % H;
  const int d = i / (W * H);
  float o[3];
  point_coor(g, bn, bn / N, d, h, w, o);
  coor[p * 3 + 0] = o[0];
  coor[p * 3 + 1] = o[1];
  coor[p * 3 + 2] = o[2];
}

// view_transformer_raw.py:267-286: voxel, filter, float32 rank.  key = -1 when
// the point is dropped.
__device__ __forceinline__ int voxel_key(const GridF& gr, const float* c,
                                         int64_t ib) {
  const float fx = (c[0] - gr.lo[0]) / gr.step[0];
  const float fy = (c[1] - gr.lo[1]) / gr.step[1];
  const float fz = (c[2] - gr.lo[2]) / gr.step[2];
  const long long ix = (long long)fx, iy = (long long)fy, iz = (long long)fz;
  const bool ok = (ix >= 0) && ((float)ix < gr.size[0]) && (iy >= 0) &&
                  ((float)iy < gr.size[1]) && (iz >= 0) &&
                  ((float)iz < gr.size[2]);
  if (!ok) return -1;
  float r = (float)ib * (gr.size[2] * gr.size[1] * gr.size[0]);
  r += (float)iz * (gr.size[1] * gr.size[0]);
  r += (float)iy * gr.size[0] + (float)ix;
  return (int)r;
}

// grid = (points of one camera / kBlock, B*N cameras): the camera index is
// workgroup-uniform, so the 3x3 matrices come in by scalar loads.  The
// returning histogram atomic doubles as the point's arrival slot in its voxel.
struct CameraRaw {  // FROM == 2: the reference's per-camera inputs
  const float* sensor2ego;  // [B,N,4,4]
  const float* cam2imgs;    // [B,N,3,3]
  const float* post_rots;   // [B,N,3,3]
};

// FROM: 0 = geometry from prepared matrices, 1 = given coordinates, 2 = geometry
// from the raw camera tensors (the algebra of k_camera_matrices runs once per
// workgroup into LDS: one launch less, same arithmetic, same bits).
template <int FROM>
__global__ __launch_bounds__(kBlock) void k_voxel_keys(
    Geometry g, CameraRaw cr, const float* __restrict__ coor, GridF gr, int N, int D,
    int H, int W, int64_t n_bins, int* __restrict__ keys, int* __restrict__ slots,
    int* __restrict__ hist, const float* __restrict__ depth_w, float depth_eps,
    const int2* __restrict__ win) {
  __shared__ float cam[21];  // post_rots_inv[9], combine[9], trans[3]
  const int bn = blockIdx.y;
  if constexpr (FROM == 2) {
    if (threadIdx.x == 0) {
      float pri[9], kin[9];
      inv3_f64(cr.post_rots + bn * 9, pri);
      inv3_f64(cr.cam2imgs + bn * 9, kin);
      const float* s = cr.sensor2ego + bn * 16;
      for (int r = 0; r < 3; ++r) {
        for (int cidx = 0; cidx < 3; ++cidx) {
          float acc = 0.f;
          acc = acc + s[r * 4 + 0] * kin[0 * 3 + cidx];
          acc = acc + s[r * 4 + 1] * kin[1 * 3 + cidx];
          acc = acc + s[r * 4 + 2] * kin[2 * 3 + cidx];
          cam[9 + r * 3 + cidx] = acc;
        }
        cam[18 + r] = s[r * 4 + 3];
      }
      for (int k = 0; k < 9; ++k) cam[k] = pri[k];
    }
    __syncthreads();
  }
  const int dhw = D * H * W;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const bool valid = i < dhw;   // no early return: the wave votes below
  const int ic = valid ? i : dhw - 1;
  const int64_t p = (int64_t)bn * dhw + ic;
  float c[3];
  if constexpr (FROM == 1) {
    c[0] = coor[p * 3 + 0];
    c[1] = coor[p * 3 + 1];
    c[2] = coor[p * 3 + 2];
  } else {
    const int w = ic % W;
    const int h = (ic / W) % H;
    const int d = ic / (W * H);
    if constexpr (FROM == 2) {
      // point_coor with this camera's matrices from LDS
      const float* pt = g.post_trans + bn * 3;
      const float fx = g.xs[w] - pt[0];
      const float fy = g.ys[h] - pt[1];
      const float fz = g.ds[d] - pt[2];
      float pp[3], q[3];
      mat3_vec(cam, fx, fy, fz, pp);
      const float cx = pp[0] * pp[2], cy = pp[1] * pp[2], cz = pp[2];
      mat3_vec(cam + 9, cx, cy, cz, q);
      q[0] += cam[18];
      q[1] += cam[19];
      q[2] += cam[20];
      mat3_vec(g.bda + (bn / N) * 9, q[0], q[1], q[2], c);
    } else {
      point_coor(g, bn, bn / N, d, h, w, c);
    }
  }
  int key = voxel_key(gr, c, bn / N);
  if (key >= n_bins) key = -1;  // cannot happen for consistent grids; be safe
  // opt-in sparse lift: a point whose depth weight is below the caller's threshold
  // (the clamped tail of VEON's soft two-hot depth, ~1e-7 per bin) is dropped here
  // and never sorted, ranked or pooled
  if (depth_w != nullptr && key >= 0 && depth_w[p] < depth_eps) key = -1;
  // two-hot lift by construction (csrc/depth_ops.hip k_two_hot_window): the depth
  // weights exist only as a per-pixel window + tail; a point is kept iff its bin is in
  // the pixel's kept window, or outside the unclamped window of a pixel whose tail
  // weight passed the threshold
  if (win != nullptr && key >= 0) {
    const int hw = H * W;
    const int k = ic / hw;
    const int2 wv = win[(int64_t)bn * hw + (ic - k * hw)];
    const int k0 = wv.x & 0xffff, nk = wv.x >> 16;
    const int q0 = wv.y & 0xffff, nq = (wv.y >> 16) & 0x7fff;
    const bool in_kept = (unsigned)(k - q0) < (unsigned)nq;
    const bool in_tail = (wv.y < 0) && !((unsigned)(k - k0) < (unsigned)nk);
    if (!(in_kept || in_tail)) key = -1;
  }
  if (!valid) key = -2;
  // Neighbouring pixels of one image row mostly fall into the same voxel: a run of
  // consecutive lanes with one key takes ONE returning atomic (by its first lane,
  // for the whole run) instead of one per point.
  const int lane = threadIdx.x & 63;
  const int prev = __shfl_up(key, 1);
  const bool head = lane == 0 || prev != key;
  const unsigned long long hm = __ballot(head);
  const unsigned long long upto = (2ull << lane) - 1ull;   // lanes 0..lane
  const int head_lane = 63 - __clzll((long long)(hm & upto));
  const unsigned long long above = hm & ~upto;
  const int next = above ? __ffsll((long long)above) - 1 : 64;
  int base = 0;
  if (head && key >= 0) base = atomicAdd(hist + key, next - head_lane);
  base = __shfl(base, head_lane);
  if (valid) {
    keys[p] = key;
    if (key >= 0) slots[p] = base + (lane - head_lane);
  }
}

// ---- exclusive scan over the histogram (count and non-empty flag) ----------
struct Pair {
  int pts;
  int ivs;
};

__device__ __forceinline__ Pair block_reduce(Pair v, Pair* sm) {
  for (int off = 32; off > 0; off >>= 1) {
    v.pts += __shfl_down(v.pts, off);
    v.ivs += __shfl_down(v.ivs, off);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sm[w] = v;
  __syncthreads();
  Pair t{0, 0};
  if (threadIdx.x == 0) {
    for (int i = 0; i < kBlock / 64; ++i) {
      t.pts += sm[i].pts;
      t.ivs += sm[i].ivs;
    }
  }
  return t;  // valid in thread 0
}

__global__ __launch_bounds__(kBlock) void k_scan_reduce(
    const int* __restrict__ hist, int64_t n_bins, Pair* __restrict__ block_sums) {
  __shared__ Pair sm[kBlock / 64];
  const int64_t base = (int64_t)blockIdx.x * kScanBlock + threadIdx.x * kScanItems;
  Pair v{0, 0};
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t i = base + k;
    const int c = i < n_bins ? hist[i] : 0;
    v.pts += c;
    v.ivs += c > 0;
  }
  const Pair t = block_reduce(v, sm);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = t;
}

// per block: offset = sum of the block sums before this block (a few hundred
// values, added up by the block itself: no separate single-block scan launch),
// local exclusive scan, then: vstart (dense voxel table, n_bins + 1 entries), the
// interval arrays, the plan entries of the block's 16 tiles, counts (last
// block); the histogram is left zeroed.
__global__ __launch_bounds__(kBlock) void k_scan_emit(
    int* __restrict__ hist, int64_t n_bins, const Pair* __restrict__ block_sums,
    int* __restrict__ vstart, int* __restrict__ interval_starts,
    int* __restrict__ interval_lengths, int4* __restrict__ plan,
    int64_t vpb, int64_t tiles_per_batch, int* __restrict__ counts) {
  __shared__ Pair sm[kBlock];
  __shared__ Pair red[kBlock / 64];
  __shared__ Pair boff_s;
  {
    Pair v{0, 0};
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += kBlock) {
      v.pts += block_sums[i].pts;
      v.ivs += block_sums[i].ivs;
    }
    const Pair t = block_reduce(v, red);
    if (threadIdx.x == 0) boff_s = t;
  }
  const int64_t base = (int64_t)blockIdx.x * kScanBlock + threadIdx.x * kScanItems;
  int c[kScanItems];
  Pair v{0, 0};
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t i = base + k;
    c[k] = i < n_bins ? hist[i] : 0;
    if (i < n_bins) hist[i] = 0;
    v.pts += c[k];
    v.ivs += c[k] > 0;
  }
  sm[threadIdx.x] = v;
  __syncthreads();
  for (int off = 1; off < kBlock; off <<= 1) {
    Pair add{0, 0};
    if (threadIdx.x >= off) add = sm[threadIdx.x - off];
    __syncthreads();
    sm[threadIdx.x].pts += add.pts;
    sm[threadIdx.x].ivs += add.ivs;
    __syncthreads();
  }
  const Pair boff = boff_s;
  Pair run{boff.pts + sm[threadIdx.x].pts - v.pts,
           boff.ivs + sm[threadIdx.x].ivs - v.ivs};
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t i = base + k;
    if (i < n_bins) {
      vstart[i] = run.pts;
      if (c[k] > 0) {
        interval_starts[run.ivs] = run.pts;
        interval_lengths[run.ivs] = c[k];
      }
    }
    run.pts += c[k];
    run.ivs += c[k] > 0;
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kBlock - 1) {
    counts[0] = run.pts;  // P_kept
    counts[1] = run.ivs;  // n_intervals
    vstart[n_bins] = run.pts;
  }
  // plan: tiles are 64 consecutive voxel ranks of ONE batch element.  When the
  // voxel count per batch is a multiple of 64 (and so of the scan block's tile
  // grid) a tile is 16 consecutive threads' bins.
  if (plan != nullptr) {
    // thread handles bins [base, base+4); tile = 16 threads
    const int tl = threadIdx.x & 15;  // position inside the tile
    // exclusive prefix at the tile's first thread and totals over the tile
    const int first = threadIdx.x - tl;
    const Pair pre{boff.pts + (first > 0 ? sm[first - 1].pts : 0),
                   boff.ivs + (first > 0 ? sm[first - 1].ivs : 0)};
    const Pair end{boff.pts + sm[first + 15].pts, boff.ivs + sm[first + 15].ivs};
    if (tl == 0) {
      const int64_t bin0 = (int64_t)blockIdx.x * kScanBlock + (int64_t)first * kScanItems;
      if (bin0 < n_bins) {
        const int64_t b = bin0 / vpb;
        const int64_t t = b * tiles_per_batch + (bin0 - b * vpb) / kTileV;
        plan[t] = make_int4(pre.ivs, end.ivs - pre.ivs, pre.pts, end.pts - pre.pts);
      }
    }
  }
}

// points -> slot range of their voxel, at the arrival slot the histogram atomic
// handed out (order inside the voxel is then fixed by k_rank_in_bin)
__global__ __launch_bounds__(kBlock) void k_scatter(
    const int* __restrict__ keys, const int* __restrict__ slots, int64_t P,
    const int* __restrict__ bin_start, int* __restrict__ tmp_point) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= P) return;
  const int key = keys[p];
  if (key < 0) return;
  tmp_point[bin_start[key] + slots[p]] = (int)p;
}

// stable order inside each interval: final slot = start + #points of the
// interval with a smaller index.  One lane per kept point; the slot range the
// block's 256 points' voxels cover is staged in LDS once (what does not fit -- a
// voxel with thousands of points -- is read from global memory).  Quadratic in the
// voxel's length: 498k points in ONE voxel take ~10 ms (tests/
// test_view_transformer_gpu.py::test_degenerate_single_voxel...).
constexpr int kRankSpan = 3072;

__global__ __launch_bounds__(kBlock) void k_rank_in_bin(
    const int* __restrict__ keys, const int* __restrict__ tmp_point,
    const int* __restrict__ counts, const int* __restrict__ bin_start, int D, int HW,
    int* __restrict__ ranks_bev,
    int* __restrict__ ranks_depth, int* __restrict__ ranks_feat,
    const int2* __restrict__ win, int K) {
  __shared__ int sp[kRankSpan];
  __shared__ int lo_s, hi_s;
  const int kept = counts[0];
  const int64_t q0 = (int64_t)blockIdx.x * kBlock;
  if (q0 >= kept) return;   // block-uniform
  const int64_t q = q0 + threadIdx.x;
  const bool valid = q < kept;
  int p = 0, key = 0, start = 0, len = 0;
  if (valid) {
    p = tmp_point[q];
    key = keys[p];
    start = bin_start[key];
    len = bin_start[key + 1] - start;
  }
  if (threadIdx.x == 0) lo_s = start;
  if (valid && (q == kept - 1 || threadIdx.x == kBlock - 1)) hi_s = start + len;
  __syncthreads();
  const int lo = lo_s;
  const int hi = min(hi_s, lo + kRankSpan);
  for (int i = threadIdx.x; i < hi - lo; i += kBlock) sp[i] = tmp_point[lo + i];
  __syncthreads();
  if (!valid) return;
  const int end = start + len;
  int rank = 0;
  const int s_end = min(end, hi) - lo;
  for (int i = start - lo; i < s_end; ++i) rank += sp[i] < p;
  for (int i = max(start, hi); i < end; ++i) rank += tmp_point[i] < p;
  const int slot = start + rank;
  ranks_bev[slot] = key;
  // pixel index (b,n,h,w) of point (b,n,d,h,w): view_transformer_raw.py:262-265
  const int pix = (p / (D * HW)) * HW + p % HW;
  ranks_feat[slot] = pix;
  if (win == nullptr) {
    ranks_depth[slot] = p;
  } else {
    // index into the compact weight table: slot 0 = the tail, 1 + j = window bin j
    const int k = (p / HW) % D;
    const int wx = win[pix].x;
    const int j = k - (wx & 0xffff);
    ranks_depth[slot] = pix * K + ((unsigned)j < (unsigned)(wx >> 16) ? 1 + j : 0);
  }
}

inline int launch_status() {
  return hipGetLastError() == hipSuccess ? VEON_OK : VEON_ERR_LAUNCH;
}

struct Workspace {
  int* keys;       // [P]
  int* slots;      // [P]  arrival index of the point inside its voxel
  int* tmp_point;  // [P]
  int* hist;       // [n_bins]      (zero on entry, left zero)
  int* bin_start;  // [n_bins + 1]  (used when the caller passes no vstart)
  Pair* block_sums;  // [n_scan_blocks]
  int64_t bytes;
};

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

inline Workspace carve(void* base, int64_t P, int64_t n_bins) {
  Workspace w;
  char* p = static_cast<char*>(base);
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    char* r = p ? p + off : nullptr;
    off += align_up(bytes, 256);
    return r;
  };
  w.keys = reinterpret_cast<int*>(take(P * 4));
  w.slots = reinterpret_cast<int*>(take(P * 4));
  w.tmp_point = reinterpret_cast<int*>(take(P * 4));
  w.hist = reinterpret_cast<int*>(take(n_bins * 4));
  w.bin_start = reinterpret_cast<int*>(take((n_bins + 1) * 4));
  const int64_t n_scan_blocks = (n_bins + kScanBlock - 1) / kScanBlock;
  w.block_sums = reinterpret_cast<Pair*>(take(n_scan_blocks * (int64_t)sizeof(Pair)));
  w.bytes = off;
  return w;
}

}  // namespace

extern "C" {

int veon_camera_matrices(int BN, const float* sensor2ego, const float* cam2imgs,
                         const float* post_rots, float* post_rots_inv,
                         float* combine, float* trans, void* stream) {
  if (BN <= 0 || !sensor2ego || !cam2imgs || !post_rots || !post_rots_inv ||
      !combine || !trans)
    return VEON_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_camera_matrices, dim3((BN + kBlock - 1) / kBlock),
                     dim3(kBlock), 0, static_cast<hipStream_t>(stream), BN,
                     sensor2ego, cam2imgs, post_rots, post_rots_inv, combine,
                     trans);
  return launch_status();
}

int veon_sensor2keyego(int B, int N, const float* sensor2ego, const float* ego2global,
                       float* out, void* stream) {
  if (B <= 0 || N <= 0 || !sensor2ego || !ego2global || !out) return VEON_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_sensor2keyego, dim3((unsigned)((B * N + 63) / 64)), dim3(64), 0,
                     static_cast<hipStream_t>(stream), B, N, sensor2ego, ego2global, out);
  return launch_status();
}

int veon_lidar_coor(int B, int N, int D, int H, int W, const float* xs,
                    const float* ys, const float* ds,
                    const float* post_rots_inv, const float* post_trans,
                    const float* combine, const float* trans, const float* bda,
                    float* coor, void* stream) {
  if (B <= 0 || N <= 0 || D <= 0 || H <= 0 || W <= 0) return VEON_ERR_BAD_ARG;
  if (!xs || !ys || !ds || !post_rots_inv || !post_trans || !combine || !trans ||
      !bda || !coor)
    return VEON_ERR_BAD_ARG;
  const int64_t P = (int64_t)B * N * D * H * W;
  if (P > 0x7fffffffLL || B * N > 65535) return VEON_ERR_BAD_ARG;
  Geometry g{xs, ys, ds, post_rots_inv, post_trans, combine, trans, bda};
  const int64_t dhw = (int64_t)D * H * W;
  hipLaunchKernelGGL(k_lidar_coor,
                     dim3((unsigned)((dhw + kBlock - 1) / kBlock), (unsigned)(B * N)),
                     dim3(kBlock), 0, static_cast<hipStream_t>(stream), g, N, D,
                     H, W, coor);
  return launch_status();
}

int64_t veon_lss_prepare_workspace_bytes(int64_t num_points,
                                         int64_t num_voxels_total) {
  if (num_points <= 0 || num_voxels_total <= 0) return 0;
  return carve(nullptr, num_points, num_voxels_total).bytes;
}

static int prepare_impl(int B, int N, int D, int H, int W, const float* coor,
                        const float* xs, const float* ys, const float* ds,
                        const float* post_rots_inv, const float* post_trans,
                        const float* combine, const float* trans, const float* bda,
                        const float* sensor2ego, const float* cam2imgs,
                        const float* post_rots, const float* grid_lower,
                        const float* grid_interval, const float* grid_size,
                        int64_t voxels_per_batch, void* workspace,
                        int64_t workspace_bytes, int hist_is_zero, int* ranks_bev,
                        int* ranks_depth, int* ranks_feat, int* interval_starts,
                        int* interval_lengths, int* plan, int* vstart, int* counts,
                        const float* depth_w, float depth_eps, const int* win_i, int K,
                        void* stream) {
  if (B <= 0 || N <= 0 || D <= 0 || H <= 0 || W <= 0 || voxels_per_batch <= 0)
    return VEON_ERR_BAD_ARG;
  if (!grid_lower || !grid_interval || !grid_size || !workspace || !ranks_bev ||
      !ranks_depth || !ranks_feat || !interval_starts || !interval_lengths ||
      !counts)
    return VEON_ERR_BAD_ARG;
  const bool raw = sensor2ego != nullptr;
  if (!coor) {
    if (!xs || !ys || !ds || !post_trans || !bda) return VEON_ERR_BAD_ARG;
    if (raw ? (!cam2imgs || !post_rots) : (!post_rots_inv || !combine || !trans))
      return VEON_ERR_BAD_ARG;
  }
  const int64_t P = (int64_t)B * N * D * H * W;
  const int64_t n_bins = voxels_per_batch * B;
  if (P > 0x7fffffffLL || n_bins > 0x7ffffffeLL || B * N > 65535)
    return VEON_ERR_BAD_ARG;
  // the plan emitted by the scan needs tiles aligned to the scan blocks
  if (plan && (voxels_per_batch % kTileV != 0)) return VEON_ERR_BAD_ARG;
  if (plan && (reinterpret_cast<uintptr_t>(plan) & 15u)) return VEON_ERR_BAD_ARG;
  const Workspace w = carve(workspace, P, n_bins);
  if (w.bytes > workspace_bytes) return VEON_ERR_WORKSPACE;
  const int2* win = reinterpret_cast<const int2*>(win_i);
  if (win && (K <= 0 || coor || (reinterpret_cast<uintptr_t>(win) & 7u) ||
              (int64_t)B * N * H * W * K > 0x7fffffffLL))
    return VEON_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  GridF gr;
  for (int i = 0; i < 3; ++i) {
    gr.lo[i] = grid_lower[i];
    gr.step[i] = grid_interval[i];
    gr.size[i] = grid_size[i];
  }
  Geometry g{xs, ys, ds, post_rots_inv, post_trans, combine, trans, bda};
  CameraRaw cr{sensor2ego, cam2imgs, post_rots};
  const int n_scan_blocks = (int)((n_bins + kScanBlock - 1) / kScanBlock);
  if (!hist_is_zero &&
      hipMemsetAsync(w.hist, 0, (size_t)n_bins * 4, s) != hipSuccess)
    return VEON_ERR_LAUNCH;
  int* table = vstart ? vstart : w.bin_start;
  const unsigned pb = (unsigned)((P + kBlock - 1) / kBlock);
  const int64_t dhw = (int64_t)D * H * W;
  const dim3 kgrid((unsigned)((dhw + kBlock - 1) / kBlock), (unsigned)(B * N));
  if (coor)
    hipLaunchKernelGGL(k_voxel_keys<1>, kgrid, dim3(kBlock), 0, s, g, cr, coor, gr, N,
                       D, H, W, n_bins, w.keys, w.slots, w.hist, depth_w, depth_eps, win);
  else if (raw)
    hipLaunchKernelGGL(k_voxel_keys<2>, kgrid, dim3(kBlock), 0, s, g, cr, coor, gr, N,
                       D, H, W, n_bins, w.keys, w.slots, w.hist, depth_w, depth_eps, win);
  else
    hipLaunchKernelGGL(k_voxel_keys<0>, kgrid, dim3(kBlock), 0, s, g, cr, coor, gr, N,
                       D, H, W, n_bins, w.keys, w.slots, w.hist, depth_w, depth_eps, win);
  const int64_t tiles_per_batch = voxels_per_batch / kTileV;
  hipLaunchKernelGGL(k_scan_reduce, dim3(n_scan_blocks), dim3(kBlock), 0, s,
                     w.hist, n_bins, w.block_sums);
  hipLaunchKernelGGL(k_scan_emit, dim3(n_scan_blocks), dim3(kBlock), 0, s, w.hist,
                     n_bins, w.block_sums, table, interval_starts, interval_lengths,
                     reinterpret_cast<int4*>(plan), voxels_per_batch, tiles_per_batch,
                     counts);
  hipLaunchKernelGGL(k_scatter, dim3(pb), dim3(kBlock), 0, s, w.keys, w.slots, P,
                     table, w.tmp_point);
  hipLaunchKernelGGL(k_rank_in_bin, dim3(pb), dim3(kBlock), 0, s, w.keys,
                     w.tmp_point, counts, table, D, H * W, ranks_bev, ranks_depth,
                     ranks_feat, win, K);
  return launch_status();
}

int veon_lss_prepare(int B, int N, int D, int H, int W, const float* coor,
                     const float* xs, const float* ys, const float* ds,
                     const float* post_rots_inv, const float* post_trans,
                     const float* combine, const float* trans, const float* bda,
                     const float* grid_lower, const float* grid_interval,
                     const float* grid_size, int64_t voxels_per_batch,
                     void* workspace, int64_t workspace_bytes, int* ranks_bev,
                     int* ranks_depth, int* ranks_feat, int* interval_starts,
                     int* interval_lengths, int* plan, int* counts,
                     void* stream) {
  return prepare_impl(B, N, D, H, W, coor, xs, ys, ds, post_rots_inv, post_trans,
                      combine, trans, bda, nullptr, nullptr, nullptr, grid_lower,
                      grid_interval, grid_size, voxels_per_batch, workspace,
                      workspace_bytes, 0, ranks_bev, ranks_depth, ranks_feat,
                      interval_starts, interval_lengths, plan, nullptr, counts, nullptr,
                      0.f, nullptr, 0, stream);
}

int veon_lss_prepare_cameras(int B, int N, int D, int H, int W, const float* xs,
                             const float* ys, const float* ds,
                             const float* sensor2ego, const float* cam2imgs,
                             const float* post_rots, const float* post_trans,
                             const float* bda, const float* grid_lower,
                             const float* grid_interval, const float* grid_size,
                             int64_t voxels_per_batch, void* workspace,
                             int64_t workspace_bytes, int hist_is_zero,
                             int* ranks_bev, int* ranks_depth, int* ranks_feat,
                             int* interval_starts, int* interval_lengths, int* plan,
                             int* vstart, int* counts, void* stream) {
  if (!sensor2ego) return VEON_ERR_BAD_ARG;
  return prepare_impl(B, N, D, H, W, nullptr, xs, ys, ds, nullptr, post_trans, nullptr,
                      nullptr, bda, sensor2ego, cam2imgs, post_rots, grid_lower,
                      grid_interval, grid_size, voxels_per_batch, workspace,
                      workspace_bytes, hist_is_zero, ranks_bev, ranks_depth,
                      ranks_feat, interval_starts, interval_lengths, plan, vstart,
                      counts, nullptr, 0.f, nullptr, 0, stream);
}

int veon_lss_prepare_cameras_sparse(
    int B, int N, int D, int H, int W, const float* xs, const float* ys, const float* ds,
    const float* sensor2ego, const float* cam2imgs, const float* post_rots,
    const float* post_trans, const float* bda, const float* grid_lower,
    const float* grid_interval, const float* grid_size, int64_t voxels_per_batch,
    void* workspace, int64_t workspace_bytes, int hist_is_zero, int* ranks_bev,
    int* ranks_depth, int* ranks_feat, int* interval_starts, int* interval_lengths,
    int* plan, int* vstart, int* counts, const float* depth_weights, float depth_eps,
    void* stream) {
  if (!sensor2ego || !depth_weights) return VEON_ERR_BAD_ARG;
  return prepare_impl(B, N, D, H, W, nullptr, xs, ys, ds, nullptr, post_trans, nullptr,
                      nullptr, bda, sensor2ego, cam2imgs, post_rots, grid_lower,
                      grid_interval, grid_size, voxels_per_batch, workspace,
                      workspace_bytes, hist_is_zero, ranks_bev, ranks_depth,
                      ranks_feat, interval_starts, interval_lengths, plan, vstart,
                      counts, depth_weights, depth_eps, nullptr, 0, stream);
}

int veon_lss_prepare_cameras_twohot(
    int B, int N, int D, int H, int W, const float* xs, const float* ys, const float* ds,
    const float* sensor2ego, const float* cam2imgs, const float* post_rots,
    const float* post_trans, const float* bda, const float* grid_lower,
    const float* grid_interval, const float* grid_size, int64_t voxels_per_batch,
    void* workspace, int64_t workspace_bytes, int hist_is_zero, int* ranks_bev,
    int* ranks_depth, int* ranks_feat, int* interval_starts, int* interval_lengths,
    int* plan, int* vstart, int* counts, const int* win, int window_slots, void* stream) {
  if (!sensor2ego || !win || window_slots <= 0) return VEON_ERR_BAD_ARG;
  return prepare_impl(B, N, D, H, W, nullptr, xs, ys, ds, nullptr, post_trans, nullptr,
                      nullptr, bda, sensor2ego, cam2imgs, post_rots, grid_lower,
                      grid_interval, grid_size, voxels_per_batch, workspace,
                      workspace_bytes, hist_is_zero, ranks_bev, ranks_depth,
                      ranks_feat, interval_starts, interval_lengths, plan, vstart,
                      counts, nullptr, 0.f, win, window_slots, stream);
}

}  // extern "C"
