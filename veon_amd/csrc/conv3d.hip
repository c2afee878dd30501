// 3x3x3 Conv3d body of VEON's 3D alignment network on the MI355X matrix cores.
//
// The step right after the lift: AlignNetOcc3D runs
// ResBlock3D x N on the max-pooled lift volume
// (mmdet3d/models/semantic_net/side_adapter/align_net_occ3d.py:224-228, 363-399:
// Conv3d 3x3x3 pad 1 no bias -> BN3d -> ReLU -> Conv3d -> BN3d, + identity,
// ReLU; 256 -> 256 channels on 8 x 100 x 100 voxels, 283 GFLOP per conv).
//
// Implicit GEMM without an im2col and without boundary tests: the volume lives
// channels-last in a ZERO-PADDED grid [B][Z+2][Y+2][X+2][C] (bf16).  In the
// linear row index m of that grid every filter tap is a constant row offset
//     off(dz,dy,dx) = ((dz-1)*(Y+2) + (dy-1))*(X+2) + (dx-1),
// so   out[m][n] = sum_tap sum_c in[m + off(tap)][c] * w[n][tap][c]
// is the dense GEMM of vit_block.hip whose activation slab simply starts at a
// different row for every group of Cin/64 k-steps.  Outputs at halo rows are
// written as zeros, so a conv's output buffer is directly the next conv's
// padded input; tiles that lie entirely in a z-halo plane skip the contraction.
// The cost is the halo's share of the rows that do run (4 % at 8x100x100).
// Rows m + off can leave [0, M) for halo m only; the caller provides
// veon_conv3d_guard_rows() readable rows before and after the grid instead of
// per-lane clamps in the k-loop.
//
// Epilogue (fused, fp32): y = acc*scale[n] + shift[n] (BatchNorm3d in eval mode),
// optional + residual (bf16, same padded layout), optional ReLU or GELU(erf),
// -> bf16.
#include "mfma_common.h"

namespace {

constexpr int CBK = 64;

// Optional extras of the epilogue (both null in the 3-D body): a SECOND residual image
// added before the activation, and a second output that receives relu(result) -- the
// DPT fusion blocks' "x0 + RCU1(x1)" and the ReLU every ResidualConvUnit applies to its
// input (util/blocks.py:49-83), which otherwise cost an elementwise pass each.
struct ConvExtra {
  const bf16_t* resid2;
  bf16_t* out_relu;
};

// Tile = (WM*16*MT) voxels x (64*WN) features, WM x WN waves of (16*MT) x 64 each.
template <int WM, int WN, int MT, int ACT, bool RESID>
__global__ __launch_bounds__(64 * WM * WN) void k_conv3d_k3(
    const bf16_t* __restrict__ in, const bf16_t* __restrict__ W,
    const float* __restrict__ scale, const float* __restrict__ shift,
    const bf16_t* __restrict__ resid, bf16_t* __restrict__ out, int planes,
    int Zp, int Yp, int Xp, int Cin, int Cout, int kd, int abl, int stride, int Ypi,
    int Xpi, ConvExtra ex) {
  // stride 2 (kd == 1 only): planes / Yp / Xp describe the OUTPUT grid, Ypi / Xpi the
  // padded input image; output pixel (y, x) reads the 3x3 neighbourhood of input
  // pixel (2y, 2x).  The DMA source of a tile row is a per-lane pointer anyway, so the
  // row map costs nothing in the k-loop.  stride 1: Ypi == Yp, Xpi == Xp.
  constexpr int BM = WM * 16 * MT;
  constexpr int CBN = 64 * WN;
  constexpr int NW = WM * WN;              // waves
  constexpr int NT = 64 * NW;              // threads
  // DMA pieces (8 rows = 1 KiB each) are dealt round-robin to the waves:
  // wave w issues pieces w, w + NW, ... of each slab
  constexpr int APIECES = BM / 8, WPIECES = CBN / 8;
  constexpr int AP = (APIECES + NW - 1) / NW;  // per wave, last may be absent
  constexpr int WP = (WPIECES + NW - 1) / NW;
  constexpr int A_ELEMS = BM * CBK;
  constexpr int CW_ELEMS = CBN * CBK;
  constexpr int BUF_ELEMS = A_ELEMS + CW_ELEMS;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];  // [2][A|W]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fg = lane >> 4;
  const int YX = Yp * Xp;
  const int M = planes * YX;  // planes = B * Zp
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * CBN;
  // kd = 3: 3x3x3 taps on a grid padded in z, y and x; kd = 1: the 2-D case,
  // 3x3 taps on [B][Yp][Xp] images (Zp = planes per image = 1, no z halo)
  const int ntaps = 9 * kd;
  const int K = ntaps * Cin;

  // tile entirely inside z-halo planes: nothing to contract, store zeros
  if (kd == 3) {
    const int p0 = m0 / YX;
    const int mlast = (m0 + BM - 1 < M ? m0 + BM - 1 : M - 1);
    const int p1 = mlast / YX;
    const int z0 = p0 % Zp, z1 = p1 % Zp;
    const bool h0 = z0 == 0 || z0 == Zp - 1, h1 = z1 == 0 || z1 == Zp - 1;
    if (h0 && h1 && p1 - p0 <= 1) {
      // BM rows x CBN features of bf16, 16 bytes per lane-store
      constexpr int CPR = CBN / 8;  // 16-byte chunks per row
      for (int i = tid; i < BM * CPR; i += NT) {
        const int r = i / CPR, c8 = (i % CPR) * 8;
        if (m0 + r < M && n0 + c8 < Cout)
          *reinterpret_cast<uint4*>(out + (int64_t)(m0 + r) * Cout + n0 + c8) =
              make_uint4(0u, 0u, 0u, 0u);
      }
      return;
    }
  }

  // DMA map as k_gemm_bf16: a wave instruction fills 8 rows of a slab; lane l
  // lands in row r = 8*piece + l/8, physical chunk l%8, and fetches logical
  // chunk (l%8) ^ (r&7).  Activation rows are NOT clamped (guard rows).
  const bf16_t* srcA[AP];
  const bf16_t* srcW[WP];
#pragma unroll
  for (int j = 0; j < AP; ++j) {
    const int r = (wave + j * NW) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (r & 7);
    int64_t row = m0 + r;
    if (stride != 1) {
      const int mm = m0 + r < M ? m0 + r : M - 1;
      const int p = mm / YX, rem = mm - p * YX;
      const int yo = rem / Xp, xo = rem - yo * Xp;
      int yi = stride * (yo - 1) + 1, xi = stride * (xo - 1) + 1;   // padded coordinates
      yi = yi < 1 ? 1 : (yi > Ypi - 2 ? Ypi - 2 : yi);   // halo rows: any valid pixel
      xi = xi < 1 ? 1 : (xi > Xpi - 2 ? Xpi - 2 : xi);   // (their result is zeroed)
      row = ((int64_t)p * Ypi + yi) * Xpi + xi;
    }
    srcA[j] = in + row * Cin + c * 8;
  }
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    const int r = (wave + j * NW) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz_w(r);
    const int gn = n0 + r < Cout ? n0 + r : Cout - 1;
    srcW[j] = W + (int64_t)gn * K + c * 8;
  }
  const int cpk = Cin / CBK;  // k-steps per filter tap
  auto a_off = [&](int kt) -> int64_t {
    const int tap = kt / cpk;
    const int cc = (kt - tap * cpk) * CBK;
    const int dz = kd == 3 ? tap / 9 - 1 : 0, dy = (tap / 3) % 3, dx = tap % 3;
    const int off = (dz * Ypi + (dy - 1)) * Xpi + (dx - 1);
    return (int64_t)off * Cin + cc;
  };
  auto dma = [&](int buf, int kt) {
    bf16_t* dA = smem + buf * BUF_ELEMS;
    bf16_t* dW = dA + A_ELEMS;
    const int64_t ao = a_off(kt);
    // ablation (tools/body_bench.py, wrong results): 1 = activation slab only for
    // every third tap, 2 = weight slab only once
    const bool doA = !(abl & 1) || (kt / cpk) % 3 == 0;
    const bool doW = !(abl & 2) || kt == 0;
#pragma unroll
    for (int j = 0; j < AP; ++j)
      if (doA && wave + j * NW < APIECES)  // wave-uniform
        __builtin_amdgcn_global_load_lds((gptr_t)(srcA[j] + ao),
                                         (lptr_t)(dA + (wave + j * NW) * 512), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < WP; ++j)
      if (doW && wave + j * NW < WPIECES)
        __builtin_amdgcn_global_load_lds((gptr_t)(srcW[j] + kt * CBK),
                                         (lptr_t)(dW + (wave + j * NW) * 512), 16, 0, 0);
  };

  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int offA[MT], offW[4];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int ra = wm * (16 * MT) + i * 16 + fr;
    offA[i] = ra * CBK + ((fg ^ (ra & 7)) * 8);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // tiles are paired, weight rows interleaved in groups of four: the accumulator
    // rows 4 fg .. 4 fg + 3 of tiles 2p and 2p + 1 are the 8 consecutive features
    // 32 p + 8 fg + 0..7 (16-byte epilogue accesses instead of 8-byte ones)
    const int rw = wn * 64 + (i / 2) * 32 + (fr / 4) * 8 + (i & 1) * 4 + (fr & 3);
    offW[i] = rw * CBK + ((fg ^ swz_w(rw)) * 8);
  }

  const int nk = ntaps * cpk;
  auto compute = [&](int buf) {
    const bf16_t* tA = smem + buf * BUF_ELEMS;
    const bf16_t* tW = tA + A_ELEMS;
#pragma unroll
    for (int ks = 0; ks < CBK / 32; ++ks) {
      bf16x8 fa[MT], fw[4];
#pragma unroll
      for (int i = 0; i < MT; ++i)
        fa[i] = *reinterpret_cast<const bf16x8*>(tA + (offA[i] ^ (ks * 32)));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        fw[j] = *reinterpret_cast<const bf16x8*>(tW + (offW[j] ^ (ks * 32)));
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          // acc[i][j][reg] = C[voxel i*16 + fr][feature j*16 + 4*fg + reg]
          acc[i][j] = mfma_16x16x32(fw[j], fa[i],
                                                               acc[i][j]);
    }
  };
  dma(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) dma(buf ^ 1, kt + 1);
    if (!(abl & 4)) compute(buf);
    __syncthreads();  // next slab landed (vmcnt drained) and this one released
  }

  // epilogue: lane owns 8 consecutive features of one voxel per tile pair
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + wm * (16 * MT) + i * 16 + fr;
    if (m >= M) continue;
    const int p = m / YX, rem = m - p * YX;
    const int y = rem / Xp, x = rem - y * Xp, z = p % Zp;
    const bool interior = (kd == 1 || (z >= 1 && z <= Zp - 2)) && y >= 1 &&
                          y <= Yp - 2 && x >= 1 && x <= Xp - 2;
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
      const int n = n0 + wn * 64 + p2 * 32 + fg * 8;  // Cout % 8 == 0
      if (n >= Cout) continue;
      const f32x4 a0 = acc[i][2 * p2], a1 = acc[i][2 * p2 + 1];
      float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      if (scale != nullptr) {
        const float4 s0 = *reinterpret_cast<const float4*>(scale + n);
        const float4 s1 = *reinterpret_cast<const float4*>(scale + n + 4);
        v[0] *= s0.x; v[1] *= s0.y; v[2] *= s0.z; v[3] *= s0.w;
        v[4] *= s1.x; v[5] *= s1.y; v[6] *= s1.z; v[7] *= s1.w;
      }
      if (shift != nullptr) {
        const float4 b0 = *reinterpret_cast<const float4*>(shift + n);
        const float4 b1 = *reinterpret_cast<const float4*>(shift + n + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
        v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      }
      if (RESID) {
        const bf16x8 r8 =
            *reinterpret_cast<const bf16x8*>(resid + (int64_t)m * Cout + n);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += bf2f((bf16_t)r8[k]);
      }
      if (ex.resid2 != nullptr) {
        const bf16x8 r8 =
            *reinterpret_cast<const bf16x8*>(ex.resid2 + (int64_t)m * Cout + n);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += bf2f((bf16_t)r8[k]);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (ACT == 1) v[k] = fmaxf(v[k], 0.f);
        if (ACT == 2) v[k] = gelu_erf(v[k]);
        if (!interior) v[k] = 0.f;
      }
      const uint4 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]),
                       pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
      *reinterpret_cast<uint4*>(out + (int64_t)m * Cout + n) = o;
      if (ex.out_relu != nullptr) {
        // relu of the ROUNDED result: what a separate pass over `out` would produce
        const uint4 q = {pack_bf16(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)),
                         pack_bf16(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)),
                         pack_bf16(fmaxf(v[4], 0.f), fmaxf(v[5], 0.f)),
                         pack_bf16(fmaxf(v[6], 0.f), fmaxf(v[7], 0.f))};
        *reinterpret_cast<uint4*>(ex.out_relu + (int64_t)m * Cout + n) = q;
      }
    }
  }
}

// The same tile with the activation slab shared by the three x-taps of a filter
// row: the taps (dz, dy, -1 / 0 / +1) read rows m + off - 1, m + off, m + off + 1 of
// the padded grid, i.e. ONE slab of BM + 2 rows shifted by a row.  The slab is
// fetched once per (dz, dy, channel chunk) and the MFMA operand reads step through it
// (the XOR swizzle is keyed on the slab row, so a shifted read stays conflict-free);
// only the weight slab changes per tap.  L2 -> LDS traffic per three k-steps drops
// from 3 (BM + BN) to (BM + 2) + 3 BN rows: 1.6x less at 336 x 256 -- the loads
// alone took 183 us of this kernel's 268 at the body's shape (tools/body_bench.py
// ablations), the arithmetic alone 220.
// Tile = (WM*16*MT) voxels x (64*WN) features, WM x WN waves of (16*MT) x 64 each.
template <int WM, int WN, int MT, int ACT, bool RESID>
__global__ __launch_bounds__(64 * WM * WN) void k_conv3d_k3_ax(
    const bf16_t* __restrict__ in, const bf16_t* __restrict__ W,
    const float* __restrict__ scale, const float* __restrict__ shift,
    const bf16_t* __restrict__ resid, bf16_t* __restrict__ out, int planes,
    int Zp, int Yp, int Xp, int Cin, int Cout, int kd, ConvExtra ex) {
  constexpr int BM = WM * 16 * MT;
  constexpr int CBN = 64 * WN;
  constexpr int NW = WM * WN;              // waves
  constexpr int NT = 64 * NW;              // threads
  // DMA pieces (8 rows = 1 KiB each) are dealt round-robin to the waves:
  // wave w issues pieces w, w + NW, ... of each slab
  constexpr int AROWS = BM + 8;   // BM + 2 used, rounded up to whole 8-row pieces
  constexpr int APIECES = AROWS / 8, WPIECES = CBN / 8;
  constexpr int AP = (APIECES + NW - 1) / NW;  // per wave, last may be absent
  constexpr int WP = (WPIECES + NW - 1) / NW;
  constexpr int A_ELEMS = AROWS * CBK;
  constexpr int CW_ELEMS = CBN * CBK;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];  // [2][A] [2][W]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fg = lane >> 4;
  const int YX = Yp * Xp;
  const int M = planes * YX;  // planes = B * Zp
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * CBN;
  // kd = 3: 3x3x3 taps on a grid padded in z, y and x; kd = 1: the 2-D case,
  // 3x3 taps on [B][Yp][Xp] images (Zp = planes per image = 1, no z halo)
  const int K = 9 * kd * Cin;

  // tile entirely inside z-halo planes: nothing to contract, store zeros
  if (kd == 3) {
    const int p0 = m0 / YX;
    const int mlast = (m0 + BM - 1 < M ? m0 + BM - 1 : M - 1);
    const int p1 = mlast / YX;
    const int z0 = p0 % Zp, z1 = p1 % Zp;
    const bool h0 = z0 == 0 || z0 == Zp - 1, h1 = z1 == 0 || z1 == Zp - 1;
    if (h0 && h1 && p1 - p0 <= 1) {
      // BM rows x CBN features of bf16, 16 bytes per lane-store
      constexpr int CPR = CBN / 8;  // 16-byte chunks per row
      for (int i = tid; i < BM * CPR; i += NT) {
        const int r = i / CPR, c8 = (i % CPR) * 8;
        if (m0 + r < M && n0 + c8 < Cout)
          *reinterpret_cast<uint4*>(out + (int64_t)(m0 + r) * Cout + n0 + c8) =
              make_uint4(0u, 0u, 0u, 0u);
      }
      return;
    }
  }

  // DMA map as k_gemm_bf16: a wave instruction fills 8 rows of a slab; lane l
  // lands in row r = 8*piece + l/8, physical chunk l%8, and fetches logical
  // chunk (l%8) ^ (r&7).  Activation rows are NOT clamped (guard rows).
  // The DMA is buffer_load ... lds (MUBUF), not global_load_lds: the latter is a
  // FLAT-encoded instruction that may touch LDS, after which hipcc's waitcnt pass
  // turns every counted s_waitcnt lgkmcnt(N) of the fragment pipeline below into
  // lgkmcnt(0) ("pending flat").  Per-lane 32-bit BYTE offsets in VGPRs, the slab's
  // tap offset in an SGPR (the launcher checks the extents: < 2^31 elements).  The
  // activation descriptor starts ((Yp + 1) Xp + 2) rows before `in` (inside the
  // guard rows), so that the most negative tap offset is still a positive offset.
  // (the slab starts one row before the tile, m0 - 1: that row is part of the bias
  // too, so that the per-lane offsets and the tap offsets are all non-negative)
  const int abias = ((Yp + 1) * Xp + 2) * Cin;
  const rsrc_t rsA = make_rsrc(in - abias);
  const rsrc_t rsW = make_rsrc(W);
  int srcA[AP], srcW[WP];
#pragma unroll
  for (int j = 0; j < AP; ++j) {
    const int r = (wave + j * NW) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (r & 7);
    srcA[j] = 2 * ((m0 + r) * Cin + c * 8);
  }
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    const int r = (wave + j * NW) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz_w(r);
    const int gn = n0 + r < Cout ? n0 + r : Cout - 1;
    srcW[j] = 2 * (gn * K + c * 8);
  }
  const int cpk = Cin / CBK;  // channel chunks
  // group g = (dz, dy, channel chunk): one activation slab, three weight slabs
  const int ngroups = 3 * kd * cpk;
  auto dma_a = [&](int g) {
    bf16_t* dA = smem + (g & 1) * A_ELEMS;
    const int zy = g / cpk, cc = (g - zy * cpk) * CBK;
    const int dz = kd == 3 ? zy / 3 - 1 : 0, dy = zy % 3 - 1;
    const int ao = 2 * (((dz * Yp + dy) * Xp - 1) * Cin + cc + abias);
#pragma unroll
    for (int j = 0; j < AP; ++j)
      if (wave + j * NW < APIECES)  // wave-uniform
        buffer_load_lds16(rsA, (lptr_t)(dA + (wave + j * NW) * 512), srcA[j], ao);
  };
  auto dma_w = [&](int st) {   // st = 3 g + dx
    bf16_t* dW = smem + 2 * A_ELEMS + (st & 1) * CW_ELEMS;
    const int g = st / 3, dx = st - g * 3;
    const int zy = g / cpk, cc = (g - zy * cpk) * CBK;
    const int wk = 2 * ((zy * 3 + dx) * Cin + cc);
#pragma unroll
    for (int j = 0; j < WP; ++j)
      if (wave + j * NW < WPIECES)
        buffer_load_lds16(rsW, (lptr_t)(dW + (wave + j * NW) * 512), srcW[j], wk);
  };

  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int offW[4];
  const int ra0 = wm * (16 * MT) + fr;   // + 16 i + dx: the slab row of this lane
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // tiles are paired, weight rows interleaved in groups of four: the accumulator
    // rows 4 fg .. 4 fg + 3 of tiles 2p and 2p + 1 are the 8 consecutive features
    // 32 p + 8 fg + 0..7 (16-byte epilogue accesses instead of 8-byte ones)
    const int rw = wn * 64 + (i / 2) * 32 + (fr / 4) * 8 + (i & 1) * 4 + (fr & 3);
    offW[i] = rw * CBK + ((fg ^ swz_w(rw)) * 8);
  }

  // One k-step = 2 * MT activation fragments against 2 x 4 weight fragments.  The
  // activation fragments run through a ring of kPD + 1 registers: the read of
  // fragment q + kPD is issued BEFORE the four MFMAs of fragment q, so its LDS
  // latency passes under 4 * kPD MFMAs (the compiler's own order was "read two,
  // s_waitcnt lgkmcnt(0), eight MFMAs": every read fully exposed).  The weight
  // fragments of the second half are re-read one by one as the last MFMAs of the
  // first half release them.  sched_barrier pins the order; the waits the compiler
  // inserts are counted (LDS returns in order).
  constexpr int kPD = 2;
  constexpr int NQ = (CBK / 32) * MT;
  auto compute = [&](int g, int st, int dx) {
    const bf16_t* tA = smem + (g & 1) * A_ELEMS;
    const bf16_t* tW = smem + 2 * A_ELEMS + (st & 1) * CW_ELEMS;
    const int ra = ra0 + dx;   // + 16 i: the low three bits of the row do not change
    const bf16_t* pA = tA + (ra * CBK + ((fg ^ (ra & 7)) * 8));
    auto lda = [&](int q) {   // q = ks * MT + i, compile-time after unrolling
      const int ks = q / MT, i = q - ks * MT;
      // (x + i * 16 * CBK) ^ 32 == (x ^ 32) + i * 16 * CBK: bit 5 is not touched
      return *reinterpret_cast<const bf16x8*>(
          tA + (((int)(pA - tA) ^ (ks * 32)) + i * 16 * CBK));
    };
    bf16x8 fw[4], fa[kPD + 1];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      fw[j] = *reinterpret_cast<const bf16x8*>(tW + offW[j]);
#pragma unroll
    for (int q = 0; q < kPD; ++q) fa[q] = lda(q);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (q + kPD < NQ) fa[(q + kPD) % (kPD + 1)] = lda(q + kPD);
      __builtin_amdgcn_sched_barrier(0);
      const int i = q % MT;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // acc[i][j][reg] = C[voxel i*16 + fr][feature j*16 + 4*fg + reg]
        acc[i][j] = mfma_16x16x32(fw[j], fa[q % (kPD + 1)], acc[i][j]);
        if (q == MT - 1 && CBK / 32 > 1)   // fw[j] is free: fetch its second half
          fw[j] = *reinterpret_cast<const bf16x8*>(tW + (offW[j] ^ 32));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  dma_a(0);
  dma_w(0);
  __syncthreads();
  for (int g = 0; g < ngroups; ++g) {
#pragma unroll 1   // (unrolled, the three bodies' operand offsets spill at 168 VGPRs)
    for (int dx = 0; dx < 3; ++dx) {
      const int st = 3 * g + dx;
      if (dx == 0 && g + 1 < ngroups) dma_a(g + 1);
      if (st + 1 < 3 * ngroups) dma_w(st + 1);
      compute(g, st, dx);
      __syncthreads();  // next slabs landed (vmcnt drained) and these released
    }
  }

  // epilogue: lane owns 8 consecutive features of one voxel per tile pair
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + wm * (16 * MT) + i * 16 + fr;
    if (m >= M) continue;
    const int p = m / YX, rem = m - p * YX;
    const int y = rem / Xp, x = rem - y * Xp, z = p % Zp;
    const bool interior = (kd == 1 || (z >= 1 && z <= Zp - 2)) && y >= 1 &&
                          y <= Yp - 2 && x >= 1 && x <= Xp - 2;
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
      const int n = n0 + wn * 64 + p2 * 32 + fg * 8;  // Cout % 8 == 0
      if (n >= Cout) continue;
      const f32x4 a0 = acc[i][2 * p2], a1 = acc[i][2 * p2 + 1];
      float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      if (scale != nullptr) {
        const float4 s0 = *reinterpret_cast<const float4*>(scale + n);
        const float4 s1 = *reinterpret_cast<const float4*>(scale + n + 4);
        v[0] *= s0.x; v[1] *= s0.y; v[2] *= s0.z; v[3] *= s0.w;
        v[4] *= s1.x; v[5] *= s1.y; v[6] *= s1.z; v[7] *= s1.w;
      }
      if (shift != nullptr) {
        const float4 b0 = *reinterpret_cast<const float4*>(shift + n);
        const float4 b1 = *reinterpret_cast<const float4*>(shift + n + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
        v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      }
      if (RESID) {
        const bf16x8 r8 =
            *reinterpret_cast<const bf16x8*>(resid + (int64_t)m * Cout + n);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += bf2f((bf16_t)r8[k]);
      }
      if (ex.resid2 != nullptr) {
        const bf16x8 r8 =
            *reinterpret_cast<const bf16x8*>(ex.resid2 + (int64_t)m * Cout + n);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += bf2f((bf16_t)r8[k]);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (ACT == 1) v[k] = fmaxf(v[k], 0.f);
        if (ACT == 2) v[k] = gelu_erf(v[k]);
        if (!interior) v[k] = 0.f;
      }
      const uint4 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]),
                       pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
      *reinterpret_cast<uint4*>(out + (int64_t)m * Cout + n) = o;
      if (ex.out_relu != nullptr) {
        // relu of the ROUNDED result: what a separate pass over `out` would produce
        const uint4 q = {pack_bf16(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)),
                         pack_bf16(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)),
                         pack_bf16(fmaxf(v[4], 0.f), fmaxf(v[5], 0.f)),
                         pack_bf16(fmaxf(v[6], 0.f), fmaxf(v[7], 0.f))};
        *reinterpret_cast<uint4*>(ex.out_relu + (int64_t)m * Cout + n) = q;
      }
    }
  }
}

// Planar (B,C,Z,Y,X) fp32 or bf16 -> interior of the padded channels-last bf16
// grid.  One workgroup = one (b,z,y) row x 64 channels: reads are x-contiguous per
// channel, writes channel-contiguous per voxel, transposed through LDS.  PZ = 1:
// the grid is padded in z too (3-D volumes); PZ = 0: images, [B*Z][Y+2][X+2][C].
template <typename TP>
__device__ __forceinline__ float planar_load(const TP* p);
template <>
__device__ __forceinline__ float planar_load<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float planar_load<bf16_t>(const bf16_t* p) { return bf2f(*p); }
__device__ __forceinline__ void planar_store(float* p, float v) { *p = v; }
__device__ __forceinline__ void planar_store(bf16_t* p, float v) { *p = f2bf(v); }

template <typename TP, int PZ>
__global__ __launch_bounds__(256) void k_volume_pack(
    const TP* __restrict__ in, bf16_t* __restrict__ out, int C, int Z, int Y,
    int X) {
  __shared__ float t[64][65];
  const int Yp = Y + 2, Xp = X + 2;
  const int row = blockIdx.x;  // (b, z, y)
  const int y = row % Y, z = (row / Y) % Z, b = row / (Y * Z);
  const int c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t cstride = (int64_t)Z * Y * X;
  const TP* ib = in + ((int64_t)b * C) * cstride + ((int64_t)z * Y + y) * X;
  bf16_t* ob =
      out + ((((int64_t)b * (Z + 2 * PZ) + z + PZ) * Yp + y + 1) * Xp + 1) * C;
  for (int x0 = 0; x0 < X; x0 += 64) {
    for (int cc = ty; cc < 64; cc += 4)
      t[cc][tx] = (x0 + tx < X && c0 + cc < C)
                      ? planar_load<TP>(ib + (int64_t)(c0 + cc) * cstride + x0 + tx)
                      : 0.f;
    __syncthreads();
    for (int xx = ty; xx < 64; xx += 4)
      if (x0 + xx < X && c0 + tx < C)
        ob[(int64_t)(x0 + xx) * C + c0 + tx] = f2bf(t[tx][xx]);
    __syncthreads();
  }
}

// interior of the padded channels-last bf16 grid -> planar (B,C,Z,Y,X)
template <typename TP, int PZ>
__global__ __launch_bounds__(256) void k_volume_unpack(
    const bf16_t* __restrict__ in, TP* __restrict__ out, int C, int Z, int Y,
    int X) {
  __shared__ float t[64][65];
  const int Yp = Y + 2, Xp = X + 2;
  const int row = blockIdx.x;
  const int y = row % Y, z = (row / Y) % Z, b = row / (Y * Z);
  const int c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t cstride = (int64_t)Z * Y * X;
  TP* ob = out + ((int64_t)b * C) * cstride + ((int64_t)z * Y + y) * X;
  const bf16_t* ib =
      in + ((((int64_t)b * (Z + 2 * PZ) + z + PZ) * Yp + y + 1) * Xp + 1) * C;
  for (int x0 = 0; x0 < X; x0 += 64) {
    for (int xx = ty; xx < 64; xx += 4)
      t[xx][tx] = (x0 + xx < X && c0 + tx < C)
                      ? bf2f(ib[(int64_t)(x0 + xx) * C + c0 + tx])
                      : 0.f;
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4)
      if (x0 + tx < X && c0 + cc < C)
        planar_store(ob + (int64_t)(c0 + cc) * cstride + x0 + tx, t[tx][cc]);
    __syncthreads();
  }
}

// Bilinear resize (align_corners = True, F.interpolate semantics) between two
// padded channels-last bf16 images; writes the interior only, so the zero halo
// of the destination survives.  One lane = 8 channels (16 B) of one output
// pixel; coordinates and blends in fp32 as PyTorch does for bf16 inputs.
__global__ __launch_bounds__(256) void k_resize_bilinear_padded(
    const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int B, int C, int Yi,
    int Xi, int Yo, int Xo, float sy, float sx) {
  const int c8 = C / 8;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * Yo * Xo * c8;
  if (idx >= total) return;
  const int cc = (int)(idx % c8) * 8;
  int64_t p = idx / c8;
  const int xo = (int)(p % Xo);
  p /= Xo;
  const int yo = (int)(p % Yo);
  const int b = (int)(p / Yo);
  const float fy = sy * yo, fx = sx * xo;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < Yi - 1), x1 = x0 + (x0 < Xi - 1);
  const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
  const int Xip = Xi + 2, Xop = Xo + 2;
  const bf16_t* ib = in + ((int64_t)b * (Yi + 2) * Xip) * C + cc;
  auto px = [&](int y, int x) {
    return *reinterpret_cast<const bf16x8*>(ib + ((int64_t)(y + 1) * Xip + x + 1) * C);
  };
  const bf16x8 a = px(y0, x0), bq = px(y0, x1), c = px(y1, x0), d = px(y1, x1);
  unsigned o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = 2 * k + h;
      v[h] = hy * (hx * bf2f((bf16_t)a[e]) + lx * bf2f((bf16_t)bq[e])) +
             ly * (hx * bf2f((bf16_t)c[e]) + lx * bf2f((bf16_t)d[e]));
    }
    o[k] = pack_bf16(v[0], v[1]);
  }
  bf16_t* op = out + (((int64_t)b * (Yo + 2) + yo + 1) * Xop + xo + 1) * C + cc;
  *reinterpret_cast<uint4*>(op) = make_uint4(o[0], o[1], o[2], o[3]);
}

// Token rows of a ViT -> padded image, with the pixel shuffle of a stride == kernel
// transposed convolution folded in: the GEMM that applies ConvTranspose2d(k = s,
// stride = s) to the tokens leaves, per token (y, x), the s*s output pixels side by
// side in one row ([i][j][C]); output pixel (s*y + i, s*x + j) is columns
// (i*s + j)*C .. +C of that row.  ``skip`` leading rows per image (the class token)
// are passed over.  s == 1 is the plain tokens -> image pack.  One lane = 8 channels.
__global__ __launch_bounds__(256) void k_tokens_to_image(
    const bf16_t* __restrict__ src, int64_t row_elems, int T, int skip, int h, int w,
    int s, int C, bf16_t* __restrict__ out, int B) {
  const int c8 = C / 8;
  const int Yo = h * s, Xo = w * s;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * Yo * Xo * c8;
  if (idx >= total) return;
  const int cc = (int)(idx % c8) * 8;
  int64_t p = idx / c8;
  const int xo = (int)(p % Xo);
  p /= Xo;
  const int yo = (int)(p % Yo);
  const int b = (int)(p / Yo);
  const int y = yo / s, i = yo - y * s, x = xo / s, j = xo - x * s;
  const bf16_t* ip = src + ((int64_t)b * T + skip + (int64_t)y * w + x) * row_elems +
                     (int64_t)(i * s + j) * C + cc;
  bf16_t* op = out + (((int64_t)b * (Yo + 2) + yo + 1) * (Xo + 2) + xo + 1) * C + cc;
  *reinterpret_cast<uint4*>(op) = *reinterpret_cast<const uint4*>(ip);
}

// out(y, x) = in(step*y, step*x) between padded images: a stride-``step`` 3x3
// convolution is the stride-1 one sampled at every ``step``-th pixel.
__global__ __launch_bounds__(256) void k_image_subsample(
    const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int B, int C, int Yi,
    int Xi, int Yo, int Xo, int step) {
  const int c8 = C / 8;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * Yo * Xo * c8;
  if (idx >= total) return;
  const int cc = (int)(idx % c8) * 8;
  int64_t p = idx / c8;
  const int xo = (int)(p % Xo);
  p /= Xo;
  const int yo = (int)(p % Yo);
  const int b = (int)(p / Yo);
  const bf16_t* ip =
      in + (((int64_t)b * (Yi + 2) + yo * step + 1) * (Xi + 2) + xo * step + 1) * C + cc;
  bf16_t* op = out + (((int64_t)b * (Yo + 2) + yo + 1) * (Xo + 2) + xo + 1) * C + cc;
  *reinterpret_cast<uint4*>(op) = *reinterpret_cast<const uint4*>(ip);
}

// out[b][y][x] = act(sum_c in[b][y][x][c] * w[c] + bias): the last 1x1 conv of a
// dense-prediction head (C -> 1) with its activation, straight from the padded
// channels-last image to a planar fp32 map.  One lane per pixel; a wave reads a
// contiguous run of rows.  act: 0 none, 1 ReLU, 2 sigmoid.
template <int C>
__global__ __launch_bounds__(256) void k_image_dot(
    const bf16_t* __restrict__ in, const float* __restrict__ w, float bias,
    float* __restrict__ out, int B, int Y, int X, int act) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * Y * X) return;
  const int x = (int)(idx % X);
  const int y = (int)((idx / X) % Y);
  const int b = (int)(idx / ((int64_t)X * Y));
  const bf16_t* p = in + (((int64_t)b * (Y + 2) + y + 1) * (X + 2) + x + 1) * C;
  float acc = bias;
#pragma unroll
  for (int c = 0; c < C; c += 8) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p + c);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc = fmaf(bf2f((bf16_t)v[k]), w[c + k], acc);
  }
  if (act == 1) acc = fmaxf(acc, 0.f);
  if (act == 2) acc = 1.f / (1.f + __expf(-acc));
  out[idx] = acc;
}

// LayerNorm over the channels of every pixel of a padded channels-last bf16 image
// (the LayerNorms between the 3x3 convs of the HSA network's ConvBlock,
// highres_side_adaptor.py:31-52, which the reference reaches through two
// permutes).  One wave per padded row, up to two 16-byte chunks per lane
// (C <= 1024), statistics in fp32 (mean, then the centred sum of squares).
// TOKENS = false: -> padded bf16 image, halo rows written as zeros (ready for the
// next conv); TOKENS = true: -> compact fp32 tokens (B, Y*X, C), halo skipped.
template <bool TOKENS>
__global__ __launch_bounds__(256) void k_image_layernorm(
    const bf16_t* __restrict__ in, const float* __restrict__ gamma,
    const float* __restrict__ beta, void* __restrict__ out, int B, int Y, int X, int C,
    float eps, const float* __restrict__ resid) {
  // resid (TOKENS only, may be null): fp32 tokens added to the result, the
  // `ConvBlock(ln_3(x)) + x` of the adaptor block (highres_side_adaptor.py:122)
  const int lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int Yp = Y + 2, Xp = X + 2;
  if (m >= (int64_t)B * Yp * Xp) return;
  const int x = (int)(m % Xp), y = (int)((m / Xp) % Yp), b = (int)(m / ((int64_t)Xp * Yp));
  const bool interior = x >= 1 && x <= X && y >= 1 && y <= Y;
  const int nchunk = C / 8;
  const bool has0 = lane < nchunk, has1 = lane + 64 < nchunk;
  if (!interior) {
    if (!TOKENS) {
      const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
      bf16_t* o = static_cast<bf16_t*>(out) + m * C;
      if (has0) *reinterpret_cast<bf16x8*>(o + lane * 8) = zero;
      if (has1) *reinterpret_cast<bf16x8*>(o + (lane + 64) * 8) = zero;
    }
    return;
  }
  float v[16];
  float sum = 0.f;
  {
    const bf16_t* p = in + m * C;
    bf16x8 c0 = {0, 0, 0, 0, 0, 0, 0, 0}, c1 = c0;
    if (has0) c0 = *reinterpret_cast<const bf16x8*>(p + lane * 8);
    if (has1) c1 = *reinterpret_cast<const bf16x8*>(p + (lane + 64) * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = bf2f((bf16_t)c0[k]);
      v[8 + k] = bf2f((bf16_t)c1[k]);
      sum += v[k] + v[8 + k];
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
  const float mean = sum / C;
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float a = has0 ? v[k] - mean : 0.f, c = has1 ? v[8 + k] - mean : 0.f;
    sq += a * a + c * c;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) sq += __shfl_xor(sq, d);
  const float rstd = rsqrtf(sq / C + eps);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (!(h == 0 ? has0 : has1)) continue;
    const int c0 = (lane + 64 * h) * 8;
    float r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      r[k] = (v[8 * h + k] - mean) * rstd * gamma[c0 + k] + beta[c0 + k];
    if (TOKENS) {
      const int64_t to = (((int64_t)b * Y + (y - 1)) * X + (x - 1)) * C + c0;
      if (resid != nullptr) {
        const float4 a = *reinterpret_cast<const float4*>(resid + to);
        const float4 e = *reinterpret_cast<const float4*>(resid + to + 4);
        r[0] += a.x; r[1] += a.y; r[2] += a.z; r[3] += a.w;
        r[4] += e.x; r[5] += e.y; r[6] += e.z; r[7] += e.w;
      }
      float* o = static_cast<float*>(out) + to;
      *reinterpret_cast<float4*>(o) = float4{r[0], r[1], r[2], r[3]};
      *reinterpret_cast<float4*>(o + 4) = float4{r[4], r[5], r[6], r[7]};
    } else {
      bf16x8 o8;
#pragma unroll
      for (int k = 0; k < 8; ++k) o8[k] = (short)f2bf(r[k]);
      *reinterpret_cast<bf16x8*>(static_cast<bf16_t*>(out) + m * C + c0) = o8;
    }
  }
}

}  // namespace

extern "C" {

int64_t veon_conv3d_guard_rows(int Y, int X) {
  if (Y <= 0 || X <= 0) return 0;
  // largest tap offset + the overhang of the last (up to 512-row) tile
  return (int64_t)(Y + 2) * (X + 2) + (X + 2) + 1 + 512;
}

static int g_conv_abl = 0;

struct ConvTile { int wm, wn, mt; };

// Tile of the 3x3(x3) conv launch for one problem (host-only logic, also exported as
// veon_conv_tile_choice so that the CPU tests pin it).  Y, X: the output grid.
static ConvTile conv_pick_tile(int kd, int B, int Z, int Y, int X, int Cin, int Cout,
                               int stride) {
  const int pz = kd == 3 ? 1 : 0;
  const int64_t M = (int64_t)B * (Z + 2 * pz) * (Y + 2) * (X + 2);
  // Tile choice.  256 features wide when the layer has them (the activation
  // slab is then fetched once, not once per 128-feature column); the height is
  // the candidate with the fewest rounds x rows over the 256 CUs, counting only
  // the tiles that do work (z-halo planes are skipped).  One workgroup per CU,
  // 12-16 waves of small per-wave tiles: measured faster than 8 waves of
  // bigger ones, and a 3-stage counted-vmcnt DMA ring was not faster than this.
  struct Tile { int wm, wn, mt; };
  static const Tile wide[] = {{4, 4, 3}, {4, 4, 4}, {3, 4, 7}};   // 192/256/336 x 256
  static const Tile narrow[] = {{4, 2, 1}, {4, 2, 2}};             // 64/128 x 128, 8 waves
  static const Tile slim[] = {{8, 1, 1}, {8, 1, 2}};               // 128/256 x 64, 8 waves
  static const Tile mid[] = {{4, 3, 2}, {4, 3, 3}};                // 128/192 x 192, 12 waves
  // 192-wide tiles when they divide the features and 256-wide ones do not
  // (e.g. 384 = 2 x 192 exactly, but 256 + a half-empty second column)
  const bool is_mid = Cout % 192 == 0 && Cout % 256 != 0;
  const bool is_wide = !is_mid && Cout >= 256;
  const Tile* cands = is_mid ? mid : is_wide ? wide : (Cout <= 64 ? slim : narrow);
  const int ncand = is_wide ? 3 : 2;
  const int64_t active = (int64_t)B * Z * (Y + 2) * (X + 2);  // rows off the z-halo
  int wm = cands[0].wm, wn = cands[0].wn, mt = cands[0].mt;
  int64_t best = -1;
  for (int i = 0; i < ncand; ++i) {
    const int64_t rows = cands[i].wm * 16 * cands[i].mt;
    const int64_t cols = (Cout + 64 * cands[i].wn - 1) / (64 * cands[i].wn);
    const int64_t tiles = ((active + rows - 1) / rows + B) * cols;  // + straddlers
    const int64_t slots = (is_wide || is_mid) ? kNumCU : 2 * kNumCU;  // resident workgroups
    const int64_t cost = ((tiles + slots - 1) / slots) * rows;
    if (best < 0 || cost < best) {
      best = cost;
      wm = cands[i].wm; wn = cands[i].wn; mt = cands[i].mt;
    }
  }
  // 128-feature tiles: 128 rows (16 MFMAs per wave and k-step) beat 64 once the grid
  // has several rounds either way -- measured 128 vs 166 us at 6 x 144 x 400 pixels,
  // 40 vs 46 at 6 x 72 x 200 (tools/dpt_probe2.py); the rounds x rows estimate above
  // cannot see that
  if (!is_wide && !is_mid && Cout > 64 && M >= 65536) mt = 2;
  // experiment knob (tools/dpt_probe2.py): bits 8..15 of veon_conv_debug_set force the
  // tile height (mt) of the 128- and 64-feature tile classes
  if (((g_conv_abl >> 8) & 0xff) != 0 && !is_wide && !is_mid) mt = (g_conv_abl >> 8) & 0xff;
  // Under-filled grids (end of round 2, tools/conv_tile_sweep.py).  The choice above
  // minimises rounds x rows and assumes the grid fills the chip; the small convs of
  // the 256x704 path do not (HSA 384 -> 384 on 6 x 16 x 44 pixels: 90 tiles of
  // 128 x 192; the DPT stride-2 conv 768 -> 768: 48 tiles of 192 x 256 on 256 CUs), and
  // then a SMALLER tile is faster because a workgroup's k-loop is serial: 46 -> 35 us
  // and 118 -> 69 us.  For those grids every instantiated tile is priced with
  //   T = k-steps x max(floor, share x max(MFMA time / 0.6, LDS-DMA bytes / 45 GB/s))
  // (floor 0.45 us = one DMA round trip per k-step; share = tiles per CU, whole
  // rounds for the one-workgroup-per-CU tiles), which reproduces the sweep within
  // ~15 %, and the cheapest wins.
  {
    const int64_t rows0 = wm * 16 * mt;
    const int64_t tiles0 = ((active + rows0 - 1) / rows0 + B) * ((Cout + 64 * wn - 1) / (64 * wn));
    if (tiles0 * 4 < 3 * kNumCU) {
      static const Tile all[] = {{3, 4, 7}, {4, 4, 4}, {4, 4, 3}, {4, 3, 3}, {4, 3, 2}, {4, 2, 4},
                                 {4, 2, 3}, {4, 2, 2}, {4, 2, 1}, {8, 1, 2}, {8, 1, 1}};
      const int cmax = (int)((Cout + 63) / 64) * 64;
      const double ksteps = 9.0 * kd * (Cin / CBK);
      const double arow = stride == 1 ? 1.0 / 3.0 : 1.0;   // slab shared by three x-taps
      double bestT = -1.0;
      for (const Tile& t : all) {
        const int r = t.wm * 16 * t.mt, c = 64 * t.wn;
        if (c > cmax && t.wn > 1) continue;
        const int lds_t = 2 * (r + 8 + c) * CBK * (int)sizeof(bf16_t);
        if (lds_t > 160 * 1024) continue;
        const double tiles = (double)(((active + r - 1) / r + B) * ((Cout + c - 1) / c));
        const bool one_per_cu = lds_t > 80 * 1024;
        double share = tiles / kNumCU;
        if (one_per_cu) share = (double)((int64_t)((tiles + kNumCU - 1) / kNumCU));
        if (share < 1.0) share = 1.0;
        const double mfma = r * (double)c * 128.0 / 9.8e6 / 0.6;
        const double dma = (arow * r + c) * 128.0 / 45.0e3;
        double step = share * (mfma > dma ? mfma : dma);
        if (step < 0.45) step = 0.45;
        const double T = ksteps * step;
        if (bestT < 0 || T < bestT) {
          bestT = T;
          wm = t.wm; wn = t.wn; mt = t.mt;
        }
      }
    }
  }
  // experiment knob (tools/conv_tile_sweep.py): bits 16..27 of veon_conv_debug_set force
  // the whole tile (wm, wn, mt: four bits each; must be one of the instantiated shapes)
  if (((g_conv_abl >> 16) & 0xfff) != 0) {
    wm = (g_conv_abl >> 16) & 15;
    wn = (g_conv_abl >> 20) & 15;
    mt = (g_conv_abl >> 24) & 15;
  }
  return ConvTile{wm, wn, mt};
}

static int conv_k3_impl(int kd, const void* in_padded, const void* w_bf16,
                        const float* scale, const float* shift,
                        const void* resid_padded, void* out_padded, int B, int Z,
                        int Y, int X, int Cin, int Cout, int relu, void* stream,
                        int stride = 1, int Yin = 0, int Xin = 0,
                        ConvExtra ex = ConvExtra{nullptr, nullptr}) {
  // Y, X: the OUTPUT grid; Yin, Xin: the input image of a strided 2-D conv
  if ((ex.resid2 && !al16(ex.resid2)) || (ex.out_relu && !al16(ex.out_relu)) ||
      (ex.out_relu && ex.out_relu == out_padded))
    return VEON_ERR_BAD_ARG;
  if (stride == 1) { Yin = Y; Xin = X; }
  if (stride < 1 || stride > 2 || (stride == 2 && kd != 1) || Yin <= 0 || Xin <= 0)
    return VEON_ERR_BAD_ARG;
  const int pz = kd == 3 ? 1 : 0;  // z halo planes on each side
  if (B <= 0 || Z <= 0 || Y <= 0 || X <= 0 || Cin <= 0 || Cout <= 0 ||
      Cin % CBK != 0 || Cout % 8 != 0 || !in_padded || !w_bf16 || !out_padded)
    return VEON_ERR_BAD_ARG;
  if (!al16(in_padded) || !al16(w_bf16) || !al16(out_padded) ||
      (scale && !al16(scale)) || (shift && !al16(shift)) ||
      (resid_padded && !al16(resid_padded)))
    return VEON_ERR_BAD_ARG;
  const int64_t M = (int64_t)B * (Z + 2 * pz) * (Y + 2) * (X + 2);
  if (M > 0x3fffffffLL) return VEON_ERR_BAD_ARG;
  const ConvTile tile = conv_pick_tile(kd, B, Z, Y, X, Cin, Cout, stride);
  const int wm = tile.wm, wn = tile.wn, mt = tile.mt;
  const int bm = wm * 16 * mt;
  const int64_t ncol = (Cout + 64 * wn - 1) / (64 * wn);
  const dim3 grid((unsigned)ncol, (unsigned)((M + bm - 1) / bm));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bf16_t* I = static_cast<const bf16_t*>(in_padded);
  const bf16_t* Wt = static_cast<const bf16_t*>(w_bf16);
  const bf16_t* R = static_cast<const bf16_t*>(resid_padded);
  bf16_t* O = static_cast<bf16_t*>(out_padded);
  const int planes = B * (Z + 2 * pz);
  // the slab-sharing kernel addresses activations and weights by 32-bit offsets
  // k_conv3d_k3_ax addresses both operands with 32-bit BYTE offsets (bf16: 2^30 elements)
  const bool fits32 = (M + 2 * veon_conv3d_guard_rows(Y, X)) * Cin < 0x3fffffffLL &&
                      (int64_t)Cout * 9 * kd * Cin < 0x3fffffffLL;
#define VEON_LAUNCH_CONV(WM, WN, MT, ACT, RESID)                              \
  do {                                                                         \
    constexpr int ldsx =                                                       \
        2 * (WM * 16 * MT + 8 + 64 * WN) * CBK * (int)sizeof(bf16_t);          \
    if (ldsx <= 160 * 1024 && !(g_conv_abl & 8) && fits32 && stride == 1) {    \
      static const hipError_t attrx = hipFuncSetAttribute(                     \
          reinterpret_cast<const void*>(&k_conv3d_k3_ax<WM, WN, MT, ACT, RESID>), \
          hipFuncAttributeMaxDynamicSharedMemorySize, ldsx);                   \
      if (attrx != hipSuccess) return VEON_ERR_LAUNCH;                         \
      hipLaunchKernelGGL((k_conv3d_k3_ax<WM, WN, MT, ACT, RESID>), grid,      \
                         dim3(64 * WM * WN), ldsx, s, I, Wt, scale, shift, R, O, \
                         planes, Z + 2 * pz, Y + 2, X + 2, Cin, Cout, kd, ex); \
      break;                                                                   \
    }                                                                          \
    constexpr int lds =                                                        \
        2 * (WM * 16 * MT + 64 * WN) * CBK * (int)sizeof(bf16_t);              \
    static const hipError_t attr = hipFuncSetAttribute(                        \
        reinterpret_cast<const void*>(&k_conv3d_k3<WM, WN, MT, ACT, RESID>),  \
        hipFuncAttributeMaxDynamicSharedMemorySize, lds);                      \
    if (attr != hipSuccess) return VEON_ERR_LAUNCH;                            \
    hipLaunchKernelGGL((k_conv3d_k3<WM, WN, MT, ACT, RESID>), grid,           \
                       dim3(64 * WM * WN), lds, s, I, Wt, scale, shift, R, O,  \
                       planes, Z + 2 * pz, Y + 2, X + 2, Cin, Cout, kd,        \
                       g_conv_abl, stride, Yin + 2, Xin + 2, ex);              \
  } while (0)
#define VEON_TILE_IS(a, b, c) (wm == a && wn == b && mt == c)
#define VEON_LAUNCH_CONV_T(ACT, RESID)                                        \
  do {                                                                         \
    if (VEON_TILE_IS(3, 4, 7)) VEON_LAUNCH_CONV(3, 4, 7, ACT, RESID);         \
    else if (VEON_TILE_IS(4, 4, 3)) VEON_LAUNCH_CONV(4, 4, 3, ACT, RESID);    \
    else if (VEON_TILE_IS(4, 4, 4)) VEON_LAUNCH_CONV(4, 4, 4, ACT, RESID);    \
    else if (VEON_TILE_IS(4, 2, 2)) VEON_LAUNCH_CONV(4, 2, 2, ACT, RESID);    \
    else if (VEON_TILE_IS(4, 3, 2)) VEON_LAUNCH_CONV(4, 3, 2, ACT, RESID);    \
    else if (VEON_TILE_IS(4, 3, 3)) VEON_LAUNCH_CONV(4, 3, 3, ACT, RESID);    \
    else if (VEON_TILE_IS(8, 1, 1)) VEON_LAUNCH_CONV(8, 1, 1, ACT, RESID);    \
    else if (VEON_TILE_IS(8, 1, 2)) VEON_LAUNCH_CONV(8, 1, 2, ACT, RESID);    \
    else if (VEON_TILE_IS(8, 1, 3)) VEON_LAUNCH_CONV(8, 1, 3, ACT, RESID);    \
    else if (VEON_TILE_IS(8, 1, 4)) VEON_LAUNCH_CONV(8, 1, 4, ACT, RESID);    \
    else if (VEON_TILE_IS(4, 2, 3)) VEON_LAUNCH_CONV(4, 2, 3, ACT, RESID);    \
    else if (VEON_TILE_IS(4, 2, 4)) VEON_LAUNCH_CONV(4, 2, 4, ACT, RESID);    \
    else VEON_LAUNCH_CONV(4, 2, 1, ACT, RESID);                               \
  } while (0)
  if (relu == 1) {
    if (R) VEON_LAUNCH_CONV_T(1, true); else VEON_LAUNCH_CONV_T(1, false);
  } else if (relu == 2) {
    if (R) VEON_LAUNCH_CONV_T(2, true); else VEON_LAUNCH_CONV_T(2, false);
  } else if (relu == 0) {
    if (R) VEON_LAUNCH_CONV_T(0, true); else VEON_LAUNCH_CONV_T(0, false);
  } else {
    return VEON_ERR_BAD_ARG;
  }
#undef VEON_LAUNCH_CONV_T
#undef VEON_TILE_IS
#undef VEON_LAUNCH_CONV
  return launch_status();
}

void veon_conv_debug_set(int flags) { g_conv_abl = flags; }

int veon_conv_tile_choice(int kd, int B, int Z, int Y, int X, int Cin, int Cout, int stride) {
  if ((kd != 1 && kd != 3) || B <= 0 || Z <= 0 || Y <= 0 || X <= 0 || Cin <= 0 || Cout <= 0 ||
      Cin % CBK != 0 || stride < 1 || stride > 2)
    return -1;
  const ConvTile t = conv_pick_tile(kd, B, Z, Y, X, Cin, Cout, stride);
  return (t.wm * 16 * t.mt) | ((64 * t.wn) << 16);
}

int veon_conv3d_k3_bf16(const void* in_padded, const void* w_bf16,
                        const float* scale, const float* shift,
                        const void* resid_padded, void* out_padded, int B, int Z,
                        int Y, int X, int Cin, int Cout, int relu, void* stream) {
  return conv_k3_impl(3, in_padded, w_bf16, scale, shift, resid_padded, out_padded,
                      B, Z, Y, X, Cin, Cout, relu, stream);
}

int veon_conv2d_k3_bf16(const void* in_padded, const void* w_bf16,
                        const float* scale, const float* shift,
                        const void* resid_padded, void* out_padded, int B, int Y,
                        int X, int Cin, int Cout, int relu, void* stream) {
  return conv_k3_impl(1, in_padded, w_bf16, scale, shift, resid_padded, out_padded,
                      B, 1, Y, X, Cin, Cout, relu, stream);
}

int veon_conv2d_k3_bf16_ex(const void* in_padded, const void* w_bf16, const float* scale,
                           const float* shift, const void* resid_padded,
                           const void* resid2_padded, void* out_padded,
                           void* out_relu_padded, int B, int Y, int X, int Cin, int Cout,
                           int relu, void* stream) {
  ConvExtra ex;
  ex.resid2 = static_cast<const bf16_t*>(resid2_padded);
  ex.out_relu = static_cast<bf16_t*>(out_relu_padded);
  return conv_k3_impl(1, in_padded, w_bf16, scale, shift, resid_padded, out_padded, B, 1,
                      Y, X, Cin, Cout, relu, stream, 1, 0, 0, ex);
}

int veon_conv2d_k3s2_bf16(const void* in_padded, const void* w_bf16, const float* scale,
                          const float* shift, const void* resid_padded, void* out_padded,
                          int B, int Yin, int Xin, int Cin, int Cout, int act,
                          void* stream) {
  if (Yin <= 0 || Xin <= 0) return VEON_ERR_BAD_ARG;
  return conv_k3_impl(1, in_padded, w_bf16, scale, shift, resid_padded, out_padded, B, 1,
                      (Yin + 1) / 2, (Xin + 1) / 2, Cin, Cout, act, stream, 2, Yin, Xin);
}

static int pack_impl(bool unpack, int pz, int planar_bf16, const void* planar,
                     const void* padded, int B, int C, int Z, int Y, int X,
                     void* stream) {
  if (B <= 0 || C <= 0 || Z <= 0 || Y <= 0 || X <= 0 || !planar || !padded)
    return VEON_ERR_BAD_ARG;
  const int64_t rows = (int64_t)B * Z * Y;
  if (rows > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  const dim3 grid((unsigned)rows, (unsigned)((C + 63) / 64));
  hipStream_t s = static_cast<hipStream_t>(stream);
#define VEON_PACK(TP, PZ)                                                       \
  do {                                                                          \
    if (unpack)                                                                 \
      hipLaunchKernelGGL((k_volume_unpack<TP, PZ>), grid, dim3(256), 0, s,      \
                         static_cast<const bf16_t*>(padded),                    \
                         static_cast<TP*>(const_cast<void*>(planar)), C, Z, Y, X); \
    else                                                                        \
      hipLaunchKernelGGL((k_volume_pack<TP, PZ>), grid, dim3(256), 0, s,        \
                         static_cast<const TP*>(planar),                        \
                         static_cast<bf16_t*>(const_cast<void*>(padded)), C, Z, \
                         Y, X);                                                 \
  } while (0)
  if (planar_bf16) {
    if (pz) VEON_PACK(bf16_t, 1); else VEON_PACK(bf16_t, 0);
  } else {
    if (pz) VEON_PACK(float, 1); else VEON_PACK(float, 0);
  }
#undef VEON_PACK
  return launch_status();
}

int veon_volume_pack_bf16(const float* ncdhw, void* padded, int B, int C, int Z,
                          int Y, int X, void* stream) {
  return pack_impl(false, 1, 0, ncdhw, padded, B, C, Z, Y, X, stream);
}

int veon_volume_unpack_f32(const void* padded, float* ncdhw, int B, int C, int Z,
                           int Y, int X, void* stream) {
  return pack_impl(true, 1, 0, ncdhw, padded, B, C, Z, Y, X, stream);
}

int veon_image_pack_bf16(const void* nchw, int nchw_is_bf16, void* padded, int B,
                         int C, int Y, int X, void* stream) {
  return pack_impl(false, 0, nchw_is_bf16, nchw, padded, B, C, 1, Y, X, stream);
}

int veon_image_unpack(const void* padded, void* nchw, int nchw_is_bf16, int B,
                      int C, int Y, int X, void* stream) {
  return pack_impl(true, 0, nchw_is_bf16, nchw, padded, B, C, 1, Y, X, stream);
}

int veon_image_resize_bilinear(const void* in_padded, void* out_padded, int B,
                               int C, int Yi, int Xi, int Yo, int Xo,
                               void* stream) {
  if (B <= 0 || C <= 0 || C % 8 != 0 || Yi <= 0 || Xi <= 0 || Yo <= 0 || Xo <= 0 ||
      !in_padded || !out_padded || !al16(in_padded) || !al16(out_padded))
    return VEON_ERR_BAD_ARG;
  const int64_t total = (int64_t)B * Yo * Xo * (C / 8);
  const int64_t blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  const float sy = Yo > 1 ? (float)(Yi - 1) / (float)(Yo - 1) : 0.f;
  const float sx = Xo > 1 ? (float)(Xi - 1) / (float)(Xo - 1) : 0.f;
  hipLaunchKernelGGL(k_resize_bilinear_padded, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream),
                     static_cast<const bf16_t*>(in_padded),
                     static_cast<bf16_t*>(out_padded), B, C, Yi, Xi, Yo, Xo, sy, sx);
  return launch_status();
}

int veon_tokens_to_image(const void* rows, int64_t row_elems, int tokens_per_image,
                         int skip, int h, int w, int s, int C, void* out_padded, int B,
                         void* stream) {
  if (B <= 0 || h <= 0 || w <= 0 || s <= 0 || C <= 0 || C % 8 != 0 || skip < 0 ||
      !rows || !out_padded || !al16(rows) || !al16(out_padded) || row_elems % 8 != 0 ||
      row_elems < (int64_t)s * s * C || tokens_per_image < skip + h * w)
    return VEON_ERR_BAD_ARG;
  const int64_t total = (int64_t)B * h * s * w * s * (C / 8);
  const int64_t blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_tokens_to_image, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(rows),
                     row_elems, tokens_per_image, skip, h, w, s, C,
                     static_cast<bf16_t*>(out_padded), B);
  return launch_status();
}

int veon_image_subsample(const void* in_padded, void* out_padded, int B, int C, int Yi,
                         int Xi, int step, void* stream) {
  if (B <= 0 || C <= 0 || C % 8 != 0 || Yi <= 0 || Xi <= 0 || step <= 0 || !in_padded ||
      !out_padded || !al16(in_padded) || !al16(out_padded))
    return VEON_ERR_BAD_ARG;
  const int Yo = (Yi + step - 1) / step, Xo = (Xi + step - 1) / step;
  const int64_t total = (int64_t)B * Yo * Xo * (C / 8);
  const int64_t blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_image_subsample, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream),
                     static_cast<const bf16_t*>(in_padded),
                     static_cast<bf16_t*>(out_padded), B, C, Yi, Xi, Yo, Xo, step);
  return launch_status();
}

int veon_image_dot(const void* in_padded, const float* w, float bias, float* out,
                   int B, int C, int Y, int X, int act, void* stream) {
  if (B <= 0 || Y <= 0 || X <= 0 || !in_padded || !w || !out || !al16(in_padded) ||
      act < 0 || act > 2)
    return VEON_ERR_BAD_ARG;
  const int64_t total = (int64_t)B * Y * X;
  const int64_t blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bf16_t* I = static_cast<const bf16_t*>(in_padded);
  if (C == 32)
    hipLaunchKernelGGL(k_image_dot<32>, dim3((unsigned)blocks), dim3(256), 0, s, I, w,
                       bias, out, B, Y, X, act);
  else if (C == 64)
    hipLaunchKernelGGL(k_image_dot<64>, dim3((unsigned)blocks), dim3(256), 0, s, I, w,
                       bias, out, B, Y, X, act);
  else
    return VEON_ERR_BAD_ARG;
  return launch_status();
}

int veon_image_layernorm_bf16(const void* in_padded, const float* gamma,
                              const float* beta, void* out, int out_tokens_f32, int B,
                              int C, int Y, int X, float eps, const float* resid_tokens,
                              void* stream) {
  if (resid_tokens && (!out_tokens_f32 || !al16(resid_tokens))) return VEON_ERR_BAD_ARG;
  if (B <= 0 || Y <= 0 || X <= 0 || C <= 0 || C % 8 != 0 || C > 1024 || !in_padded ||
      !gamma || !beta || !out || in_padded == out)
    return VEON_ERR_BAD_ARG;
  if (!al16(in_padded) || !al16(out)) return VEON_ERR_BAD_ARG;
  const int64_t M = (int64_t)B * (Y + 2) * (X + 2);
  const int64_t blocks = (M + 3) / 4;
  if (blocks > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bf16_t* I = static_cast<const bf16_t*>(in_padded);
  if (out_tokens_f32)
    hipLaunchKernelGGL(k_image_layernorm<true>, dim3((unsigned)blocks), dim3(256), 0, s, I,
                       gamma, beta, out, B, Y, X, C, eps, resid_tokens);
  else
    hipLaunchKernelGGL(k_image_layernorm<false>, dim3((unsigned)blocks), dim3(256), 0, s, I,
                       gamma, beta, out, B, Y, X, C, eps, nullptr);
  return launch_status();
}

}  // extern "C"
