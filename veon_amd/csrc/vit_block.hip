// ViT encoder block kernels for MI355X (gfx950 / CDNA4): bf16 operands on the
// matrix cores (v_mfma_f32_16x16x32_bf16), fp32 accumulation, fp32 residual
// stream.  These are the dense contractions of the image encoders that front
// the lift: DINOv2 blocks of DepthAnythingV2
// (mmdet3d/models/depth_anything/dinov2_layers/block.py:85-110,
//  attention.py:56-69, mlp.py:40-46, layer_scale.py) and the CLIP residual
// attention blocks (semantic_net/clip_utils/visual.py, attn_helper.py).
//
//   k_layernorm      x fp32 [T,d] -> bf16 [T,d]                  (HBM-bound)
//   k_gemm_bf16      C[M,N] = A[M,K] . W[N,K]^T  (+bias, and one of: ->bf16,
//                    GELU->bf16, QuickGELU->bf16, *gamma + residual -> fp32)
//   k_attention      softmax(q k^T [+ bias]) v, flash style, head_dim 64
//
// Fragment convention (guide section 3): for mfma_f32_16x16x32_bf16 lane l
// holds A[row l&15][k = 8(l>>4)+j] and B[k = 8(l>>4)+j][col l&15], j = 0..7;
// D[row 4(l>>4)+reg][col l&15].  Both GEMM operands are K-contiguous rows, so
// both fragments are 16-byte LDS reads.  The kernels feed the WEIGHT (or K, or
// V^T) as the MFMA "A" operand and the ACTIVATION (or Q, or P) as "B": the
// accumulator then holds 4 consecutive output features of ONE token per lane
// (16-byte epilogue accesses, per-token softmax statistics stay lane-local).
#include <stdlib.h>

#include <type_traits>

#include "mfma_common.h"

namespace {

// ------------------------------------------------------------------ LayerNorm
// one wave per token row; d <= 64 * kMaxPerLane
constexpr int kLnMaxPerLane = 32;

__global__ __launch_bounds__(256) void k_layernorm(
    const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, bf16_t* __restrict__ out, int T, int d,
    float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= T) return;
  const float* xr = x + (int64_t)row * d;
  float v[kLnMaxPerLane];
  float s = 0.f;
  const int n = (d + 63) / 64;
#pragma unroll
  for (int i = 0; i < kLnMaxPerLane; ++i) {
    if (i < n) {
      const int c = i * 64 + lane;
      v[i] = c < d ? xr[c] : 0.f;
      s += v[i];
    }
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < kLnMaxPerLane; ++i) {
    if (i < n) {
      const int c = i * 64 + lane;
      const float t = c < d ? v[i] - mean : 0.f;
      q += t * t;
    }
  }
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
  const float rstd = rsqrtf(q / (float)d + eps);
  bf16_t* o = out + (int64_t)row * d;
#pragma unroll
  for (int i = 0; i < kLnMaxPerLane; ++i) {
    if (i < n) {
      const int c = i * 64 + lane;
      if (c < d) o[c] = f2bf((v[i] - mean) * rstd * gamma[c] + beta[c]);
    }
  }
}

// Rows of `ld` floats of which the first d are the token (the rest is padding that the
// GEMMs need: K and N in multiples of 64): statistics over d, padding written as zeros.
// The side-adapter ViT of SAN has d = 240 (side_adaptor_in_veon.py:194-241).
__global__ __launch_bounds__(256) void k_layernorm_padded(
    const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, bf16_t* __restrict__ out, int T, int d, int ld,
    float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= T) return;
  const float* xr = x + (int64_t)row * ld;
  float v[kLnMaxPerLane];
  float s = 0.f;
  const int n = (ld + 63) / 64;
#pragma unroll
  for (int i = 0; i < kLnMaxPerLane; ++i) {
    if (i < n) {
      const int c = i * 64 + lane;
      v[i] = c < d ? xr[c] : 0.f;
      s += v[i];
    }
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < kLnMaxPerLane; ++i) {
    if (i < n) {
      const int c = i * 64 + lane;
      const float t = c < d ? v[i] - mean : 0.f;
      q += t * t;
    }
  }
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
  const float rstd = rsqrtf(q / (float)d + eps);
  bf16_t* o = out + (int64_t)row * ld;
#pragma unroll
  for (int i = 0; i < kLnMaxPerLane; ++i) {
    if (i < n) {
      const int c = i * 64 + lane;
      if (c < d) o[c] = f2bf((v[i] - mean) * rstd * gamma[c] + beta[c]);
      else if (c < ld) o[c] = f2bf(0.f);
    }
  }
}

// Vector form for d % 256 == 0 (ViT-B/L: 768, 1024): a lane owns float4 chunks
// c = 4*(i*64 + lane), so the row comes in by 16-byte loads (1 KiB per wave
// instruction) and leaves by 8-byte bf16x4 stores.  Same arithmetic order per
// element as k_layernorm except the lane-local partial sums.
template <int NV>  // float4 chunks per lane = d / 256
__global__ __launch_bounds__(256) void k_layernorm_v4(
    const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, bf16_t* __restrict__ out, int T, float eps) {
  constexpr int d = NV * 256;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= T) return;
  const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * d);
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = xr[i * 64 + lane];
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean,
                e = v[i].w - mean;
    q += (a * a + b * b) + (c * c + e * e);
  }
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
  const float rstd = rsqrtf(q / (float)d + eps);
  uint2* o = reinterpret_cast<uint2*>(out + (int64_t)row * d);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float4 g = g4[i * 64 + lane], b = b4[i * 64 + lane];
    o[i * 64 + lane] =
        uint2{pack_bf16((v[i].x - mean) * rstd * g.x + b.x,
                        (v[i].y - mean) * rstd * g.y + b.y),
              pack_bf16((v[i].z - mean) * rstd * g.z + b.z,
                        (v[i].w - mean) * rstd * g.w + b.w)};
  }
}

// ----------------------------------------------------------------------- GEMM
// (WM*16*MT) x (64*WN) output tile, BK = 64, WM x WN waves, each wave
// (16*MT) tokens x 64 features = MT x 4 MFMA tiles; the launcher picks
// 64 x 128 or 128 x 128 with eight waves.  Operands go global -> LDS directly
// (global_load_lds_dwordx4: no VGPR staging, 1 KiB per wave instruction) into a
// double-buffered image whose 16-byte chunks are XOR-swizzled by the row
// (chunk ^= row & 7): the DMA writes LDS linearly, so the permutation is applied
// to the per-lane SOURCE address, and again on the fragment reads, which makes
// every ds_read_b128 of the 16x16x32 operand maps bank-conflict free.
constexpr int BK = 64;

enum { EPI_BF16 = 0, EPI_GELU = 1, EPI_QUICKGELU = 2, EPI_RESID = 3,
       EPI_AFFINE = 4, EPI_AFFINE_RELU = 5,
       EPI_AFFINE_SIGM = 6 };  // sigmoid(affine) - 0.5 = tanh(x/2)/2 (PredHead3DSem)

__device__ __forceinline__ float quick_gelu(float x) {
  // x * sigmoid(1.702 x); v_exp + v_rcp (1 ulp each, the output is bf16) instead of
  // __expf and an IEEE division (ten more instructions per element)
  return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
}


template <int EPI>
__device__ __forceinline__ void gemm_epilogue_store(f32x4 a, int m, int n, int N,
                                                    const float* __restrict__ bias,
                                                    const float* __restrict__ gamma,
                                                    float* __restrict__ resid,
                                                    bf16_t* __restrict__ out) {
  float v[4] = {a[0], a[1], a[2], a[3]};
  if (EPI == EPI_AFFINE || EPI == EPI_AFFINE_RELU || EPI == EPI_AFFINE_SIGM) {
    if (gamma != nullptr) {
      const float4 g4 = *reinterpret_cast<const float4*>(gamma + n);
      v[0] *= g4.x; v[1] *= g4.y; v[2] *= g4.z; v[3] *= g4.w;
    }
  }
  if (bias != nullptr) {
    const float4 b4 = *reinterpret_cast<const float4*>(bias + n);
    v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
  }
  if (EPI == EPI_RESID) {
    float* rp = resid + (int64_t)m * N + n;
    float4 r4 = *reinterpret_cast<const float4*>(rp);
    if (gamma != nullptr) {
      const float4 g4 = *reinterpret_cast<const float4*>(gamma + n);
      v[0] *= g4.x; v[1] *= g4.y; v[2] *= g4.z; v[3] *= g4.w;
    }
    r4.x += v[0]; r4.y += v[1]; r4.z += v[2]; r4.w += v[3];
    *reinterpret_cast<float4*>(rp) = r4;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (EPI == EPI_GELU) v[k] = gelu_erf(v[k]);
      if (EPI == EPI_QUICKGELU) v[k] = quick_gelu(v[k]);
      if (EPI == EPI_AFFINE_RELU) v[k] = fmaxf(v[k], 0.f);
      if (EPI == EPI_AFFINE_SIGM) v[k] = 0.5f * tanhf(0.5f * v[k]);
    }
    bf16x4 o;
    o[0] = (short)f2bf(v[0]); o[1] = (short)f2bf(v[1]);
    o[2] = (short)f2bf(v[2]); o[3] = (short)f2bf(v[3]);
    *reinterpret_cast<bf16x4*>(out + (int64_t)m * N + n) = o;
  }
}

// Two adjacent MFMA tiles whose weight rows were interleaved in groups of four
// (see weight_row): a lane holds EIGHT consecutive features of one token, so the
// epilogue moves 16 bytes of bf16 (32 of fp32) per lane and a wave instruction
// covers whole 64-byte segments instead of 32-byte halves.
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_store8(f32x4 a, f32x4 b, int m, int n, int N,
                                                     const float* __restrict__ bias,
                                                     const float* __restrict__ gamma,
                                                     float* __restrict__ resid,
                                                     bf16_t* __restrict__ out) {
  if (n + 8 > N) {  // N % 8 == 4: the second half falls off the matrix
    gemm_epilogue_store<EPI>(a, m, n, N, bias, gamma, resid, out);
    return;
  }
  float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  auto scale8 = [&](const float* g) {
    const float4 g0 = *reinterpret_cast<const float4*>(g + n);
    const float4 g1 = *reinterpret_cast<const float4*>(g + n + 4);
    v[0] *= g0.x; v[1] *= g0.y; v[2] *= g0.z; v[3] *= g0.w;
    v[4] *= g1.x; v[5] *= g1.y; v[6] *= g1.z; v[7] *= g1.w;
  };
  // same order of operations as gemm_epilogue_store
  if ((EPI == EPI_AFFINE || EPI == EPI_AFFINE_RELU || EPI == EPI_AFFINE_SIGM) &&
      gamma != nullptr)
    scale8(gamma);
  if (bias != nullptr) {
    const float4 b0 = *reinterpret_cast<const float4*>(bias + n);
    const float4 b1 = *reinterpret_cast<const float4*>(bias + n + 4);
    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
    v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
  }
  if (EPI == EPI_RESID && gamma != nullptr) scale8(gamma);
  if (EPI == EPI_RESID) {
    float* rp = resid + (int64_t)m * N + n;
    float4 r0 = *reinterpret_cast<const float4*>(rp);
    float4 r1 = *reinterpret_cast<const float4*>(rp + 4);
    r0.x += v[0]; r0.y += v[1]; r0.z += v[2]; r0.w += v[3];
    r1.x += v[4]; r1.y += v[5]; r1.z += v[6]; r1.w += v[7];
    *reinterpret_cast<float4*>(rp) = r0;
    *reinterpret_cast<float4*>(rp + 4) = r1;
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (EPI == EPI_GELU) v[k] = gelu_erf(v[k]);
      if (EPI == EPI_QUICKGELU) v[k] = quick_gelu(v[k]);
      if (EPI == EPI_AFFINE_RELU) v[k] = fmaxf(v[k], 0.f);
      if (EPI == EPI_AFFINE_SIGM) v[k] = 0.5f * tanhf(0.5f * v[k]);
    }
    uint4 o;
    o.x = pack_bf16(v[0], v[1]);
    o.y = pack_bf16(v[2], v[3]);
    o.z = pack_bf16(v[4], v[5]);
    o.w = pack_bf16(v[6], v[7]);
    *reinterpret_cast<uint4*>(out + (int64_t)m * N + n) = o;
  }
}

// Weight row (inside the wave's 16 NT feature rows) that MFMA tile t feeds as its
// row r: tiles are paired, rows interleaved in groups of four, so that the
// accumulator rows 4 fg .. 4 fg + 3 of tiles 2p and 2p+1 are the 8 consecutive
// features 32 p + 8 fg + 0..7; an unpaired last tile keeps the plain order.
template <int NT>
__device__ __forceinline__ int weight_row(int t, int r) {
  if (2 * (t / 2) + 1 < NT) return (t / 2) * 32 + (r / 4) * 8 + (t & 1) * 4 + (r & 3);
  return t * 16 + r;
}

template <int EPI, int WM, int WN, int MT>
__global__ __launch_bounds__(64 * WM * WN) void k_gemm_bf16(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
    const float* __restrict__ bias, const float* __restrict__ gamma,
    float* __restrict__ resid, bf16_t* __restrict__ out, int M, int N, int K) {
  // one array for all staging (a second __shared__ object beside a DMA target
  // can make hipcc drain vmcnt before every ds_read: guide section 5 item 4a)
  constexpr int BM = WM * 16 * MT;      // tile height
  constexpr int TBN = 64 * WN;          // tile width
  constexpr int NW = WM * WN;           // waves
  constexpr int A_ELEMS = BM * BK;      // activation slab per buffer
  constexpr int TW_ELEMS = TBN * BK;    // weight slab per buffer
  constexpr int BUF_ELEMS = A_ELEMS + TW_ELEMS;
  constexpr int APIECES = BM / 8, WPIECES = TBN / 8;  // 1 KiB DMA pieces
  constexpr int AP = (APIECES + NW - 1) / NW, WP = (WPIECES + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];  // [2][A|W]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * TBN;
  const int fr = lane & 15, fg = lane >> 4;

  // DMA map: a wave instruction fills 64 LDS chunks = 8 rows of a slab; the
  // pieces of each slab are dealt round-robin to the waves (wave w issues pieces
  // w, w + NW, ...); lane l lands in row r = 8*piece + l/8, physical chunk l%8,
  // so it must fetch logical chunk (l%8) ^ (r&7).  Rows beyond M / N are
  // clamped (their outputs are dropped).
  const bf16_t* srcA[AP];
  const bf16_t* srcW[WP];
#pragma unroll
  for (int j = 0; j < AP; ++j) {
    const int r = (wave + j * NW) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (r & 7);
    const int gm = m0 + r < M ? m0 + r : M - 1;
    srcA[j] = A + (int64_t)gm * K + c * 8;
  }
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    const int r = (wave + j * NW) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz_w(r);
    const int gn = n0 + r < N ? n0 + r : N - 1;
    srcW[j] = W + (int64_t)gn * K + c * 8;
  }
  auto dma = [&](int buf, int k0) {
    bf16_t* dA = smem + buf * BUF_ELEMS;
    bf16_t* dW = dA + A_ELEMS;
#pragma unroll
    for (int j = 0; j < AP; ++j)
      if (wave + j * NW < APIECES)  // wave-uniform
        __builtin_amdgcn_global_load_lds((gptr_t)(srcA[j] + k0),
                                         (lptr_t)(dA + (wave + j * NW) * 512), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < WP; ++j)
      if (wave + j * NW < WPIECES)
        __builtin_amdgcn_global_load_lds((gptr_t)(srcW[j] + k0),
                                         (lptr_t)(dW + (wave + j * NW) * 512), 16, 0, 0);
  };

  f32x4 acc[MT][4];  // [token tile][feature tile]
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bf16 elements) for ks = 0; ks = 1 flips chunk bit 2
  int offA[MT], offW[4];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int ra = wm * (16 * MT) + i * 16 + fr;
    offA[i] = ra * BK + ((fg ^ (ra & 7)) * 8);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rw = wn * 64 + weight_row<4>(i, fr);
    offW[i] = rw * BK + ((fg ^ swz_w(rw)) * 8);
  }

  const int nk = K / BK;
  dma(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) dma(buf ^ 1, (kt + 1) * BK);
    const bf16_t* tA = smem + buf * BUF_ELEMS;
    const bf16_t* tW = tA + A_ELEMS;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
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
          // weight rows as MFMA "A", token rows as "B":
          // acc[i][j][reg] = C[token i*16 + fr][feature j*16 + 4*fg + reg]
          acc[i][j] = mfma_16x16x32(fw[j], fa[i],
                                                               acc[i][j]);
    }
    __syncthreads();  // next slab landed (vmcnt drained) and this one released
  }

  // epilogue: tiles are paired (weight_row): a lane owns 8 consecutive features
  // of one token per pair -> 16-byte bf16 / 32-byte fp32 accesses
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + wm * (16 * MT) + i * 16 + fr;
    if (m >= M) continue;
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
      const int n = n0 + wn * 64 + p2 * 32 + fg * 8;
      if (n >= N) continue;
      gemm_epilogue_store8<EPI>(acc[i][2 * p2], acc[i][2 * p2 + 1], m, n, N, bias, gamma,
                                resid, out);
    }
  }
}

// ------------------------------------------------- GEMM, big tiles + DMA ring
// The 64/128-row tiles above move (1/BM + 1/BN) operand bytes per flop through
// L2 -> LDS and re-read them from LDS at 1.25 / 0.75 fragment reads per MFMA:
// at M = 5406 that, not the matrix pipe, sets their time (440 MB of L2 traffic
// for the ViT-B qkv GEMM = the 34 us it takes).  This variant spends the
// registers on the accumulator instead: a wave owns (16 MT) tokens x (16 NT)
// features ((MT + NT) / (MT NT) fragment reads per MFMA: 0.375 for 8 x 4), a
// workgroup of WM x WN = 8 waves a (WM 16 MT) x (WN 16 NT) tile, ONE workgroup
// per CU, and the operands arrive through a ring of S LDS stages filled by
// global_load_lds with S - 1 K-steps in flight: per K-step one counted
// s_waitcnt vmcnt (never 0 in steady state), one raw s_barrier (LDS-DMA stays in
// flight across it), the refill of the stage freed one step ago, then the
// fragment reads and MFMAs of the step.  Same swizzled LDS image, fragment maps
// and epilogues as k_gemm_bf16.
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// SK = 2: split-K.  Workgroups 2t and 2t + 1 (neighbours in the XCD-contiguous order:
// same XCD, speed only) compute the two K halves of tile t.  Whoever draws ticket 0
// from sync[2t] stores its accumulators to the tile's fp32 slab (plain 16-byte stores,
// every wave drains them, barrier, ONE agent-scope release by lane 0, then the flag
// sync[2t + 1]); whoever draws ticket 1 polls that flag (relaxed), takes ONE agent-scope
// acquire, adds the slab to its own registers and runs the epilogue, then resets both
// words.  The first arriver is resident by then, so the wait cannot deadlock; a + b is
// commutative in fp32, so the result does not depend on who came first.  sync[] must be
// zero at launch and is left zero.
template <int EPI, int WM, int WN, int MT, int NT, int S, int SK = 1>
__global__ __launch_bounds__(64 * WM * WN) void k_gemm_ring(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
    const float* __restrict__ bias, const float* __restrict__ gamma,
    float* __restrict__ resid, bf16_t* __restrict__ out, int M, int N, int K,
    int grid_n, int abl, float* __restrict__ slab = nullptr, int* __restrict__ sync = nullptr) {
  constexpr int NW = WM * WN;
  constexpr int BM = WM * 16 * MT, TBN = WN * 16 * NT;
  constexpr int A_ELEMS = BM * BK, STAGE_ELEMS = (BM + TBN) * BK;
  constexpr int PIECES = (BM + TBN) / 8;  // 1 KiB DMA pieces per stage
  static_assert(PIECES % NW == 0, "every wave issues the same number of pieces");
  constexpr int P = PIECES / NW;           // per wave and stage
  static_assert(P * (S - 1) < 64, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];  // [S][A | W]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // blocks b and b + 8 share an XCD (speed only): give every XCD a contiguous run
  // of tiles, neighbours in n first, so that the token rows a tile row re-reads
  // stay in one L2
  const int nwg = gridDim.x;
  const int q8 = nwg / 8, r8 = nwg % 8, xcd = blockIdx.x % 8;
  const int wgl = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) +
                  blockIdx.x / 8;
  const int wg = wgl / SK, slice = wgl % SK;
  const int kbase = slice * (K / SK);   // this workgroup's K range (SK = 1: all of K)
  const int m0 = (wg / grid_n) * BM, n0 = (wg % grid_n) * TBN;
  const int fr = lane & 15, fg = lane >> 4;

  // DMA map: piece p = wave + j*NW covers LDS rows 8p .. 8p+7 of the stage image
  // (token rows first, then weight rows); lane l lands in row 8p + l/8, physical
  // chunk l%8, so it fetches logical chunk (l%8) ^ key(row).  The DMA is
  // buffer_load ... lds (per-lane BYTE offsets, k0 in an SGPR; the launcher checks
  // that both matrices stay below 2^31 bytes): after a FLAT-encoded
  // global_load_lds hipcc's waitcnt pass downgrades every counted lgkmcnt of the
  // fragment pipeline below to lgkmcnt(0).
  const rsrc_t rsA = make_rsrc(A), rsW = make_rsrc(W);
  int src[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const int r = (wave + j * NW) * 8 + (lane >> 3);
    if (r < BM) {
      const int c = (lane & 7) ^ swz_a(r);
      const int gm = m0 + r < M ? m0 + r : M - 1;
      src[j] = 2 * (gm * K + c * 8);
    } else {
      const int c = (lane & 7) ^ swz_w(r - BM);
      const int gn = n0 + (r - BM) < N ? n0 + (r - BM) : N - 1;
      src[j] = 2 * (gn * K + c * 8);
    }
  }
  auto piece = [&](int j, bf16_t* d, int k0) {
    // a piece is 8 rows, BM a multiple of 8: token or weight rows, wave-uniform
    if ((wave + j * NW) * 8 < BM)
      buffer_load_lds16(rsA, (lptr_t)(d + (wave + j * NW) * 512), src[j], 2 * k0);
    else
      buffer_load_lds16(rsW, (lptr_t)(d + (wave + j * NW) * 512), src[j], 2 * k0);
  };
  auto dma = [&](int stage, int k0) {
    bf16_t* d = smem + stage * STAGE_ELEMS;
#pragma unroll
    for (int j = 0; j < P; ++j) piece(j, d, k0);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // token fragments: row wm*16*MT + 16 i + fr, whose low three bits do not depend
  // on i -> one base and compile-time offsets
  const int ra0 = wm * (16 * MT) + fr;
  const int baseA = ra0 * BK + ((fg ^ swz_a(ra0)) * 8);
  int offW[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int rw = wn * (16 * NT) + weight_row<NT>(i, fr);
    offW[i] = A_ELEMS + rw * BK + ((fg ^ swz_w(rw)) * 8);
  }

  const int nk = K / SK / BK;
  // prologue: S - 1 stages in flight (fewer when K is short)
#pragma unroll
  for (int t = 0; t < S - 1; ++t)
    if (t < nk) dma(t, kbase + t * BK);
  // With S >= 3 the refill of a K-step is not issued as one burst behind the
  // barrier (every wave would sit in the memory pipe's issue queue while the
  // matrix pipe idles) but one piece after every MPP MFMAs: the queued MFMAs
  // cover the issue slot of the DMA.  The refilled stage is needed two barriers
  // later, so its late pieces still have a whole K-step to land.
  constexpr bool ILV = S >= 3;
  constexpr int KS = BK / 32;
  constexpr int MFMAS = KS * MT * NT;
  static_assert(MFMAS >= P, "at least one MFMA per DMA piece");
  constexpr int MPP = MFMAS / P;
  // Fragment pipeline of a K-step: the token fragments run through a ring of
  // kPD + 1 registers, fragment q + kPD is read BEFORE the NT MFMAs of fragment q
  // (counted lgkmcnt in steady state); the weight fragments are double-buffered
  // over the two halves of the K-step (the second half is read while the first
  // computes).  With 8 waves on the CU (2 per SIMD) nothing else hides that latency.
  constexpr int kPD = 2, RS = kPD + 1, NQ = KS * MT;
  for (int kt = 0; kt < nk; ++kt) {
    // my pieces of K-step kt have landed when at most S - 2 younger stages are
    // outstanding; at the tail fewer were issued, so drain
    if (kt + S - 1 <= nk) wait_vmcnt<P * (S - 2)>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // everyone's pieces landed, stage kt-1 is free
    const bool refill = (kt + S - 1 < nk) && !(abl & 2);
    const int rstage = (kt + S - 1) % S, rk0 = kbase + (kt + S - 1) * BK;
    if (!ILV && refill) dma(rstage, rk0);
    const bf16_t* st = smem + (kt % S) * STAGE_ELEMS;
    bf16_t* rd = smem + rstage * STAGE_ELEMS;
    if (abl & 8) {
      if (ILV && refill) dma(rstage, rk0);
      continue;
    }
    auto lda = [&](int q) {   // q = ks * MT + i, compile-time after unrolling
      const int ks = q / MT, i = q - ks * MT;
      return *reinterpret_cast<const bf16x8*>(st + ((baseA ^ (ks * 32)) + i * 16 * BK));
    };
    bf16x8 fw[2][NT], fa[RS];
#pragma unroll
    for (int j = 0; j < NT; ++j)
      fw[0][j] = *reinterpret_cast<const bf16x8*>(st + offW[j]);
#pragma unroll
    for (int q = 0; q < kPD; ++q) fa[q] = lda(q);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int ks = q / MT, i = q - ks * MT;
      if (q + kPD < NQ) fa[(q + kPD) % RS] = lda(q + kPD);
      if (i == 0 && ks + 1 < KS) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
          fw[(ks + 1) & 1][j] =
              *reinterpret_cast<const bf16x8*>(st + (offW[j] ^ ((ks + 1) * 32)));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[i][j] = mfma_16x16x32(fw[ks & 1][j], fa[q % RS], acc[i][j]);
        if (ILV) {
          const int c = q * NT + j;  // compile-time after unrolling
          if (c % MPP == MPP - 1 && c / MPP < P) {
            if (refill) piece(c / MPP, rd, rk0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (SK == 2) {
    // (the ticket travels through the staging array itself: a second __shared__ object
    //  beside a DMA target can make hipcc drain vmcnt before every fragment read)
    int* ticket_s = reinterpret_cast<int*>(smem);
    __syncthreads();   // every wave is out of the main loop (LDS no longer read)
    if (tid == 0)
      *ticket_s = __hip_atomic_fetch_add(&sync[2 * wg], 1, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = *ticket_s;
    // slab image: accumulator (i, j) of wave w, lane l at float4 index
    // ((w * MT + i) * NT + j) * 64 + l: every wave instruction moves 1 KiB contiguous
    float4* sl = reinterpret_cast<float4*>(slab) +
                 ((int64_t)wg * NW + wave) * (MT * NT * 64) + lane;
    if (ticket == 0) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          sl[(i * NT + j) * 64] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2],
                                              acc[i][j][3]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&sync[2 * wg + 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    if (tid == 0) {
      while (__hip_atomic_load(&sync[2 * wg + 1], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT) == 0)
        __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float4 p = sl[(i * NT + j) * 64];
        acc[i][j][0] += p.x;
        acc[i][j][1] += p.y;
        acc[i][j][2] += p.z;
        acc[i][j][3] += p.w;
      }
    if (tid == 0) {   // leave the words zero for the next launch
      __hip_atomic_store(&sync[2 * wg], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sync[2 * wg + 1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (abl & 1) {  // ablation: no epilogue stores
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + wm * (16 * MT) + i * 16 + fr;
    if (m >= M) continue;
#pragma unroll
    for (int p2 = 0; p2 < NT / 2; ++p2) {
      const int n = n0 + wn * (16 * NT) + p2 * 32 + fg * 8;
      if (n >= N) continue;
      gemm_epilogue_store8<EPI>(acc[i][2 * p2], acc[i][2 * p2 + 1], m, n, N, bias, gamma,
                                resid, out);
    }
    if (NT & 1) {
      const int n = n0 + wn * (16 * NT) + (NT - 1) * 16 + fg * 4;
      if (n < N)
        gemm_epilogue_store<EPI>(acc[i][NT - 1], m, n, N, bias, gamma, resid, out);
    }
  }
}

// --------------------------------------- GEMM, big tiles + ring of K = 32 granules
// k_gemm_ring holds S = 2 whole K-steps (64 deep) of a 256 x 256 tile in LDS, so only ONE
// step is in flight while the other is consumed: every step pays the L2 latency of its
// refill (measured 2.0 us per step against 0.85 us of MFMA work).  Here the same tile is
// streamed in granules of K = 32 (one MFMA depth): 32 KB per granule, FOUR slots, THREE
// granules in flight (counted vmcnt, raw barrier: LDS-DMA stays in flight across it),
// i.e. 1.3 us of matrix work between the issue of a granule and its use.  The LDS image
// of a granule is [rows][32] (64-byte rows); a DMA piece is 16 rows; the 16-byte chunk
// of a row is XOR-ed with a key of the row so that every ds_read_b128 lane group (four
// NON-contiguous groups of 16 lanes: {0-3,12-15,20-27}, ...) touches 16 different
// 16-byte slots of the 256-byte bank row:
//   token rows (16 consecutive rows per fragment):         key = (-(row >> 2)) & 3
//   weight rows (paired tiles: rows {0-3, 8-11, 16-19, 24-27} + 4 t): key = (-(row >> 3)) & 3
// (derivation in DESIGN 4b).  Same fragment maps, epilogues and tile order as k_gemm_ring.
constexpr int GK = 32;   // granule depth
__device__ __forceinline__ int gkey_a(int row) { return (-(row >> 2)) & 3; }
__device__ __forceinline__ int gkey_w(int row) { return (-(row >> 3)) & 3; }

template <int EPI, int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(64 * WM * WN) void k_gemm_g32(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
    const float* __restrict__ bias, const float* __restrict__ gamma,
    float* __restrict__ resid, bf16_t* __restrict__ out, int M, int N, int K,
    int grid_n) {
  constexpr int NW = WM * WN, SLOTS = 4, AHEAD = SLOTS - 1;
  constexpr int BM = WM * 16 * MT, TBN = WN * 16 * NT;
  static_assert(NT % 2 == 0, "paired weight tiles (16-byte epilogues, the weight key)");
  constexpr int A_ELEMS = BM * GK, SLOT_ELEMS = (BM + TBN) * GK;
  constexpr int PIECES = (BM + TBN) / 16;  // 1 KiB DMA pieces (16 rows) per granule
  static_assert(PIECES % NW == 0, "every wave issues the same number of pieces");
  constexpr int P = PIECES / NW;
  static_assert(P * AHEAD < 64, "vmcnt range");
  static_assert(SLOTS * SLOT_ELEMS * 2 <= 160 * 1024, "LDS");
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];  // [SLOTS][A | W]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int nwg = gridDim.x;
  const int q8 = nwg / 8, r8 = nwg % 8, xcd = blockIdx.x % 8;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) +
                 blockIdx.x / 8;
  const int m0 = (wg / grid_n) * BM, n0 = (wg % grid_n) * TBN;
  const int fr = lane & 15, fg = lane >> 4;

  // DMA map: piece p = wave + j*NW covers rows 16p .. 16p+15 of the granule image;
  // lane l lands in row 16p + l/4, physical chunk l%4 = logical chunk (l%4) ^ key(row)
  const rsrc_t rsA = make_rsrc(A), rsW = make_rsrc(W);
  int src[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const int r = (wave + j * NW) * 16 + (lane >> 2);
    if (r < BM) {
      const int c = (lane & 3) ^ gkey_a(r);
      const int gm = m0 + r < M ? m0 + r : M - 1;
      src[j] = 2 * (gm * K + c * 8);
    } else {
      const int c = (lane & 3) ^ gkey_w(r - BM);
      const int gn = n0 + (r - BM) < N ? n0 + (r - BM) : N - 1;
      src[j] = 2 * (gn * K + c * 8);
    }
  }
  auto piece = [&](int j, bf16_t* d, int k0) {
    if ((wave + j * NW) * 16 < BM)   // a piece is token rows or weight rows, wave-uniform
      buffer_load_lds16(rsA, (lptr_t)(d + (wave + j * NW) * 512), src[j], 2 * k0);
    else
      buffer_load_lds16(rsW, (lptr_t)(d + (wave + j * NW) * 512), src[j], 2 * k0);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment addresses (elements inside a slot): token row ra0 + 16 i shares its key
  // bits with ra0 only through (row >> 2) & 3, which 16 i does not change
  const int ra0 = wm * (16 * MT) + fr;
  const int baseA = ra0 * GK + ((fg ^ gkey_a(ra0)) * 8);
  int offW[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int rw = wn * (16 * NT) + weight_row<NT>(i, fr);
    offW[i] = A_ELEMS + rw * GK + ((fg ^ gkey_w(rw)) * 8);
  }

  const int ng = K / GK;
#pragma unroll
  for (int t = 0; t < AHEAD; ++t)
    if (t < ng) {
      bf16_t* d = smem + t * SLOT_ELEMS;
#pragma unroll
      for (int j = 0; j < P; ++j) piece(j, d, t * GK);
    }
  constexpr int MFMAS = MT * NT;
  static_assert(MFMAS >= P, "at least one MFMA per DMA piece");
  constexpr int MPP = MFMAS / P;
  constexpr int kPD = 2, RS = kPD + 1;
  for (int kt = 0; kt < ng; ++kt) {
    // granule kt has landed when at most min(AHEAD - 1, ng - 1 - kt) younger ones are
    // outstanding
    const int younger = ng - 1 - kt;
    if (younger >= AHEAD - 1) wait_vmcnt<P * (AHEAD - 1)>();
    else if (younger == 1) wait_vmcnt<P>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // everyone's pieces landed; slot of kt-1 is free
    const bool refill = kt + AHEAD < ng;
    bf16_t* rd = smem + ((kt + AHEAD) % SLOTS) * SLOT_ELEMS;
    const int rk0 = (kt + AHEAD) * GK;
    const bf16_t* st = smem + (kt % SLOTS) * SLOT_ELEMS;
    bf16x8 fw[NT], fa[RS];
#pragma unroll
    for (int j = 0; j < NT; ++j) fw[j] = *reinterpret_cast<const bf16x8*>(st + offW[j]);
#pragma unroll
    for (int q = 0; q < kPD; ++q)
      fa[q] = *reinterpret_cast<const bf16x8*>(st + baseA + q * 16 * GK);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (i + kPD < MT)
        fa[(i + kPD) % RS] =
            *reinterpret_cast<const bf16x8*>(st + baseA + (i + kPD) * 16 * GK);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[i][j] = mfma_16x16x32(fw[j], fa[i % RS], acc[i][j]);
        const int c = i * NT + j;  // compile-time after unrolling
        if (c % MPP == MPP - 1 && c / MPP < P) {
          if (refill) piece(c / MPP, rd, rk0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + wm * (16 * MT) + i * 16 + fr;
    if (m >= M) continue;
#pragma unroll
    for (int p2 = 0; p2 < NT / 2; ++p2) {
      const int n = n0 + wn * (16 * NT) + p2 * 32 + fg * 8;
      if (n >= N) continue;
      gemm_epilogue_store8<EPI>(acc[i][2 * p2], acc[i][2 * p2 + 1], m, n, N, bias, gamma,
                                resid, out);
    }
  }
}

// ------------------------------------------------------------------ attention
// qkv: [B, T, 3, H, 64] bf16 (the packed output of the qkv projection, q already
// scaled by head_dim^-0.5 through the weights; LOG2Q: also by log2(e), so that the
// scores arrive in the exp2 domain).  out: [B, T, H*64] bf16.
// Optional additive bias [B or 1][H or 1][T][T] fp32 (CLIP tail; strides given).
// Workgroup = 256 threads = 4 waves, 32 queries per wave.  K/V tiles of 64 keys
// go global -> LDS by DMA (global_load_lds_dwordx4, no VGPR staging) into a
// double buffer whose 16-byte chunks are XOR-swizzled by the row exactly as in
// the GEMM; one barrier per tile.  S^T = K.Q^T so the softmax statistics of a
// query are lane-local; K fragments are 16-B LDS reads, V^T fragments come from
// the hardware transposing read ds_read_b64_tr_b16.
// With head_dim 64 the loop is bound by vector issue (32 exp2 against 36 MFMAs per
// wave and tile), so the softmax is reduced to what cannot be avoided:
//  * the scores are formed RELATIVE to a per-query reference maximum: it is the C
//    operand of the first S^T MFMA (LOG2Q), so exp2 applies to the accumulator as
//    it is -- no scale, no subtraction;
//  * the reference moves only when a tile stands more than kAttRise above it
//    (guide T13); the usual tile has no rescale of O at all;
//  * the row sums come from the matrix core (a fifth V^T "d tile" of ones).
// Per wave and tile that is 32 v_exp + 16 v_cvt_pk + ~20 v_max3 + a dozen others
// (was ~200 vector instructions with the running maximum of round 2).
constexpr int HD = 64;        // head dim
constexpr int QT = 2;         // 16-query tiles per wave
constexpr int AQ = 16 * QT;   // queries per wave
constexpr int AK = 64;        // keys per LDS tile
constexpr int KV_ELEMS = AK * HD;  // one operand tile (8 KiB)
constexpr float kAttRise = 8.f;    // log2 units a score may stand above the reference maximum

typedef bf16x4 __attribute__((address_space(3))) lds_bf16x4;

// max over the four 16-lane rows of a wave without touching LDS: the gfx950
// row-swap instructions exchange halves (permlane32) / odd and even rows
// (permlane16) of two registers, so two copies of x come back as x and its
// partner.  Written as asm: through the builtin, hipcc 7.2 folds
// max(result0, result1) to result0 (seen in the ISA: both v_max dropped, rows
// disagree on their maximum).  The s_nop are the VALU-write -> swap and swap ->
// VALU-read wait states the compiler would otherwise insert itself.
__device__ __forceinline__ float max_over_rows(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  a = b = fmaxf(a, b);
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}

template <bool HAS_BIAS, bool LOG2Q>
__global__ __launch_bounds__(256, 3) void k_attention(
    const bf16_t* __restrict__ qkv, const float* __restrict__ bias,
    int64_t bias_sb, int64_t bias_sh, bf16_t* __restrict__ out, int T, int H) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[4 * KV_ELEMS];  // [buf][K|V]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = (blockIdx.x * 4 + wave) * AQ;
  const int64_t tok_stride = (int64_t)3 * H * HD;
  const bf16_t* qb = qkv + (int64_t)b * T * tok_stride + (int64_t)h * HD;
  constexpr float kLog2e = 1.4426950408889634f;

  // DMA map (as the GEMM): wave instruction j (0..1) of wave w fills LDS rows
  // (w*2 + j)*8 .. +8 of a tile; lane l lands in row r = base + l/8, physical
  // chunk l%8, so it fetches logical chunk (l%8) ^ (r&7).  Keys beyond T are
  // clamped to the last row (masked out of the softmax below).
  int dr[2], dc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    dr[j] = (wave * 2 + j) * 8 + (lane >> 3);
    dc[j] = ((lane & 7) ^ (dr[j] & 7)) * 8;
  }
  const rsrc_t rsQ = make_rsrc(qb);   // K rows at +H*HD elements, V rows at +2*H*HD
  auto dma = [&](int buf, int k0) {
    bf16_t* dK = smem + buf * 2 * KV_ELEMS;
    bf16_t* dV = dK + KV_ELEMS;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int key = k0 + dr[j] < T ? k0 + dr[j] : T - 1;
      // byte offset inside this image's qkv rows (the launcher checks < 2^31)
      const int off = 2 * (key * (int)tok_stride + dc[j]);
      const int slot = (wave * 2 + j) * 512;  // bf16 elements: 64 lanes x 8
      // buffer_load ... lds, not global_load_lds: keeps hipcc's counted lgkmcnt
      // waits on the fragment reads
      buffer_load_lds16(rsQ, (lptr_t)(dK + slot), off, 2 * H * HD);
      buffer_load_lds16(rsQ, (lptr_t)(dV + slot), off, 4 * H * HD);
    }
  };
  dma(0, 0);

  // Q fragments (B operand of S^T = K . Q^T): lane holds Q[q = fr][d = 8fg + j]
  bf16x8 qf[QT][2];
#pragma unroll
  for (int i = 0; i < QT; ++i) {
    const int q = q0 + i * 16 + fr;
    const int qc = q < T ? q : T - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      qf[i][ks] = *reinterpret_cast<const bf16x8*>(qb + (int64_t)qc * tok_stride +
                                                   ks * 32 + fg * 8);
  }
  f32x4 o[QT][4];  // [q tile][d tile]: O^T[d = 4fg + reg][q = fr]
  // ol: the row sums, accumulated by the matrix core as a fifth "d tile" whose V^T
  // fragment is all ones (every register of a lane = the sum of query fr).
  // negm: minus the REFERENCE maximum of query fr (log2 domain), in all four
  // registers: it is the C operand of the first S^T MFMA, so the scores arrive
  // already shifted.
  f32x4 ol[QT];
  float negm[QT];
#pragma unroll
  for (int i = 0; i < QT; ++i) {
    ol[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    negm[i] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = f2bf(1.f);
  const float* brow[QT];
  if (HAS_BIAS) {
#pragma unroll
    for (int i = 0; i < QT; ++i) {
      const int q = q0 + i * 16 + fr;
      brow[i] = bias + b * bias_sb + h * bias_sh + (int64_t)(q < T ? q : T - 1) * T;
    }
  }
  // fragment read offsets (bf16 elements) inside a tile
  const int offK = fr * HD + ((fg ^ (fr & 7)) * 8);  // + kt*16*HD, ^32 for ks=1
  int offV[4];  // transposing read: lane 4q+p of a 16-lane group addresses row
                // q, columns 4p..4p+3 of a 4 x 16 block (guide T10)
  {
    const int row = 4 * fg + (fr >> 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int chunk = 2 * j + ((fr & 3) >> 1);
      offV[j] = row * HD + ((chunk ^ (row & 7)) * 8) + 4 * (fr & 1);
    }
  }

  const int nt = (T + AK - 1) / AK;
  auto tile = [&](int t, auto tail_tag, float floor_) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const int k0 = t * AK;
    const bf16_t* sK = smem + (t & 1) * 2 * KV_ELEMS;
    const bf16_t* sV = sK + KV_ELEMS;
    // S^T tiles: s[i][kt][reg] = S[q = i*16 + fr][key = kt*16 + 4fg + reg], in the
    // log2 domain and relative to the reference maximum
    f32x4 s[QT][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const bf16x8 kf0 =
          *reinterpret_cast<const bf16x8*>(sK + kt * 16 * HD + offK);
      const bf16x8 kf1 =
          *reinterpret_cast<const bf16x8*>(sK + kt * 16 * HD + (offK ^ 32));
#pragma unroll
      for (int i = 0; i < QT; ++i) {
        f32x4 a = LOG2Q ? f32x4{negm[i], negm[i], negm[i], negm[i]}
                        : f32x4{0.f, 0.f, 0.f, 0.f};
        a = mfma_16x16x32(kf0, qf[i][0], a);
        a = mfma_16x16x32(kf1, qf[i][1], a);
        s[i][kt] = a;
      }
    }
    float mx[QT];
#pragma unroll
    for (int i = 0; i < QT; ++i) {
      if (!LOG2Q) {
        // natural-log scores: scale and shift here, on register pairs and with a real
        // (-m, -m) pair (no operand crossing in the packed FMA, see the move below)
        f32x2 nn = {negm[i], negm[i]};
        asm volatile("" : "+v"(nn));
        const f32x2 ll = {kLog2e, kLog2e};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          f32x2 lo = {s[i][kt][0], s[i][kt][1]}, hi = {s[i][kt][2], s[i][kt][3]};
          lo = __builtin_elementwise_fma(lo, ll, nn);
          hi = __builtin_elementwise_fma(hi, ll, nn);
          s[i][kt] = f32x4{lo[0], lo[1], hi[0], hi[1]};
        }
      }
      if (HAS_BIAS) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int key = k0 + kt * 16 + fg * 4 + r;
            if (TAIL) key = key < T ? key : T - 1;
            s[i][kt][r] = fmaf(brow[i][key], kLog2e, s[i][kt][r]);
          }
      }
      if (TAIL) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (k0 + kt * 16 + fg * 4 + r >= T) s[i][kt][r] = -INFINITY;
      }
      float m = s[i][0][0];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, s[i][kt][r]);
      mx[i] = m;
    }
    // The reference maximum of a query moves only when a score of this tile stands
    // more than kAttRise above it (or at tile 0, which sets it).  The decision is per
    // query (the row maximum is formed over the four lane rows first, so the lanes
    // of a query agree), and when any query of the wave moves, everything that still
    // stands at the old reference -- O, the row sums, the reference and this tile's
    // scores -- moves with it, each exactly once and BEFORE any score of the tile is
    // exponentiated (guide T13's textbook order); d = 0 leaves a query as it is.
    float d[QT];
    bool any_move = false;
#pragma unroll
    for (int i = 0; i < QT; ++i) {
      const float m = max_over_rows(mx[i]);
      d[i] = (m > kAttRise || m < floor_) ? m : 0.f;
      if (d[i] == -INFINITY) d[i] = 0.f;  // a fully masked row (bias of -inf)
      any_move |= d[i] != 0.f;
    }
    if (__any(any_move)) {
#pragma unroll
      for (int i = 0; i < QT; ++i) {
        // (tile 0 may move DOWN by any amount: O and the sums are 0 there, keep corr finite)
        const float corr = __builtin_amdgcn_exp2f(fminf(-d[i], 100.f));
        // Both factors as REAL register pairs (x, x), and the updates written on
        // pairs: left to itself hipcc packs the scalar form into v_pk_add_f32 /
        // v_pk_mul_f32 with op_sel operand crossing (one half reading the other dword
        // of the source pair), and with "op_sel:[0,1]" the low halves in lanes 48-63
        // came back unshifted now and then (a wrong P for one key of one query; 300 of
        // 300 launches of B6 T901 H12 had such a tile, none in 3000 without the
        // crossing -- DESIGN 4b).  The empty asm keeps the pairs from being folded
        // back into one register.
        f32x2 dd = {d[i], d[i]}, cc = {corr, corr};
        asm volatile("" : "+v"(dd), "+v"(cc));
        auto on_pairs = [](f32x4& x, f32x2 f, bool mul) {
          f32x2 lo = {x[0], x[1]}, hi = {x[2], x[3]};
          if (mul) { lo *= f; hi *= f; } else { lo -= f; hi -= f; }
          x = f32x4{lo[0], lo[1], hi[0], hi[1]};
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) on_pairs(o[i][j], cc, true);
        on_pairs(ol[i], cc, true);
        negm[i] -= d[i];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) on_pairs(s[i][kt], dd, false);
      }
    }
    // O^T += V^T . P^T : MFMA k-slot (8fg + j) <-> key
    //   j < 4 : key tile 2*kk,   keys 4fg + j
    //   j >= 4: key tile 2*kk+1, keys 4fg + (j-4)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 pf[QT];
#pragma unroll
      for (int i = 0; i < QT; ++i) {
        typedef unsigned __attribute__((ext_vector_type(4))) u32x4;
#pragma unroll
        for (int kt = 2 * kk; kt < 2 * kk + 2; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[i][kt][r] = __builtin_amdgcn_exp2f(s[i][kt][r]);
        const u32x4 pk = {pack_bf16(s[i][2 * kk][0], s[i][2 * kk][1]),
                          pack_bf16(s[i][2 * kk][2], s[i][2 * kk][3]),
                          pack_bf16(s[i][2 * kk + 1][0], s[i][2 * kk + 1][1]),
                          pack_bf16(s[i][2 * kk + 1][2], s[i][2 * kk + 1][3])};
        pf[i] = __builtin_bit_cast(bf16x8, pk);
        ol[i] = mfma_16x16x32(ones, pf[i], ol[i]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // V^T fragment of d tile j: column d = j*16 + fr of keys {4fg..4fg+3} of
        // key tiles 2kk and 2kk+1, delivered by the transposing read
        const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_bf16x4*)(sV + (2 * kk) * 16 * HD + offV[j]));
        const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_bf16x4*)(sV + (2 * kk + 1) * 16 * HD + offV[j]));
        bf16x8 vf;
        vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
        vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
#pragma unroll
        for (int i = 0; i < QT; ++i)
          o[i][j] = mfma_16x16x32(vf, pf[i], o[i][j]);
      }
    }
  };

  __syncthreads();  // tile 0 landed
  float floor_ = INFINITY;
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) dma((t + 1) & 1, (t + 1) * AK);  // flies under this tile
    if (q0 < T) {  // waves past the last query only feed the DMA and barriers
      if ((t + 1) * AK > T)
        tile(t, std::true_type{}, floor_);
      else
        tile(t, std::false_type{}, floor_);
      // Tile 0 sets the reference (any maximum counts as "below the floor" of +inf);
      // from then on it only rises.  The empty asm keeps the compiler from peeling
      // tile 0 out of the loop, which costs 80 registers of copied accumulators.
      floor_ = -INFINITY;
      asm volatile("" : "+s"(floor_));
    }
    __syncthreads();  // next tile landed, this one fully consumed
  }
  // normalise and store: lane owns O[q = fr][d = j*16 + 4fg .. +3]
#pragma unroll
  for (int i = 0; i < QT; ++i) {
    const int q = q0 + i * 16 + fr;
    const float l = ol[i][0];
    if (q >= T) continue;
    const float inv = l > 0.f ? 1.f / l : 0.f;
    bf16_t* op = out + ((int64_t)b * T + q) * (int64_t)H * HD + (int64_t)h * HD;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint2 w2 = {pack_bf16(o[i][j][0] * inv, o[i][j][1] * inv),
                        pack_bf16(o[i][j][2] * inv, o[i][j][3] * inv)};
      *reinterpret_cast<uint2*>(op + j * 16 + fg * 4) = w2;
    }
  }
}

// Patch embedding as a GEMM operand: the non-overlapping p x p patches of an fp32
// NCHW image as bf16 rows [b][skip + y*w + x][c*p*p + i*p + j] (nn.Conv2d's weight
// order), K padded with zeros to ``kpad``, ``skip`` all-zero rows in front of every
// image (the class token's slot: the GEMM then adds only its bias there).  One lane
// = one row element; consecutive lanes walk a patch row (j), so image reads are
// contiguous runs of p floats.
__global__ __launch_bounds__(256) void k_patchify_bf16(
    const float* __restrict__ img, bf16_t* __restrict__ out, int B, int C, int H, int W,
    int p, int h, int w, int skip, int kpad) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int T = skip + h * w;
  if (idx >= (int64_t)B * T * kpad) return;
  const int col = (int)(idx % kpad);
  const int64_t row = idx / kpad;
  const int t = (int)(row % T) - skip;
  const int b = (int)(row / T);
  float v = 0.f;
  if (t >= 0 && col < C * p * p) {
    const int c = col / (p * p), r = col - c * p * p;
    const int i = r / p, j = r - i * p;
    const int y = t / w, x = t - y * w;
    v = img[(((int64_t)b * C + c) * H + y * p + i) * W + x * p + j];
  }
  out[idx] = f2bf(v);
}

// fp32 -> bf16 cast (weights / activations entering the bf16 path)
__global__ __launch_bounds__(256) void k_cast_bf16(const float* __restrict__ in,
                                                   bf16_t* __restrict__ out,
                                                   int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = f2bf(in[i]);
}


}  // namespace

// fp32 -> fp32 LayerNorm over rows of d = 128*K floats, half a wave per row (three
// float4 per lane at d = 384: the token LayerNorms of the HSA network, which run on
// 100 MB tensors at VEON's resolution).  Two-pass statistics in registers.
namespace {
// PADDED: instead of fp32 rows, write bf16 into the interior of a zero-haloed
// channels-last image [B][Y+2][X+2][D] (token t = (b, y, x)): LayerNorm + the
// ConvBlock's input staging in one pass.
struct LnAdd {        // rows [B][h*w][D] fp32 added to the last Y*X tokens of every L
  const float* add;   // (nullptr: plain LayerNorm)
  int L, h, w;
  float sy, sx;       // (float)h / Y, (float)w / X
};
template <int K, bool PADDED>
__global__ __launch_bounds__(256) void k_layernorm_f32(
    const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, void* __restrict__ out, int T, float eps, int Y,
    int X, LnAdd ad) {
  constexpr int D = 128 * K;
  const int l = threadIdx.x & 31;
  const int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (row >= T) return;  // a whole half-wave leaves; the shuffles below stay inside one
  const float4* p = reinterpret_cast<const float4*>(x + row * D);
  // optional: + the nearest-resized row of a coarser map (the HSA block's "offset")
  const float4* pa = nullptr;
  if (ad.add != nullptr) {
    const int t = (int)(row % ad.L) - (ad.L - Y * X);   // position on the Y x X token map
    if (t >= 0) {
      const int py = t / X, px = t - py * X;
      // F.interpolate(mode='nearest'): src = min(floor(dst * (in / out)), in - 1), in fp32
      const int iy = min((int)floorf(py * ad.sy), ad.h - 1);
      const int ix = min((int)floorf(px * ad.sx), ad.w - 1);
      pa = reinterpret_cast<const float4*>(
          ad.add + (((row / ad.L) * ad.h + iy) * ad.w + ix) * (int64_t)D);
    }
  }
  float4 v[K];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    v[k] = p[l + 32 * k];
    if (pa != nullptr) {
      const float4 a = pa[l + 32 * k];
      v[k].x += a.x; v[k].y += a.y; v[k].z += a.z; v[k].w += a.w;
    }
    s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  const float mean = s / D;
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, e = v[k].w - mean;
    sq += (a * a + b * b) + (c * c + e * e);
  }
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) sq += __shfl_xor(sq, d);
  const float rstd = rsqrtf(sq / D + eps);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
  int64_t orow = row;
  if (PADDED) {
    const int64_t px = row % X, py = (row / X) % Y, pb = row / ((int64_t)X * Y);
    orow = (pb * (Y + 2) + py + 1) * (X + 2) + px + 1;
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float4 g = g4[l + 32 * k], b = b4[l + 32 * k];
    float4 r;
    r.x = (v[k].x - mean) * rstd * g.x + b.x;
    r.y = (v[k].y - mean) * rstd * g.y + b.y;
    r.z = (v[k].z - mean) * rstd * g.z + b.z;
    r.w = (v[k].w - mean) * rstd * g.w + b.w;
    if (PADDED) {
      bf16x4 o4;
      o4[0] = (short)f2bf(r.x); o4[1] = (short)f2bf(r.y);
      o4[2] = (short)f2bf(r.z); o4[3] = (short)f2bf(r.w);
      reinterpret_cast<bf16x4*>(static_cast<bf16_t*>(out) + orow * D)[l + 32 * k] = o4;
    } else {
      reinterpret_cast<float4*>(static_cast<float*>(out) + orow * D)[l + 32 * k] = r;
    }
  }
}
}  // namespace

static int layernorm_f32_impl(const float* x, const float* gamma, const float* beta,
                              void* out, int T, int d, float eps, int Y, int X,
                              bool padded, void* stream, LnAdd ad = LnAdd{}) {
  if (!x || !gamma || !beta || !out || T <= 0 || d <= 0 || d % 128 != 0 || d > 1024)
    return VEON_ERR_BAD_ARG;
  if (!al16(x) || !al16(gamma) || !al16(beta) || !al16(out)) return VEON_ERR_BAD_ARG;
  if (padded && (Y <= 0 || X <= 0 || T % ((int64_t)Y * X) != 0)) return VEON_ERR_BAD_ARG;
  const dim3 grid((unsigned)((T + 7) / 8));
  hipStream_t s = static_cast<hipStream_t>(stream);
#define VEON_LN32(K)                                                            \
  do {                                                                          \
    if (padded)                                                                 \
      hipLaunchKernelGGL((k_layernorm_f32<K, true>), grid, dim3(256), 0, s, x, gamma,  \
                         beta, out, T, eps, Y, X, ad);                          \
    else                                                                        \
      hipLaunchKernelGGL((k_layernorm_f32<K, false>), grid, dim3(256), 0, s, x, gamma, \
                         beta, out, T, eps, Y, X, ad);                          \
  } while (0)
  switch (d / 128) {
    case 1: VEON_LN32(1); break;
    case 2: VEON_LN32(2); break;
    case 3: VEON_LN32(3); break;
    case 4: VEON_LN32(4); break;
    case 5: VEON_LN32(5); break;
    case 6: VEON_LN32(6); break;
    case 7: VEON_LN32(7); break;
    default: VEON_LN32(8); break;
  }
#undef VEON_LN32
  return launch_status();
}

extern "C" int veon_layernorm_f32(const float* x, const float* gamma, const float* beta,
                                  float* out, int T, int d, float eps, void* stream) {
  return layernorm_f32_impl(x, gamma, beta, out, T, d, eps, 1, 1, false, stream);
}

extern "C" int veon_layernorm_f32_add_nearest(const float* x, const float* add,
                                              const float* gamma, const float* beta,
                                              float* out, int B, int L, int d, int Y, int X,
                                              int h, int w, float eps, void* stream) {
  if (!add || B <= 0 || L <= 0 || Y <= 0 || X <= 0 || h <= 0 || w <= 0 ||
      (int64_t)Y * X > L || (int64_t)B * L > 0x7fffffffLL || !al16(add))
    return VEON_ERR_BAD_ARG;
  LnAdd ad;
  ad.add = add;
  ad.L = L;
  ad.h = h;
  ad.w = w;
  ad.sy = (float)h / (float)Y;
  ad.sx = (float)w / (float)X;
  return layernorm_f32_impl(x, gamma, beta, out, B * L, d, eps, Y, X, false, stream, ad);
}

extern "C" int veon_layernorm_f32_to_padded(const float* x, const float* gamma,
                                            const float* beta, void* out_padded, int B,
                                            int Y, int X, int d, float eps,
                                            void* stream) {
  if (B <= 0 || Y <= 0 || X <= 0 || (int64_t)B * Y * X > 0x7fffffffLL)
    return VEON_ERR_BAD_ARG;
  return layernorm_f32_impl(x, gamma, beta, out_padded, B * Y * X, d, eps, Y, X, true,
                            stream);
}

namespace {
int g_gemm_ring = -1;  // tools/gemm_bench.py: force a ring config (0 = small-tile kernel)
int g_gemm_abl = 0;    // ablation bits of the ring kernel (0 in production)

// Ring-kernel tile for a problem: 0 = keep the small-tile kernel.  Measured on
// MI355X at the encoder shapes (tools/gemm_bench.py, profiles/r02_gemm_bench.txt):
// the big tiles win where ONE round of workgroups fills most of the 256 CUs --
// 256 x 256 for the ViT-B qkv projection (29.6 vs 32.7 us), 256 x 128 for N = 1024
// (ViT-L proj / fc2) and for the HSA token heads (12.0 vs 16.1 us); with a second,
// mostly empty round or far fewer workgroups than CUs the 64/128-row tiles (three
// workgroups per CU, finer tail) stay ahead.
inline int gemm_ring_config(int M, int N, int K) {
  if ((int64_t)M * N < (int64_t)1 << 21 || K < 256) return 0;
  auto wgs = [&](int bm, int bn) {
    return (int64_t)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  };
  const int64_t lo = 160, hi = kNumCU;
  // round 3: the same tiles with SIXTEEN waves of 64 x 64 / 64 x 32 (configs 8 / 9)
  // instead of eight of 128 x 64 / 64 x 64 (1 / 7): ViT-B qkv 28.4 vs 31.3 us, ViT-L
  // fc2 (bf16 out) 58.6 vs 66.1, HSA heads 12.0 vs 13.1
  // (profiles/r03_gemm_bench_16waves.txt) -- more waves per SIMD hide the LDS-read ->
  // MFMA latency that eight could not, as in the conv kernel
  static const bool w16 = [] { const char* e = getenv("VEON_GEMM_W16"); return !e || atoi(e); }();
  if (wgs(256, 256) >= lo && wgs(256, 256) <= hi) return w16 ? 8 : 1;
  if (wgs(256, 128) >= lo && wgs(256, 128) <= hi) return w16 ? 9 : 7;
  return 0;
}

// The residual epilogue (fp32 read-modify-write of the stream, issued by every workgroup
// at the same time in a one-round grid) costs the big tiles more than their main loop
// gains -- except for the long-K fc2 of ViT-L: 63.4 us on the 16-wave 256 x 128 tile
// against 70.0 small-tile and 65.8 split-K.
inline int gemm_resid_config(int M, int N, int K) {
  const int64_t w = (int64_t)((M + 255) / 256) * ((N + 127) / 128);
  static const bool w16 = [] { const char* e = getenv("VEON_GEMM_W16"); return !e || atoi(e); }();
  return (w16 && K >= 4096 && w >= 160 && w <= kNumCU) ? 9 : 0;
}
}  // namespace

extern "C" {

int veon_vit_cast_bf16(const float* in, void* out, int64_t n, void* stream) {
  if (n < 0 || (n > 0 && (!in || !out))) return VEON_ERR_BAD_ARG;
  if (n == 0) return VEON_OK;
  hipLaunchKernelGGL(k_cast_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in,
                     static_cast<bf16_t*>(out), n);
  return launch_status();
}

int veon_vit_patchify(const float* img, void* out_bf16, int B, int C, int H, int W,
                      int patch, int skip, int kpad, void* stream) {
  if (B <= 0 || C <= 0 || patch <= 0 || H < patch || W < patch || skip < 0 ||
      kpad < C * patch * patch || !img || !out_bf16)
    return VEON_ERR_BAD_ARG;
  const int h = H / patch, w = W / patch;
  const int64_t n = (int64_t)B * (skip + h * w) * kpad;
  const int64_t blocks = (n + 255) / 256;
  if (blocks > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_patchify_bf16, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), img,
                     static_cast<bf16_t*>(out_bf16), B, C, H, W, patch, h, w, skip, kpad);
  return launch_status();
}

int veon_vit_layernorm(const float* x, const float* gamma, const float* beta,
                       void* out_bf16, int T, int d, float eps, void* stream) {
  if (T <= 0 || d <= 0 || d > 64 * kLnMaxPerLane || !x || !gamma || !beta ||
      !out_bf16)
    return VEON_ERR_BAD_ARG;
  const dim3 grid((unsigned)((T + 3) / 4));
  hipStream_t s = static_cast<hipStream_t>(stream);
  bf16_t* o = static_cast<bf16_t*>(out_bf16);
  const bool vec = al16(x) && al16(gamma) && al16(beta) &&
                   (reinterpret_cast<uintptr_t>(out_bf16) & 7u) == 0;
  if (vec && d == 768)
    hipLaunchKernelGGL(k_layernorm_v4<3>, grid, dim3(256), 0, s, x, gamma, beta, o, T, eps);
  else if (vec && d == 1024)
    hipLaunchKernelGGL(k_layernorm_v4<4>, grid, dim3(256), 0, s, x, gamma, beta, o, T, eps);
  else if (vec && d == 256)
    hipLaunchKernelGGL(k_layernorm_v4<1>, grid, dim3(256), 0, s, x, gamma, beta, o, T, eps);
  else
    hipLaunchKernelGGL(k_layernorm, grid, dim3(256), 0, s, x, gamma, beta, o, T, d, eps);
  return launch_status();
}

int veon_vit_layernorm_padded(const float* x, const float* gamma, const float* beta,
                              void* out_bf16, int T, int d, int ld, float eps,
                              void* stream) {
  if (T <= 0 || d <= 0 || ld < d || ld > 64 * kLnMaxPerLane || !x || !gamma || !beta ||
      !out_bf16)
    return VEON_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_layernorm_padded, dim3((unsigned)((T + 3) / 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, gamma, beta,
                     static_cast<bf16_t*>(out_bf16), T, d, ld, eps);
  return launch_status();
}

// fc2-shaped GEMMs (residual epilogue, long K, few output columns) leave most CUs idle
// with big tiles and drown in operand re-reads with small ones: split K in two.
// slab: tiles x BM x BN fp32 of scratch; sync: 2 ints per tile, ZERO on entry, left zero.
int64_t veon_vit_gemm_splitk_plan(int M, int N, int K, int* tile_out) {
  // -> slab bytes (0 = no split-K for this shape); *tile_out 0 = 192 x 192, 1 = 192 x 256
  if (tile_out) *tile_out = -1;
  if (K < 2048 || K % (2 * BK) != 0 || M < 1024) return 0;
  auto wgs = [&](int bm, int bn) {
    return (int64_t)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  };
  const int64_t a = 2 * wgs(192, 192), b = 2 * wgs(192, 256);
  const bool oka = a > 160 && a <= kNumCU, okb = b > 160 && b <= kNumCU;
  if (!oka && !okb) return 0;
  const bool use_b = okb && (!oka || b > a);
  if (tile_out) *tile_out = use_b ? 1 : 0;
  return (use_b ? b / 2 * 192 * 256 : a / 2 * 192 * 192) * 4;
}

int veon_vit_gemm_splitk(const void* a_bf16, const void* w_bf16, const float* bias,
                         const float* gamma, float* resid, int M, int N, int K,
                         void* slab, int64_t slab_bytes, int* sync_words,
                         int64_t sync_ints, void* stream) {
  int tile = -1;
  const int64_t need = veon_vit_gemm_splitk_plan(M, N, K, &tile);
  if (need == 0 || !a_bf16 || !w_bf16 || !resid || !slab || !sync_words ||
      slab_bytes < need || N % 4 != 0)
    return VEON_ERR_BAD_ARG;
  if (!al16(a_bf16) || !al16(w_bf16) || (bias && !al16(bias)) || (gamma && !al16(gamma)) ||
      !al16(resid) || !al16(slab))
    return VEON_ERR_BAD_ARG;
  if ((int64_t)M * K * 2 >= (1ll << 31) || (int64_t)N * K * 2 >= (1ll << 31))
    return VEON_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bf16_t* A = static_cast<const bf16_t*>(a_bf16);
  const bf16_t* W = static_cast<const bf16_t*>(w_bf16);
#define VEON_RING_SK(WM, WN, MT, NT, S)                                                 \
  do {                                                                                 \
    constexpr int bm_ = WM * 16 * MT, bn_ = WN * 16 * NT;                              \
    constexpr int lds = S * (bm_ + bn_) * BK * (int)sizeof(bf16_t);                    \
    static const hipError_t attr = hipFuncSetAttribute(                                \
        reinterpret_cast<const void*>(&k_gemm_ring<EPI_RESID, WM, WN, MT, NT, S, 2>),  \
        hipFuncAttributeMaxDynamicSharedMemorySize, lds);                              \
    if (attr != hipSuccess) return VEON_ERR_LAUNCH;                                    \
    const int gn = (N + bn_ - 1) / bn_, gm = (M + bm_ - 1) / bm_;                      \
    if ((int64_t)2 * gn * gm > sync_ints) return VEON_ERR_WORKSPACE;                   \
    hipLaunchKernelGGL((k_gemm_ring<EPI_RESID, WM, WN, MT, NT, S, 2>),                 \
                       dim3((unsigned)(2 * gn * gm)), dim3(64 * WM * WN), lds, s, A, W, \
                       bias, gamma, resid, nullptr, M, N, K, gn, 0,                    \
                       static_cast<float*>(slab), sync_words);                         \
  } while (0)
  if (tile == 1) VEON_RING_SK(2, 4, 6, 4, 2);   /* 192 x 256 */
  else VEON_RING_SK(2, 4, 6, 3, 3);              /* 192 x 192 */
#undef VEON_RING_SK
  return launch_status();
}

void veon_gemm_ring_set(int config) {
  g_gemm_ring = config & 0xff;
  if (config < 0) g_gemm_ring = -1;
  g_gemm_abl = config < 0 ? 0 : (config >> 8);
}

int veon_vit_gemm(const void* a_bf16, const void* w_bf16, const float* bias,
                  const float* gamma, float* resid, void* out_bf16, int M, int N,
                  int K, int epilogue, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || K % BK != 0 || N % 4 != 0 || !a_bf16 ||
      !w_bf16)
    return VEON_ERR_BAD_ARG;
  if (epilogue == EPI_RESID ? !resid : !out_bf16) return VEON_ERR_BAD_ARG;
  if (!al16(a_bf16) || !al16(w_bf16) || (bias && !al16(bias)) ||
      (gamma && !al16(gamma)) || (resid && !al16(resid)))
    return VEON_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bf16_t* A = static_cast<const bf16_t*>(a_bf16);
  const bf16_t* W = static_cast<const bf16_t*>(w_bf16);
  bf16_t* O = static_cast<bf16_t*>(out_bf16);
  // Tile: 64 x 128 with EIGHT waves of 16 x 64 (3 workgroups per CU by LDS, 24
  // waves per CU) unless that makes more than ~8 workgroups per CU, where the
  // 128-row tile's halved weight traffic wins.  Measured at M = 5406
  // (profiles/r01_vit_bench.txt): these GEMMs are latency/occupancy-bound --
  // more, smaller waves beat the 4-wave tiles of the same size by 5-10 %,
  // 256-wide 16-wave tiles and 160/192-row tiles that bring N = 768 down to one
  // round of workgroups were 10-20 % slower.
  // Big-tile ring kernel (k_gemm_ring) for the encoder-sized problems: the tile is
  // picked per shape so that the workgroups fill the 256 CUs in one or two rounds.
  {
    // (the fp32 read-modify-write of the residual epilogue, issued by every
    // workgroup at the same time in a one-round grid, costs the big tiles more
    // than their main loop gains: measured 30 vs 25 us on the ViT-L proj)
    int sel = g_gemm_ring >= 0 ? g_gemm_ring
                    : epilogue == EPI_RESID ? gemm_resid_config(M, N, K)
                                            : gemm_ring_config(M, N, K);
    if (epilogue == EPI_AFFINE_SIGM) sel = 0;   // small-tile kernel only
    // the ring kernel addresses both matrices with 32-bit byte offsets
    if ((int64_t)M * K * 2 >= (1ll << 31) || (int64_t)N * K * 2 >= (1ll << 31)) sel = 0;
    if (sel >= 11 && K % GK == 0) {
#define VEON_G32(EPI, WM, WN, MT, NT)                                                  \
  do {                                                                                \
    constexpr int bm_ = WM * 16 * MT, bn_ = WN * 16 * NT;                             \
    constexpr int lds = 4 * (bm_ + bn_) * GK * (int)sizeof(bf16_t);                   \
    static const hipError_t attr = hipFuncSetAttribute(                               \
        reinterpret_cast<const void*>(&k_gemm_g32<EPI, WM, WN, MT, NT>),              \
        hipFuncAttributeMaxDynamicSharedMemorySize, lds);                             \
    if (attr != hipSuccess) return VEON_ERR_LAUNCH;                                   \
    const int gn = (N + bn_ - 1) / bn_, gm = (M + bm_ - 1) / bm_;                     \
    hipLaunchKernelGGL((k_gemm_g32<EPI, WM, WN, MT, NT>), dim3((unsigned)(gn * gm)),  \
                       dim3(64 * WM * WN), lds, s, A, W, bias, gamma, resid, O, M, N, \
                       K, gn);                                                        \
  } while (0)
#define VEON_G32_SEL(EPI)                                            \
  do {                                                               \
    switch (sel) {                                                   \
      case 11: VEON_G32(EPI, 2, 4, 8, 4); break; /* 256 x 256 */     \
      case 12: VEON_G32(EPI, 4, 2, 4, 4); break; /* 256 x 128 */     \
      case 13: VEON_G32(EPI, 2, 4, 4, 4); break; /* 128 x 256 */     \
      default: VEON_G32(EPI, 2, 4, 4, 2); break; /* 128 x 128 */     \
    }                                                                \
  } while (0)
      switch (epilogue) {
        case EPI_BF16: VEON_G32_SEL(EPI_BF16); break;
        case EPI_GELU: VEON_G32_SEL(EPI_GELU); break;
        case EPI_QUICKGELU: VEON_G32_SEL(EPI_QUICKGELU); break;
        case EPI_RESID: VEON_G32_SEL(EPI_RESID); break;
        case EPI_AFFINE: VEON_G32_SEL(EPI_AFFINE); break;
        case EPI_AFFINE_RELU: VEON_G32_SEL(EPI_AFFINE_RELU); break;
        default: return VEON_ERR_BAD_ARG;
      }
#undef VEON_G32_SEL
#undef VEON_G32
      return launch_status();
    }
    if (sel > 0) {
#define VEON_RING(EPI, WM, WN, MT, NT, S)                                             \
  do {                                                                                \
    constexpr int bm_ = WM * 16 * MT, bn_ = WN * 16 * NT;                             \
    constexpr int lds = S * (bm_ + bn_) * BK * (int)sizeof(bf16_t);                   \
    static const hipError_t attr = hipFuncSetAttribute(                               \
        reinterpret_cast<const void*>(&k_gemm_ring<EPI, WM, WN, MT, NT, S>),          \
        hipFuncAttributeMaxDynamicSharedMemorySize, lds);                             \
    if (attr != hipSuccess) return VEON_ERR_LAUNCH;                                   \
    const int gn = (N + bn_ - 1) / bn_, gm = (M + bm_ - 1) / bm_;                     \
    hipLaunchKernelGGL((k_gemm_ring<EPI, WM, WN, MT, NT, S>), dim3((unsigned)(gn * gm)), \
                       dim3(64 * WM * WN), lds, s, A, W, bias, gamma, resid, O, M, N, \
                       K, gn, g_gemm_abl);                                            \
  } while (0)
#define VEON_RING_SEL(EPI)                                          \
  do {                                                              \
    switch (sel) {                                                  \
      case 1: VEON_RING(EPI, 2, 4, 8, 4, 2); break; /* 256 x 256 */ \
      case 2: VEON_RING(EPI, 2, 4, 6, 3, 3); break; /* 192 x 192 */ \
      case 3: VEON_RING(EPI, 2, 4, 4, 3, 3); break; /* 128 x 192 */ \
      case 4: VEON_RING(EPI, 2, 4, 4, 4, 3); break; /* 128 x 256 */ \
      case 5: VEON_RING(EPI, 2, 4, 8, 3, 2); break; /* 256 x 192 */ \
      case 7: VEON_RING(EPI, 4, 2, 4, 4, 3); break; /* 256 x 128 */ \
      case 8: VEON_RING(EPI, 4, 4, 4, 4, 2); break; /* 256 x 256, 16 waves of 64 x 64 */ \
      case 9: VEON_RING(EPI, 4, 4, 4, 2, 3); break; /* 256 x 128, 16 waves of 64 x 32 */ \
      case 10: VEON_RING(EPI, 4, 4, 2, 4, 3); break; /* 128 x 256, 16 waves of 32 x 64 */ \
      default: VEON_RING(EPI, 2, 4, 4, 2, 4); break; /* 128 x 128 */ \
    }                                                               \
  } while (0)
      switch (epilogue) {
        case EPI_BF16: VEON_RING_SEL(EPI_BF16); break;
        case EPI_GELU: VEON_RING_SEL(EPI_GELU); break;
        case EPI_QUICKGELU: VEON_RING_SEL(EPI_QUICKGELU); break;
        case EPI_RESID: VEON_RING_SEL(EPI_RESID); break;
        case EPI_AFFINE: VEON_RING_SEL(EPI_AFFINE); break;
        case EPI_AFFINE_RELU: VEON_RING_SEL(EPI_AFFINE_RELU); break;
        default: return VEON_ERR_BAD_ARG;
      }
#undef VEON_RING_SEL
#undef VEON_RING
      return launch_status();
    }
  }
  int wm = 4, wn = 2;
  int mt = ((int64_t)((M + 63) / 64) * ((N + 127) / 128) > 8 * kNumCU) ? 2 : 1;
  {
    // experiment knob (tools/gemm_bench.py): VEON_GEMM_SMALL=wm,wn,mt forces the small-tile
    // kernel's shape among the instantiated ones
    static const int forced = [] {
      const char* e = getenv("VEON_GEMM_SMALL");
      int a = 0, b = 0, c = 0;
      return (e && sscanf(e, "%d,%d,%d", &a, &b, &c) == 3) ? a * 100 + b * 10 + c : 0;
    }();
    if (forced) { wm = forced / 100; wn = (forced / 10) % 10; mt = forced % 10; }
  }
  const int bm = wm * 16 * mt, bn = 64 * wn;
  const dim3 grid((unsigned)((N + bn - 1) / bn), (unsigned)((M + bm - 1) / bm));
#define VEON_LAUNCH_GEMM(EPI, WM, WN, MT)                                      \
  do {                                                                         \
    constexpr int lds = 2 * (WM * 16 * MT + 64 * WN) * BK * (int)sizeof(bf16_t); \
    static const hipError_t attr = hipFuncSetAttribute(                        \
        reinterpret_cast<const void*>(&k_gemm_bf16<EPI, WM, WN, MT>),          \
        hipFuncAttributeMaxDynamicSharedMemorySize, lds);                      \
    if (attr != hipSuccess) return VEON_ERR_LAUNCH;                            \
    hipLaunchKernelGGL((k_gemm_bf16<EPI, WM, WN, MT>), grid,                   \
                       dim3(64 * WM * WN), lds, s, A, W, bias, gamma, resid,   \
                       O, M, N, K);                                            \
  } while (0)
#define VEON_T(a, b, c) (wm == a && wn == b && mt == c)
#define VEON_LAUNCH_GEMM_MT(EPI)                                               \
  do {                                                                         \
    if (VEON_T(4, 2, 2)) VEON_LAUNCH_GEMM(EPI, 4, 2, 2);                       \
    else if (VEON_T(4, 4, 1)) VEON_LAUNCH_GEMM(EPI, 4, 4, 1);                  \
    else if (VEON_T(4, 4, 2)) VEON_LAUNCH_GEMM(EPI, 4, 4, 2);                  \
    else if (VEON_T(8, 2, 1)) VEON_LAUNCH_GEMM(EPI, 8, 2, 1);                  \
    else VEON_LAUNCH_GEMM(EPI, 4, 2, 1);                                       \
  } while (0)
  switch (epilogue) {
    case EPI_BF16: VEON_LAUNCH_GEMM_MT(EPI_BF16); break;
    case EPI_GELU: VEON_LAUNCH_GEMM_MT(EPI_GELU); break;
    case EPI_QUICKGELU: VEON_LAUNCH_GEMM_MT(EPI_QUICKGELU); break;
    case EPI_RESID: VEON_LAUNCH_GEMM_MT(EPI_RESID); break;
    case EPI_AFFINE: VEON_LAUNCH_GEMM_MT(EPI_AFFINE); break;
    case EPI_AFFINE_RELU: VEON_LAUNCH_GEMM_MT(EPI_AFFINE_RELU); break;
    case EPI_AFFINE_SIGM: VEON_LAUNCH_GEMM_MT(EPI_AFFINE_SIGM); break;
    default: return VEON_ERR_BAD_ARG;
  }
#undef VEON_LAUNCH_GEMM_MT
#undef VEON_T
#undef VEON_LAUNCH_GEMM
  return launch_status();
}

static int attention_launch(const void* qkv_bf16, const float* bias,
                            int64_t bias_batch_stride, int64_t bias_head_stride,
                            void* out_bf16, int B, int T, int H, int head_dim, bool q_log2,
                            void* stream) {
  if (!qkv_bf16 || !out_bf16 || B <= 0 || T <= 0 || H <= 0) return VEON_ERR_BAD_ARG;
  if (head_dim != HD) return VEON_ERR_BAD_ARG;
  if (!al16(qkv_bf16) || !al16(out_bf16)) return VEON_ERR_BAD_ARG;
  // the K/V DMA addresses an image's rows with 32-bit byte offsets
  if ((int64_t)T * 3 * H * HD * 2 >= (1ll << 31)) return VEON_ERR_BAD_ARG;
  const dim3 grid((unsigned)((T + 4 * AQ - 1) / (4 * AQ)), (unsigned)H, (unsigned)B);
#define VEON_LAUNCH_ATT(BIAS, LOG2Q)                                           \
  hipLaunchKernelGGL((k_attention<BIAS, LOG2Q>), grid, dim3(256), 0,           \
                     static_cast<hipStream_t>(stream),                         \
                     static_cast<const bf16_t*>(qkv_bf16), bias,               \
                     bias_batch_stride, bias_head_stride,                      \
                     static_cast<bf16_t*>(out_bf16), T, H)
  if (q_log2) {
    if (bias != nullptr) VEON_LAUNCH_ATT(true, true); else VEON_LAUNCH_ATT(false, true);
  } else {
    if (bias != nullptr) VEON_LAUNCH_ATT(true, false); else VEON_LAUNCH_ATT(false, false);
  }
#undef VEON_LAUNCH_ATT
  return launch_status();
}

int veon_vit_attention(const void* qkv_bf16, const float* bias, int64_t bias_batch_stride,
                       int64_t bias_head_stride, void* out_bf16, int B, int T, int H,
                       int head_dim, void* stream) {
  return attention_launch(qkv_bf16, bias, bias_batch_stride, bias_head_stride, out_bf16, B,
                          T, H, head_dim, false, stream);
}

int veon_vit_attention_log2(const void* qkv_bf16, const float* bias,
                            int64_t bias_batch_stride, int64_t bias_head_stride,
                            void* out_bf16, int B, int T, int H, int head_dim,
                            void* stream) {
  return attention_launch(qkv_bf16, bias, bias_batch_stride, bias_head_stride, out_bf16, B,
                          T, H, head_dim, true, stream);
}


constexpr int kSplitKSyncInts = 1024;

int64_t veon_vit_block_workspace_bytes(int B, int T, int d, int mlp_dim) {
  if (B <= 0 || T <= 0 || d <= 0 || mlp_dim <= 0) return 0;
  // h [M,d] + qkv [M,3d] + o [M,d] + u [M,mlp], bf16, each 256-B aligned, + the sync
  // words of the split-K fc2 (kSplitKSyncInts ints: ZERO when the workspace is first
  // used, left zero by every call)
  const int64_t M = (int64_t)B * T;
  auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
  return al(M * d * 2) + al(M * 3 * d * 2) + al(M * d * 2) + al(M * mlp_dim * 2) +
         kSplitKSyncInts * 4;
}

int veon_vit_block(float* x, const veon_vit_block_weights* w,
                   const float* attn_bias, int64_t bias_batch_stride,
                   int64_t bias_head_stride, void* workspace,
                   int64_t workspace_bytes, int B, int T, int d, int H,
                   void* stream) {
  if (!x || !w || !workspace || B <= 0 || T <= 0 || d <= 0 || H <= 0 ||
      d != H * HD || w->mlp_dim <= 0)
    return VEON_ERR_BAD_ARG;
  if (w->act != EPI_GELU && w->act != EPI_QUICKGELU) return VEON_ERR_BAD_ARG;
  if (workspace_bytes < veon_vit_block_workspace_bytes(B, T, d, w->mlp_dim))
    return VEON_ERR_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0) return VEON_ERR_BAD_ARG;
  const int64_t M64 = (int64_t)B * T;
  if (M64 > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  const int M = (int)M64;
  auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
  char* p = static_cast<char*>(workspace);
  void* h = p;
  p += al(M64 * d * 2);
  void* qkv = p;
  p += al(M64 * 3 * d * 2);
  void* o = p;
  p += al(M64 * d * 2);
  void* u = p;
  p += al(M64 * w->mlp_dim * 2);
  int* sync_words = reinterpret_cast<int*>(p);
  int st;
  if ((st = veon_vit_layernorm(x, w->ln1_w, w->ln1_b, h, M, d, w->ln1_eps, stream)))
    return st;
  if ((st = veon_vit_gemm(h, w->w_qkv, w->b_qkv, nullptr, nullptr, qkv, M, 3 * d, d,
                          EPI_BF16, stream)))
    return st;
  if ((st = attention_launch(qkv, attn_bias, bias_batch_stride, bias_head_stride, o, B, T,
                             H, HD, w->q_log2 != 0, stream)))
    return st;
  if ((st = veon_vit_gemm(o, w->w_proj, w->b_proj, w->gamma1, x, nullptr, M, d, d,
                          EPI_RESID, stream)))
    return st;
  if ((st = veon_vit_layernorm(x, w->ln2_w, w->ln2_b, h, M, d, w->ln2_eps, stream)))
    return st;
  if ((st = veon_vit_gemm(h, w->w_fc1, w->b_fc1, nullptr, nullptr, u, M, w->mlp_dim,
                          d, w->act, stream)))
    return st;
  // fc2: split-K where it was measured to win (MI355X, M = 5406: ViT-L's 1024 x 4096
  // 66.9 vs 71.0 us; ViT-B's 768 x 3072 LOSES, 49.8 vs 41.8 us, and stays unsplit); the
  // qkv buffer is free by now and serves as the slab
  const int64_t slab_need = (g_gemm_ring < 0 && w->mlp_dim >= 4096)
                                ? veon_vit_gemm_splitk_plan(M, d, w->mlp_dim, nullptr)
                                : 0;
  // (the split-K form stays an entry point of its own; the 16-wave tile beats it here)
  if (slab_need > 0 && slab_need <= al(M64 * 3 * d * 2) &&
      gemm_resid_config(M, d, w->mlp_dim) == 0)
    return veon_vit_gemm_splitk(u, w->w_fc2, w->b_fc2, w->gamma2, x, M, d, w->mlp_dim, qkv,
                                al(M64 * 3 * d * 2), sync_words, kSplitKSyncInts, stream);
  return veon_vit_gemm(u, w->w_fc2, w->b_fc2, w->gamma2, x, nullptr, M, d,
                       w->mlp_dim, EPI_RESID, stream);
}

}  // extern "C"
