// Temporal path of the 3-D alignment network on the padded
// channels-last bf16 grid [B][Z+2][Y+2][X+2][C] that the Conv3d body uses.
//
//  * k_deform_attn: the sampling + attention core of TemporalDeformable
//    (mmdet3d/models/semantic_net/side_adapter/align_net_occ3d.py:138-196): per
//    voxel and head, 8 trilinear samples of the [key | value] rows at
//    base + tanh(offset)/size (border clamp, align_corners), dot with the query,
//    softmax over the samples, weighted sum of the values.  The reference
//    materialises a (B*heads*8, 2*hd, D, H, W) fp32 copy of the volume for
//    grid_sample (1.3 GB at VEON's shape); here it is a gather: each head's
//    [key | value] pair is one contiguous 4*hd-byte run of a row.
//  * k_warp_volume: SANInVeonTemporal.align_after_lss
//    (san_in_veon_temporal.py:325-365) as a trilinear gather with an affine
//    voxel-index map; samples outside the grid read as zero.
//
// Both are L2/HBM gather kernels (no MFMA): 16-byte loads, one contiguous run
// per corner, fp32 accumulation.
#include <hip/hip_runtime.h>

#include "mfma_common.h"
#include "veon_hip.h"

namespace {

constexpr int kSamples = 8;

__device__ __forceinline__ void fma8(float (&acc)[8], float w, const bf16x8 v) {
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = fmaf(w, bf2f((bf16_t)v[k]), acc[k]);
}

// Lane layout: a head takes 2*HD/8 lanes (the first half holds 8 key channels
// each, the second half 8 value channels each), a voxel takes heads * that many,
// a wave holds 64 / that voxels.
template <int HD>
__global__ __launch_bounds__(256) void k_deform_attn(
    const bf16_t* __restrict__ kv, const bf16_t* __restrict__ q,
    const bf16_t* __restrict__ off, bf16_t* __restrict__ out, int B, int Z, int Y,
    int X, int heads, int off_stride, float qscale) {
  constexpr int KL = HD / 8;       // key lanes (= value lanes) per head
  constexpr int LPH = 2 * KL;
  const int lpv = heads * LPH;     // lanes per voxel, divides 64 (host-checked)
  const int vpw = 64 / lpv;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t nvox = (int64_t)B * Z * Y * X;
  int64_t v = ((int64_t)blockIdx.x * 4 + wave) * vpw + lane / lpv;
  const bool live = v < nvox;
  if (!live) v = nvox - 1;         // keep the lane in the shuffles
  const int r = lane % lpv;
  const int h = r / LPH;
  const int j = r % LPH;
  const bool is_val = j >= KL;
  const int x = (int)(v % X);
  const int y = (int)((v / X) % Y);
  const int z = (int)((v / ((int64_t)X * Y)) % Z);
  const int b = (int)(v / ((int64_t)X * Y * Z));
  const int Yp = Y + 2, Xp = X + 2;
  const int64_t plane0 = (int64_t)b * (Z + 2);
  const int64_t row = ((plane0 + z + 1) * Yp + y + 1) * Xp + x + 1;
  const int C = heads * HD;

  float qv[8];
  {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(q + row * C + h * HD + (j & (KL - 1)) * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) qv[k] = is_val ? 0.f : bf2f((bf16_t)t[k]) * qscale;
  }
  // torch.linspace(-1, 1, n)[i]
  const float zn = Z > 1 ? -1.f + 2.f * z / (Z - 1) : -1.f;
  const float yn = Y > 1 ? -1.f + 2.f * y / (Y - 1) : -1.f;
  const float xn = X > 1 ? -1.f + 2.f * x / (X - 1) : -1.f;
  const bf16_t* orow = off + row * off_stride + h * kSamples * 3;
  const bf16_t* kvh = kv + h * 2 * HD + j * 8;
  const int64_t kvc = 2 * C;

  float m = -INFINITY, l = 0.f;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int s = 0; s < kSamples; ++s) {
    // The reference stacks its base grid (z, y, x) and divides by (D, H, W), and
    // grid_sample reads that last axis as (x, y, z): component 0 -- built from
    // the z index -- is the position along X, component 2 the one along Z.
    const float o0 = tanhf(bf2f(orow[s * 3 + 0]));
    const float o1 = tanhf(bf2f(orow[s * 3 + 1]));
    const float o2 = tanhf(bf2f(orow[s * 3 + 2]));
    const float gx = fminf(fmaxf(zn + o0 / Z, -1.f), 1.f);
    const float gy = fminf(fmaxf(yn + o1 / Y, -1.f), 1.f);
    const float gz = fminf(fmaxf(xn + o2 / X, -1.f), 1.f);
    const float fx = (gx + 1.f) * 0.5f * (X - 1);
    const float fy = (gy + 1.f) * 0.5f * (Y - 1);
    const float fz = (gz + 1.f) * 0.5f * (Z - 1);
    const int x0 = min((int)floorf(fx), X - 1), y0 = min((int)floorf(fy), Y - 1),
              z0 = min((int)floorf(fz), Z - 1);
    const float tx = fx - x0, ty = fy - y0, tz = fz - z0;
    const int x1 = min(x0 + 1, X - 1), y1 = min(y0 + 1, Y - 1), z1 = min(z0 + 1, Z - 1);
    float samp[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int zc = (c & 4) ? z1 : z0, yc = (c & 2) ? y1 : y0, xc = (c & 1) ? x1 : x0;
      const float w = ((c & 4) ? tz : 1.f - tz) * ((c & 2) ? ty : 1.f - ty) *
                      ((c & 1) ? tx : 1.f - tx);
      const int64_t rr = ((plane0 + zc + 1) * Yp + yc + 1) * Xp + xc + 1;
      fma8(samp, w, *reinterpret_cast<const bf16x8*>(kvh + rr * kvc));
    }
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) part = fmaf(qv[k], samp[k], part);
#pragma unroll
    for (int d = 1; d < KL; d <<= 1) part += __shfl_xor(part, d);
    const float other = __shfl_xor(part, KL);
    const float logit = is_val ? other : part;
    const float mn = fmaxf(m, logit);
    const float corr = __expf(m - mn), p = __expf(logit - mn);
    l = l * corr + p;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = acc[k] * corr + p * samp[k];
    m = mn;
  }
  if (live && is_val) {
    const float inv = 1.f / l;
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (short)f2bf(acc[k] * inv);
    *reinterpret_cast<bf16x8*>(out + row * C + h * HD + (j - KL) * 8) = o;
  }
}

// out voxel (x,y,z) <- trilinear sample of `in` at A[b] * (x,y,z,1), zero outside.
// C/8 lanes per voxel.
__global__ __launch_bounds__(256) void k_warp_volume(
    const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
    const float* __restrict__ A, int B, int Z, int Y, int X, int C) {
  const int lpv = C / 8;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t v = t / lpv;
  if (v >= (int64_t)B * Z * Y * X) return;
  const int c0 = (int)(t % lpv) * 8;
  const int x = (int)(v % X);
  const int y = (int)((v / X) % Y);
  const int z = (int)((v / ((int64_t)X * Y)) % Z);
  const int b = (int)(v / ((int64_t)X * Y * Z));
  const float* a = A + b * 12;
  const float fx = a[0] * x + a[1] * y + a[2] * z + a[3];
  const float fy = a[4] * x + a[5] * y + a[6] * z + a[7];
  const float fz = a[8] * x + a[9] * y + a[10] * z + a[11];
  const int Yp = Y + 2, Xp = X + 2;
  const int64_t plane0 = (int64_t)b * (Z + 2);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // anything further than one voxel outside contributes nothing; the compare
  // also rejects NaN and keeps the int conversion in range
  if (fx > -1.f && fx < (float)X && fy > -1.f && fy < (float)Y && fz > -1.f &&
      fz < (float)Z) {
    const int x0 = (int)floorf(fx), y0 = (int)floorf(fy), z0 = (int)floorf(fz);
    const float tx = fx - x0, ty = fy - y0, tz = fz - z0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int zc = z0 + ((c >> 2) & 1), yc = y0 + ((c >> 1) & 1), xc = x0 + (c & 1);
      if (zc < 0 || zc >= Z || yc < 0 || yc >= Y || xc < 0 || xc >= X) continue;
      const float w = ((c & 4) ? tz : 1.f - tz) * ((c & 2) ? ty : 1.f - ty) *
                      ((c & 1) ? tx : 1.f - tx);
      const int64_t rr = ((plane0 + zc + 1) * Yp + yc + 1) * Xp + xc + 1;
      fma8(acc, w, *reinterpret_cast<const bf16x8*>(in + rr * C + c0));
    }
  }
  const int64_t row = ((plane0 + z + 1) * Yp + y + 1) * Xp + x + 1;
  bf16x8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (short)f2bf(acc[k]);
  *reinterpret_cast<bf16x8*>(out + row * C + c0) = o;
}

// affine[b] = S^-1 * inv(prev2glob[b]) * cur2glob[b] * S with S = voxel index ->
// metric (diag(step), first centre): the whole coordinate chain of
// align_after_lss in voxel-index units, in double, one thread per sample -- so the
// warp needs no host round trip for 4x4 algebra.
__global__ void k_warp_affine(const float* __restrict__ cur2glob,
                              const float* __restrict__ prev2glob, int stride,
                              double fx, double fy, double fz, double sx, double sy,
                              double sz, float* __restrict__ A, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double m[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      m[i][j] = prev2glob[(int64_t)b * stride + i * 4 + j];
      m[i][4 + j] = i == j ? 1.0 : 0.0;
    }
  for (int c = 0; c < 4; ++c) {  // Gauss-Jordan, partial pivoting
    int piv = c;
    for (int r = c + 1; r < 4; ++r)
      if (fabs(m[r][c]) > fabs(m[piv][c])) piv = r;
    for (int j = 0; j < 8; ++j) {
      const double t = m[c][j];
      m[c][j] = m[piv][j];
      m[piv][j] = t;
    }
    const double inv = 1.0 / m[c][c];
    for (int j = 0; j < 8; ++j) m[c][j] *= inv;
    for (int r = 0; r < 4; ++r) {
      if (r == c) continue;
      const double f = m[r][c];
      for (int j = 0; j < 8; ++j) m[r][j] -= f * m[c][j];
    }
  }
  const double first[3] = {fx, fy, fz}, step[3] = {sx, sy, sz};
  for (int i = 0; i < 3; ++i) {
    double t[4];  // row i of inv(prev) * cur
    for (int j = 0; j < 4; ++j) {
      t[j] = 0.0;
      for (int k = 0; k < 4; ++k)
        t[j] += m[i][4 + k] * (double)cur2glob[(int64_t)b * stride + k * 4 + j];
    }
    // metric point = step * idx + first;  past idx = (T p - first) / step
    double off = t[3];
    for (int j = 0; j < 3; ++j) {
      A[b * 12 + i * 4 + j] = (float)(t[j] * step[j] / step[i]);
      off += t[j] * first[j];
    }
    A[b * 12 + i * 4 + 3] = (float)((off - first[i]) / step[i]);
  }
}

// zero the halo rows of a padded grid (after a row-wise GEMM wrote its bias there)
__global__ __launch_bounds__(256) void k_zero_halo(bf16_t* __restrict__ rows, int planes,
                                                    int Zp, int Yp, int Xp, int C) {
  const int lpr = C / 8;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t rrow = t / lpr;
  if (rrow >= (int64_t)planes * Yp * Xp) return;
  const int x = (int)(rrow % Xp);
  const int y = (int)((rrow / Xp) % Yp);
  const int z = (int)((rrow / ((int64_t)Xp * Yp)) % Zp);
  if (x == 0 || x == Xp - 1 || y == 0 || y == Yp - 1 || z == 0 || z == Zp - 1) {
    bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    *reinterpret_cast<bf16x8*>(rows + rrow * C + (t % lpr) * 8) = zero;
  }
}

}  // namespace

extern "C" {

int veon_deform_attention_bf16(const void* kv_padded, const void* q_padded,
                               const void* off_padded, void* out_padded, int B,
                               int Z, int Y, int X, int C, int heads, int samples,
                               int off_channels, void* stream) {
  if (B <= 0 || Z <= 0 || Y <= 0 || X <= 0 || heads <= 0 || C <= 0 ||
      C % heads != 0 || samples != kSamples || off_channels < heads * samples * 3 ||
      !kv_padded || !q_padded || !off_padded || !out_padded)
    return VEON_ERR_BAD_ARG;
  if (!al16(kv_padded) || !al16(q_padded) || !al16(out_padded) ||
      ((uintptr_t)off_padded & 1))
    return VEON_ERR_BAD_ARG;
  const int hd = C / heads;
  if (hd != 32 && hd != 64) return VEON_ERR_BAD_ARG;
  const int lpv = heads * 2 * (hd / 8);
  if (lpv > 64 || 64 % lpv != 0) return VEON_ERR_BAD_ARG;
  if ((int64_t)B * (Z + 2) * (Y + 2) * (X + 2) > 0x3fffffffLL) return VEON_ERR_BAD_ARG;
  const int64_t nvox = (int64_t)B * Z * Y * X;
  const int vpb = 4 * (64 / lpv);  // voxels per 256-thread workgroup
  const int64_t blocks = (nvox + vpb - 1) / vpb;
  if (blocks > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float qscale = 1.f / sqrtf((float)hd);
  const bf16_t* KV = static_cast<const bf16_t*>(kv_padded);
  const bf16_t* Q = static_cast<const bf16_t*>(q_padded);
  const bf16_t* O = static_cast<const bf16_t*>(off_padded);
  bf16_t* out = static_cast<bf16_t*>(out_padded);
  if (hd == 64)
    hipLaunchKernelGGL(k_deform_attn<64>, dim3((unsigned)blocks), dim3(256), 0, s, KV, Q,
                       O, out, B, Z, Y, X, heads, off_channels, qscale);
  else
    hipLaunchKernelGGL(k_deform_attn<32>, dim3((unsigned)blocks), dim3(256), 0, s, KV, Q,
                       O, out, B, Z, Y, X, heads, off_channels, qscale);
  return launch_status();
}

int veon_volume_warp_bf16(const void* in_padded, void* out_padded,
                          const float* affine, int B, int C, int Z, int Y, int X,
                          void* stream) {
  if (B <= 0 || C <= 0 || C % 8 != 0 || Z <= 0 || Y <= 0 || X <= 0 || !in_padded ||
      !out_padded || !affine || in_padded == out_padded)
    return VEON_ERR_BAD_ARG;
  if (!al16(in_padded) || !al16(out_padded)) return VEON_ERR_BAD_ARG;
  if ((int64_t)B * (Z + 2) * (Y + 2) * (X + 2) > 0x3fffffffLL) return VEON_ERR_BAD_ARG;
  const int64_t total = (int64_t)B * Z * Y * X * (C / 8);
  const int64_t blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_warp_volume, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream),
                     static_cast<const bf16_t*>(in_padded),
                     static_cast<bf16_t*>(out_padded), affine, B, Z, Y, X, C);
  return launch_status();
}

int veon_warp_affine(const float* cur2glob, const float* prev2glob, int mat_stride,
                     const double* first_xyz, const double* step_xyz, float* affine,
                     int B, void* stream) {
  if (B <= 0 || mat_stride < 16 || !cur2glob || !prev2glob || !first_xyz || !step_xyz ||
      !affine)
    return VEON_ERR_BAD_ARG;
  for (int i = 0; i < 3; ++i)
    if (!(step_xyz[i] > 0.0)) return VEON_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_warp_affine, dim3((B + 63) / 64), dim3(64), 0,
                     static_cast<hipStream_t>(stream), cur2glob, prev2glob, mat_stride,
                     first_xyz[0], first_xyz[1], first_xyz[2], step_xyz[0], step_xyz[1],
                     step_xyz[2], affine, B);
  return launch_status();
}

int veon_volume_zero_halo_bf16(void* padded, int B, int C, int Z, int Y, int X,
                               void* stream) {
  if (B <= 0 || C <= 0 || C % 8 != 0 || Z <= 0 || Y <= 0 || X <= 0 || !padded ||
      !al16(padded))
    return VEON_ERR_BAD_ARG;
  const int64_t total = (int64_t)B * (Z + 2) * (Y + 2) * (X + 2) * (C / 8);
  const int64_t blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_zero_halo, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<bf16_t*>(padded),
                     B * (Z + 2), Z + 2, Y + 2, X + 2, C);
  return launch_status();
}

}  // extern "C"
