// Depth preparation for VEON's lift on MI355X.
//
//   downsample_depth : block-min over the non-zero pixels of each ds x ds block,
//                      zeros counted as 1e5
//                      (mmdet3d/models/necks/view_transformer_raw.py:393-404)
//   get_two_hot_depth: softmax over D+1 bin centres of -gamma*|d - c_k| clamped
//                      at -16, last bin dropped, output (BN, D, H, W)
//                      (view_transformer_raw.py:406-429)
// and the fusion of both (AlignNetOcc3D.prepare_depth calls them back to back,
// align_net_occ3d.py:320-326): the (BN, H, W) block-min map never reaches HBM.
// Memory-bound elementwise work: one lane per output pixel, lanes run along W
// so every load and store instruction is contiguous across the wave.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/veon_hip.h"

namespace {
constexpr int kBlock = 256;

__device__ __forceinline__ float block_min_nonzero(const float* __restrict__ src,
                                                   int Wsrc, int ds) {
  float m = INFINITY;
  for (int dy = 0; dy < ds; ++dy)
    for (int dx = 0; dx < ds; ++dx) {
      float v = src[(int64_t)dy * Wsrc + dx];
      if (v == 0.0f) v = 1e5f;
      m = v < m ? v : m;
    }
  return m;
}

__global__ __launch_bounds__(kBlock) void k_downsample_depth(
    const float* __restrict__ depths, int64_t n_out, int H, int W, int ds,
    float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_out) return;
  const int h = H / ds, w = W / ds;
  const int x = (int)(i % w);
  const int y = (int)((i / w) % h);
  const int64_t bn = i / ((int64_t)w * h);
  out[i] = block_min_nonzero(depths + (bn * H + (int64_t)y * ds) * W + (int64_t)x * ds, W, ds);
}

// ds == 0: `depths` is already (BN, H, W) at output resolution.
__global__ __launch_bounds__(kBlock) void k_two_hot_depth(
    const float* __restrict__ depths, int64_t n_pix, int H, int W, int ds, int D,
    float step, float off, float gamma, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_pix) return;
  const int x = (int)(i % W);
  const int y = (int)((i / W) % H);
  const int64_t bn = i / ((int64_t)W * H);
  float d;
  if (ds > 0) {
    const int Hs = H * ds, Ws = W * ds;
    d = block_min_nonzero(depths + (bn * Hs + (int64_t)y * ds) * Ws + (int64_t)x * ds, Ws, ds);
  } else {
    d = depths[i];
  }
  // pass 1: max gap and sum of exp (the clamp makes every term >= exp(-16 - max))
  float mx = -INFINITY;
  for (int k = 0; k <= D; ++k) {
    float gap = -fabsf(d - ((float)k * step + off)) * gamma;
    if (!(gap >= -16.f)) gap = -16.f;
    mx = gap > mx ? gap : mx;
  }
  float sum = 0.f;
  for (int k = 0; k <= D; ++k) {
    float gap = -fabsf(d - ((float)k * step + off)) * gamma;
    if (!(gap >= -16.f)) gap = -16.f;
    sum += expf(gap - mx);
  }
  float* o = out + (bn * D) * (int64_t)H * W + (int64_t)y * W + x;
  for (int k = 0; k < D; ++k) {
    float gap = -fabsf(d - ((float)k * step + off)) * gamma;
    if (!(gap >= -16.f)) gap = -16.f;
    o[(int64_t)k * H * W] = expf(gap - mx) / sum;
  }
}

inline int launch_status() {
  return hipGetLastError() == hipSuccess ? VEON_OK : VEON_ERR_LAUNCH;
}
}  // namespace

extern "C" {

int veon_downsample_depth(int BN, int H, int W, int ds, const float* depths,
                          float* out, void* stream) {
  if (BN <= 0 || H <= 0 || W <= 0 || ds <= 0 || H % ds || W % ds || !depths || !out)
    return VEON_ERR_BAD_ARG;
  const int64_t n = (int64_t)BN * (H / ds) * (W / ds);
  hipLaunchKernelGGL(k_downsample_depth, dim3((unsigned)((n + kBlock - 1) / kBlock)),
                     dim3(kBlock), 0, static_cast<hipStream_t>(stream), depths, n,
                     H, W, ds, out);
  return launch_status();
}

int veon_two_hot_depth(int BN, int H, int W, int ds, int D, float lo, float step,
                       float gamma, const float* depths, float* out,
                       void* stream) {
  if (BN <= 0 || H <= 0 || W <= 0 || ds < 0 || D <= 0 || !depths || !out)
    return VEON_ERR_BAD_ARG;
  const int64_t n = (int64_t)BN * H * W;
  // bin centres as torch forms them: arange(int64) * python-double step +
  // python-double (lo + step/2), evaluated in float32 (:417-418)
  const float off = (float)((double)lo + (double)step / 2.0);
  hipLaunchKernelGGL(k_two_hot_depth, dim3((unsigned)((n + kBlock - 1) / kBlock)),
                     dim3(kBlock), 0, static_cast<hipStream_t>(stream), depths, n,
                     H, W, ds, D, step, off, gamma, out);
  return launch_status();
}

}  // extern "C"
