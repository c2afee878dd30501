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

// Compact, EXACT form of the same distribution (the (BN,D,H,W)
// tensor is never written).  A pixel's D+1 logits are -gamma*|d - c_k| where that is
// >= -16 and exactly -16 everywhere else, so its weights are: distinct values on the
// contiguous window of unclamped bins [k0, k0+nk) (|d - c_k| <= 16/gamma: at most
// floor(32/(gamma*step)) + 1 bins) and ONE value, the tail exp(-16 - mx)/sum, on every
// other bin.  Per pixel: wts[pix*K + 0] = tail, wts[pix*K + 1 + j] = weight of bin
// k0 + j (the same float expression as k_two_hot_depth: bit-identical values), and
//   win[pix].x = k0 | nk << 16               the unclamped window (bins < D only)
//   win[pix].y = k0' | nk' << 16 | t << 31   the window bins with weight >= eps
//                                            (contiguous: the weights are unimodal),
//                                            t = 1 when the tail weight is >= eps.
// The lift's prepare keeps point (pix, k) iff k is in [k0', k0'+nk') or (t and k is
// outside [k0, k0+nk)); its depth weight is wts[pix*K + (in window ? 1 + k - k0 : 0)].
// eps = 0 keeps every point: the dense lift to the bit, without the dense tensor.
__global__ __launch_bounds__(kBlock) void k_two_hot_window(
    const float* __restrict__ depths, int64_t n_pix, int H, int W, int ds, int D,
    float step, float off, float gamma, float eps, int K, int2* __restrict__ win,
    float* __restrict__ wts) {
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
  float mx = -INFINITY;
  for (int k = 0; k <= D; ++k) {
    float gap = -fabsf(d - ((float)k * step + off)) * gamma;
    if (!(gap >= -16.f)) gap = -16.f;
    mx = gap > mx ? gap : mx;
  }
  float sum = 0.f;
  int k0 = D, k1 = -1;  // unclamped bins among 0..D-1 (a NaN depth: none)
  for (int k = 0; k <= D; ++k) {
    float gap = -fabsf(d - ((float)k * step + off)) * gamma;
    const bool un = gap >= -16.f;
    if (!un) gap = -16.f;
    sum += expf(gap - mx);
    if (un && k < D) {
      k0 = k < k0 ? k : k0;
      k1 = k;
    }
  }
  int nk = k1 >= k0 ? k1 - k0 + 1 : 0;
  if (nk > K - 1) nk = K - 1;  // cannot happen for K from veon_two_hot_window_slots
  float* o = wts + i * K;
  const float tail = expf(-16.f - mx) / sum;
  o[0] = tail;
  int q0 = 0, q1 = -1;
  for (int j = 0; j < K - 1; ++j) {
    float wv = 0.f;
    if (j < nk) {
      float gap = -fabsf(d - ((float)(k0 + j) * step + off)) * gamma;
      if (!(gap >= -16.f)) gap = -16.f;
      wv = expf(gap - mx) / sum;
      if (wv >= eps) {
        if (q1 < 0) q0 = j;
        q1 = j;
      }
    }
    o[1 + j] = wv;
  }
  const int nq = q1 >= q0 ? q1 - q0 + 1 : 0;
  const unsigned t = (tail >= eps) ? 0x80000000u : 0u;
  win[i] = make_int2(k0 | (nk << 16), (int)((unsigned)(k0 + q0) | ((unsigned)nq << 16) | t));
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

int veon_two_hot_window_slots(int D, float step, float gamma) {
  if (D <= 0 || !(step > 0.f) || !(gamma > 0.f)) return 0;
  // unclamped bins: |d - c_k| <= 16/gamma, centres `step` apart (+1 for float rounding
  // of the centres), never more than the D bins there are; + 1 slot for the tail
  const double span = 32.0 / ((double)gamma * (double)step);
  int64_t nk = (int64_t)span + 2;
  if (nk > D) nk = D;
  if (nk > 0x7ffe) return 0;  // nk is packed in 15 bits
  return (int)nk + 1;
}

int veon_two_hot_window(int BN, int H, int W, int ds, int D, float lo, float step,
                        float gamma, float eps, int K, const float* depths, int* win,
                        float* wts, void* stream) {
  if (BN <= 0 || H <= 0 || W <= 0 || ds < 0 || D <= 0 || D > 0x7fff || !depths || !win ||
      !wts || !(eps >= 0.f))
    return VEON_ERR_BAD_ARG;
  if (K != veon_two_hot_window_slots(D, step, gamma) || K <= 0) return VEON_ERR_BAD_ARG;
  if (reinterpret_cast<uintptr_t>(win) & 7u) return VEON_ERR_BAD_ARG;
  const int64_t n = (int64_t)BN * H * W;
  if (n * K > 0x7fffffffLL) return VEON_ERR_BAD_ARG;  // compact index is an int32 rank
  const float off = (float)((double)lo + (double)step / 2.0);
  hipLaunchKernelGGL(k_two_hot_window, dim3((unsigned)((n + kBlock - 1) / kBlock)),
                     dim3(kBlock), 0, static_cast<hipStream_t>(stream), depths, n, H, W,
                     ds, D, step, off, gamma, eps, K, reinterpret_cast<int2*>(win), wts);
  return launch_status();
}

}  // extern "C"
