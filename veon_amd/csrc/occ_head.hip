// Tail of the occupancy path in one pass (SANInVeonTemporal.forward,
// san_in_veon_temporal.py:196-211, and VEONTemporal.simple_test,
// detectors/veon_temporal.py:219-227): trilinear upsampling (align_corners=False)
// of the class logits and of the two occupancy logits from the head's grid to the
// evaluation grid, softmax over the classes -> best class, softmax over
// (occupied, free) -> keep, class-or-free label written in the (X, Y, Z) order of
// the benchmark.
//
// The reference runs this as ~10 tensor ops over 48 MB of fp32 outputs; here every
// output element is written once and nothing else touches HBM (the 6 MB of
// low-resolution logits stay in L2): the kernel is bound by its output writes.
// One lane = one output voxel, x fastest: every channel plane is written
// coalesced; the 8-byte labels go out transposed (strided), 5 MB in all.
//
// Interpolation follows ATen's upsample_trilinear3d (area_pixel_compute_source_index
// with align_corners=False: src = scale*(dst+0.5)-0.5 clamped at 0, the nested
// w/h/t blend); the label is the first maximum of the class logits (= of their
// softmax, up to ties at rounding level).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "veon_hip.h"

namespace {

struct Strides5 {
  int64_t b, c, z, y, x;
};

struct Axis {
  int i0, i1;
  float l0, l1;
};

__device__ __forceinline__ Axis source(int dst, float scale, int in_size) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  Axis a;
  a.i0 = (int)src;
  a.i1 = a.i0 + (a.i0 < in_size - 1 ? 1 : 0);
  a.l1 = src - (float)a.i0;
  a.l0 = 1.f - a.l1;
  return a;
}

// the eight corner offsets (elements, within one batch element and channel) of a
// voxel; 32-bit: the host checks the extent
struct Corners {
  int o[8];
};

__device__ __forceinline__ Corners corners(const Strides5& s, const Axis& az,
                                           const Axis& ay, const Axis& ax) {
  const int z0 = az.i0 * (int)s.z, z1 = az.i1 * (int)s.z;
  const int y0 = ay.i0 * (int)s.y, y1 = ay.i1 * (int)s.y;
  const int x0 = ax.i0 * (int)s.x, x1 = ax.i1 * (int)s.x;
  return Corners{{z0 + y0 + x0, z0 + y0 + x1, z0 + y1 + x0, z0 + y1 + x1,
                  z1 + y0 + x0, z1 + y0 + x1, z1 + y1 + x0, z1 + y1 + x1}};
}

__device__ __forceinline__ float blend(const float* __restrict__ p, const Corners& k,
                                       const Axis& az, const Axis& ay, const Axis& ax) {
  return az.l0 * (ay.l0 * (ax.l0 * p[k.o[0]] + ax.l1 * p[k.o[1]]) +
                  ay.l1 * (ax.l0 * p[k.o[2]] + ax.l1 * p[k.o[3]])) +
         az.l1 * (ay.l0 * (ax.l0 * p[k.o[4]] + ax.l1 * p[k.o[5]]) +
                  ay.l1 * (ax.l0 * p[k.o[6]] + ax.l1 * p[k.o[7]]));
}

// VEC4: the class logits are channels-last (channel stride 1, rows 16-byte aligned,
// at least 4*ceil(Q/4) floats per row): four classes per 16-byte corner load -- the
// kernel is bound by the number of cache lines its gathers touch, 3.4x fewer so.
template <bool VEC4>
__global__ __launch_bounds__(256) void k_occ_classify(
    const float* __restrict__ sem, Strides5 ss, int Q, const float* __restrict__ bin,
    Strides5 bs, int B, int zi, int yi, int xi, int Zo, int Yo, int Xo, float scz,
    float scy, float scx, float* __restrict__ sem_out, float* __restrict__ bin_out,
    int64_t* __restrict__ cls_out) {
  // 32-bit index arithmetic (the host checks B*Zo*Yo*Xo < 2^31): 64-bit divisions
  // cost ~100 VGPRs here
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  const unsigned plane32 = (unsigned)Zo * Yo * Xo;
  if (idx >= (unsigned)B * plane32) return;
  const int b = (int)(idx / plane32);
  const unsigned v = idx - (unsigned)b * plane32;
  const int x = (int)(v % (unsigned)Xo);
  const int y = (int)((v / (unsigned)Xo) % (unsigned)Yo);
  const int z = (int)(v / ((unsigned)Xo * Yo));
  const int64_t plane = plane32;
  const Axis az = source(z, scz, zi), ay = source(y, scy, yi), ax = source(x, scx, xi);

  // One pass over the classes (a rolled loop: unrolled, the 8 x Q corner loads are
  // all hoisted and the kernel drops to one wave per SIMD with scratch).  The best
  // class of softmax(sem) is the first maximum of the logits; its probability
  // 1 / sum(exp(v - vmax)) is positive exactly when the logits hold no NaN and the
  // maximum is finite (otherwise ATen's softmax is NaN and `score > 0` is false).
  float vmax = -INFINITY;
  int cls = 0;
  bool bad = false;
  const float* sp = sem + b * ss.b;
  float* so = sem_out + (int64_t)b * Q * plane + v;
  const Corners ks = corners(ss, az, ay, ax);
  if (VEC4) {
#pragma unroll 1
    for (int c = 0; c < Q; c += 4) {
      float4 k4[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) k4[q] = *reinterpret_cast<const float4*>(sp + c + ks.o[q]);
      float val4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        auto at = [&](int q) { return reinterpret_cast<const float*>(&k4[q])[e]; };
        val4[e] = az.l0 * (ay.l0 * (ax.l0 * at(0) + ax.l1 * at(1)) +
                           ay.l1 * (ax.l0 * at(2) + ax.l1 * at(3))) +
                  az.l1 * (ay.l0 * (ax.l0 * at(4) + ax.l1 * at(5)) +
                           ay.l1 * (ax.l0 * at(6) + ax.l1 * at(7)));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (c + e < Q) {
          const float val = val4[e];
          so[(int64_t)(c + e) * plane] = val;
          bad |= val != val;
          if (val > vmax) {
            vmax = val;
            cls = c + e;
          }
        }
      }
    }
  } else {
#pragma unroll 2
    for (int c = 0; c < Q; ++c) {
      const float val = blend(sp + c * ss.c, ks, az, ay, ax);
      so[(int64_t)c * plane] = val;
      bad |= val != val;
      if (val > vmax) {
        vmax = val;
        cls = c;
      }
    }
  }
  const bool scored = !bad && vmax < INFINITY && vmax > -INFINITY;
  const float* bp = bin + b * bs.b;
  const Corners kb = corners(bs, az, ay, ax);
  const float o0 = blend(bp, kb, az, ay, ax), o1 = blend(bp + bs.c, kb, az, ay, ax);
  bin_out[((int64_t)b * 2 + 0) * plane + v] = o0;
  bin_out[((int64_t)b * 2 + 1) * plane + v] = o1;
  const float om = o0 > o1 ? o0 : o1;
  const float e0 = expf(o0 - om), e1 = expf(o1 - om);
  const bool keep = scored && (e0 / (e0 + e1) > 0.5f);
  cls_out[(((int64_t)b * Xo + x) * Yo + y) * Zo + z] = keep ? cls : Q;
}

inline int launch_status() {
  return hipGetLastError() == hipSuccess ? VEON_OK : VEON_ERR_LAUNCH;
}

}  // namespace

extern "C" int veon_occ_classify(const float* sem, const int64_t* sem_strides, int Q,
                                 const float* bin, const int64_t* bin_strides, int B,
                                 int zi, int yi, int xi, int Zo, int Yo, int Xo,
                                 float* sem_out, float* bin_out, int64_t* cls_out,
                                 void* stream) {
  if (!sem || !bin || !sem_strides || !bin_strides || !sem_out || !bin_out || !cls_out ||
      Q <= 0 || B <= 0 || zi <= 0 || yi <= 0 || xi <= 0 || Zo <= 0 ||
      Yo <= 0 || Xo <= 0)
    return VEON_ERR_BAD_ARG;
  const int64_t total = (int64_t)B * Zo * Yo * Xo;
  const int64_t blocks = (total + 255) / 256;
  if (total > 0x7fffffffLL) return VEON_ERR_BAD_ARG;
  for (const int64_t* st : {sem_strides, bin_strides}) {  // 32-bit corner offsets
    if (st[2] < 0 || st[3] < 0 || st[4] < 0 ||
        st[2] * (zi - 1) + st[3] * (yi - 1) + st[4] * (xi - 1) > 0x7fffffffLL)
      return VEON_ERR_BAD_ARG;
  }
  const Strides5 ss{sem_strides[0], sem_strides[1], sem_strides[2], sem_strides[3],
                    sem_strides[4]};
  const Strides5 bs{bin_strides[0], bin_strides[1], bin_strides[2], bin_strides[3],
                    bin_strides[4]};
  // ATen: scale = in / out in float (area_pixel_compute_scale, no scale_factor given)
  const float scz = (float)zi / (float)Zo, scy = (float)yi / (float)Yo,
              scx = (float)xi / (float)Xo;
  const int q4 = (Q + 3) / 4 * 4;
  const bool vec4 = ss.c == 1 && (reinterpret_cast<uintptr_t>(sem) & 15u) == 0 &&
                    ss.b % 4 == 0 && ss.z % 4 == 0 && ss.y % 4 == 0 && ss.x % 4 == 0 &&
                    ss.x >= q4;
  if (vec4)
    hipLaunchKernelGGL(k_occ_classify<true>, dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), sem, ss, Q, bin, bs, B, zi, yi,
                       xi, Zo, Yo, Xo, scz, scy, scx, sem_out, bin_out, cls_out);
  else
    hipLaunchKernelGGL(k_occ_classify<false>, dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), sem, ss, Q, bin, bs, B, zi, yi,
                       xi, Zo, Yo, Xo, scz, scy, scx, sem_out, bin_out, cls_out);
  return launch_status();
}
