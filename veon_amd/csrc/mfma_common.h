// Shared typedefs / helpers of the MFMA kernels (vit_block.hip, conv3d.hip).
// Fragment convention for v_mfma_f32_16x16x32_bf16 (guide section 3): lane l
// holds A[row l&15][k = 8(l>>4)+j] and B[k = 8(l>>4)+j][col l&15], j = 0..7;
// D[row 4(l>>4)+reg][col l&15].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/veon_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;

typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;

typedef __bf16 __attribute__((ext_vector_type(2))) bf16pair;
typedef float __attribute__((ext_vector_type(2))) f32x2;

// The 16-bit operand type of every kernel that includes this header is ONE
// compile-time choice: bf16 (libveon_hip.so, the default) or IEEE fp16
// (libveon_hip_f16.so, the same sources compiled with -DVEON_HALF_FP16: BASELINE
// configs[4] asks for fp16).  Kernels keep the historical names (bf16_t, f2bf,
// bf2f, pack_bf16): they mean "the half type of this build".  Both MFMA
// instructions share the 16x16x32 fragment layout above; accumulation is fp32.
#include "half_mode.h"
#ifdef VEON_HALF_FP16
typedef _Float16 __attribute__((ext_vector_type(2))) halfpair;
typedef _Float16 __attribute__((ext_vector_type(8))) half8_native;
__device__ __forceinline__ bf16_t f2bf(float f) {
  // v_cvt_f16_f32: round to nearest even, overflow to +-inf
  return __builtin_bit_cast(bf16_t, (_Float16)f);
}
__device__ __forceinline__ float bf2f(bf16_t h) {
  return (float)__builtin_bit_cast(_Float16, h);
}
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, halfpair));
}
__device__ __forceinline__ f32x4 mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_native, a),
                                                __builtin_bit_cast(half8_native, b), c, 0, 0,
                                                0);
}
#else
__device__ __forceinline__ bf16_t f2bf(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (round to nearest even, NaN kept)
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}
__device__ __forceinline__ float bf2f(bf16_t h) {
  return __uint_as_float(((unsigned)h) << 16);
}
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16pair));
}
__device__ __forceinline__ f32x4 mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
#endif

// XOR key of the 16-byte-chunk permutation inside a 128-byte LDS row (the
// "swizzle"): physical chunk = logical chunk ^ key(row), applied on the DMA source
// address and again on the fragment read.  ds_read_b128 is served in four
// NON-contiguous 16-lane groups ({0-3,12-15,20-27}, ...) over 16 slots of 16 bytes,
// slot = (row & 1) * 8 + physical chunk for 128-byte rows.
//  * token / activation rows: a fragment's 16 lanes read 16 CONSECUTIVE rows from
//    any base -> key = row & 7 is conflict-free.
//  * weight rows: the paired tiles read rows {0-3, 8-11, 16-19, 24-27} (+4) of
//    their 32-row block (weight_row: 16-byte epilogues), where row & 7 only takes
//    four values -> every weight fragment read was 2-way conflicted (8 LDS cycles
//    instead of 4; PMC SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 16/36 = 44.4 % for
//    the 64x128 GEMM tile, 16/60 = 26.7 % for the conv: profiles/r02_lds_pmc.txt).
//    swz_w takes bit 3 of the row in place of bit 2: conflict-free for those rows
//    and for the 16 consecutive rows of an unpaired tile.
__device__ __forceinline__ int swz_a(int row) { return row & 7; }
__device__ __forceinline__ int swz_w(int row) { return (row & 3) | ((row >> 1) & 4); }

// LDS DMA through the buffer path (buffer_load_dwordx4 ... lds, MUBUF): base from a
// raw buffer descriptor, per-lane byte offset in a VGPR, a uniform byte offset in an
// SGPR; lane l lands at dst + 16 l.  Unlike global_load_lds (FLAT encoding) it does
// not make hipcc's waitcnt pass fall back from counted s_waitcnt lgkmcnt(N) to
// lgkmcnt(0) on the LDS reads around it.  The descriptor type exists only in the
// device pass; the host pass just parses kernel bodies.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base) {
  // no bounds (0xffffffff records), DATA_FORMAT = 32-bit (gfx9 family dword 3)
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xffffffff, 0x00020000);
}
__device__ __forceinline__ void buffer_load_lds16(rsrc_t r, lptr_t dst, int voffset,
                                                  int soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voffset, soffset, 0, 0);
}
#else
typedef int rsrc_t;
__device__ inline rsrc_t make_rsrc(const void*) { return 0; }
__device__ inline void buffer_load_lds16(rsrc_t, lptr_t, int, int) {}
#endif

// GELU(x) = x Phi(x) = max(x, 0) - |x| * erfc(|x| / sqrt 2) / 2, erfc by Abramowitz &
// Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 rounding of the output):
// erfc(z) = (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 + 0.3275911 z).  One v_rcp, one
// v_exp, eight FMAs / multiplies: the form with 1 + erf and an IEEE division
// (__frcp_rn expands to the ten-instruction div_scale / div_fmas / div_fixup sequence)
// was 27 vector instructions per element -- 11 us of the 71 us of ViT-L's fc1 + GELU.
// The 1/2 and the 1/sqrt 2 are folded into the constants.
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.f));
  float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
  p = fmaf(p, t, 0.5f * 1.421413741f);
  p = fmaf(p, t, 0.5f * -0.284496736f);
  p = fmaf(p, t, 0.5f * 0.254829592f);
  // exp(-x^2 / 2) = exp2(-x^2 * log2(e) / 2)
  const float e = __builtin_amdgcn_exp2f(ax * ax * -0.72134752044448170f);
  return fmaxf(x, 0.f) - ax * (p * t) * e;
}
inline int launch_status() {
  return hipGetLastError() == hipSuccess ? VEON_OK : VEON_ERR_LAUNCH;
}
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr int kNumCU = 256;  // MI355X

}  // namespace
