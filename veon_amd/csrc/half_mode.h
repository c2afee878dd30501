// Half-precision flavour of this build of the library (see mfma_common.h):
// veon_half_native is the compiler's type for "round an fp32 to the build's 16-bit
// format" outside the MFMA kernels (the pool kernels' padded half output).
#pragma once
#ifdef VEON_HALF_FP16
typedef _Float16 veon_half_native;
#define VEON_HALF_MODE 1
#else
typedef __bf16 veon_half_native;
#define VEON_HALF_MODE 0
#endif
