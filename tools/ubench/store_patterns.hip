// Microbenchmark: which store shape lets a channels-first (C planes x V voxels) volume
// be written at the speed of a linear fill, on SEVERAL allocations (the pool kernel's
// time is bimodal in the physical placement of the volume, tools/mall_probe.py).
// Every variant writes the whole volume once per launch; one workgroup = one tile of
// TV voxels x all C planes unless stated.
//   build: hipcc --offload-arch=gfx950 -O3 store_patterns.hip -o store_patterns
//   run:   ./store_patterns [C=80] [nbuf=8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__global__ void fill4(float4* p, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float4 z = make_float4(0, 0, 0, 0);
  for (; i < n4; i += stride) p[i] = z;
}
// the product's shape: 64 voxels x C, wave w stores planes w, w+4, ...: 256 B per store
template <bool NT>
__global__ void cf64(float* out, int c, long vpb) {
  const long t = blockIdx.x;
  const int v = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* ob = out + t * 64 + v;
  for (int cc = w; cc < c; cc += 4) {
    if (NT) __builtin_nontemporal_store(0.f, ob + (long)cc * vpb);
    else ob[(long)cc * vpb] = 0.f;
  }
}
// same tile, wave w stores a CONTIGUOUS block of planes [w*c/4, (w+1)*c/4)
template <bool NT>
__global__ void cf64_blocked(float* out, int c, long vpb) {
  const long t = blockIdx.x;
  const int v = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* ob = out + t * 64 + v;
  const int per = (c + 3) / 4;
  for (int cc = w * per; cc < (w + 1) * per && cc < c; ++cc) {
    if (NT) __builtin_nontemporal_store(0.f, ob + (long)cc * vpb);
    else ob[(long)cc * vpb] = 0.f;
  }
}
// TV = 64 * VPL voxels per tile, VPL consecutive voxels per lane (8 / 16 B stores)
template <int VPL, bool NT>
__global__ void cfwide(float* out, int c, long vpb) {
  const long t = blockIdx.x;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* ob = out + t * (64 * VPL) + l * VPL;
  for (int cc = w; cc < c; cc += 4) {
    float* q = ob + (long)cc * vpb;
    if (VPL == 4) {
      const v4f z4 = {0.f, 0.f, 0.f, 0.f};
      if (NT) __builtin_nontemporal_store(z4, (v4f*)q);
      else *(v4f*)q = z4;
    } else {
      const v2f z2 = {0.f, 0.f};
      if (NT) __builtin_nontemporal_store(z2, (v2f*)q);
      else *(v2f*)q = z2;
    }
  }
}
// one workgroup = G consecutive 64-voxel tiles, written plane by plane: G x 256 B
// contiguous per plane (the tile loop INSIDE the plane loop)
template <int G, bool NT>
__global__ void cf64_group(float* out, int c, long vpb) {
  const long t = (long)blockIdx.x * G;
  const int v = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* ob = out + t * 64 + v;
  for (int cc = w; cc < c; cc += 4)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (NT) __builtin_nontemporal_store(0.f, ob + (long)cc * vpb + g * 64);
      else ob[(long)cc * vpb + g * 64] = 0.f;
    }
}
// grid = (tiles, 2 halves of the planes): each workgroup writes C/2 planes (the
// two-slab variant that showed no bimodality in the product kernel)
template <bool NT>
__global__ void cf64_half(float* out, int c, long vpb) {
  const long t = blockIdx.x;
  const int h = blockIdx.y;
  const int v = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* ob = out + t * 64 + v;
  const int c0 = h * (c / 2), c1 = h ? c : c / 2;
  for (int cc = c0 + w; cc < c1; cc += 4) {
    if (NT) __builtin_nontemporal_store(0.f, ob + (long)cc * vpb);
    else ob[(long)cc * vpb] = 0.f;
  }
}

// the product's 64-voxel tile with an explicit cache policy on the store instruction
#define ASM_STORE(POL) asm volatile("global_store_dword %0, %1, off " POL :: "v"(q), "v"(0.f) : "memory")
template <int POL>
__global__ void cf64_pol(float* out, int c, long vpb) {
  const long t = blockIdx.x;
  const int v = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* ob = out + t * 64 + v;
  for (int cc = w; cc < c; cc += 4) {
    float* q = ob + (long)cc * vpb;
    if (POL == 0) ASM_STORE("");
    if (POL == 1) ASM_STORE("nt");
    if (POL == 2) ASM_STORE("sc0");
    if (POL == 3) ASM_STORE("sc1");
    if (POL == 4) ASM_STORE("sc0 sc1");
    if (POL == 5) ASM_STORE("sc0 nt");
    if (POL == 6) ASM_STORE("sc1 nt");
    if (POL == 7) ASM_STORE("sc0 sc1 nt");
  }
}

template <class F> float timeit(F f, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; i++) f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < iters; i++) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms / iters * 1e3f;
}

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 80;
  const int nbuf = argc > 2 ? atoi(argv[2]) : 8;
  const long vpb = 640000;
  const size_t n = (size_t)C * vpb;
  std::vector<float*> bufs(nbuf);
  for (auto& p : bufs) CK(hipMalloc(&p, n * 4));
  const int it = 60;
  printf("C = %d, %.1f MB per volume, %d buffers; us per launch\n", C, n * 4 / 1e6, nbuf);
#define ROW(name, launch)                                    \
  do {                                                       \
    printf("%-28s", name);                                   \
    for (float* p : bufs) { printf(" %6.1f", timeit([&] { launch; }, it)); } \
    printf("\n"); fflush(stdout);                            \
  } while (0)
  ROW("fill4 linear", hipLaunchKernelGGL(fill4, dim3(8192), dim3(256), 0, 0, (float4*)p, n / 4));
  ROW("cf64 nt (product)", hipLaunchKernelGGL(cf64<true>, dim3(vpb / 64), dim3(256), 0, 0, p, C, vpb));
  ROW("cf64 plain", hipLaunchKernelGGL(cf64<false>, dim3(vpb / 64), dim3(256), 0, 0, p, C, vpb));
  ROW("cf64 asm (none)", hipLaunchKernelGGL(cf64_pol<0>, dim3(vpb / 64), dim3(256), 0, 0, p, C, vpb));
  ROW("cf64 asm nt", hipLaunchKernelGGL(cf64_pol<1>, dim3(vpb / 64), dim3(256), 0, 0, p, C, vpb));
  ROW("cf64 asm sc0", hipLaunchKernelGGL(cf64_pol<2>, dim3(vpb / 64), dim3(256), 0, 0, p, C, vpb));
  ROW("cf64 asm sc1", hipLaunchKernelGGL(cf64_pol<3>, dim3(vpb / 64), dim3(256), 0, 0, p, C, vpb));
  ROW("cf64 asm sc0 sc1", hipLaunchKernelGGL(cf64_pol<4>, dim3(vpb / 64), dim3(256), 0, 0, p, C, vpb));
  ROW("cf64 asm sc0 nt", hipLaunchKernelGGL(cf64_pol<5>, dim3(vpb / 64), dim3(256), 0, 0, p, C, vpb));
  ROW("cf64 asm sc1 nt", hipLaunchKernelGGL(cf64_pol<6>, dim3(vpb / 64), dim3(256), 0, 0, p, C, vpb));
  ROW("cf64 asm sc0 sc1 nt", hipLaunchKernelGGL(cf64_pol<7>, dim3(vpb / 64), dim3(256), 0, 0, p, C, vpb));
  ROW("cf64 blocked planes nt", hipLaunchKernelGGL(cf64_blocked<true>, dim3(vpb / 64), dim3(256), 0, 0, p, C, vpb));
  ROW("cf64 blocked planes plain", hipLaunchKernelGGL(cf64_blocked<false>, dim3(vpb / 64), dim3(256), 0, 0, p, C, vpb));
  ROW("cf128 float2 nt", hipLaunchKernelGGL((cfwide<2, true>), dim3(vpb / 128), dim3(256), 0, 0, p, C, vpb));
  ROW("cf128 float2 plain", hipLaunchKernelGGL((cfwide<2, false>), dim3(vpb / 128), dim3(256), 0, 0, p, C, vpb));
  ROW("cf256 float4 nt", hipLaunchKernelGGL((cfwide<4, true>), dim3(vpb / 256), dim3(256), 0, 0, p, C, vpb));
  ROW("cf256 float4 plain", hipLaunchKernelGGL((cfwide<4, false>), dim3(vpb / 256), dim3(256), 0, 0, p, C, vpb));
  ROW("cf64 x2 tiles nt", hipLaunchKernelGGL((cf64_group<2, true>), dim3(vpb / 128), dim3(256), 0, 0, p, C, vpb));
  ROW("cf64 x4 tiles nt", hipLaunchKernelGGL((cf64_group<4, true>), dim3(vpb / 256), dim3(256), 0, 0, p, C, vpb));
  ROW("cf64 x4 tiles plain", hipLaunchKernelGGL((cf64_group<4, false>), dim3(vpb / 256), dim3(256), 0, 0, p, C, vpb));
  ROW("cf64 half planes nt", hipLaunchKernelGGL(cf64_half<true>, dim3(vpb / 64, 2), dim3(256), 0, 0, p, C, vpb));
  ROW("cf64 half planes plain", hipLaunchKernelGGL(cf64_half<false>, dim3(vpb / 64, 2), dim3(256), 0, 0, p, C, vpb));
  return 0;
}
