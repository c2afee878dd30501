// Microbenchmark: write-only HBM ceiling on MI355X for the store shapes the
// pool kernel can use.  Build: hipcc --offload-arch=gfx950 -O3 fill.hip -o fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

// (a) grid-stride float4 stores
__global__ void fill4(float4* p, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  float4 z = make_float4(0,0,0,0);
  for (; i < n4; i += stride) p[i] = z;
}
// (a') nontemporal
__global__ void fill4nt(float4* p, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    float* q = (float*)(p + i);
    __builtin_nontemporal_store(0.f, q); __builtin_nontemporal_store(0.f, q+1);
    __builtin_nontemporal_store(0.f, q+2); __builtin_nontemporal_store(0.f, q+3);
  }
}
// (b) dword per lane, each block writes a contiguous chunk
__global__ void fill1(float* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = 0.f;
}
// (c) the pool kernel's cf store shape: tile of 64 voxels x C channels, wave w
// stores channel cc: 256 B contiguous, channel stride = vpb floats.
__global__ void fill_cf(float* out, int c, long vpb) {
  long t = blockIdx.x;
  int v = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* ob = out + t * 64 + v;
  for (int cc = w; cc < c; cc += 4) ob[(long)cc * vpb] = 0.f;
}
// (d) cf shape with 256-voxel tiles and float4 per lane: wave covers 256 voxels x 1 channel = 1 KB
__global__ void fill_cf4(float* out, int c, long vpb) {
  long t = blockIdx.x;
  int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  float4* ob = (float4*)(out + t * 256) + l;
  float4 z = make_float4(0,0,0,0);
  for (int cc = w; cc < c; cc += 4) ob[(long)cc * (vpb/4)] = z;
}
// (e) each block writes one contiguous 20 KB region (channels-last tile) with float4
__global__ void fill_cl(float4* out, int per_tile4) {
  float4* ob = out + (size_t)blockIdx.x * per_tile4;
  float4 z = make_float4(0,0,0,0);
  for (int i = threadIdx.x; i < per_tile4; i += 256) ob[i] = z;
}

template <class F> float timeit(F f, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; i++) f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < iters; i++) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

int main() {
  const int C = 80; const long vpb = 640000; const size_t n = (size_t)C * vpb;
  float* p; CK(hipMalloc(&p, n * 4));
  const double gb = n * 4 / 1e9;
  int iters = 200;
  for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
    float ms = timeit([&]{ hipLaunchKernelGGL(fill4, dim3(blocks), dim3(256), 0, 0, (float4*)p, n/4); }, iters);
    printf("fill4   grid %6d: %.2f us  %.0f GB/s\n", blocks, ms*1e3, gb/ms*1e3);
  }
  for (int blocks : {2048, 8192}) {
    float ms = timeit([&]{ hipLaunchKernelGGL(fill4nt, dim3(blocks), dim3(256), 0, 0, (float4*)p, n/4); }, iters);
    printf("fill4nt grid %6d: %.2f us  %.0f GB/s\n", blocks, ms*1e3, gb/ms*1e3);
  }
  for (int blocks : {2048, 8192, 32768}) {
    float ms = timeit([&]{ hipLaunchKernelGGL(fill1, dim3(blocks), dim3(256), 0, 0, p, n); }, iters);
    printf("fill1   grid %6d: %.2f us  %.0f GB/s\n", blocks, ms*1e3, gb/ms*1e3);
  }
  { float ms = timeit([&]{ hipLaunchKernelGGL(fill_cf, dim3(vpb/64), dim3(256), 0, 0, p, C, vpb); }, iters);
    printf("fill_cf  (64 vox x C, dword): %.2f us  %.0f GB/s\n", ms*1e3, gb/ms*1e3); }
  { float ms = timeit([&]{ hipLaunchKernelGGL(fill_cf4, dim3(vpb/256), dim3(256), 0, 0, p, C, vpb); }, iters);
    printf("fill_cf4 (256 vox x C, float4): %.2f us  %.0f GB/s\n", ms*1e3, gb/ms*1e3); }
  { float ms = timeit([&]{ hipLaunchKernelGGL(fill_cl, dim3(vpb/64), dim3(256), 0, 0, (float4*)p, 64*C/4); }, iters);
    printf("fill_cl  (64 vox x C contiguous 20KB): %.2f us  %.0f GB/s\n", ms*1e3, gb/ms*1e3); }
  { float ms = timeit([&]{ hipMemsetAsync(p, 0, n*4, 0); }, iters);
    printf("hipMemsetAsync: %.2f us  %.0f GB/s\n", ms*1e3, gb/ms*1e3); }
  // larger than the 256 MiB infinity cache: 1 GiB
  float* q; CK(hipMalloc(&q, (size_t)1<<30));
  { size_t n4 = ((size_t)1<<30)/16; float ms = timeit([&]{ hipLaunchKernelGGL(fill4, dim3(8192), dim3(256), 0, 0, (float4*)q, n4); }, 50);
    printf("fill4 1GiB: %.2f us  %.0f GB/s\n", ms*1e3, 1.073741824/ms*1e3); }
  return 0;
}
