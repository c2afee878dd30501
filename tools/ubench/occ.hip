// Residency census: how many 256-thread workgroups with L bytes of dynamic LDS
// are co-resident per CU?  Each WG records start/end wall clock; all WGs spin
// ~20 us so that residency == max overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void spin(unsigned long long* st, int us) {
  extern __shared__ float lds[];
  unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) st[blockIdx.x * 2] = t0;
  lds[threadIdx.x] = (float)t0;
  while (wall_clock64() - t0 < (unsigned long long)us * 100) { __builtin_amdgcn_s_sleep(10); }
  __syncthreads();
  if (threadIdx.x == 0) st[blockIdx.x * 2 + 1] = wall_clock64() + (lds[5] == -1.f);
}
int main() {
  const int n = 256 * 10;
  unsigned long long* d; hipMalloc(&d, n * 16);
  std::vector<unsigned long long> h(n * 2);
  for (int lds : {1024, 8192, 16384, 17000, 20480, 21504, 23552, 24576, 29000, 32768, 40960}) {
    int occ = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin, 256, lds);
    hipMemset(d, 0, n * 16);
    hipLaunchKernelGGL(spin, dim3(n), dim3(256), lds, 0, d, 30);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
    // max overlap
    std::vector<std::pair<unsigned long long,int>> ev;
    for (int i = 0; i < n; i++) { ev.push_back({h[2*i], 1}); ev.push_back({h[2*i+1], -1}); }
    std::sort(ev.begin(), ev.end());
    int cur = 0, mx = 0; for (auto& e : ev) { cur += e.second; mx = std::max(mx, cur); }
    printf("lds %6d: api occupancy %d/CU, census max resident %d (%.2f/CU)\n", lds, occ, mx, mx / 256.0);
  }
  return 0;
}
