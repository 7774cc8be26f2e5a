// How much of a launch of 17 408 one-wave workgroups is spent starting workgroups?  Same work as a persistent grid of 4 096.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(64, 4) void spin_kernel(unsigned long long* out, int frames_per_wg, int clocks, int n_items, unsigned* ctr) {
  __shared__ unsigned lds[2528];  // 10 112 B: 16 workgroups per CU
  volatile unsigned* l = lds;
  unsigned long long acc = 0;
  if (ctr) {  // dynamic: take items until none is left
    for (;;) {
      unsigned p = 0;
      if (threadIdx.x == 0) p = atomicAdd(ctr, 1u);
      p = __builtin_amdgcn_readfirstlane(p);
      if ((int)p >= n_items) break;
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)clocks) { l[threadIdx.x] = (unsigned)acc; acc += l[(threadIdx.x + 1) & 63]; }
    }
  } else {
    for (int f = 0; f < frames_per_wg; f++) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)clocks) { l[threadIdx.x] = (unsigned)acc; acc += l[(threadIdx.x + 1) & 63]; }
    }
  }
  if (acc == 0x1234567) out[0] = acc;
}
int main() {
  unsigned long long* out; unsigned* ctr;
  hipMalloc(&out, 8); hipMalloc(&ctr, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int clocks_list[] = {2500, 5000, 10000};  // s_memtime ticks (100 MHz): 25, 50, 100 us?  -- printed as measured
  for (int clocks : clocks_list) {
    float ms[3];
    for (int mode = 0; mode < 3; mode++) {
      float best = 1e9;
      for (int rep = 0; rep < 5; rep++) {
        hipMemset(ctr, 0, 4);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(spin_kernel, dim3(16384), dim3(64), 0, 0, out, 1, clocks, 0, (unsigned*)nullptr);
        if (mode == 1) hipLaunchKernelGGL(spin_kernel, dim3(4096), dim3(64), 0, 0, out, 4, clocks, 0, (unsigned*)nullptr);
        if (mode == 2) hipLaunchKernelGGL(spin_kernel, dim3(4096), dim3(64), 0, 0, out, 0, clocks, 16384, ctr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1); if (t < best) best = t;
      }
      ms[mode] = best;
    }
    printf("spin %5d ticks per frame: 16384 workgroups %.1f us | 4096 persistent x 4 frames %.1f us | 4096 persistent, atomic queue %.1f us\n",
           clocks, ms[0] * 1e3, ms[1] * 1e3, ms[2] * 1e3);
  }
  return 0;
}
