// cadence.hip -- VALU issue cadence of one wave as a function of how many OTHER waves are resident on its SIMD,
// and of whether those are issuing VALU, sleeping (s_sleep) or parked at a barrier-like LDS wait.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

// role 0: worker (timed VALU stream).  role 1: busy (same VALU stream, untimed).  role 2: sleeper.
__global__ __launch_bounds__(256) void k(unsigned long long *out, int iters, int workers_layers, int other_role) {
  const int layer = blockIdx.x / 256;  // the dispatcher fills the 256 CUs layer by layer (checked by placement.hip)
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = threadIdx.x + i;
  float w = 1e-9f;
  if (layer < workers_layers || other_role == 1) {
    unsigned long long t0 = wall_clock64(), c0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
    }
    unsigned long long t1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[blockIdx.x] = (layer < workers_layers) ? (t1 - t0) : 0; out[8192 + blockIdx.x] = c1 - c0; }
  } else {
    // sleep for roughly as long as the workers run (their time is printed; over-sleeping only lengthens the launch)
    for (int it = 0; it < iters / 24; it++) __builtin_amdgcn_s_sleep(127);
    if (threadIdx.x == 0) out[blockIdx.x] = 0;
  }
  float r = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) r += a[i];
  if (r == 12345.f) out[0] = 1;
}

int main() {
  unsigned long long *d; hipMalloc(&d, 8 * 16384);
  const int iters = 60000;
  printf("ns per v_add_f32 for a timed wave (median over worker workgroups); wall_clock64 = 100 MHz\n");
  for (int other_role = 1; other_role <= 2; other_role++)
    for (int total = 1; total <= 8; total++) {
      int workers = (other_role == 1) ? total : 1;
      if (other_role == 2 && total == 1) continue;
      int blocks = 256 * total;
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, workers, other_role);
      hipDeviceSynchronize();
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, workers, other_role);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(blocks);
      hipMemcpy(h.data(), d, 8 * blocks, hipMemcpyDeviceToHost);
      std::vector<unsigned long long> hc(blocks);
      hipMemcpy(hc.data(), d + 8192, 8 * blocks, hipMemcpyDeviceToHost);
      double ghz = (double)hc[0] / (h[0] * 10.0);
      std::vector<double> t;
      for (int b = 0; b < 256 * workers; b++) t.push_back(h[b] * 10.0 / (iters * 16.0));
      std::sort(t.begin(), t.end());
      printf("%d waves/SIMD resident, %d issuing VALU, %d %s:  %.2f ns per instr per wave (min %.2f max %.2f)  -> SIMD: %.2f ns per instr; clock64/wall = %.3f GHz, %.2f clk per instr per wave\n",
             total, workers, total - workers, other_role == 2 ? "sleeping" : "-", t[t.size() / 2], t.front(), t.back(),
             t[t.size() / 2] / workers, ghz, (double)hc[0] / (iters * 16.0));
    }
}
