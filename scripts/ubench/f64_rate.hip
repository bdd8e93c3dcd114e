// f64_rate.hip -- VALU issue rate of the double-precision ops a float64 pass 3 would use (v_add_f64, v_mul_f64, v_fma_f64,
// v_cvt_f64_f32, v_cvt_f32_f64) against v_fma_f32 / v_log_f32, at 1..4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o f64_rate f64_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s) {
  double a[16];
  float f[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    a[i] = (double)threadIdx.x + i;
    f[i] = (float)threadIdx.x + i;
  }
  double w = (double)s, w2 = 1.0 + (double)s;
  float wf = s;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(wf), "v"(wf));
      else if (MODE == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(w));
      else if (MODE == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(w2));
      else if (MODE == 3) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w), "v"(w2));
      else if (MODE == 4) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
      else if (MODE == 5) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
      else if (MODE == 6) asm volatile("v_log_f32 %0, %0" : "+v"(f[i]));
      else if (MODE == 7) asm volatile("v_frexp_mant_f32 %0, %0" : "+v"(f[i]));
      else if (MODE == 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(wf));
    }
  }
  double r = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) r += a[i] + (double)f[i];
  out[blockIdx.x * 256 + threadIdx.x] = (float)r;
}

template <int MODE>
double run(int waves_per_simd, int iters, float *d) {
  int blocks = 256 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1e-9f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1e-9f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3;
}

int main() {
  float *d;
  hipMalloc(&d, 256 * 4 * 256 * 4);
  const int iters = 50000;
  const char *names[] = {"v_fma_f32", "v_add_f64", "v_mul_f64", "v_fma_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_log_f32", "v_frexp_mant_f32", "v_add_f32"};
  for (int w = 1; w <= 4; w++) {
    double t[9];
    t[0] = run<0>(w, iters, d); t[1] = run<1>(w, iters, d); t[2] = run<2>(w, iters, d); t[3] = run<3>(w, iters, d); t[4] = run<4>(w, iters, d);
    t[5] = run<5>(w, iters, d); t[6] = run<6>(w, iters, d); t[7] = run<7>(w, iters, d); t[8] = run<8>(w, iters, d);
    for (int m = 0; m < 9; m++)
      printf("waves/SIMD %d  %-18s %8.3f ms   %.2f ns per wave-instruction per SIMD (v_fma_f32 = %.2f)\n", w, names[m], t[m] * 1e3,
             t[m] * 1e9 / ((double)iters * 16.0 * w), t[0] * 1e9 / ((double)iters * 16.0 * w));
  }
  return 0;
}
