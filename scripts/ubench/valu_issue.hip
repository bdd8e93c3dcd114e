// valu_issue.hip -- how fast does ONE wave (and 2, 3 waves) per SIMD issue scalar vs packed f32 VALU?
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s) {
  // 16 independent accumulators (pairs), like the 16 complex values of a butterfly
  v2f a[16];
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = v2f{(float)threadIdx.x + i, (float)i};
  v2f w = v2f{s, -s};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (MODE == 0) {  // scalar: 2 v_fma_f32 per complex value
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(w.x), "v"(w.y));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(w.x), "v"(w.y));
      } else if (MODE == 1) {  // packed: 1 v_pk_fma_f32
        asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(w));
      } else if (MODE == 2) {  // packed add with swap + neg modifiers (a + (-i) w)
        asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(w));
      } else if (MODE == 3) {  // scalar adds
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(w.y));
        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(w.x));
      } else if (MODE == 4) {  // dependent chain packed: each op depends on the previous (latency)
        asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a[0]) : "v"(w));
      } else {                 // dependent chain scalar
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[0].x) : "v"(w.x), "v"(w.y));
      }
    }
  }
  v2f r = a[0];
#pragma unroll
  for (int i = 1; i < 16; i++) r += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y;
}

template <int MODE>
double run(int waves_per_simd, int iters, float *d) {
  int cus = 256;
  int blocks = cus * waves_per_simd;  // one 256-thread workgroup = one wave on each of the CU's 4 SIMDs
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1e-9f);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1e-9f);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1e-9f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1e-9f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3;
}

int main() {
  float *d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4 * 2);
  const int iters = 100000;
  const char *names[] = {"scalar 2xv_fma_f32", "packed v_pk_fma_f32", "packed v_pk_add_f32 op_sel/neg", "scalar v_add+v_sub",
                         "dependent v_pk_fma_f32", "dependent v_fma_f32"};
  printf("complex-element ops per SIMD per ns (one 'op' = one complex value updated; 16 per inner loop)\n");
  for (int w = 1; w <= 4; w++) {
    double t[6];
    t[0] = run<0>(w, iters, d); t[1] = run<1>(w, iters, d); t[2] = run<2>(w, iters, d); t[3] = run<3>(w, iters, d);
    t[4] = run<4>(w, iters, d); t[5] = run<5>(w, iters, d);
    for (int m = 0; m < 6; m++) {
      double ops = (double)iters * 16 * w;  // per SIMD
      printf("waves/SIMD %d  %-32s %8.3f ms   %6.3f elem-ops/ns/SIMD  (%.2f ns per op per wave)\n", w, names[m], t[m] * 1e3,
             ops / (t[m] * 1e9), t[m] * 1e9 / (iters * 16.0));
    }
  }
  return 0;
}
