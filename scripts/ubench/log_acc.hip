// log_acc.hip -- accuracy of the dB map  d = (5/log2 10) * log2(P)  (utility.cpp:86-98 computes it in double and rounds
// once) in three device forms, against the correctly rounded value:
//   A  k * v_log_f32(P)                                   (round 1-2 product form)
//   B  exponent split: P = m 2^e, m in [1, 2):  fma(k, v_log_f32(m), k_hi e) + k_lo e   (log error scaled to [0, 1))
//   C  B with the mantissa folded to [sqrt(1/2), sqrt 2) so that log2(m) is in [-1/2, 1/2)
// Reports, per decade of P, max and rms error in units of the result's ulp.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ float db_a(float p) { return 1.50514997831990597607f * __builtin_amdgcn_logf(p); }
__device__ __forceinline__ float db_b(float p) {
  const float m = __builtin_amdgcn_frexp_mantf(p) * 2.0f;           // [1, 2)
  const float e = (float)(__builtin_amdgcn_frexp_expf(p) - 1);
  const float KH = 1.505126953125f;                                  // 12 significant bits of k: KH * e is exact for |e| < 4096
  const float KL = (float)(1.50514997831990597607 - 1.505126953125);
  return __builtin_fmaf(KL, e, __builtin_fmaf(1.50514997831990597607f, __builtin_amdgcn_logf(m), KH * e));
}
__device__ __forceinline__ float db_c(float p) {
  float m = __builtin_amdgcn_frexp_mantf(p);                         // [1/2, 1)
  int ei = __builtin_amdgcn_frexp_expf(p);
  const bool lo = m < 0.70710678118654752440f;
  m = lo ? m * 2.0f : m;                                             // [sqrt 1/2, sqrt 2)
  const float e = (float)(lo ? ei - 1 : ei);
  const float KH = 1.505126953125f;
  const float KL = (float)(1.50514997831990597607 - 1.505126953125);
  return __builtin_fmaf(KL, e, __builtin_fmaf(1.50514997831990597607f, __builtin_amdgcn_logf(m), KH * e));
}
__global__ void k(const float *p, float *a, float *b, float *c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    a[i] = db_a(p[i]);
    b[i] = db_b(p[i]);
    c[i] = db_c(p[i]);
  }
}
int main() {
  const int n = 1 << 22;
  std::vector<float> p(n), a(n), b(n), c(n);
  srand(1);
  float *dp, *da, *db, *dc;
  hipMalloc(&dp, 4 * n); hipMalloc(&da, 4 * n); hipMalloc(&db, 4 * n); hipMalloc(&dc, 4 * n);
  for (int dec = -6; dec <= 10; dec += 2) {
    for (int i = 0; i < n; i++) p[i] = (float)std::pow(10.0, dec + 2.0 * (rand() / (double)RAND_MAX));
    hipMemcpy(dp, p.data(), 4 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dp, da, db, dc, n);
    hipMemcpy(a.data(), da, 4 * n, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, 4 * n, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, 4 * n, hipMemcpyDeviceToHost);
    double mx[3] = {0, 0, 0}, sq[3] = {0, 0, 0}, mrel[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) {
      const double ref = 5.0 * std::log10((double)p[i]);
      const float rf = (float)ref;
      const double ulp = std::fabs((double)std::nextafterf(rf, INFINITY) - (double)rf);
      const float *v[3] = {&a[i], &b[i], &c[i]};
      for (int m = 0; m < 3; m++) {
        const double e = std::fabs((double)*v[m] - ref) / ulp;
        if (e > mx[m]) mx[m] = e;
        sq[m] += e * e;
        const double rel = std::fabs((double)*v[m] - ref) * std::log(10.0) / 5.0;  // as a relative power error
        if (rel > mrel[m]) mrel[m] = rel;
      }
    }
    printf("P in 1e%+d..1e%+d (dB %6.1f..%6.1f): A max %.2f rms %.2f ulp (rel power %.2e) | B max %.2f rms %.2f (%.2e) | C max %.2f rms %.2f (%.2e)\n", dec, dec + 2,
           5.0 * dec, 5.0 * (dec + 2), mx[0], std::sqrt(sq[0] / n), mrel[0], mx[1], std::sqrt(sq[1] / n), mrel[1], mx[2], std::sqrt(sq[2] / n), mrel[2]);
  }
  return 0;
}
