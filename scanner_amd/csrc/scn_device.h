// scn_device.h -- device helpers shared by the kernel files: register-resident complex type,
// in-register 16/8-point DFTs, buffer descriptors, wave reductions.  Internal; .hip files only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scn_kernels.h"

typedef scn_v2f v2f;  // memory / LDS element (8 B)
typedef float v16f __attribute__((ext_vector_type(16)));

// Register-resident complex value.  Deliberately two independent floats, not an
// ext_vector: on gfx950 a v_pk_*_f32 costs the same 4 issue cycles as two scalar ops, and
// hipcc's packed complex multiply is 3 packed ops + a move + wait states (~14 cycles)
// against 8 for mul/mul/fma/fma, so scalar arithmetic is the faster form here.
struct cf {
  float x, y;
};
__device__ __forceinline__ cf operator+(cf a, cf b) { return cf{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf operator-(cf a, cf b) { return cf{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cf operator*(cf a, float s) { return cf{a.x * s, a.y * s}; }
__device__ __forceinline__ cf from_v2f(v2f v) { return cf{v.x, v.y}; }
__device__ __forceinline__ v2f to_v2f(cf c) { return v2f{c.x, c.y}; }

namespace {

// 10*log2(sqrt(P))/log2(10) == (5/log2(10)) * log2(P)
__device__ __forceinline__ float power_db(cf x) {
  float p = __builtin_fmaf(x.y, x.y, x.x * x.x);
  return 1.50514997831990597607f * __builtin_amdgcn_logf(p);
}

__device__ __forceinline__ cf cmul(cf a, cf w) {
  return cf{__builtin_fmaf(-a.y, w.y, a.x * w.x), __builtin_fmaf(a.y, w.x, a.x * w.y)};
}
// a * (1 - i) * h  and  a * (-1 - i) * h  (W16^2, W16^6 with h = sqrt(1/2))
__device__ __forceinline__ cf mul_w2(cf a, float h) { return cf{(a.x + a.y) * h, (a.y - a.x) * h}; }
__device__ __forceinline__ cf mul_w6(cf a, float h) { return cf{(a.y - a.x) * h, -(a.x + a.y) * h}; }

// (x0,x1,x2,x3) -> DFT4 with W4 = -i, results left in (x0,x1,x2,x3) = (X0,X1,X2,X3)
__device__ __forceinline__ void radix4(cf &x0, cf &x1, cf &x2, cf &x3) {
  cf t0 = x0 + x2, t1 = x0 - x2, t2 = x1 + x3, t3 = x1 - x3;
  x0 = t0 + t2;
  x2 = t0 - t2;
  x1 = cf{t1.x + t3.y, t1.y - t3.x};  // t1 - i*t3
  x3 = cf{t1.x - t3.y, t1.y + t3.x};  // t1 + i*t3
}

// In-register 16-point forward DFT (radix 4 x 4).  On return X[k] sits in v[OUT16(k)].
#define OUT16(k) (4 * ((k) & 3) + ((k) >> 2))
__device__ __forceinline__ void fft16(cf v[16]) {
  const float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f;
  const float H = 0.70710678118654752440f;
#pragma unroll
  for (int n0 = 0; n0 < 4; n0++) radix4(v[n0], v[n0 + 4], v[n0 + 8], v[n0 + 12]);
  // v[n0 + 4*k0] *= W16^(n0*k0)
  v[5] = cmul(v[5], cf{C1, -S1});           // W^1
  v[9] = mul_w2(v[9], H);                   // W^2
  v[13] = cmul(v[13], cf{S1, -C1});         // W^3
  v[6] = mul_w2(v[6], H);                   // W^2
  v[10] = cf{v[10].y, -v[10].x};            // W^4 = -i
  v[14] = mul_w6(v[14], H);                 // W^6
  v[7] = cmul(v[7], cf{S1, -C1});           // W^3
  v[11] = mul_w6(v[11], H);                 // W^6
  v[15] = cmul(v[15], cf{-C1, S1});         // W^9
#pragma unroll
  for (int k0 = 0; k0 < 4; k0++) radix4(v[4 * k0], v[4 * k0 + 1], v[4 * k0 + 2], v[4 * k0 + 3]);
}

// In-register 8-point forward DFT (radix 4 x 2).  On return X[k] sits in z[OUT8(k)].
#define OUT8(k) (2 * ((k) & 3) + ((k) >> 2))
__device__ __forceinline__ void fft8(cf z[8]) {
  const float H = 0.70710678118654752440f;
  radix4(z[0], z[2], z[4], z[6]);  // even samples -> z[2*k0]
  radix4(z[1], z[3], z[5], z[7]);  // odd samples  -> z[2*k0+1]
  z[3] = mul_w2(z[3], H);          // W8^1
  z[5] = cf{z[5].y, -z[5].x};      // W8^2 = -i
  z[7] = mul_w6(z[7], H);          // W8^3
#pragma unroll
  for (int k0 = 0; k0 < 4; k0++) {
    cf a = z[2 * k0], b = z[2 * k0 + 1];
    z[2 * k0] = a + b;      // X[k0]
    z[2 * k0 + 1] = a - b;  // X[k0 + 4]
  }
}

// ---- global memory access through buffer descriptors ---------------------------------
// A raw buffer resource (SGPR descriptor, wave-uniform base) + one per-lane VGPR offset
// + scalar/immediate offsets: the 16 strided accesses of a thread cost no address VGPRs.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

// sum over the 64 lanes of a wave (result in every lane)
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Wave-wide reductions in 6 DPP steps (row_shr 1/2/4/8, then row_bcast15 / row_bcast31);
// the result is read from lane 63 and returned as a wave-uniform scalar.
#define SCN_DPP_STEP(OP, x, ctrl, rmask) x = x OP (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rmask, 0xf, true)
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t x) {
  SCN_DPP_STEP(|, x, 0x111, 0xf);
  SCN_DPP_STEP(|, x, 0x112, 0xf);
  SCN_DPP_STEP(|, x, 0x114, 0xf);
  SCN_DPP_STEP(|, x, 0x118, 0xf);
  SCN_DPP_STEP(|, x, 0x142, 0xa);  // row_bcast15 into rows 1 and 3
  SCN_DPP_STEP(|, x, 0x143, 0xc);  // row_bcast31 into rows 2 and 3
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ uint32_t wave_add_u32(uint32_t x) {
  SCN_DPP_STEP(+, x, 0x111, 0xf);
  SCN_DPP_STEP(+, x, 0x112, 0xf);
  SCN_DPP_STEP(+, x, 0x114, 0xf);
  SCN_DPP_STEP(+, x, 0x118, 0xf);
  SCN_DPP_STEP(+, x, 0x142, 0xa);
  SCN_DPP_STEP(+, x, 0x143, 0xc);
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

}  // namespace
