// scn_wave.hip -- EXPERIMENT: one wavefront per 4096-point FFT.
//
// 4096 = 64 x 64: lane t (0..63) owns column t (samples 64a + t), does a 64-point DFT in registers,
// applies W_4096^{t p}, transposes through a wave-private LDS tile (64 x 65 complex), does the second
// 64-point DFT and writes bins k = t + 64 q.  No workgroup barriers, one LDS exchange instead of
// two; 512 VGPRs per lane hold the data (128), the next buffer's prefetched samples (128), the
// window taps (64) and 14 base twiddles.  One 64-thread workgroup = one wave; 33 KiB LDS each
// -> 4 per CU = one per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../scn_device.h"
#include "scn_w64.inc"

namespace {

// In-register 64-point forward DFT: 16 radix-4 butterflies over the top digit, twiddle W64^{i k0},
// then four 16-point DFTs.  On return X[k] sits in v[OUT64(k)].
#define OUT64(k) (16 * ((k) & 3) + OUT16((k) >> 2))
__device__ __forceinline__ void fft64(cf v[64]) {
#pragma unroll
  for (int i = 0; i < 16; i++) radix4(v[i], v[i + 16], v[i + 32], v[i + 48]);
#pragma unroll
  for (int k0 = 1; k0 < 4; k0++)
#pragma unroll
    for (int i = 1; i < 16; i++) {
      const int m = i * k0;  // < 64
      v[i + 16 * k0] = (m == 16) ? cf{v[i + 16 * k0].y, -v[i + 16 * k0].x}
                                 : cmul(v[i + 16 * k0], cf{kW64C[m], -kW64S[m]});
    }
#pragma unroll
  for (int k0 = 0; k0 < 4; k0++) fft16(v + 16 * k0);
}

}  // namespace

#ifndef SCN_WAVE_V
#define SCN_WAVE_V 1
#endif
__global__ __launch_bounds__(64, SCN_WAVE_V == 1 ? 1 : 2) void scn_fft4096_wave_kernel(ScnFftArgs args) {
  constexpr uint32_t N = 4096, PITCH = 65;
  constexpr bool PF = SCN_WAVE_V == 1;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v2f *lds = reinterpret_cast<v2f *>(smem_raw);
  const uint32_t t = threadIdx.x;

  cf twa[8], twb[8];  // W^{t p0}, W^{8 t p1}
#pragma unroll
  for (int j = 1; j < 8; j++) {
    twa[j] = from_v2f(args.twiddle[(t * j) & (N - 1)]);
    twb[j] = from_v2f(args.twiddle[(8 * t * j) & (N - 1)]);
  }
  float win[64];
#pragma unroll
  for (int a = 0; a < 64; a++) win[a] = args.window[64 * a + t];

  v2f raw[64];
  if (PF && blockIdx.x < args.n_buffers) {
    __amdgpu_buffer_rsrc_t r0 = make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)blockIdx.x * N * 8u, N * 8u);
#pragma unroll
    for (int a = 0; a < 64; a++)
      raw[a] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r0, t * 8u, a * 512u, 2));
  }
  for (uint32_t buf = blockIdx.x; buf < args.n_buffers; buf += gridDim.x) {
    cf v[64];
    if (!PF) {
      __amdgpu_buffer_rsrc_t rc = make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)buf * N * 8u, N * 8u);
#pragma unroll
      for (int a = 0; a < 64; a++)
        raw[a] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rc, t * 8u, a * 512u, 2));
    }
#pragma unroll
    for (int a = 0; a < 64; a++) v[a] = from_v2f(raw[a]) * win[a];
    const uint32_t nxt = buf + gridDim.x;
    if (PF && nxt < args.n_buffers) {
      __amdgpu_buffer_rsrc_t rn = make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)nxt * N * 8u, N * 8u);
#pragma unroll
      for (int a = 0; a < 64; a++)
        raw[a] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rn, t * 8u, a * 512u, 2));
    }
    fft64(v);
#pragma unroll
    for (int p = 0; p < 64; p++) {
      const int p0 = p & 7, p1 = p >> 3;
      if (p0 && p1) v[OUT64(p)] = cmul(v[OUT64(p)], cmul(twa[p0], twb[p1]));
      else if (p0) v[OUT64(p)] = cmul(v[OUT64(p)], twa[p0]);
      else if (p1) v[OUT64(p)] = cmul(v[OUT64(p)], twb[p1]);
    }
#if SCN_WAVE_V == 1
#pragma unroll
    for (int p = 0; p < 64; p++) lds[p * PITCH + t] = to_v2f(v[OUT64(p)]);
    __syncthreads();  // single-wave workgroup: orders the LDS writes before the reads
#pragma unroll
    for (int c = 0; c < 64; c++) v[c] = from_v2f(lds[t * PITCH + c]);
    __syncthreads();
#else
    // two-phase transpose through a 32-row tile: rows 0..31 feed lanes 0..31, rows 32..63 feed lanes 32..63
    cf u[64];
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
      for (int p = 0; p < 32; p++) lds[p * PITCH + t] = to_v2f(v[OUT64(32 * h + p)]);
      __syncthreads();
      if ((t >> 5) == (uint32_t)h) {
#pragma unroll
        for (int c = 0; c < 64; c++) u[c] = from_v2f(lds[(t & 31) * PITCH + c]);
      }
      __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < 64; c++) v[c] = u[c];
#endif
    fft64(v);
    __amdgpu_buffer_rsrc_t rout = make_rsrc(args.power_db + (size_t)buf * N, args.power_db ? 4u * N : 0u);
#pragma unroll
    for (int q = 0; q < 64; q++) {
      const cf x = v[OUT64(q)];
      const float d = 1.50514997831990597607f * __builtin_amdgcn_logf(__builtin_fmaf(x.y, x.y, x.x * x.x));
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d), rout, t * 4u, 256u * q, 2);
    }
  }
}

hipError_t scn_launch_fft4096_wave(const ScnFftArgs &a, int num_cus, hipStream_t s) {
  const size_t lds = (SCN_WAVE_V == 1 ? 64 : 32) * 65 * 8;
  int grid = num_cus * (SCN_WAVE_V == 1 ? 4 : 8);
  if ((uint32_t)grid > a.n_buffers) grid = (int)a.n_buffers;
  hipLaunchKernelGGL(scn_fft4096_wave_kernel, dim3(grid), dim3(64), lds, s, a);
  return hipGetLastError();
}
