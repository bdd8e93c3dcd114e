// c5_single.hip -- memory-and-ALU SKELETON of a one-kernel 65536-point Welch step with no work buffer, to decide by
// measurement whether that form is worth building against the shipped pair (scn_welch_cols_kernel + scn_welch_rows_kernel: 127 us per
// 32-PSD step).  Decimation in frequency by 16:
//     X[16 q + r] = FFT_4096 over m of  y_r[m] = W_N^{m r} * sum_j x[m + 4096 j] w[m + 4096 j] W_16^{j r}
// Four workgroups per PSD (x `parts` shares of its K segments); workgroup r0 owns the residues r = r0 + 4 c, c = 0 .. 3: for every m it
// loads the 16 samples and window taps x[m + 4096 j], does the radix-4 x radix-4 front end for its four residues, the twiddle, leaves
// four 4096-point streams in 128 KiB of LDS, transforms each in place (three radix-16 passes, 256 threads x 16 points, the in-register
// fft16 of scn_device.h) and accumulates |X|^2 in 64 registers per thread over its segments.  The segment is read by four workgroups.
// OPTIMISTIC on purpose: pass twiddles are a few per-thread constants (no table loads), LDS layouts are plain strides (whatever bank
// conflicts they have, a real kernel would have to beat), no dB map, no combine kernel for parts > 1, outputs stored once per PSD.  Not
// a transform: the values are not checked -- only that every load, flop and LDS exchange of the form is there.
// usage: c5_single [psds 32] [parts 1] [reps 20]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../scanner_amd/csrc/scn_device.h"

constexpr unsigned N = 65536, HOP = N / 2, K = 16, M4 = 4096;

__global__ __launch_bounds__(256, 1) void c5_single_kernel(const v2f *x, const float *win, const v2f *tw, float *out, unsigned n_psd, unsigned parts) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v2f *lds = reinterpret_cast<v2f *>(smem_raw);  // [4 streams][4096 + pad]
  constexpr unsigned SP = M4 + 16;               // stream pitch (slots)
  const unsigned t = threadIdx.x;
  const unsigned r0 = blockIdx.x & 3u, psd = (blockIdx.x >> 2) % n_psd, part = (blockIdx.x >> 2) / n_psd;
  const unsigned s_lo = K * part / parts, s_hi = K * (part + 1) / parts;
  // W_16^{b r0}, b = 1 .. 3 (wave-uniform), and three pass twiddles per thread (stand-ins for the 15 a real kernel keeps)
  cf w16[4], twp[3];
  for (int b = 1; b < 4; b++) w16[b] = from_v2f(tw[(4096u * b * r0) & (N - 1)]);
  for (int p = 0; p < 3; p++) twp[p] = from_v2f(tw[(16u * t * (p + 1)) & (N - 1)]);
  float acc[4][16];
#pragma unroll
  for (int c = 0; c < 4; c++)
#pragma unroll
    for (int q = 0; q < 16; q++) acc[c][q] = 0.f;

  for (unsigned s = s_lo; s < s_hi; s++) {
    const char *seg = reinterpret_cast<const char *>(x) + (size_t)(psd * K + s) * HOP * 8u;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(seg, N * 8u), rw = make_rsrc(win, N * 4u), rt = make_rsrc(tw, N * 8u);
    // ---- front end: 16 values of m per thread, m = t + 256 u
    v2f raw[16];
    float wv[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      raw[j] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rx, t * 8u, j * 32768u, 2));
      wv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, t * 4u, j * 16384u, 0));
    }
    for (unsigned u = 0; u < 16; u++) {
      const unsigned m = t + 256u * u;
      cf v[16];
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = from_v2f(raw[j]) * wv[j];
      if (u + 1 < 16) {  // the next m's samples and taps, fetched while this one is worked on
#pragma unroll
        for (int j = 0; j < 16; j++) {
          raw[j] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rx, (m + 256u) * 8u, j * 32768u, 2));
          wv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, (m + 256u) * 4u, j * 16384u, 0));
        }
      }
      // the twiddles W_N^{m r}, r = r0 + 4 c: four loads from the L2-resident table
      cf wm[4];
#pragma unroll
      for (int c = 0; c < 4; c++) wm[c] = from_v2f(__builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rt, ((m * (r0 + 4u * c)) & (N - 1)) * 8u, 0, 0)));
      // stage 1 (over a, j = 4 a + b): t_b = sum_a v[4a + b] W_4^{a r0} -- one output of a radix-4 per b (r0 picks which; all four
      // cost the same three complex additions with trivial twiddles: take output r0 of radix4)
      cf tb[4];
#pragma unroll
      for (int b = 0; b < 4; b++) {
        cf a0 = v[b], a1 = v[4 + b], a2 = v[8 + b], a3 = v[12 + b];
        radix4(a0, a1, a2, a3);
        tb[b] = r0 == 0 ? a0 : r0 == 1 ? a1 : r0 == 2 ? a2 : a3;  // (wave-uniform select)
      }
      // stage 2 (over b): y_c = sum_b W_16^{b r0} W_4^{b c} t_b
      tb[1] = cmul(tb[1], w16[1]);
      tb[2] = cmul(tb[2], w16[2]);
      tb[3] = cmul(tb[3], w16[3]);
      radix4(tb[0], tb[1], tb[2], tb[3]);
#pragma unroll
      for (int c = 0; c < 4; c++) lds[c * SP + m] = to_v2f(cmul(tb[c], wm[c]));
    }
    __syncthreads();
    // ---- four 4096-point transforms in place: three radix-16 passes each (strides 256, 16, 1)
#pragma unroll
    for (int c = 0; c < 4; c++) {  // (unrolled: the accumulators are registers, not an indexed private array)
      v2f *st = lds + c * SP;
      cf v[16];
#pragma unroll
      for (int a = 0; a < 16; a++) v[a] = from_v2f(st[t + 256u * a]);
      fft16(v);
#pragma unroll
      for (int p = 0; p < 16; p++) st[t + 256u * p] = to_v2f(p ? cmul(v[OUT16(p)], twp[p % 3]) : v[OUT16(p)]);
      __syncthreads();
      const unsigned hi = t >> 4, lo = t & 15u;
#pragma unroll
      for (int a = 0; a < 16; a++) v[a] = from_v2f(st[256u * hi + lo + 16u * a]);
      fft16(v);
#pragma unroll
      for (int p = 0; p < 16; p++) st[256u * hi + lo + 16u * p] = to_v2f(p ? cmul(v[OUT16(p)], twp[(p + 1) % 3]) : v[OUT16(p)]);
      __syncthreads();
#pragma unroll
      for (int a = 0; a < 16; a++) v[a] = from_v2f(st[16u * t + a]);
      fft16(v);
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const cf z = v[OUT16(q)];
        acc[c][q] += __builtin_fmaf(z.y, z.y, z.x * z.x);
      }
    }
    __syncthreads();
  }
  // one store per bin per PSD and part (a real kernel: dB, or partial sums for the combine kernel)
  float *o = out + ((size_t)part * n_psd + psd) * N + r0 * 16384u;
#pragma unroll
  for (int c = 0; c < 4; c++)
#pragma unroll
    for (int q = 0; q < 16; q++) o[c * 4096u + q * 256u + t] = acc[c][q];
}

#define CK(call)                                                                    \
  do {                                                                              \
    hipError_t e_ = (call);                                                         \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));             \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

int main(int argc, char **argv) {
  const unsigned psds = argc > 1 ? (unsigned)atoi(argv[1]) : 32u, parts = argc > 2 ? (unsigned)atoi(argv[2]) : 1u, reps = argc > 3 ? (unsigned)atoi(argv[3]) : 20u;
  const size_t samples = ((size_t)psds * K + 1u) * HOP;
  const unsigned R = 12;  // rotate the input streams past the Infinity Cache (12 x 134 MB)
  std::vector<char *> xs(R);
  for (unsigned r = 0; r < R; r++) {
    CK(hipMalloc(&xs[r], samples * 8u));
    CK(hipMemset(xs[r], 0x3c + (int)r, samples * 8u));
  }
  float *win, *out;
  v2f *tw;
  CK(hipMalloc(&win, N * 4u));
  CK(hipMalloc(&tw, N * 8u));
  CK(hipMalloc(&out, (size_t)parts * psds * N * 4u));
  CK(hipMemset(win, 0x3c, N * 4u));
  CK(hipMemset(tw, 0x3c, N * 8u));
  const size_t lds = 4u * (M4 + 16u) * 8u;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(c5_single_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const unsigned grid = 4u * psds * parts;
  for (unsigned k = 0; k < 5; k++) hipLaunchKernelGGL(c5_single_kernel, dim3(grid), dim3(256), lds, 0, reinterpret_cast<const v2f *>(xs[k % R]), win, tw, out, psds, parts);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (unsigned k = 0; k < reps; k++) hipLaunchKernelGGL(c5_single_kernel, dim3(grid), dim3(256), lds, 0, reinterpret_cast<const v2f *>(xs[k % R]), win, tw, out, psds, parts);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, new_samples = (double)psds * K * HOP;
  printf("c5_single: %u PSDs x %u parts = %u workgroups of 256 threads, %zu B of LDS each: %.1f us per step, %.1f Gsamples/s of new samples "
         "(the shipped pair: 127 us per 32-PSD step = 130 Gsamples/s)\n",
         psds, parts, grid, lds, us, new_samples / us / 1e3);
  return 0;
}
