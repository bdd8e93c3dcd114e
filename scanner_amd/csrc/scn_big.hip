// scn_big.hip -- plain 65 536- and 32 768-point plans: the per-buffer path (K1 .. K5) through a four-step 256 x 256 (256 x 128) FFT.
//
// The reference plans an FFT for whatever --count it is given (fft.cpp:4-11, scan.cpp:85).  512 KiB per buffer does not fit a
// CU's LDS, so this size cannot use the fused single-pass kernels; until round 3 it ran the staged double-precision path of
// scn_generic.hip (six passes over the batch, 10 Gsamples/s).  The Welch pipeline (scn_welch.hip) already owned a two-kernel
// four-step transform of exactly this length; these are its two kernels with the per-buffer prologue and epilogue of the
// fused path instead of the Welch ones:
//   n = 256 n1 + n2,  k = k1 + 256 k2
//   scn_big_dc_kernel<KIND>     with correctDC on integer samples only: the buffer's integer sums (utility.cpp:15-26)
//   scn_big_cols_kernel<KIND>   K1 (utility.cpp:9-84) + K2 (process.cpp:28-34), 16 x 256-pt FFTs over n1 per workgroup, twiddle W_N^(n2 k1), Y[k1][n2]
//                               into a tiled work buffer (16 x 16 blocks of 2 KiB: whole blocks move on both sides)
//   scn_big_rows_kernel<HITS, SPEC>  the 256-pt FFTs along n2 (fft.cpp:20-25) -- in double, see there --, K4 (the dB map of scn_device.h), K5 (mask, strict >,
//                               records into the buffer's region: scn_record_hits with the buffer's device counter; one global
//                               atomic per wave that holds a hit).  A thread ends up with bins k1 + 256 (p + 16 q): 4-byte
//                               pieces 1 KiB apart, so the dB values cross the workgroup's LDS once more and leave as 64-byte
//                               runs (16 consecutive k1 per k2 -- the last stage of any four-step produces the HIGH digit).
// HBM traffic per sample: raw in + 8 B out + 8 B in + 4 B out (cfloat: 28 B against the algorithmic 12).
// 32768 points: n = 128 n1 + n2, k = k1 + 256 k2 -- the same column kernel over 8 column tiles, and a row kernel of its own
// (scn_big_rows32k_kernel: 128-point rows as 16 x 8, two 16-row tiles per workgroup).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scn_device.h"

#ifndef SCN_BIG_AUX_IN_POLICY
#define SCN_BIG_AUX_IN_POLICY 2
#endif
namespace {
// cache policy of the column kernel's sample loads: non-temporal (every sample is read once) -- until round 5 they were the one
// streaming access of the library left at the default policy.  One box, two rounds, us per step default -> nt: 65536-pt cfloat
// 195.0 -> 160.7 / 162.1, int16 193.6 / 189.2 -> 168.8 / 168.6, 32768-pt cfloat 191.5 / 195.3 -> 172.8 / 177.6 (profiles/r05_bignt_ab.txt)
constexpr int BIG_AUX_IN = SCN_BIG_AUX_IN_POLICY;
constexpr uint32_t BN = 65536;
constexpr uint32_t BP = 272;  // LDS row pitch (slots): 16 rows of 256 + 16, as in scn_welch.hip

template <int KIND>
struct BigRaw;
template <>
struct BigRaw<SCN_K_FLOAT_COMPLEX> {
  static constexpr uint32_t kBytes = 8;
  typedef v2f raw_t;
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t s, uint32_t a, uint32_t n) {  // sample s + (n / 16) a
    return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, s * 8u, a * (n / 2u), BIG_AUX_IN));
  }
  static __device__ __forceinline__ cf conv(raw_t r, int, int) { return from_v2f(r); }
  static __device__ __forceinline__ void ints(raw_t, int &re, int &im) { re = im = 0; }  // (no DC removal for float samples)
};
template <>
struct BigRaw<SCN_K_SHORT_COMPLEX> {
  static constexpr uint32_t kBytes = 4;
  typedef int raw_t;
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t s, uint32_t a, uint32_t n) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, s * 4u, a * (n / 4u), BIG_AUX_IN);
  }
  // float(source) * onebymax with the scale folded into the window tap (utility.cpp:81-82; onebymax is +-2^-k: exact)
  // (source - dc in int arithmetic with wrap-around, as the oracle's conv1)
  static __device__ __forceinline__ void ints(raw_t r, int &re, int &im) { re = (int)(short)(r & 0xffff); im = r >> 16; }
  static __device__ __forceinline__ cf conv(raw_t r, int dc_re, int dc_im) {
    int re, im;
    ints(r, re, im);
    return cf{(float)(int)((uint32_t)re - (uint32_t)dc_re), (float)(int)((uint32_t)im - (uint32_t)dc_im)};
  }
};
template <>
struct BigRaw<SCN_K_SHORT> {  // planar: I[n] then Q[n]
  static constexpr uint32_t kBytes = 4;
  typedef int raw_t;
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t s, uint32_t a, uint32_t n) {
    const int re = (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, s * 2u, a * (n / 8u), BIG_AUX_IN);
    const int im = (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, s * 2u, n * 2u + a * (n / 8u), BIG_AUX_IN);
    return (re & 0xffff) | (im << 16);
  }
  static __device__ __forceinline__ void ints(raw_t r, int &re, int &im) { BigRaw<SCN_K_SHORT_COMPLEX>::ints(r, re, im); }
  static __device__ __forceinline__ cf conv(raw_t r, int dc_re, int dc_im) { return BigRaw<SCN_K_SHORT_COMPLEX>::conv(r, dc_re, dc_im); }
};
template <>
struct BigRaw<SCN_K_BYTE_COMPLEX> {
  static constexpr uint32_t kBytes = 2;
  typedef int raw_t;
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t s, uint32_t a, uint32_t n) {
    return (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, s * 2u, a * (n / 8u), BIG_AUX_IN);
  }
  static __device__ __forceinline__ void ints(raw_t r, int &re, int &im) { re = (int)(signed char)(r & 0xff); im = (int)(signed char)((r >> 8) & 0xff); }
  static __device__ __forceinline__ cf conv(raw_t r, int dc_re, int dc_im) {
    int re, im;
    ints(r, re, im);
    return cf{(float)(int)((uint32_t)re - (uint32_t)dc_re), (float)(int)((uint32_t)im - (uint32_t)dc_im)};
  }
};
}  // namespace

// ---- DC removal (utility.cpp:9-84 with correctDC): the buffer's integer sums first -- one more pass over the raw samples (2..4 B
// per sample on top of the 22..24 the transform moves) --, grid = 4 parts x n_buffers, 16-byte loads, one atomic pair per workgroup.  int32 sums wrap
// like the reference's accumulators; the column kernel divides (int32 /= uint32, utility.cpp:77-78).
template <int KIND, uint32_t N>
__global__ __launch_bounds__(256) void scn_big_dc_kernel(ScnBigArgs args) {
  typedef BigRaw<KIND> L;
  typedef int v4i __attribute__((__vector_size__(16)));
  __shared__ int part_sums[8];
  const uint32_t b = blockIdx.x >> 2, part = blockIdx.x & 3u, t = threadIdx.x;
  const __amdgpu_buffer_rsrc_t r = make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)b * L::kBytes * N, L::kBytes * N);
  int sr = 0, si = 0;
  if constexpr (KIND == SCN_K_SHORT) {  // planar: a quarter of the I plane and the same quarter of the Q plane
    constexpr uint32_t PB = 2u * N / 4u, NV = PB / 16u / 256u;
    v4i wi[NV], wq[NV];
#pragma unroll
    for (uint32_t a = 0; a < NV; a++) {
      wi[a] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r, (a * 256u + t) * 16u, part * PB, 0));
      wq[a] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r, (a * 256u + t) * 16u, 2u * N + part * PB, 0));
    }
#pragma unroll
    for (uint32_t a = 0; a < NV; a++)
#pragma unroll
      for (int c = 0; c < 4; c++) {
        sr += (int)(short)(wi[a][c] & 0xffff) + (wi[a][c] >> 16);
        si += (int)(short)(wq[a][c] & 0xffff) + (wq[a][c] >> 16);
      }
  } else {
    constexpr uint32_t PB = L::kBytes * N / 4u, NV = PB / 16u / 256u;
    v4i w[NV];
#pragma unroll
    for (uint32_t a = 0; a < NV; a++) w[a] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r, (a * 256u + t) * 16u, part * PB, 0));
#pragma unroll
    for (uint32_t a = 0; a < NV; a++)
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int d = w[a][c];
        if constexpr (KIND == SCN_K_SHORT_COMPLEX) {
          sr += (int)(short)(d & 0xffff);
          si += d >> 16;
        } else {  // int8 pairs: two samples per dword
          sr += (int)(signed char)(d & 0xff) + (int)(signed char)((d >> 16) & 0xff);
          si += (int)(signed char)((d >> 8) & 0xff) + (d >> 24);
        }
      }
  }
  sr = wave_sum(sr);
  si = wave_sum(si);
  if ((t & 63u) == 0) {
    part_sums[t >> 6] = sr;
    part_sums[4 + (t >> 6)] = si;
  }
  __syncthreads();
  if (t < 2) atomicAdd(&args.dc_sums[2 * b + t], part_sums[4 * t] + part_sums[4 * t + 1] + part_sums[4 * t + 2] + part_sums[4 * t + 3]);
}

// ---- columns: grid = CT column tiles x G workgroups (CT = N / 4096: 16 or 8); workgroup (j, g) owns tile j for buffers g, g + G, ... --
// n = NC n1 + n2 with NC = N / 256 columns: thread (hi, lo) of tile j transforms column n2 = 16 j + lo over n1 = 16 a + hi.
template <int KIND, uint32_t N, bool DC>
__global__ __launch_bounds__(256, 3) void scn_big_cols_kernel(ScnBigArgs args) {
  typedef BigRaw<KIND> L;
  constexpr uint32_t NC = N / 256u, CT = NC / 16u;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v2f *lds = reinterpret_cast<v2f *>(smem_raw);
  const uint32_t t = threadIdx.x;
  const uint32_t j = blockIdx.x % CT, g = blockIdx.x / CT, G = gridDim.x / CT;
  const uint32_t hi = t >> 4, lo = t & 15u;  // pass 1: (b, c)   pass 2: (p, c)
  const uint32_t n2 = 16u * j + lo;

  cf twa[16], twb[16];
  float win[16];
#pragma unroll
  for (int p = 1; p < 16; p++) twa[p] = from_v2f(args.twiddle[((N / 256u) * hi * p) & (N - 1)]);  // W_256^{b p}
#pragma unroll
  for (int q = 0; q < 16; q++) twb[q] = from_v2f(args.twiddle[(n2 * (hi + 16u * q)) & (N - 1)]);  // W_N^{n2 k1}
#pragma unroll
  for (int a = 0; a < 16; a++) win[a] = args.window[NC * (16u * a + hi) + n2] * args.scale;

  v2f *w1 = lds + t;             // + p*BP
  v2f *r1 = lds + hi * BP + lo;  // + b*16
  const uint32_t s0 = NC * hi + n2;    // sample index of a = 0; + 16 NC a
  const uint32_t st_voff = t * 8u;     // block (i = q, j) of the tiled work buffer: + (CT q + j) * 2048

  auto in_rsrc = [&](uint32_t b) {
    const bool ok = b < args.n_buffers;
    return make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)(ok ? b : 0u) * L::kBytes * N, ok ? L::kBytes * N : 0u);
  };
  typename L::raw_t raw[16];
  {
    const __amdgpu_buffer_rsrc_t r0 = in_rsrc(g);
#pragma unroll
    for (int a = 0; a < 16; a++) raw[a] = L::load(r0, s0, a, N);
  }
  for (uint32_t b = g; b < args.n_buffers; b += G) {
    __amdgpu_buffer_rsrc_t rwk = make_rsrc(reinterpret_cast<char *>(args.work) + (size_t)b * N * 8u + j * 2048u, N * 8u - j * 2048u);
    if (j == 0 && t == 0 && args.per_buffer_hits) args.per_buffer_hits[b] = 0;  // the row kernel (the next launch) counts into it
    int dc_re = 0, dc_im = 0;
    if (DC) {
      dc_re = (int)((uint32_t)args.dc_sums[2 * b] / N);  // int32 /= uint32, utility.cpp:77-78
      dc_im = (int)((uint32_t)args.dc_sums[2 * b + 1] / N);
    }
    cf v[16];
#pragma unroll
    for (int a = 0; a < 16; a++) v[a] = L::conv(raw[a], dc_re, dc_im) * win[a];
    const __amdgpu_buffer_rsrc_t rn = in_rsrc(b + G);
#pragma unroll
    for (int a = 0; a < 8; a++) raw[a] = L::load(rn, s0, a, N);
    fft16(v);
#pragma unroll
    for (int p = 0; p < 16; p++) {
      cf y = v[OUT16(p)];
      if (p) y = cmul(y, twa[p]);
      w1[p * BP] = to_v2f(y);
    }
    __syncthreads();
#pragma unroll
    for (int bb = 0; bb < 16; bb++) v[bb] = from_v2f(r1[bb * 16]);
#pragma unroll
    for (int a = 8; a < 16; a++) raw[a] = L::load(rn, s0, a, N);
    fft16(v);
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const cf y = cmul(v[OUT16(q)], twb[q]);
      typedef unsigned u2 __attribute__((__vector_size__(8)));
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, to_v2f(y)), rwk, st_voff, q * (CT * 2048u), 0);
    }
    __syncthreads();
  }
}

// ---- rows: grid = 16 row tiles x n_buffers; workgroup (i, b) finishes the 4096 bins k1 in [16 i, 16 i + 16) of buffer b ----
template <bool HITS, bool SPEC>
__global__ __launch_bounds__(256, 3) void scn_big_rows_kernel(ScnBigArgs args) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v2f *lds = reinterpret_cast<v2f *>(smem_raw);
  // the dB tile on its way out (re-uses the exchange area): [k2 (256)][16] floats with the k1 index XOR-swizzled by bits 1..3 of
  // k2, slot(k2, k1) = 16 k2 + (k1 ^ (k2 & 14)).  Written by 32-lane groups of (two k1) x (16 k2 = lo + 16 q) and read by
  // groups of (two consecutive k2) x (16 k1): both hit 32 distinct banks.  (Round 3 used a [k2][17] pitch: lane (k1, lo = 0)
  // and lane (k1 + 1, lo = 15) then share a bank on the way in, rows k2 and k2 + 1 overlap by one bank on the way out --
  // 2-way both times, SQ_LDS_BANK_CONFLICT = 17 % of the kernel's LDS cycles.)
  float *lds_out = reinterpret_cast<float *>(smem_raw);
  const uint32_t t = threadIdx.x, lane = t & 63u;
  const uint32_t i = blockIdx.x & 15u, b = blockIdx.x >> 4;
  const uint32_t hi = t >> 4, lo = t & 15u;  // pass 1: (rho, b)   pass 2: (rho, p)
  const uint32_t k1 = 16u * i + hi;

  // The row transform runs in DOUBLE (the float exchange between its two passes excepted).  At this size a strong tone's
  // partial sums in the last levels are 256x the input, and their float rounding lands on the other 255 bins of the tone's
  // row: the first, all-float form of this kernel read 1.0e-5 .. 1.9e-5 on eleven buffers (peak / mean power ~2e4) of a
  // 30-minute fuzz -- over the parity bar, where the staged double-precision path it replaced never was.  v_add_f64 /
  // v_fma_f64 issue at 0.85x / 0.7x the float rate on gfx950 and this kernel waits for its work-buffer reads anyway.
  // Emulated (48 strong-tone buffers): max 6.9e-6 all float, 3.0e-6 with this, 2.2e-6 with a double exchange as well.
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(reinterpret_cast<const char *>(args.work) + (size_t)b * BN * 8u + i * 32768u, BN * 8u - i * 32768u);
  v2f raw[16];
#pragma unroll
  for (int a = 0; a < 16; a++) raw[a] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rw, t * 8u, a * 2048u, 2));
  // W_256^(b p), p = 1 .. 15, b = lo: from the plan's double table (a float twiddle would put the float error right back)
  scn_v2d twa[16];
#pragma unroll
  for (int p = 1; p < 16; p++) twa[p] = args.tw256[(lo * p) & 255u];
  v2f *w1 = lds + hi * BP + lo * 17u;  // + p        (row rho, slot b*17 + p)
  v2f *r1 = lds + hi * BP + lo;        // + b*17
  cd v[16];
#pragma unroll
  for (int a = 0; a < 16; a++) v[a] = to_cd(raw[a]);
  fft16_d(v);
#pragma unroll
  for (int p = 0; p < 16; p++) {
    cd y = v[OUT16(p)];
    if (p) y = cmul_d(y, twa[p]);
    w1[p] = v2f{(float)y.x, (float)y.y};
  }
  __syncthreads();
#pragma unroll
  for (int bb = 0; bb < 16; bb++) v[bb] = to_cd(r1[bb * 17]);
  fft16_d(v);
  __syncthreads();  // every exchange read done: the area becomes the output tile

  // this thread's 16 bins: k = k1 + 256 k2, k2 = lo + 16 q
  v16f pw;
  float gmax[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const cd xq = v[OUT16(q)];
    const float p = (float)__builtin_fma(xq.y, xq.y, xq.x * xq.x);
    pw[q] = p;
    gmax[q >> 2] = fmaxf(gmax[q >> 2], p);
    if constexpr (SPEC) lds_out[(lo + 16u * q) * 16u + (hi ^ (lo & 14u))] = db_of_power(p);
  }
  if constexpr (SPEC) {
    __syncthreads();
    // 64-byte runs: lanes 0..15 take the 16 consecutive k1 of one k2, four k2 per wave instruction
    __amdgpu_buffer_rsrc_t rout = make_rsrc(args.power_db + (size_t)b * BN, args.power_db ? 4u * BN : 0u);
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const uint32_t k2 = (t >> 4) + 16u * u, kk1 = t & 15u;
      const float d = lds_out[k2 * 16u + (kk1 ^ ((t >> 4) & 14u))];
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d), rout, (16u * i + kk1 + 256u * k2) * 4u, 0, 2);
    }
  }
  if constexpr (HITS) {
    // K5 (process.cpp:46-62): mask of this thread's bins, candidates in linear power, the decision on the dB value
    uint32_t keepmask = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const uint32_t jj = k1 + 256u * (lo + 16u * q);
      const uint32_t ii = jj ^ (BN / 2);  // (j + N/2) % N, process.cpp:47
      const bool keep = !(jj < args.dc_ignore || (BN - jj) < args.dc_ignore) && !(ii < args.i_lo || ii > args.i_hi);
      keepmask |= keep ? (1u << q) : 0u;
    }
    const float pmax = fmaxf(fmaxf(gmax[0], gmax[1]), fmaxf(gmax[2], gmax[3]));
    if (__ballot(pmax > args.p_lo))
      scn_record_hits_device_counter<16>(pw, gmax, keepmask, args, reinterpret_cast<int *>(args.per_buffer_hits + b), b, lane,
                          [&](int q) -> uint32_t { return (k1 + 256u * (lo + 16u * (uint32_t)q)) ^ (BN / 2); });
  }
}

// ---- rows, 32768 points: grid = 8 pairs of row tiles x n_buffers; workgroup (i, b) finishes the 32 rows k1 in [32 i, 32 i + 32) ----
// 128-point rows as 16 x 8: n2 = 8 a + b, k2 = p + 16 q.  Pass 1: thread (rho, b) = (t >> 3, t & 7): 16-pt DFT over a, * W_128^(b p);
// pass 2: thread (rho, pl): 8-pt DFTs over b for p = pl and pl + 8.  In double like the 65536-point rows (float exchange).
template <bool HITS, bool SPEC>
__global__ __launch_bounds__(256, 3) void scn_big_rows32k_kernel(ScnBigArgs args) {
  // exchange: [rho][p][b] at rho*RP + 9 p + b, RP = 152 = 8 mod 16: a ds_write_b64 is served in groups of 16 lanes = (two
  // rho) x (8 b), whose slots rho RP + b then cover 16 distinct bank pairs; the pass-2 reads' 32 lanes = (four rho) x (8 p) sit
  // at slots 24 rho + 9 p mod 32, all distinct.  (Round 3: RP = 162: 2-way on the writes and on the reads.)
  constexpr uint32_t N = 32768u, RP = 152u;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v2f *lds = reinterpret_cast<v2f *>(smem_raw);
  // the dB tile on its way out: [k2 (128)][32] floats, slot(k2, k1) = 32 k2 + (k1 ^ ((k2 & 7) << 2)): written by 32-lane
  // groups of (four k1) x (8 k2 = pl + ...) -- k1's upper bits XOR the 8 values of pl: 32 distinct banks --, read a whole k2
  // row per group.  (Round 3: pitch 33 -> bank = pl + rho, up to 4-way: 41 % of the kernel's LDS cycles were conflict cycles.)
  float *lds_out = reinterpret_cast<float *>(smem_raw);
  const uint32_t t = threadIdx.x, lane = t & 63u;
  const uint32_t i = blockIdx.x & 7u, b = blockIdx.x >> 3;
  const uint32_t rho = t >> 3, bq = t & 7u;
  const uint32_t k1 = 32u * i + rho;

  // work buffer: [row tile k1 / 16 (16)][column tile n2 / 16 (8)][16 k1][16 n2]: this thread's Y[k1][8 a + bq], a = 0 .. 15
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(reinterpret_cast<const char *>(args.work) + (size_t)b * N * 8u, N * 8u);
  const uint32_t ld_voff = ((k1 >> 4) * 8u * 256u + (k1 & 15u) * 16u + bq) * 8u;  // + (a / 2) * 2048 + (a & 1) * 64
  v2f raw[16];
#pragma unroll
  // (default cache policy, not non-temporal: the loads for a = 2 j and 2 j + 1 read the two 64-byte halves of the same 128-byte
  // lines, and with the nt hint the second one fetched a third of them from HBM again: FETCH_SIZE 3.6e8 B per launch for 2.7e8)
  for (int a = 0; a < 16; a++) raw[a] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rw, ld_voff, (a >> 1) * 2048u + (a & 1) * 64u, 0));
  scn_v2d twa[16];  // W_128^(b p): entry 2 b p of the plan's W_256 table (double)
#pragma unroll
  for (int p = 1; p < 16; p++) twa[p] = args.tw256[(2u * bq * p) & 255u];
  cd v[16];
#pragma unroll
  for (int a = 0; a < 16; a++) v[a] = to_cd(raw[a]);
  fft16_d(v);
  v2f *w1 = lds + rho * RP + bq;  // + 9 p
#pragma unroll
  for (int p = 0; p < 16; p++) {
    cd y = v[OUT16(p)];
    if (p) y = cmul_d(y, twa[p]);
    w1[9 * p] = v2f{(float)y.x, (float)y.y};
  }
  __syncthreads();
  // pass 2: 8-point DFTs over b (radix 4 x 2 in double) for p = pl, pl + 8
  const uint32_t pl = bq;
  cd z[2][8];
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int bb = 0; bb < 8; bb++) z[h][bb] = to_cd(lds[rho * RP + 9u * (pl + 8u * h) + bb]);
  __syncthreads();  // every exchange read done: the area becomes the output tile
  const double H = 0.70710678118654752440;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    cd *zz = z[h];
    radix4_d(zz[0], zz[2], zz[4], zz[6]);  // even samples -> zz[2 k0]
    radix4_d(zz[1], zz[3], zz[5], zz[7]);  // odd samples  -> zz[2 k0 + 1]
    zz[3] = mul_w2_d(zz[3], H);            // W8^1
    zz[5] = cd{zz[5].y, -zz[5].x};         // W8^2 = -i
    zz[7] = mul_w6_d(zz[7], H);            // W8^3
#pragma unroll
    for (int k0 = 0; k0 < 4; k0++) {
      const cd a = zz[2 * k0], c = zz[2 * k0 + 1];
      zz[2 * k0] = a + c;      // X[k0]
      zz[2 * k0 + 1] = a - c;  // X[k0 + 4]
    }
  }
  // this thread's 16 bins: output o = 8 h + q holds k = k1 + 256 k2, k2 = pl + 8 h + 16 q; X[q] sits in z[h][OUT8(q)]
  auto k2_of = [&](int o) -> uint32_t { return pl + 8u * ((uint32_t)o >> 3) + 16u * ((uint32_t)o & 7u); };
  v16f pw;
  float gmax[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int o = 0; o < 16; o++) {
    const cd xq = z[o >> 3][OUT8(o & 7)];
    const float p = (float)__builtin_fma(xq.y, xq.y, xq.x * xq.x);
    pw[o] = p;
    gmax[o >> 2] = fmaxf(gmax[o >> 2], p);
    if constexpr (SPEC) lds_out[k2_of(o) * 32u + (rho ^ (pl << 2))] = db_of_power(p);
  }
  if constexpr (SPEC) {
    __syncthreads();
    // 128-byte runs: lanes 0..31 take the 32 consecutive k1 of one k2, two k2 per wave instruction
    __amdgpu_buffer_rsrc_t rout = make_rsrc(args.power_db + (size_t)b * N, args.power_db ? 4u * N : 0u);
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const uint32_t k2 = (t >> 5) + 8u * u, kk1 = t & 31u;
      const float d = lds_out[k2 * 32u + (kk1 ^ ((t >> 5) << 2))];
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d), rout, (32u * i + kk1 + 256u * k2) * 4u, 0, 2);
    }
  }
  if constexpr (HITS) {
    uint32_t keepmask = 0;
#pragma unroll
    for (int o = 0; o < 16; o++) {
      const uint32_t jj = k1 + 256u * k2_of(o);
      const uint32_t ii = jj ^ (N / 2);  // (j + N/2) % N, process.cpp:47
      const bool keep = !(jj < args.dc_ignore || (N - jj) < args.dc_ignore) && !(ii < args.i_lo || ii > args.i_hi);
      keepmask |= keep ? (1u << o) : 0u;
    }
    const float pmax = fmaxf(fmaxf(gmax[0], gmax[1]), fmaxf(gmax[2], gmax[3]));
    if (__ballot(pmax > args.p_lo))
      scn_record_hits_device_counter<16>(pw, gmax, keepmask, args, reinterpret_cast<int *>(args.per_buffer_hits + b), b, lane,
                                       [&](int o) -> uint32_t { return (k1 + 256u * k2_of(o)) ^ (N / 2); });
  }
}

bool scn_big_size_supported(uint32_t n) { return n == BN || n == 32768u; }

template <uint32_t N>
static hipError_t launch_cols(int kind, bool dc, const ScnBigArgs &a, uint32_t grid, size_t lds, hipStream_t s) {
  if (dc && kind != SCN_K_FLOAT_COMPLEX) {
    hipError_t e = hipMemsetAsync(a.dc_sums, 0, sizeof(int) * 2 * a.n_buffers, s);
    if (e != hipSuccess) return e;
    const dim3 gd(4u * a.n_buffers);
    switch (kind) {
      case SCN_K_SHORT_COMPLEX:
        hipLaunchKernelGGL((scn_big_dc_kernel<SCN_K_SHORT_COMPLEX, N>), gd, dim3(256), 0, s, a);
        hipLaunchKernelGGL((scn_big_cols_kernel<SCN_K_SHORT_COMPLEX, N, true>), dim3(grid), dim3(256), lds, s, a);
        break;
      case SCN_K_SHORT:
        hipLaunchKernelGGL((scn_big_dc_kernel<SCN_K_SHORT, N>), gd, dim3(256), 0, s, a);
        hipLaunchKernelGGL((scn_big_cols_kernel<SCN_K_SHORT, N, true>), dim3(grid), dim3(256), lds, s, a);
        break;
      case SCN_K_BYTE_COMPLEX:
        hipLaunchKernelGGL((scn_big_dc_kernel<SCN_K_BYTE_COMPLEX, N>), gd, dim3(256), 0, s, a);
        hipLaunchKernelGGL((scn_big_cols_kernel<SCN_K_BYTE_COMPLEX, N, true>), dim3(grid), dim3(256), lds, s, a);
        break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  switch (kind) {
    case SCN_K_FLOAT_COMPLEX: hipLaunchKernelGGL((scn_big_cols_kernel<SCN_K_FLOAT_COMPLEX, N, false>), dim3(grid), dim3(256), lds, s, a); break;
    case SCN_K_SHORT_COMPLEX: hipLaunchKernelGGL((scn_big_cols_kernel<SCN_K_SHORT_COMPLEX, N, false>), dim3(grid), dim3(256), lds, s, a); break;
    case SCN_K_SHORT: hipLaunchKernelGGL((scn_big_cols_kernel<SCN_K_SHORT, N, false>), dim3(grid), dim3(256), lds, s, a); break;
    case SCN_K_BYTE_COMPLEX: hipLaunchKernelGGL((scn_big_cols_kernel<SCN_K_BYTE_COMPLEX, N, false>), dim3(grid), dim3(256), lds, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t scn_launch_big(uint32_t n, int kind, bool dc, bool hits, bool spec, const ScnBigArgs &a, int num_cus, hipStream_t s) {
  if (a.n_buffers == 0) return hipSuccess;
  if ((!hits && !spec) || !scn_big_size_supported(n)) return hipErrorInvalidValue;
  if (dc && kind != SCN_K_FLOAT_COMPLEX && !a.dc_sums) return hipErrorInvalidValue;
  hipError_t e = hipSuccess;
  // (the per-buffer counters are zeroed by the column kernel's tile-0 workgroups: a memset of its own was a third dependent
  //  launch, ~5 us of a 200 us step)
  const uint32_t ct = n / 4096u;  // column tiles
  const size_t lds = 16 * BP * sizeof(v2f);
  uint32_t G = (uint32_t)(num_cus * 3) / ct;
  if (G < 1) G = 1;
  if (G > a.n_buffers) G = a.n_buffers;
  e = n == BN ? launch_cols<BN>(kind, dc, a, ct * G, lds, s) : launch_cols<32768u>(kind, dc, a, ct * G, lds, s);
  if (e != hipSuccess) return e;
  if (n == BN) {
    const dim3 grid(16u * a.n_buffers);
    if (hits && spec) hipLaunchKernelGGL((scn_big_rows_kernel<true, true>), grid, dim3(256), lds, s, a);
    else if (hits) hipLaunchKernelGGL((scn_big_rows_kernel<true, false>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((scn_big_rows_kernel<false, true>), grid, dim3(256), lds, s, a);
  } else {
    const dim3 grid(8u * a.n_buffers);
    const size_t lds32 = 32 * 152 * sizeof(v2f);  // 38 KiB: the exchange (the 128 x 32 float output tile fits inside)
    if (hits && spec) hipLaunchKernelGGL((scn_big_rows32k_kernel<true, true>), grid, dim3(256), lds32, s, a);
    else if (hits) hipLaunchKernelGGL((scn_big_rows32k_kernel<true, false>), grid, dim3(256), lds32, s, a);
    else hipLaunchKernelGGL((scn_big_rows32k_kernel<false, true>), grid, dim3(256), lds32, s, a);
  }
  return hipGetLastError();
}
