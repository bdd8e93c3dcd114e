// scn_mixed.hip -- fused kernels for the sizes N = 2^a 3^b 5^c that are not powers of two (1000, 3000, 5000, 6000, 10000, 12000
// ...: scn_mixed_plans.h; two forms, up to 10000 points and beyond).
//
// The reference hands any --count to FFTW (fft.cpp:4-11), which runs these sizes with radix-3 / radix-5 codelets at the speed
// of their power-of-two neighbours.  Until round 5 they went through Bluestein's algorithm on the staged path of
// scn_generic.hip -- three double-precision transforms of the next power of two through HBM, 0.013 of the roofline at 1000
// points, 50 times slower than the 1024-point kernel.  Here they get the single pass of the power-of-two sizes: one launch
// does K1 .. K5 with the transform staged in LDS, HBM traffic = raw sample in + one float out.
//
//   N = R1 R2 R3        n = T1 a + R3 b + c  (T1 = R2 R3)        k = p + R1 q + R1 R2 r
//   pass 1  thread tau = R3 b + c (T1 of them):  DFT_R1 over a of x[T1 a + tau] w[..] -> * W_N^(tau p)      -> L1(p, tau) = P1 p + tau
//   pass 2  thread (p, c) (R1 R3 of them):       DFT_R2 over b                        -> * W_(R2 R3)^(c q) -> L2(c, kl) = P2 c + kl, kl = p + R1 q
//   pass 3  thread kl (R1 R2 of them):           DFT_R3 over c                        -> X[kl + R1 R2 r] -> dB, mask, threshold
// (index algebra and LDS pitches: scripts/mixed_plan.py, which emulates the three passes against numpy.fft and picks, per
// size, the most balanced radices and the pitches with the fewest bank conflicts; the in-register DFTs of any 5-smooth length:
// scn_mixed_dft.h, held to the DFT sum on the CPU by tests/cpp/test_mixed_dft.cpp).  R1 is the smallest radix, so pass 1 has
// the most threads: the workgroup is T1 threads rounded up to whole waves, every thread is ONE pass-1 thread whose R1 - 1
// twiddles and R1 window taps stay in registers for the whole launch, and passes 2 and 3 run on the leading R1 R3 / R1 R2
// threads.  One workgroup per buffer, persistent over buffers, the next buffer's samples prefetched into registers across the
// passes -- the structure of scn_fft_kernel (scn_kernels.hip), whose K1 loaders, dB map and hit path these kernels share
// (scn_device.h).  Coalescing: a wave's loads are consecutive tau, its stores consecutive kl.
// Output modes as everywhere (HITS, SPEC template parameters); integer-mean DC removal (utility.cpp:70-79) is a runtime flag
// here (one uniform branch per buffer): 12 kernels per size instead of 21.
#include <hip/hip_runtime.h>

#include <atomic>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "scn_device.h"
#include "scn_kernels.h"
#include "scn_mixed_dft.h"
#include "scn_mixed_plans.h"

namespace {

template <uint32_t N_, uint32_t R1_, uint32_t R2_, uint32_t R3_, uint32_t PAD1, uint32_t PAD2>
struct GeoMixed {
  static constexpr uint32_t N = N_, R1 = R1_, R2 = R2_, R3 = R3_;
  static_assert(R1 * R2 * R3 == N && R1 <= R2 && R1 <= R3 && R2 <= 32 && R3 <= 32, "three radices, the smallest first");
  static constexpr uint32_t T1 = R2 * R3, V2 = R1 * R3, V3 = R1 * R2;   // threads of the three passes
  static constexpr uint32_t W = (T1 + 63u) / 64u * 64u;                // workgroup: T1 rounded up to whole waves
  // At most 512 threads, i.e. two waves per SIMD and 256 registers per thread: R1 - 1 twiddles + R1 window taps + R1 prefetched
  // samples + a 25-point DFT in flight need ~240 at 10000 points.  Every 5-smooth size above ~10600 has a pass with more than
  // 512 threads (three radices <= 32 with N / R <= 512 for each need N <= 22^3); built with 640 .. 832 threads (168 / 128
  // registers) the 12000 .. 16000-point kernels spilled 40 .. 170 VGPRs, with two virtual threads per thread 80 .. 250
  // (profiles/r05_experiments.md): those sizes have a form of their own, GeoMixedBig below.
  static_assert(W <= 512, "one workgroup of at most two waves per SIMD per buffer");
  static constexpr uint32_t P1 = T1 + PAD1, P2 = V3 + PAD2;
  static constexpr uint32_t EXCH = (R1 * P1 > R3 * P2) ? R1 * P1 : R3 * P2;  // 8-byte slots
  static constexpr uint32_t LDS_BYTES = EXCH * 8u + T1 * 8u + 32u * 4u + 2u * 4u + 8u;
  static constexpr int NB = R3 <= 16 ? 16 : 32;  // outputs per thread as the hit path sees them (a power of two; the rest are zero)
};

template <int NB>
struct PowVec;
template <>
struct PowVec<16> {
  typedef v16f type;
};
template <>
struct PowVec<32> {
  typedef float type __attribute__((ext_vector_type(32)));
};

}  // namespace

template <class G, int KIND, bool HITS, bool SPEC>
__global__ __launch_bounds__(G::W) void scn_fft_mixed_kernel(ScnFftArgs args, uint32_t correct_dc) {
  static_assert(HITS || SPEC, "a kernel that reports nothing");
  constexpr int AUX_LD = SCN_AUX_LD;
  constexpr int AUX_ST = SCN_AUX_ST;
  constexpr uint32_t N = G::N, R1 = G::R1, R2 = G::R2, R3 = G::R3, T1 = G::T1, V2 = G::V2, V3 = G::V3, P1 = G::P1, P2 = G::P2;
  constexpr int NB = G::NB;
  typedef RawLoader<KIND> L;
  typedef typename PowVec<NB>::type pow_t;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v2f *lds = reinterpret_cast<v2f *>(smem_raw);
  v2f *lds_tw2 = lds + G::EXCH;                           // [R2][R3]: W_(R2 R3)^(c q) at q R3 + c
  int *lds_cnt = reinterpret_cast<int *>(lds_tw2 + T1);  // [32] DC-sum scratch (re[16], im[16])
  int *lds_hits = lds_cnt + 32;                          // [2] hit counters, alternating per buffer

  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  const bool a1 = t < T1, a2 = t < V2, a3 = t < V3;     // this thread takes part in pass 1 / 2 / 3
  const uint32_t p2 = t / R3, c2 = t % R3;               // pass-2 identity (p, c); also the (q, c) of the table entry it fills
  const uint32_t t_ld = a1 ? t : 0x10000000u;            // lanes past T1 read beyond the descriptor's range: zeros, no memory access

  typename L::raw_t raw[R1];
  if (blockIdx.x < args.n_buffers) {
    const __amdgpu_buffer_rsrc_t r0 = make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)blockIdx.x * L::kBufBytes(N), L::kBufBytes(N));
#pragma unroll
    for (uint32_t a = 0; a < R1; a++) raw[a] = L::template load<AUX_LD>(r0, N, t_ld, T1 * a);
  }
  // persistent per-thread constants: W_N^(tau p) (table rows of T1, scn_mixed_layout) and the window taps with the ENOB scale folded in
  cf tw1[R1];
  float win[R1];
#pragma unroll
  for (uint32_t p = 1; p < R1; p++) tw1[p] = from_v2f(args.tw1_table[(p - 1) * T1 + (a1 ? t : 0u)]);
#pragma unroll
  for (uint32_t a = 0; a < R1; a++) win[a] = args.window[T1 * a + (a1 ? t : 0u)] * args.scale;
  if (a1) lds_tw2[t] = args.twiddle[(R1 * p2 * c2) % N];  // entry q R3 + c = t: W_(R2 R3)^(c q) = W_N^(R1 c q)
  if (t == 0) lds_hits[0] = lds_hits[1] = 0;
  __syncthreads();

  const uint32_t st_voff = a3 ? t * 4u : 0x80000000u;  // output r of this thread is bin j = t + V3 r
  constexpr uint32_t HALF = N / 2u;                    // process.cpp:47 j = (i + N/2) % N  <=>  i = (j - N/2) mod N
  auto bin_i = [&](int r) -> uint32_t {
    const uint32_t j = t + V3 * (uint32_t)r;
    return j >= HALF ? j - HALF : j + (N - HALF);
  };
  uint32_t keepmask = 0;  // K5 mask of this thread's R3 bins (process.cpp:46-52)
  if (HITS && a3) {
#pragma unroll
    for (uint32_t r = 0; r < R3; r++) {
      const uint32_t j = t + V3 * r, i = bin_i((int)r);
      const bool keep = !(j < args.dc_ignore || (N - j) < args.dc_ignore) && !(i < args.i_lo || i > args.i_hi);
      keepmask |= keep ? (1u << r) : 0u;
    }
  }
  uint32_t par = 0;
  uint32_t prev = 0xffffffffu;  // the buffer whose recorders may still be running

  for (uint32_t buf = blockIdx.x; buf < args.n_buffers; buf += gridDim.x) {
    const uint32_t nxt = buf + gridDim.x;
    const bool more = nxt < args.n_buffers;
    // ---- K1 + K2 ----
    int dc_re = 0, dc_im = 0;
    if (KIND != SCN_K_FLOAT_COMPLEX && correct_dc) {  // integer mean with the reference's int32 /= uint32 quirk (utility.cpp:77-78)
      int sr = 0, si = 0;
#pragma unroll
      for (uint32_t a = 0; a < R1; a++) {  // (lanes past T1 hold zeros)
        int re, im;
        L::ints(raw[a], re, im);
        sr += re;
        si += im;
      }
      sr = wave_sum(sr);
      si = wave_sum(si);
      if (lane == 0) {
        lds_cnt[wave] = sr;
        lds_cnt[16 + wave] = si;
      }
      __syncthreads();
      sr = si = 0;
#pragma unroll
      for (uint32_t w = 0; w < G::W / 64u; w++) {
        sr += lds_cnt[w];
        si += lds_cnt[16 + w];
      }
      dc_re = (int)((uint32_t)sr / N);
      dc_im = (int)((uint32_t)si / N);
    }
    cf v[R1];
#pragma unroll
    for (uint32_t a = 0; a < R1; a++) v[a] = L::conv(raw[a], dc_re, dc_im, 1.0f) * win[a];
    // the next buffer of this workgroup, branch-free (zero records past the end), in three groups spread over the passes
    const __amdgpu_buffer_rsrc_t rn =
        make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)(more ? nxt : buf) * L::kBufBytes(N), more ? L::kBufBytes(N) : 0u);
    auto prefetch = [&](uint32_t a_lo, uint32_t a_hi) {
#pragma unroll
      for (uint32_t a = 0; a < R1; a++)
        if (a >= a_lo && a < a_hi) raw[a] = L::template load<AUX_LD>(rn, N, t_ld, T1 * a);
    };
    constexpr uint32_t PF0 = (3u * R1) / 8u, PF1 = (11u * R1) / 16u;
    prefetch(0, PF0);

    // ---- pass 1: DFT over a, twiddle W_N^(tau p), scatter to row p ----
    scn_dft<(int)R1>(v);
    if (a1) {
#pragma unroll
      for (uint32_t p = 0; p < R1; p++) lds[p * P1 + t] = to_v2f(p ? cmul(v[p], tw1[p]) : v[p]);
    }
    __syncthreads();  // barrier 1
    if (HITS) {
      if (t == 0 && prev != 0xffffffffu) {  // every wave is past the barrier: the previous buffer's recorders are done
        args.per_buffer_hits[prev] = (uint32_t)lds_hits[par ^ 1];
        if (args.host_hits) args.host_hits[prev] = (uint32_t)lds_hits[par ^ 1];
        lds_hits[par ^ 1] = 0;
      }
    }
    prefetch(PF0, PF1);

    // ---- pass 2: thread (p, c): DFT over b, twiddle W_(R2 R3)^(c q) ----
    cf u[R2];
    if (a2) {
#pragma unroll
      for (uint32_t b = 0; b < R2; b++) u[b] = from_v2f(lds[p2 * P1 + R3 * b + c2]);
      scn_dft<(int)R2>(u);
#pragma unroll
      for (uint32_t q = 1; q < R2; q++) u[q] = cmul(u[q], from_v2f(lds_tw2[q * R3 + c2]));
    }
    __syncthreads();  // barrier 2: every exchange-1 read done before the area is re-used
    prefetch(PF1, R1);
    if (a2) {
#pragma unroll
      for (uint32_t q = 0; q < R2; q++) lds[c2 * P2 + p2 + R1 * q] = to_v2f(u[q]);
    }
    __syncthreads();  // barrier 3

    // ---- pass 3: DFT over c; K4: the thread's linear powers stay in `pw` for the hit path, the spectrum gets the dB map of
    //      scn_device.h -- product form inline, exact form stored over it for the strong bins in the waves that hold one ----
    pow_t pw;
    float gmax[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // the largest power of each group of NB / 4 outputs
#pragma unroll
    for (int o = 0; o < NB; o++) pw[o] = 0.0f;
    __amdgpu_buffer_rsrc_t rout = make_rsrc(args.power_db + (size_t)buf * N, (SPEC && args.power_db) ? 4u * N : 0u);
    if (a3) {
      cf z[R3];
#pragma unroll
      for (uint32_t c = 0; c < R3; c++) z[c] = from_v2f(lds[c * P2 + t]);
      scn_dft<(int)R3>(z);
#pragma unroll
      for (uint32_t r = 0; r < R3; r++) {
        const float q = power_of(z[r]);
        pw[r] = q;
        gmax[r / (NB / 4)] = fmaxf(gmax[r / (NB / 4)], q);
        if constexpr (SPEC) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, db_fast(q)), rout, st_voff, 4u * V3 * r, AUX_ST);
      }
    }
    const float pmax = fmaxf(fmaxf(gmax[0], gmax[1]), fmaxf(gmax[2], gmax[3]));
    if constexpr (SPEC) {
      if (__ballot(pmax >= SCN_P_EXACT_FROM)) {
#pragma unroll
        for (int g = 0; g < 4; g++) {  // the groups first, then the outputs of a group that holds a strong bin (see scn_fft_kernel)
          if (__ballot(gmax[g] >= SCN_P_EXACT_FROM)) {
#pragma unroll
            for (int r = g * (NB / 4); r < (g + 1) * (NB / 4); r++) {
              if (r < (int)R3) {
                const float q = pw[r];
                if (__ballot(q >= SCN_P_EXACT_FROM)) {
                  const float d = db_exact(q);
                  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d), rout, q >= SCN_P_EXACT_FROM ? st_voff : 0x80000000u, 4u * V3 * (uint32_t)r, AUX_ST);
                }
              }
            }
          }
        }
      }
    }
    __syncthreads();  // barrier 4: exchange area free again
    if (HITS) {
      // after the barrier: a wave that holds detections does not hold up the others (see scn_fft_kernel)
      if (__ballot(pmax > args.p_lo))
        scn_record_hits_lanes<NB, false, true>(pw, gmax, keepmask, args, &lds_hits[par], buf, lane, bin_i);
      prev = buf;
      par ^= 1;
    }
  }
  if (HITS) {
    __syncthreads();  // last buffer's recorders done
    if (t == 0 && prev != 0xffffffffu) {
      args.per_buffer_hits[prev] = (uint32_t)lds_hits[par ^ 1];
      if (args.host_hits) args.host_hits[prev] = (uint32_t)lds_hits[par ^ 1];
    }
  }
}

// ------------------------------------------------------------------------------------
// The sizes beyond 10000 (12000, 12288, 14400, 15000, 16000): scn_fft_mixed_big_kernel.
//
// Every factorisation of these sizes into three radices <= 32 has a pass with more than 512 threads (see GeoMixed), so a thread
// plays TWO virtual threads (tv = t and t + W) of every pass in a workgroup of <= 448 threads -- two waves per SIMD, 256
// registers.  Three things keep that inside the register file where the straightforward form spilled 80 .. 250 VGPRs:
//   * the two virtual threads run ONE AFTER THE OTHER in a loop that is not unrolled, so hipcc cannot interleave the two 24- /
//     25-point DFTs and keep both live;
//   * nothing of a virtual thread survives a barrier in registers: pass 2 works IN PLACE -- thread (p, c) reads L1(p, R3 b + c),
//     b < R2, and writes its outputs back to the very slots it read, L1(p, R3 q + c); no other thread touches them in that pass,
//     so there is no barrier between its reads and its writes -- and pass 3 then reads L1(p, R3 q + c), c < R3, as thread
//     kl = p + R1 q (scripts/mixed_plan.py emulates exactly this against numpy.fft).  One exchange area, three barriers per
//     buffer; hits are recorded inside pass 3's loop;
//   * no register-resident constants and no prefetch: pass-1 twiddles and window taps are read per buffer from their
//     (L2-resident) tables, the samples when they are needed.
// PASS 3 RUNS IN DOUBLE, and the smallest radix goes last so that it fits (a 16 .. 24-point DFT in double: 64 .. 96 registers of data):
// all in float, these sizes put the parity metric's tail ON the bar -- a strong tone's partial sums are rounded in the last pass and
// the error lands on the other bins of its column, as at 16384 points (scn_kernels.hip) -- : 1024 strong-tone buffers per format
// read up to 9.8e-6 (16000 points) and 9.1e-6 (14400), a ten-minute fuzz on these sizes found two spectra over 1e-5 (1.3e-5 at 12288
// points, 1.1e-5 at 16000); the sizes up to 10000 stay within 6e-6 (profiles/r05_experiments.md section 4).
// One workgroup per CU (94 .. 125 KiB of LDS): these kernels trade speed for fitting at all -- against Bluestein's three
// double-precision transforms through HBM.
// ------------------------------------------------------------------------------------
namespace {
template <uint32_t N_, uint32_t R1_, uint32_t R2_, uint32_t R3_, uint32_t PAD1>
struct GeoMixedBig {
  static constexpr uint32_t N = N_, R1 = R1_, R2 = R2_, R3 = R3_;
  static_assert(R1 * R2 * R3 == N && R3 <= R1 && R3 <= R2 && R1 <= 32 && R2 <= 32, "three radices, the smallest LAST (pass 3 runs in double)");
  static constexpr uint32_t T1 = R2 * R3, V2 = R1 * R3, V3 = R1 * R2;
  static constexpr uint32_t VMAX = T1 > V2 ? (T1 > V3 ? T1 : V3) : (V2 > V3 ? V2 : V3);
  static constexpr uint32_t W = ((VMAX + 1u) / 2u + 63u) / 64u * 64u;  // two virtual threads per thread
  static_assert(VMAX > 512 && W <= 512, "the sizes one virtual thread per thread cannot carry");
  static constexpr uint32_t P1 = T1 + PAD1;
  static constexpr uint32_t EXCH = R1 * P1;
  static constexpr uint32_t LDS_BYTES = EXCH * 8u + T1 * 8u + 32u * 4u + 2u * 4u + 8u;
  static constexpr int NB = R3 <= 16 ? 16 : 32;
};
}  // namespace

template <class G, int KIND, bool HITS, bool SPEC>
__global__ __launch_bounds__(G::W) void scn_fft_mixed_big_kernel(ScnFftArgs args, uint32_t correct_dc) {
  static_assert(HITS || SPEC, "a kernel that reports nothing");
  constexpr int AUX_LD = SCN_AUX_LD;
  constexpr int AUX_ST = SCN_AUX_ST;
  constexpr uint32_t N = G::N, R1 = G::R1, R2 = G::R2, R3 = G::R3, T1 = G::T1, V2 = G::V2, V3 = G::V3, P1 = G::P1, W = G::W;
  constexpr int NB = G::NB;
  constexpr uint32_t HALF = N / 2u;
  typedef RawLoader<KIND> L;
  typedef typename PowVec<NB>::type pow_t;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v2f *lds = reinterpret_cast<v2f *>(smem_raw);
  v2f *lds_tw2 = lds + G::EXCH;                           // [R2][R3]: W_(R2 R3)^(c q) at q R3 + c
  int *lds_cnt = reinterpret_cast<int *>(lds_tw2 + T1);  // [32] DC-sum scratch
  int *lds_hits = lds_cnt + 32;                          // [2]
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;

#pragma nounroll
  for (uint32_t h = 0; h < 2; h++) {  // entry q R3 + c = tv: W_(R2 R3)^(c q) = W_N^(R1 c q)
    const uint32_t tv = t + h * W;
    if (tv < T1) lds_tw2[tv] = args.twiddle[(R1 * (tv / R3) * (tv % R3)) % N];
  }
  if (t == 0) lds_hits[0] = lds_hits[1] = 0;
  __syncthreads();
  uint32_t par = 0, prev = 0xffffffffu;

  for (uint32_t buf = blockIdx.x; buf < args.n_buffers; buf += gridDim.x) {
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)buf * L::kBufBytes(N), L::kBufBytes(N));
    int dc_re = 0, dc_im = 0;
    if (KIND != SCN_K_FLOAT_COMPLEX && correct_dc) {  // a pass of its own over the raw samples: the integer sums (utility.cpp:70-79)
      int sr = 0, si = 0;
#pragma nounroll
      for (uint32_t h = 0; h < 2; h++) {
        const uint32_t tv = t + h * W, t_ld = tv < T1 ? tv : 0x10000000u;
#pragma unroll
        for (uint32_t a = 0; a < R1; a++) {
          int re, im;
          L::ints(L::template load<AUX_LD>(rc, N, t_ld, T1 * a), re, im);
          sr += re;
          si += im;
        }
      }
      sr = wave_sum(sr);
      si = wave_sum(si);
      if (lane == 0) {
        lds_cnt[wave] = sr;
        lds_cnt[16 + wave] = si;
      }
      __syncthreads();
      sr = si = 0;
#pragma unroll
      for (uint32_t w = 0; w < W / 64u; w++) {
        sr += lds_cnt[w];
        si += lds_cnt[16 + w];
      }
      dc_re = (int)((uint32_t)sr / N);  // int32 /= uint32, utility.cpp:77-78
      dc_im = (int)((uint32_t)si / N);
    }
    // ---- pass 1: DFT over a of x[T1 a + tv] w[..], twiddle W_N^(tv p), to row p ----
#pragma nounroll
    for (uint32_t h = 0; h < 2; h++) {
      const uint32_t tv = t + h * W;
      const bool on = tv < T1;
      const uint32_t t_ld = on ? tv : 0x10000000u, tt = on ? tv : 0u;
      cf v[R1];
#pragma unroll
      for (uint32_t a = 0; a < R1; a++) v[a] = L::conv(L::template load<AUX_LD>(rc, N, t_ld, T1 * a), dc_re, dc_im, 1.0f) * (args.window[T1 * a + tt] * args.scale);
      scn_dft<(int)R1>(v);
      if (on) {
#pragma unroll
        for (uint32_t p = 0; p < R1; p++) lds[p * P1 + tv] = to_v2f(p ? cmul(v[p], from_v2f(args.tw1_table[(p - 1) * T1 + tv])) : v[p]);
      }
    }
    __syncthreads();  // barrier 1
    if (HITS) {
      if (t == 0 && prev != 0xffffffffu) {  // every wave is past the barrier: the previous buffer's recorders are done
        args.per_buffer_hits[prev] = (uint32_t)lds_hits[par ^ 1];
        if (args.host_hits) args.host_hits[prev] = (uint32_t)lds_hits[par ^ 1];
        lds_hits[par ^ 1] = 0;
      }
    }
    // ---- pass 2, in place: thread (p, c): DFT over b, twiddle W_(R2 R3)^(c q), back to the slots it read ----
#pragma nounroll
    for (uint32_t h = 0; h < 2; h++) {
      const uint32_t tv = t + h * W;
      if (tv < V2) {
        const uint32_t p2 = tv / R3, c2 = tv % R3;
        v2f *row = lds + p2 * P1 + c2;
        cf u[R2];
#pragma unroll
        for (uint32_t b = 0; b < R2; b++) u[b] = from_v2f(row[R3 * b]);
        scn_dft<(int)R2>(u);
#pragma unroll
        for (uint32_t q = 0; q < R2; q++) row[R3 * q] = to_v2f(q ? cmul(u[q], from_v2f(lds_tw2[q * R3 + c2])) : u[q]);
      }
    }
    __syncthreads();  // barrier 2
    // ---- pass 3: thread kl = p + R1 q: DFT over c of L1(p, R3 q + c) -> X[kl + R1 R2 r]; K4, K5 ----
    __amdgpu_buffer_rsrc_t rout = make_rsrc(args.power_db + (size_t)buf * N, (SPEC && args.power_db) ? 4u * N : 0u);
#pragma nounroll
    for (uint32_t h = 0; h < 2; h++) {
      const uint32_t tv = t + h * W;
      const bool on = tv < V3;
      const uint32_t st_voff = on ? tv * 4u : 0x80000000u;
      pow_t pw;
      float gmax[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int o = 0; o < NB; o++) pw[o] = 0.0f;
      uint32_t keepmask = 0;
      if (on) {
        const v2f *col = lds + (tv % R1) * P1 + R3 * (tv / R1);
        cd z[R3];  // in double: see the header of this kernel
#pragma unroll
        for (uint32_t c = 0; c < R3; c++) z[c] = to_cd(col[c]);
        scn_dft<(int)R3>(z);
#pragma unroll
        for (uint32_t r = 0; r < R3; r++) {
          const float q = (float)__builtin_fma(z[r].y, z[r].y, z[r].x * z[r].x);
          pw[r] = q;
          gmax[r / (NB / 4)] = fmaxf(gmax[r / (NB / 4)], q);
          if constexpr (SPEC) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, db_fast(q)), rout, st_voff, 4u * V3 * r, AUX_ST);
          if constexpr (HITS) {  // K5 mask of bin j = tv + V3 r (process.cpp:46-52)
            const uint32_t j = tv + V3 * r, i = j >= HALF ? j - HALF : j + (N - HALF);
            const bool keep = !(j < args.dc_ignore || (N - j) < args.dc_ignore) && !(i < args.i_lo || i > args.i_hi);
            keepmask |= keep ? (1u << r) : 0u;
          }
        }
      }
      const float pmax = fmaxf(fmaxf(gmax[0], gmax[1]), fmaxf(gmax[2], gmax[3]));
      if constexpr (SPEC) {
        if (__ballot(pmax >= SCN_P_EXACT_FROM)) {
#pragma unroll
          for (int g = 0; g < 4; g++) {
            if (__ballot(gmax[g] >= SCN_P_EXACT_FROM)) {
#pragma unroll
              for (int r = g * (NB / 4); r < (g + 1) * (NB / 4); r++) {
                if (r < (int)R3) {
                  const float q = pw[r];
                  if (__ballot(q >= SCN_P_EXACT_FROM)) {
                    const float d = db_exact(q);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d), rout, q >= SCN_P_EXACT_FROM ? st_voff : 0x80000000u, 4u * V3 * (uint32_t)r, AUX_ST);
                  }
                }
              }
            }
          }
        }
      }
      if constexpr (HITS) {
        if (__ballot(pmax > args.p_lo))
          scn_record_hits_lanes<NB, false, true>(pw, gmax, keepmask, args, &lds_hits[par], buf, lane, [&](int r) -> uint32_t {
            const uint32_t j = tv + V3 * (uint32_t)r;
            return j >= HALF ? j - HALF : j + (N - HALF);
          });
      }
    }
    __syncthreads();  // barrier 3: exchange area free again, this buffer's recorders done
    if (HITS) {
      prev = buf;
      par ^= 1;
    }
  }
  if (HITS) {
    if (t == 0 && prev != 0xffffffffu) {
      args.per_buffer_hits[prev] = (uint32_t)lds_hits[par ^ 1];
      if (args.host_hits) args.host_hits[prev] = (uint32_t)lds_hits[par ^ 1];
    }
  }
}

// ------------------------------------------------------------------------------------
// host-side launchers.  scanner_amd/build.py compiles this file once per SCN_MIXED_TU value (0 .. 7), side by side, each
// translation unit instantiating the sizes whose row in scn_mixed_plans.h has that unit number; without SCN_MIXED_TU one unit
// holds everything.
// ------------------------------------------------------------------------------------
#ifndef SCN_MIXED_TU
#define SCN_MIXED_TU -1
#endif
#define SCN_MIXED_IN_TU(x) (SCN_MIXED_TU == -1 || SCN_MIXED_TU == (x))

namespace {
template <class G, int KIND, bool BIG>
hipError_t launch_mixed_kind(const ScnFftArgs &a, bool dc, bool hits, bool spec, int num_cus, hipStream_t s, hipEvent_t stop) {
  void (*k)(ScnFftArgs, uint32_t);
  if constexpr (BIG)
    k = !hits ? scn_fft_mixed_big_kernel<G, KIND, false, true> : spec ? scn_fft_mixed_big_kernel<G, KIND, true, true> : scn_fft_mixed_big_kernel<G, KIND, true, false>;
  else
    k = !hits ? scn_fft_mixed_kernel<G, KIND, false, true> : spec ? scn_fft_mixed_kernel<G, KIND, true, true> : scn_fft_mixed_kernel<G, KIND, true, false>;
  if (G::LDS_BYTES > 65536u) {  // per function and per device: simply set on every launch (see launch_kind, scn_kernels.hip)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (e != hipSuccess) return e;
  }
  // one resident wave of persistent workgroups: what fits a CU by registers, LDS and waves (asked of the runtime: the register
  // count of these kernels is the compiler's choice)
  // (asked once per kernel: one static per instantiation and output mode, atomic because several host threads -- one plan each --
  // may come through here at once; whoever asks first stores the answer, a second asker stores the same one: the figure is a property
  // of the kernel's code object and the GPU model, and the GPUs of a node are alike)
  static std::atomic<int> per_cu_of_mode[3];
  std::atomic<int> &cached = per_cu_of_mode[!hits ? 0 : spec ? 1 : 2];
  int per_cu = cached.load(std::memory_order_relaxed);
  if (per_cu < 1) {
    int q = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, k, (int)G::W, G::LDS_BYTES);
    if (e != hipSuccess) return e;
    per_cu = q < 1 ? 1 : q;
    cached.store(per_cu, std::memory_order_relaxed);
  }
  int grid = num_cus * per_cu;
  if ((uint32_t)grid > a.n_buffers) grid = (int)a.n_buffers;
  const uint32_t cdc = dc ? 1u : 0u;
  if (stop) hipExtLaunchKernelGGL(k, dim3(grid), dim3(G::W), G::LDS_BYTES, s, nullptr, stop, 0, a, cdc);
  else hipLaunchKernelGGL(k, dim3(grid), dim3(G::W), G::LDS_BYTES, s, a, cdc);
  return hipGetLastError();
}

template <class G, bool BIG = false>
hipError_t launch_mixed(int kind, bool dc, bool hits, bool spec, const ScnFftArgs &args, int num_cus, hipStream_t stream, hipEvent_t stop) {
  switch (kind) {
    case SCN_K_FLOAT_COMPLEX: return launch_mixed_kind<G, SCN_K_FLOAT_COMPLEX, BIG>(args, false, hits, spec, num_cus, stream, stop);
    case SCN_K_SHORT_COMPLEX: return launch_mixed_kind<G, SCN_K_SHORT_COMPLEX, BIG>(args, dc, hits, spec, num_cus, stream, stop);
    case SCN_K_SHORT: return launch_mixed_kind<G, SCN_K_SHORT, BIG>(args, dc, hits, spec, num_cus, stream, stop);
    case SCN_K_BYTE_COMPLEX: return launch_mixed_kind<G, SCN_K_BYTE_COMPLEX, BIG>(args, dc, hits, spec, num_cus, stream, stop);
    default: return hipErrorInvalidValue;
  }
}
}  // namespace

#define SCN_MIXED_ARGS int kind, bool dc, bool hits, bool spec, const ScnFftArgs &args, int num_cus, hipStream_t stream, hipEvent_t stop
hipError_t scn_launch_mixed_tu0(uint32_t n, SCN_MIXED_ARGS);
hipError_t scn_launch_mixed_tu1(uint32_t n, SCN_MIXED_ARGS);
hipError_t scn_launch_mixed_tu2(uint32_t n, SCN_MIXED_ARGS);
hipError_t scn_launch_mixed_tu3(uint32_t n, SCN_MIXED_ARGS);
hipError_t scn_launch_mixed_tu4(uint32_t n, SCN_MIXED_ARGS);
hipError_t scn_launch_mixed_tu5(uint32_t n, SCN_MIXED_ARGS);
hipError_t scn_launch_mixed_tu6(uint32_t n, SCN_MIXED_ARGS);
hipError_t scn_launch_mixed_tu7(uint32_t n, SCN_MIXED_ARGS);

// (`if constexpr` on the template parameter: a discarded branch instantiates nothing -- with a plain `if` every translation unit
//  would compile every size's kernels)
#define SCN_MIXED_CASE(N, R1, R2, R3, PAD1, PAD2, UNIT)  \
  if constexpr (UNIT == TU) {                             \
    if (n == N) return launch_mixed<GeoMixed<N, R1, R2, R3, PAD1, PAD2>>(kind, dc, hits, spec, args, num_cus, stream, stop); \
  }
#define SCN_MIXED_BIG_CASE(N, R1, R2, R3, PAD1, UNIT)     \
  if constexpr (UNIT == TU) {                             \
    if (n == N) return launch_mixed<GeoMixedBig<N, R1, R2, R3, PAD1>, true>(kind, dc, hits, spec, args, num_cus, stream, stop); \
  }
namespace {
template <int TU>
hipError_t launch_mixed_unit(uint32_t n, SCN_MIXED_ARGS) {
  SCN_MIXED_PLANS(SCN_MIXED_CASE)
  SCN_MIXED_BIG_PLANS(SCN_MIXED_BIG_CASE)
  return hipErrorInvalidValue;
}
}  // namespace
#if SCN_MIXED_IN_TU(0)
hipError_t scn_launch_mixed_tu0(uint32_t n, SCN_MIXED_ARGS) { return launch_mixed_unit<0>(n, kind, dc, hits, spec, args, num_cus, stream, stop); }
#endif
#if SCN_MIXED_IN_TU(1)
hipError_t scn_launch_mixed_tu1(uint32_t n, SCN_MIXED_ARGS) { return launch_mixed_unit<1>(n, kind, dc, hits, spec, args, num_cus, stream, stop); }
#endif
#if SCN_MIXED_IN_TU(2)
hipError_t scn_launch_mixed_tu2(uint32_t n, SCN_MIXED_ARGS) { return launch_mixed_unit<2>(n, kind, dc, hits, spec, args, num_cus, stream, stop); }
#endif
#if SCN_MIXED_IN_TU(3)
hipError_t scn_launch_mixed_tu3(uint32_t n, SCN_MIXED_ARGS) { return launch_mixed_unit<3>(n, kind, dc, hits, spec, args, num_cus, stream, stop); }
#endif
#if SCN_MIXED_IN_TU(4)
hipError_t scn_launch_mixed_tu4(uint32_t n, SCN_MIXED_ARGS) { return launch_mixed_unit<4>(n, kind, dc, hits, spec, args, num_cus, stream, stop); }
#endif
#if SCN_MIXED_IN_TU(5)
hipError_t scn_launch_mixed_tu5(uint32_t n, SCN_MIXED_ARGS) { return launch_mixed_unit<5>(n, kind, dc, hits, spec, args, num_cus, stream, stop); }
#endif
#if SCN_MIXED_IN_TU(6)
hipError_t scn_launch_mixed_tu6(uint32_t n, SCN_MIXED_ARGS) { return launch_mixed_unit<6>(n, kind, dc, hits, spec, args, num_cus, stream, stop); }
#endif
#if SCN_MIXED_IN_TU(7)
hipError_t scn_launch_mixed_tu7(uint32_t n, SCN_MIXED_ARGS) { return launch_mixed_unit<7>(n, kind, dc, hits, spec, args, num_cus, stream, stop); }
#endif

#if SCN_MIXED_IN_TU(0)
// the plan of size n, or false: its smallest radix R1 (the tw1 table has R1 - 1 rows) and the pass-1 thread count T1 (its row length)
#define SCN_MIXED_LOOKUP(N, R1, R2, R3, PAD1, PAD2, UNIT) \
  if (n == N) {                                            \
    if (rows) *rows = R1 - 1;                              \
    if (threads) *threads = R2 * R3;                       \
    return true;                                           \
  }
#define SCN_MIXED_BIG_LOOKUP(N, R1, R2, R3, PAD1, UNIT) SCN_MIXED_LOOKUP(N, R1, R2, R3, PAD1, 0, UNIT)
bool scn_mixed_layout(uint32_t n, uint32_t *rows, uint32_t *threads) {
  SCN_MIXED_PLANS(SCN_MIXED_LOOKUP)
  SCN_MIXED_BIG_PLANS(SCN_MIXED_BIG_LOOKUP)
  return false;
}
bool scn_mixed_size_supported(uint32_t n) { return scn_mixed_layout(n, nullptr, nullptr); }

#define SCN_MIXED_UNIT_OF(N, R1, R2, R3, PAD1, PAD2, UNIT) \
  if (n == N) unit = UNIT;
hipError_t scn_launch_mixed(uint32_t n, int kind, bool dc, bool hits, bool spec, const ScnFftArgs &args, int num_cus, hipStream_t stream,
                            hipEvent_t stop) {
  if (!hits && !spec) return hipErrorInvalidValue;
  if (args.n_buffers == 0) return stop ? hipEventRecord(stop, stream) : hipSuccess;
  int unit = -1;
#define SCN_MIXED_BIG_UNIT_OF(N, R1, R2, R3, PAD1, UNIT) SCN_MIXED_UNIT_OF(N, R1, R2, R3, PAD1, 0, UNIT)
  SCN_MIXED_PLANS(SCN_MIXED_UNIT_OF)
  SCN_MIXED_BIG_PLANS(SCN_MIXED_BIG_UNIT_OF)
  switch (unit) {
    case 0: return scn_launch_mixed_tu0(n, kind, dc, hits, spec, args, num_cus, stream, stop);
    case 1: return scn_launch_mixed_tu1(n, kind, dc, hits, spec, args, num_cus, stream, stop);
    case 2: return scn_launch_mixed_tu2(n, kind, dc, hits, spec, args, num_cus, stream, stop);
    case 3: return scn_launch_mixed_tu3(n, kind, dc, hits, spec, args, num_cus, stream, stop);
    case 4: return scn_launch_mixed_tu4(n, kind, dc, hits, spec, args, num_cus, stream, stop);
    case 5: return scn_launch_mixed_tu5(n, kind, dc, hits, spec, args, num_cus, stream, stop);
    case 6: return scn_launch_mixed_tu6(n, kind, dc, hits, spec, args, num_cus, stream, stop);
    case 7: return scn_launch_mixed_tu7(n, kind, dc, hits, spec, args, num_cus, stream, stop);
    default: return hipErrorInvalidValue;
  }
}
#endif
