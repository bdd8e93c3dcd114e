// scn_kernels.hip -- hand-written CDNA4 (gfx950) kernels for the spectrum-scan hot path.
//
// One launch does, for every raw IQ buffer of a batch, what the reference does per
// buffer on a CPU thread (process.cpp:293-299 + messageQueue.h:190-237):
//   K1 int8/int16/float IQ -> complex float   (utility.cpp:9-84)
//   K2 window multiply                        (process.cpp:28-34)
//   K3 N-point forward FFT                    (fft.cpp:20-25)
//   K4 10*log2(sqrt(re^2+im^2))/log2(10)      (utility.cpp:86-98)
//   K5 fftshift-indexed mask + threshold      (process.cpp:46-62)
// fused, with the FFT staged entirely in LDS.  HBM traffic per complex sample is the
// algorithmic minimum: raw sample in (8/4/2 B) + one float out (4 B).
//
// FFT structure (N = 4096 shown; see DESIGN.md for the other sizes):
//   4096 = 16 x 16 x 16, one workgroup of 256 threads per buffer, 16 points per thread.
//   n = 256a + 16b + c,  k = p + 16q + 256r
//   pass 1  thread (b,c): 16-pt DFT over a (inputs 256a + t, coalesced), twiddle W_4096^{t p}
//   pass 2  thread (p,c): 16-pt DFT over b, twiddle W_256^{c q}
//   pass 3  thread (p,q): 16-pt DFT over c, outputs k = t + 256 r (coalesced)
// Two LDS exchanges (ds_write_b64 / ds_read_b64) of the whole buffer through padded row
// layouts that are bank-conflict-free for all four access patterns and address every slot
// as per-thread base + immediate offset (no address VGPRs); 4 barriers per FFT.
// Workgroups are persistent over buffers (grid-stride), so the pass-1 twiddles and the
// window coefficients a thread needs live in registers for the whole launch.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>

#include "scn_device.h"
#include "scn_kernels.h"

// Build-time split: every kernel here is a template, compiled where a launcher instantiates it.  scanner_amd/build.py compiles
// this file once per SCN_TU value, side by side, each translation unit instantiating one group of sizes (a single hipcc run
// over all 330 specialisations takes a minute, most of it in the 8192- and 16384-point kernels); without SCN_TU (-1) one
// translation unit holds everything.
#ifndef SCN_TU
#define SCN_TU -1
#endif
#define SCN_IN_TU(x) (SCN_TU == -1 || SCN_TU == (x))
#define SCN_TU_COUNT 8  // 0: 16 / 32 points, 1: 64 / 128, 2: 256 / 512, 3: 1024 / 2048, 4: 4096, 5: 8192, 6: 16384, 7: everything that is not a fused FFT kernel

// (RawLoader<KIND>, the wire-format loaders of K1, and the cache policy of the two streams live in scn_device.h: the mixed-radix
// kernels of scn_mixed.hip use them too)

// ------------------------------------------------------------------------------------
// Fused kernel for N = 256*M points, M in {4, 8, 16} (N = 1024 / 2048 / 4096).
// One workgroup of T = 16*M threads per buffer, persistent over buffers, 16 points per thread.
//
//   n = T*a + M*b + c          k = p + 16q + 256r         a,b,p,q in [0,16), c,r in [0,M)
//   pass 1  thread t = M*b + c:  16-pt DFT over a of x[T*a + t]*w[T*a + t] -> *W_N^{t p} -> LDS row p
//   pass 2  thread (p, c):       16-pt DFT over b                          -> *W_{16M}^{c q} -> LDS
//   pass 3  M-pt DFT over c for each (p,q): thread t does the 16/M butterflies kl = t + T*u, outputs k = kl + 256 r
//
// LDS (complex = 8 B slots, every address = per-thread base + immediate, all four access patterns
// bank-conflict-free -- checked with SQ_LDS_BANK_CONFLICT and the bank model of the guide):
//   exchange 1   L1(p, col) = p*P1 + col,   P1 = T + M
//       write (pass 1): fixed p, lanes col = t                -> consecutive slots
//       read  (pass 2): thread (p, c), fixed b: p*P1 + b*M + c -> the 32/M rows a 32-lane group
//                       touches sit 8M bytes apart in bank space
//   exchange 2   L2(c, kl) = c*P2 + kl,     P2 = 256 + 16/M
//       write (pass 2): thread (p, c), fixed q: c*P2 + p + 16q -> 16 lanes on 16 distinct bank pairs
//       read  (pass 3): fixed c, lanes kl consecutive
//   then the pass-2 twiddle table [q][c] (16*M entries), DC-sum scratch, the hit counter.
//
// The next buffer's 16 loads are issued in three groups (6 / 5 / 5) over the first three barrier-separated phases of the
// current buffer: all 16 in front of pass 1 stalled the wave ~1600 cycles at issue whenever the memory pipeline was backed
// up (-3 .. -5 us per launch, profiles/r01_floors.md; 4/4/4/4 and 8/4/4 are within noise of 6/5/5).
//
// Work distribution over the persistent workgroups.  Float input: static grid-stride assignment (buf = blockIdx.x +
// k*gridDim.x).  The integer formats from 4096 points up (scn_uses_queue, scn_kernels.h): every workgroup starts on buffer
// blockIdx.x and then takes buffers from a device-scope queue, one iteration ahead so that the prefetch knows its target.
// With the static form the workgroups of one launch finish far apart (dispatch order, 10 or 11 buffers each, uneven
// speeds: 36..55 us in a 55 us int16 launch).  For the memory-bound float path that tail is not idle capacity (the
// remaining workgroups run faster: measured neutral); for the compute-bound integer formats and the 8192-point kernel
// (8 buffers per workgroup) it is.
// The queue is sharded 8 ways: one returning device-scope atomic saturates at ~88 per us on one word, so 8192
// dequeues on one head would take longer than the launch (measured: 74 -> 115 us).  Workgroups land on XCD
// blockIdx.x % 8, so a shard is pulled by one XCD; shard x owns the buffers b = 8 j + x.  The launcher never
// starts more workgroups than buffers, so workgroup g takes buffer g statically (shard g % 8, j = g / 8) and
// head x hands out j = j0, j0 + 1, ... with j0 = the number of workgroups in shard x.  Heads are never reset:
// work_base[x] is head x's value before the launch (host-tracked; a launch adds exactly the number of buffers of
// shard x, because every workgroup stops at its first index past the end).
// Measured per wire format (one box, single stream, us per launch static -> queue): int16 4096-pt 64.1 -> 61.2,
// int8 63.9 -> 59.4, 8192-pt int16 98.5 -> 89.9, float 4096-pt 76.1 -> 77.0, 8192-pt float 83.8 -> 85.8.
// The dequeue must stay ONE plain global_atomic_add whose result is collected an iteration later.  LLVM's AMDGPU
// atomic optimizer rewrites any atomic with a provably uniform address into a wave reduction followed at once
// by s_waitcnt vmcnt(0) + readfirstlane (wave 0 then sits out the atomic's ~3 us round trip on every buffer:
// 74 -> 116 us), so the address gets an opaque per-lane zero offset.
// ------------------------------------------------------------------------------------
#define SCN_WORK_QUEUE_SETUP()                                                                         \
  const uint32_t wq_shard = blockIdx.x & 7u;                                                           \
  uint32_t *const wq_head = args.work_counter + 32u * wq_shard; /* one 128-byte line per head */       \
  const uint32_t wq_base = args.work_base[wq_shard];                                                   \
  const uint32_t wq_j0 = (gridDim.x - wq_shard + 7u) >> 3;                                             \
  uint32_t wq_zero;                                                                                    \
  asm("v_mov_b32 %0, 0" : "=v"(wq_zero));                                                              \
  auto wq_take = [&]() -> uint32_t { return atomicAdd(wq_head + wq_zero, 1u); };                       \
  auto wq_buffer = [&](uint32_t taken) -> uint32_t { return 8u * (wq_j0 + (taken - wq_base)) + wq_shard; }

template <int M>
struct Geo {
  static_assert(M == 4 || M == 8 || M == 16, "1024, 2048 or 4096 points");
  static constexpr uint32_t N = 256u * M;
  static constexpr uint32_t T = 16u * M;
  static constexpr uint32_t P1 = T + M;
  static constexpr uint32_t P2 = 256u + 16u / M;
  static constexpr uint32_t EXCH = (16u * P1 > M * P2) ? 16u * P1 : M * P2;  // slots
  static constexpr uint32_t LDS_BYTES = EXCH * 8u + T * 8u + 16u * 4u + 2u * 4u + 64u * 4u + 8u;
  static constexpr uint32_t WAVES = T / 64;
  // the register prefetch of the next buffer costs a wave per SIMD (4 -> 3): VGPR budget 168
  static constexpr uint32_t WAVES_PER_SIMD = 3;
  static constexpr uint32_t WG_PER_CU = (WAVES_PER_SIMD * 4u) / WAVES;
};

// natural output index o of a thread -> register that holds it after pass 3
template <int M>
__device__ __forceinline__ constexpr int out_reg(int o) {
  return M == 4 ? o : M == 8 ? (o & ~7) + OUT8(o & 7) : OUT16(o & 15);
}

// Output modes (template parameters HITS, SPEC): spectrum only (false, true), spectrum + hits (true, true), hits only
// (true, false).  The hits-only kernel has NO store instructions and no per-bin v_log_f32: it compares the linear power with a
// guarded linear threshold (ScnFftArgs::p_lo, a shade below 10^(threshold/5)) and evaluates the dB map -- the same function
// of the bin's power as in the other modes, so that `magnitudes[j] > m_threshold` (process.cpp:54) decides identically --
// only in the waves that hold a candidate.
template <int M, int KIND, bool DC, bool HITS, bool SPEC>
__global__ __launch_bounds__(16 * M, Geo<M>::WAVES_PER_SIMD) void scn_fft_kernel(ScnFftArgs args) {
  static_assert(HITS || SPEC, "a kernel that reports nothing");
  typedef Geo<M> G;
  constexpr int AUX_LD = SCN_AUX_LD;
  constexpr int AUX_ST = SCN_AUX_ST;
  constexpr bool DYN = scn_uses_queue(KIND, G::N);
  constexpr uint32_t N = G::N, T = G::T, P1 = G::P1, P2 = G::P2;
  typedef RawLoader<KIND> L;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v2f *lds = reinterpret_cast<v2f *>(smem_raw);
  v2f *lds_tw2 = lds + G::EXCH;                              // [16][M]
  int *lds_cnt = reinterpret_cast<int *>(lds_tw2 + T);      // [32] DC-sum scratch (re[16], im[16]: up to 16 waves)
  int *lds_hits = lds_cnt + 32;                             // [2] hit counters, alternating per buffer
  uint32_t *lds_next = reinterpret_cast<uint32_t *>(lds_hits + 2);  // [1] the buffer this workgroup takes after the next one

  const uint32_t t = threadIdx.x;
  const uint32_t p2 = t / M, c2 = t % M;  // pass-2 identity (p, c); also the (q, c) of the table entry below
  const uint32_t lane = t & 63, wave = t >> 6;

  // the first buffer's samples go out before anything else: their latency then overlaps the
  // ~31 table loads below instead of following them
  typename L::raw_t raw[16];
  if (blockIdx.x < args.n_buffers) {
    __amdgpu_buffer_rsrc_t r0 = make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)blockIdx.x * L::kBufBytes(N), L::kBufBytes(N));
#pragma unroll
    for (int a = 0; a < 16; a++) raw[a] = L::template load<AUX_LD>(r0, N, t, T * a);
  }

  // persistent per-thread constants: pass-1 twiddles W_N^(t*p) and window taps
  cf tw1[16];
#pragma unroll
  for (int p = 1; p < 16; p++) tw1[p] = from_v2f(args.tw1_table[(p - 1) * T + t]);
  // Window taps with the ENOB scale folded in: onebymax (utility.cpp:65) is +-2^-k, so
  // float(s)*scale*w and float(s)*(scale*w) round identically -- one multiply per component saved.
  float win[16];
#pragma unroll
  for (int a = 0; a < 16; a++) win[a] = args.window[T * a + t] * args.scale;
  // pass-2 twiddles W_{16M}^(c*q) = W_N^(16 c q), table [q][c] shared by the workgroup
  lds_tw2[t] = args.twiddle[(16 * p2 * c2) & (N - 1)];
  SCN_WORK_QUEUE_SETUP();
  if (t == 0) {
    lds_hits[0] = lds_hits[1] = 0;
    lds_next[0] = DYN ? wq_buffer(wq_take()) : blockIdx.x + gridDim.x;
  }
  __syncthreads();

  v2f *w1 = lds + t;                      // + p*P1
  v2f *r1 = lds + p2 * P1 + c2;           // + b*M
  v2f *w2 = lds + c2 * P2 + p2;           // + 16*q
  v2f *r3 = lds + t;                      // + c*P2 + T*u
  const v2f *tw2 = lds_tw2 + c2;          // + q*M
  const uint32_t st_voff = t * 4u;        // global store offset of output o: voffset (per lane) + scalar part

  // output o = u*M + r of this thread is bin j = t + T*u + 256*r
  auto joff_of = [](int o) -> uint32_t { return T * ((uint32_t)o / M) + 256u * ((uint32_t)o % M); };
  // K5 mask of this thread's 16 output bins (process.cpp:46-52): depends on (t, o) only
  uint32_t keepmask = 0;
  if (HITS) {
#pragma unroll
    for (int o = 0; o < 16; o++) {
      const uint32_t j = t + joff_of(o);
      const uint32_t i = j ^ (N / 2);  // (j + N/2) % N, process.cpp:47
      const bool keep = !(j < args.dc_ignore || (N - j) < args.dc_ignore) && !(i < args.i_lo || i > args.i_hi);
      keepmask |= keep ? (1u << o) : 0u;
    }
  }
  uint32_t par = 0;             // which of the two LDS hit counters the buffer in flight uses
  uint32_t prev = 0xffffffffu;  // the buffer whose recorders may still be running (none yet)

  uint32_t buf = blockIdx.x;
  uint32_t nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_next[0]);
  while (buf < args.n_buffers) {
    const bool more = nxt < args.n_buffers;
    // ---- K1 + K2: convert, window (the samples were fetched during the previous buffer's passes) ----
    int dc_re = 0, dc_im = 0;
    if (DC) {
      // integer mean with the reference's int32 /= uint32 quirk (utility.cpp:77-78)
      int sr = 0, si = 0;
#pragma unroll
      for (int a = 0; a < 16; a++) {
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
      for (uint32_t w = 0; w < G::WAVES; w++) {
        sr += lds_cnt[w];
        si += lds_cnt[16 + w];
      }
      dc_re = (int)((uint32_t)sr / N);
      dc_im = (int)((uint32_t)si / N);
    }

    cf v[16];
#pragma unroll
    for (int a = 0; a < 16; a++) v[a] = L::conv(raw[a], dc_re, dc_im, 1.0f) * win[a];
    // Take the buffer after the next one; the answer is only needed at the end of this iteration.  Issued here,
    // behind the convert: a grab ahead of it sits in a divergent branch and the merged wait count then makes
    // wave 0 sit out the atomic's round trip where it waits for the samples.
    uint32_t taken = 0;
    if (DYN && t == 0 && more) taken = wq_take();
    // The raw registers are free again: fetch the next buffer of this workgroup while this one is
    // transformed.  Branch-free (past the last buffer the descriptor has zero records: the loads return
    // zeros without touching memory), so every load sits in the same basic block as the butterflies and
    // can be issued between them.
    const __amdgpu_buffer_rsrc_t rn =
        make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)(more ? nxt : buf) * L::kBufBytes(N), more ? L::kBufBytes(N) : 0u);
    auto prefetch = [&](int a_lo, int a_hi) {
#pragma unroll
      for (int a = 0; a < 16; a++)
        if (a >= a_lo && a < a_hi) raw[a] = L::template load<AUX_LD>(rn, N, t, T * a);
    };
    prefetch(0, 6);

    // ---- pass 1: DFT over a, twiddle W_N^(t p), scatter to row p ----
    fft16(v);
#pragma unroll
    for (int p = 0; p < 16; p++) {
      cf y = v[OUT16(p)];
      if (p) y = cmul(y, tw1[p]);
      w1[p * P1] = to_v2f(y);
    }
    __syncthreads();  // barrier 1
    if (HITS) {
      // every wave has passed the barrier above, so the previous buffer's recorders are done
      if (t == 0 && prev != 0xffffffffu) {
        args.per_buffer_hits[prev] = (uint32_t)lds_hits[par ^ 1];
        if (args.host_hits) args.host_hits[prev] = (uint32_t)lds_hits[par ^ 1];
        lds_hits[par ^ 1] = 0;
      }
    }

    // ---- pass 2: thread (p, c): DFT over b, twiddle W_{16M}^(c q) ----
#pragma unroll
    for (int b = 0; b < 16; b++) v[b] = from_v2f(r1[b * M]);
    prefetch(6, 11);
    fft16(v);
#pragma unroll
    for (int q = 1; q < 16; q++) v[OUT16(q)] = cmul(v[OUT16(q)], from_v2f(tw2[q * M]));
    __syncthreads();  // barrier 2: every exchange-1 read done before the area is re-used
    prefetch(11, 16);
#pragma unroll
    for (int q = 0; q < 16; q++) w2[q * 16] = to_v2f(v[OUT16(q)]);
    __syncthreads();  // barrier 3

    // ---- pass 3: M-point DFT over c ----
#pragma unroll
    for (int u = 0; u < 16 / M; u++)
#pragma unroll
      for (int c = 0; c < M; c++) v[u * M + c] = from_v2f(r3[c * P2 + T * u]);
    if constexpr (M == 16) fft16(v);
    if constexpr (M == 8) {
      fft8(v);
      fft8(v + 8);
    }
    if constexpr (M == 4) {
#pragma unroll
      for (int u = 0; u < 4; u++) radix4(v[4 * u], v[4 * u + 1], v[4 * u + 2], v[4 * u + 3]);
    }

    // ---- K4 + K5 ----
    // The thread's 16 LINEAR powers stay in `pw` for the hit path; the spectrum gets the dB map of scn_device.h: its product
    // form inline, and -- only in waves that hold a bin from SCN_P_EXACT_FROM up -- the exact form stored over it for those
    // bins (the other lanes' stores go to an out-of-range offset and are dropped by the descriptor's range check: no branch
    // per bin).
    v16f pw;
    float gmax[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // the largest power of each group of four outputs (max ignores NaN)
    __amdgpu_buffer_rsrc_t rout = make_rsrc(args.power_db + (size_t)buf * N, (SPEC && args.power_db) ? 4u * N : 0u);
#pragma unroll
    for (int o = 0; o < 16; o++) {
      const float q = power_of(v[out_reg<M>(o)]);
      pw[o] = q;
      gmax[o >> 2] = fmaxf(gmax[o >> 2], q);
      if constexpr (SPEC) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, db_fast(q)), rout, st_voff, 4u * joff_of(o), AUX_ST);
    }
    const float pmax = fmaxf(fmaxf(gmax[0], gmax[1]), fmaxf(gmax[2], gmax[3]));  // pre-filter of the hit path and of the exact half
    if constexpr (SPEC) {
      if (__ballot(pmax >= SCN_P_EXACT_FROM)) {
        // Strong bins are neighbours -- a tone's main lobe sits in ONE or two output indices o of a wave -- and the bench
        // input has such a wave in most buffers, so this path must stay short: all 16 outputs at ~14 operations each made
        // the wave its workgroup's straggler (+4 us per C2 launch), 16 wave-wide tests still +1 .. 4 us; so: the groups of
        // four first, then the outputs of a group that holds one.
#pragma unroll
        for (int g = 0; g < 4; g++) {
          if (__ballot(gmax[g] >= SCN_P_EXACT_FROM)) {
#pragma unroll
            for (int o = 4 * g; o < 4 * g + 4; o++) {
              const float q = pw[o];  // NB: never __builtin_bit_cast a vector ELEMENT: clang reads element 0 for every o
              if (__ballot(q >= SCN_P_EXACT_FROM)) {
                const float d = db_exact(q);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d), rout, q >= SCN_P_EXACT_FROM ? st_voff : 0x80000000u, 4u * joff_of(o), AUX_ST);
              }
            }
          }
        }
      }
    }
    if (t == 0) lds_next[0] = !more ? 0xffffffffu : DYN ? wq_buffer(taken) : nxt + gridDim.x;
    __syncthreads();  // barrier 4: exchange area free again; lds_next visible (it is rewritten three barriers from now)
    const uint32_t after = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_next[0]);
    if (HITS) {
      // Recording runs AFTER the barrier: a wave that holds detections does not stall the other
      // waves of its workgroup, they go on to the next buffer and meet it at that buffer's first
      // barrier (its loads are in flight meanwhile).  Only such a wave evaluates the per-bin test.
      if (__ballot(pmax > args.p_lo))
        scn_record_hits_lanes<16, false, true>(pw, gmax, keepmask, args, &lds_hits[par], buf, lane, [&](int o) -> uint32_t { return (t + joff_of(o)) ^ (N / 2); });
      prev = buf;
      par ^= 1;
    }
    buf = nxt;
    nxt = after;
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
// 256 and 512 points (round 3): SEVERAL buffers per workgroup.
//
// The reference plans whatever --count it is given (fft.cpp:4-11, scan.cpp:85); below 1024 points the 16 x 16 x M form has
// M = 2 or 1, i.e. 32 or 16 threads per buffer -- half or a quarter of a wave.  A 256-thread workgroup therefore carries
// SLOTS = 8 (512 points) or 16 (256 points) consecutive buffers per iteration, each in its own LDS region, through the same
// three passes as scn_fft_kernel (n = T a + M b + c, k = p + 16 q + 256 r, the same padded layouts: P1 = T + M,
// P2 = 256 + 16 / M, conflict-free within the 16-lane groups a ds_*_b64 is served in):
//   pass 1  thread t = M b + c of a slot: 16-pt DFT over a of x[T a + t] w[T a + t] -> * W_N^(t p) -> LDS row p
//   pass 2  thread (p, c): 16-pt DFT over b -> * W_(16 M)^(c q) -> LDS
//   pass 3  M = 2: eight radix-2 butterflies per thread (kl = t + 32 u -> bins kl, kl + 256);  M = 1: there is none -- 256 = 16 x 16,
//           pass 2's outputs are the spectrum (thread t stores its own bins t + 16 q: 64-byte runs per buffer)
// One buffer descriptor spans the workgroup's SLOTS consecutive buffers (a wave holds two or four of them, so a per-buffer
// descriptor would not be wave-uniform); slots past the end of the batch read zeros and store nothing (range check).
// Hits: a wave is no longer one buffer; a buffer's records take their places by a prefix sum over its T lanes, no atomic
// (scn_record_hits_segment).  These sizes ran the staged double-precision path (scn_generic.hip) at 40 Gsamples/s.
// ------------------------------------------------------------------------------------
namespace {
// K4 + K5 of the kernels in which a wave holds several buffers (scn_fft_small_kernel, scn_fft_tiny_kernel).  Thread t of a
// buffer's T lanes owns 16 outputs; output o is bin t + joff_of(o), its power power_at(o).  The dB map as everywhere
// (db_of_power: product form, exact form for strong bins -- evaluated in the waves that hold one); the values stay in
// registers, a hit is a dB value above the threshold (strict >, process.cpp:54) inside the keep mask, and the buffer's
// records take their places by a prefix sum over its T lanes (scn_record_hits_segment).  A hits-only kernel forms the dB
// values only in the waves that hold a candidate by linear power.
// store(o, d, pred): where output o's dB value goes if pred (global memory for 256 / 512 points, the LDS tile below that).
template <int T, bool SPEC, bool HITS, typename POWER, typename JOFF, typename STORE>
__device__ __forceinline__ void scn_small_db_store_record(POWER power_at, JOFF joff_of, STORE store, const ScnFftArgs &args, uint32_t buf, bool valid,
                                                          uint32_t keepmask, uint32_t t, uint32_t n) {
  v16f pw, dbv;
  float gmax[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int o = 0; o < 16; o++) {
    const float q = power_at(o);
    pw[o] = q;
    gmax[o >> 2] = fmaxf(gmax[o >> 2], q);
    if constexpr (SPEC) {
      // (the store takes the value just computed, in the loop that computes it: written as a second loop `dbv[o] = db_fast(pw[o]);
      //  store(dbv[o])`, hipcc 7.2 stored element 0's value sixteen times -- caught by tests/test_dispatch_gpu.py)
      const float d = db_fast(q);
      dbv[o] = d;
      store(o, d, true);
    }
  }
  const float pmax = fmaxf(fmaxf(gmax[0], gmax[1]), fmaxf(gmax[2], gmax[3]));
  const bool any_cand = HITS && __ballot(valid && pmax > args.p_lo) != 0ull;  // wave-uniform
  if (SPEC || any_cand) {
    if constexpr (!SPEC) {
#pragma unroll
      for (int o = 0; o < 16; o++) dbv[o] = db_fast(pw[o]);
    }
    if (__ballot(pmax >= SCN_P_EXACT_FROM)) {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        if (__ballot(gmax[g] >= SCN_P_EXACT_FROM)) {
#pragma unroll
          for (int o = 4 * g; o < 4 * g + 4; o++) {
            const float q = pw[o];
            if (__ballot(q >= SCN_P_EXACT_FROM)) {
              const float d = db_exact(q);
              dbv[o] = q >= SCN_P_EXACT_FROM ? d : dbv[o];
              if constexpr (SPEC) store(o, d, q >= SCN_P_EXACT_FROM);
            }
          }
        }
      }
    }
  }
  if constexpr (HITS) {
    uint32_t total = 0;
    if (any_cand) {
      uint32_t hm = 0;
#pragma unroll
      for (int o = 0; o < 16; o++) hm |= (dbv[o] > args.threshold) ? (1u << o) : 0u;
      hm &= valid ? keepmask : 0u;
      total = scn_record_hits_segment<T, 16>(dbv, hm, args, buf, t, [&](int o) -> uint32_t { return (t + joff_of(o)) ^ (n / 2); });
    }
    if (t == 0 && valid) {
      args.per_buffer_hits[buf] = total;
      if (args.host_hits) args.host_hits[buf] = total;
    }
  }
}

template <int M>
struct GeoSmall {
  static constexpr uint32_t N = 256u * M, T = 16u * M, SLOTS = 256u / T;
  static constexpr uint32_t P1 = T + M, P2 = 256u + 16u / M;
  static constexpr uint32_t EXCH = (16u * P1 > M * P2) ? 16u * P1 : M * P2;  // slots per buffer
  static constexpr uint32_t LDS_BYTES = SLOTS * EXCH * 8u + T * 8u + 16u;
  static constexpr uint32_t WG_PER_CU = 3;
};
}  // namespace

template <int M, int KIND, bool DC, bool HITS, bool SPEC>
__global__ __launch_bounds__(256, 3) void scn_fft_small_kernel(ScnFftArgs args) {
  static_assert(M == 1 || M == 2, "256 or 512 points");
  static_assert(HITS || SPEC, "a kernel that reports nothing");
  typedef GeoSmall<M> G;
  constexpr int AUX_LD = SCN_AUX_LD;
  constexpr int AUX_ST = SCN_AUX_ST;
  constexpr uint32_t N = G::N, T = G::T, SLOTS = G::SLOTS, P1 = G::P1, P2 = G::P2;
  typedef RawLoader<KIND> L;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const uint32_t tid = threadIdx.x, t = tid % T, slot = tid / T;
  v2f *lds = reinterpret_cast<v2f *>(smem_raw) + slot * G::EXCH;                     // this buffer's exchange area
  v2f *lds_tw2 = reinterpret_cast<v2f *>(smem_raw) + SLOTS * G::EXCH;                // [16][M], shared by the slots
  const uint32_t p2 = t / M, c2 = t % M;
  const uint32_t buf_bytes = L::kBufBytes(N);

  // the iteration's SLOTS consecutive buffers under one descriptor (zero records past the end of the batch)
  auto in_rsrc = [&](uint32_t first) {
    const bool ok = first < args.n_buffers;
    const uint32_t nb = ok ? (args.n_buffers - first < SLOTS ? args.n_buffers - first : SLOTS) : 0u;
    return make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)(ok ? first : 0u) * buf_bytes, nb * buf_bytes);
  };
  // planar int16 reads I and Q of a buffer n samples apart: the loader takes the buffer's own sample count.  The slot's offset
  // goes into the PER-LANE index, the part that depends on `a` only into the wave-uniform one: a wave holds several slots, and
  // until round 4 the slot offset sat in the uniform operand -- hipcc then wraps every load in a waterfall loop (one trip per
  // distinct slot in the wave: four at 256 points, 80 s_nop and 32 v_readfirstlane per iteration)
  typename L::raw_t raw[16];
  auto load_group = [&](const __amdgpu_buffer_rsrc_t &r, int a_lo, int a_hi) {
#pragma unroll
    for (int a = 0; a < 16; a++)
      if (a >= a_lo && a < a_hi) raw[a] = L::template load<AUX_LD>(r, N, t + slot * (KIND == SCN_K_SHORT ? 2u * N : N), T * a);
  };
  uint32_t first = blockIdx.x * SLOTS;
  load_group(in_rsrc(first), 0, 16);

  cf tw1[16];
#pragma unroll
  for (int p = 1; p < 16; p++) tw1[p] = from_v2f(args.tw1_table[(p - 1) * T + t]);
  float win[16];
#pragma unroll
  for (int a = 0; a < 16; a++) win[a] = args.window[T * a + t] * args.scale;
  if (slot == 0) lds_tw2[t] = args.twiddle[(16 * p2 * c2) & (N - 1)];  // W_(16 M)^(c q) = W_N^(16 c q), entry q*M + c
  __syncthreads();

  v2f *w1 = lds + t;                      // + p*P1
  v2f *r1 = lds + p2 * P1 + c2;           // + b*M
  v2f *w2 = lds + c2 * P2 + p2;           // + 16*q
  v2f *r3 = lds + t;                      // + c*P2 + T*u
  const v2f *tw2 = lds_tw2 + c2;          // + q*M
  auto joff_of = [](int o) -> uint32_t { return T * ((uint32_t)o / M) + 256u * ((uint32_t)o % M); };  // output o is bin t + joff_of(o)
  uint32_t keepmask = 0;
  if (HITS) {
#pragma unroll
    for (int o = 0; o < 16; o++) {
      const uint32_t j = t + joff_of(o);
      const uint32_t i = j ^ (N / 2);  // (j + N/2) % N, process.cpp:47
      const bool keep = !(j < args.dc_ignore || (N - j) < args.dc_ignore) && !(i < args.i_lo || i > args.i_hi);
      keepmask |= keep ? (1u << o) : 0u;
    }
  }

  for (; first < args.n_buffers; first += gridDim.x * SLOTS) {
    const uint32_t buf = first + slot;
    const bool valid = buf < args.n_buffers;
    int dc_re = 0, dc_im = 0;
    if (DC) {
      // integer mean with the reference's int32 /= uint32 quirk (utility.cpp:77-78), summed over the T lanes of this buffer
      int sr = 0, si = 0;
#pragma unroll
      for (int a = 0; a < 16; a++) {
        int re, im;
        L::ints(raw[a], re, im);
        sr += re;
        si += im;
      }
#pragma unroll
      for (uint32_t off = T / 2; off > 0; off >>= 1) {
        sr += __shfl_xor(sr, (int)off, 64);
        si += __shfl_xor(si, (int)off, 64);
      }
      dc_re = (int)((uint32_t)sr / N);
      dc_im = (int)((uint32_t)si / N);
    }
    cf v[16];
#pragma unroll
    for (int a = 0; a < 16; a++) v[a] = L::conv(raw[a], dc_re, dc_im, 1.0f) * win[a];
    const __amdgpu_buffer_rsrc_t rn = in_rsrc(first + gridDim.x * SLOTS);
    load_group(rn, 0, 6);

    // ---- pass 1 ----
    fft16(v);
#pragma unroll
    for (int p = 0; p < 16; p++) {
      cf y = v[OUT16(p)];
      if (p) y = cmul(y, tw1[p]);
      w1[p * P1] = to_v2f(y);
    }
    __syncthreads();
    // ---- pass 2 ----
#pragma unroll
    for (int b = 0; b < 16; b++) v[b] = from_v2f(r1[b * M]);
    load_group(rn, 6, 11);
    fft16(v);
    if constexpr (M == 1) {
      // 256 = 16 x 16: pass 2's outputs ARE the spectrum (thread t holds bins t + 16 q in v[OUT16(q)]): no twiddles (all
      // ones), no second exchange (until round 4 it was written and read back by the same thread: 32 LDS accesses, 15 complex
      // multiplies by one and two barriers per buffer for nothing)
      load_group(rn, 11, 16);
    } else {
#pragma unroll
      for (int q = 1; q < 16; q++) v[OUT16(q)] = cmul(v[OUT16(q)], from_v2f(tw2[q * M]));
      __syncthreads();  // every exchange-1 read done before the area is re-used
      load_group(rn, 11, 16);
#pragma unroll
      for (int q = 0; q < 16; q++) w2[q * 16] = to_v2f(v[OUT16(q)]);
      __syncthreads();
      // ---- pass 3: eight radix-2 butterflies per thread (kl = t + 32 u -> bins kl, kl + 256) ----
#pragma unroll
      for (int u = 0; u < 16 / M; u++)
#pragma unroll
        for (int c = 0; c < M; c++) v[u * M + c] = from_v2f(r3[c * P2 + T * u]);
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const cf a = v[2 * u], b = v[2 * u + 1];
        v[2 * u] = a + b;      // r = 0
        v[2 * u + 1] = a - b;  // r = 1
      }
    }
    // ---- K4, K5 (a wave holds several buffers: scn_small_db_store_record) ----
    const uint32_t nvalid = args.n_buffers - first < SLOTS ? args.n_buffers - first : SLOTS;
    __amdgpu_buffer_rsrc_t rout = make_rsrc(args.power_db + (size_t)first * N, (SPEC && args.power_db) ? nvalid * 4u * N : 0u);
    const uint32_t st_voff = (slot * N + t) * 4u;
    scn_small_db_store_record<(int)T, SPEC, HITS>(
        [&](int o) -> float { return power_of(v[M == 1 ? OUT16(o) : o]); }, joff_of,
        [&](int o, float d, bool pred) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d), rout, pred ? st_voff : 0x80000000u, 4u * joff_of(o), AUX_ST); },
        args, buf, valid, keepmask, t, N);
    __syncthreads();  // exchange areas free again
  }
}

// ------------------------------------------------------------------------------------
// 16, 32, 64 and 128 points (round 4): N = 16 R with R = 1, 2, 4, 8 THREADS per buffer.
//
// fft.cpp:4-11 plans whatever --count it is given; these sizes ran the staged double-precision path (39 Gsamples/s at 128
// points).  n = R a + c, k = p + 16 r:
//   X[p + 16 r] = sum_c W_R^(c r) * ( W_N^(c p) * sum_a W_16^(a p) x[R a + c] w[R a + c] )
//   pass 1  thread c of a buffer: the 16-point DFT over a of its 16 samples, * W_N^(c p) -> LDS [p][c]
//   pass 2  thread c': the R-point DFTs over c for ITS 16 / R values of p = c' + R m; output o = R m + r is bin c' + R m + 16 r
//   R = 1: one thread holds the whole buffer, there is no exchange at all.
// A 256-thread workgroup carries SLOTS = 256 / R consecutive buffers per iteration: one CHUNK of 4096 samples, contiguous in
// memory on both sides.  A thread's own samples and bins are runs of R elements per buffer (one element at 16 points), so the
// chunk crosses LDS on its way in and on its way out and every global access is the 1024-point kernel's: thread tid loads
// elements 256 a + tid of the chunk (512 / 256 / 128 bytes per wave instruction) into a tile [slot][N + R] -- the R elements of
// padding make the gather of x[R a + c] conflict-free for every R --, gathers its 16 samples from there; the dB values go
// back through a tile of the same shape and leave as 256-byte runs.  (The first form of this kernel loaded and stored its
// runs of R directly: 744 / 214 / 119 us per 33.5 M samples at 16 / 64 / 128 points against 74 at 1024.)  One LDS region
// serves as raw tile, exchange and dB tile in turn: six barriers per iteration (four at 16 points).
// Hits by the prefix sum over the buffer's lanes, like scn_fft_small_kernel.  The exchange is [p][R + 1] per buffer with
// R + 16 slots of padding between buffers: both sides conflict-free (16-lane write groups, 32-lane read groups; found by
// enumeration).
// ------------------------------------------------------------------------------------
namespace {
template <int R>
struct GeoTiny {
  static constexpr uint32_t N = 16u * R, T = R, SLOTS = 256u / R;
  static constexpr uint32_t P1 = R + 1u;
  static constexpr uint32_t EXCH = R == 1 ? 0u : 16u * P1 + R;  // 8-byte slots per buffer
  static constexpr uint32_t PITCH = N + R;                       // elements per buffer in the raw tile and in the dB tile
  static constexpr uint32_t TILE_BYTES = SLOTS * PITCH * 8u;     // (raw samples of up to 8 bytes)
  static constexpr uint32_t LDS_BYTES = (TILE_BYTES > SLOTS * EXCH * 8u ? TILE_BYTES : SLOTS * EXCH * 8u) + 16u;
  static constexpr uint32_t WG_PER_CU = 3;
};

// the chunk's raw elements through the LDS tile, per wire format: thread tid loads element 256 a + tid (a < 16; the planar
// format: ONE 32-bit load of the two neighbouring 16-bit elements 512 a + 2 tid (+1) of the chunk's I / Q blocks -- N is even, so
// both lie in the same block of the same buffer; as two 16-bit loads packed in registers the prefetch held 32 registers
// instead of 16 until the pack and these kernels spilled 9 .. 16 VGPRs), stage() writes what it loaded to [slot][PITCH],
// gather() returns sample j of buffer `slot` in RawLoader's register form.
template <int KIND, int R>
struct TinyStage {
  typedef GeoTiny<R> G;
  typedef RawLoader<KIND> L;
  static constexpr bool PLANAR = KIND == SCN_K_SHORT;
  typedef typename L::raw_t reg_t;  // (planar: two 16-bit elements packed)
  template <int AUX>
  static __device__ __forceinline__ reg_t load(__amdgpu_buffer_rsrc_t r, uint32_t tid, int a) {
    if constexpr (PLANAR) {
      return __builtin_amdgcn_raw_buffer_load_b32(r, tid * 4u, 1024u * (uint32_t)a, AUX);
    } else {
      return L::template load<AUX>(r, G::N, tid, 256u * (uint32_t)a);
    }
  }
  static __device__ __forceinline__ void stage(char *tile, uint32_t tid, int a, reg_t v) {
    if constexpr (PLANAR) {  // per buffer: I[N] then Q[N]; tile: I at [slot][0 .. N), Q at [slot][PITCH .. PITCH + N), 16-bit elements
      unsigned short *t16 = reinterpret_cast<unsigned short *>(tile);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const uint32_t e = 512u * (uint32_t)a + 2u * tid + h, s = e / (2u * G::N), j = e % (2u * G::N);
        t16[s * 2u * G::PITCH + (j < G::N ? j : G::PITCH + j - G::N)] = (unsigned short)(h ? (uint32_t)v >> 16 : (uint32_t)v & 0xffffu);
      }
    } else {
      const uint32_t e = 256u * (uint32_t)a + tid, idx = (e / G::N) * G::PITCH + e % G::N;
      if constexpr (KIND == SCN_K_FLOAT_COMPLEX) reinterpret_cast<v2f *>(tile)[idx] = v;
      else if constexpr (KIND == SCN_K_SHORT_COMPLEX) reinterpret_cast<int *>(tile)[idx] = v;
      else reinterpret_cast<unsigned short *>(tile)[idx] = (unsigned short)v;
    }
  }
  static __device__ __forceinline__ typename L::raw_t gather(const char *tile, uint32_t slot, uint32_t j) {
    if constexpr (PLANAR) {
      const unsigned short *t16 = reinterpret_cast<const unsigned short *>(tile) + slot * 2u * G::PITCH;
      return (int)((uint32_t)t16[j] | ((uint32_t)t16[G::PITCH + j] << 16));
    } else if constexpr (KIND == SCN_K_FLOAT_COMPLEX) {
      return reinterpret_cast<const v2f *>(tile)[slot * G::PITCH + j];
    } else if constexpr (KIND == SCN_K_SHORT_COMPLEX) {
      return reinterpret_cast<const int *>(tile)[slot * G::PITCH + j];
    } else {
      return (int)reinterpret_cast<const unsigned short *>(tile)[slot * G::PITCH + j];
    }
  }
};
}  // namespace

template <int R, int KIND, bool DC, bool HITS, bool SPEC>
__global__ __launch_bounds__(256, 3) void scn_fft_tiny_kernel(ScnFftArgs args) {
  static_assert(R == 1 || R == 2 || R == 4 || R == 8, "16, 32, 64 or 128 points");
  static_assert(HITS || SPEC, "a kernel that reports nothing");
  typedef GeoTiny<R> G;
  typedef TinyStage<KIND, R> S;
  constexpr int AUX_LD = SCN_AUX_LD;
  constexpr int AUX_ST = SCN_AUX_ST;
  constexpr uint32_t N = G::N, SLOTS = G::SLOTS, P1 = G::P1, PITCH = G::PITCH;
  typedef RawLoader<KIND> L;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const uint32_t tid = threadIdx.x, t = tid % R, slot = tid / R;
  v2f *lds = reinterpret_cast<v2f *>(smem_raw) + slot * G::EXCH;  // this buffer's exchange area
  float *tile_out = reinterpret_cast<float *>(smem_raw);          // the dB tile [slot][PITCH]
  const uint32_t buf_bytes = L::kBufBytes(N);

  auto in_rsrc = [&](uint32_t first) {  // the iteration's chunk: SLOTS consecutive buffers (zero records past the end of the batch)
    const bool ok = first < args.n_buffers;
    const uint32_t nb = ok ? (args.n_buffers - first < SLOTS ? args.n_buffers - first : SLOTS) : 0u;
    return make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)(ok ? first : 0u) * buf_bytes, nb * buf_bytes);
  };
  typename S::reg_t raw[16];  // the NEXT chunk's elements 256 a + tid, fetched while this one is transformed
  auto load_group = [&](const __amdgpu_buffer_rsrc_t &r, int a_lo, int a_hi) {
#pragma unroll
    for (int a = 0; a < 16; a++)
      if (a >= a_lo && a < a_hi) raw[a] = S::template load<AUX_LD>(r, tid, a);
  };
  uint32_t first = blockIdx.x * SLOTS;
  load_group(in_rsrc(first), 0, 16);

  cf tw1[16];
  if constexpr (R > 1) {
#pragma unroll
    for (int p = 1; p < 16; p++) tw1[p] = from_v2f(args.tw1_table[(p - 1) * R + t]);  // W_N^(c p)
  }
  float win[16];
#pragma unroll
  for (int a = 0; a < 16; a++) win[a] = args.window[R * a + t] * args.scale;
  auto joff_of = [](int o) -> uint32_t { return R * ((uint32_t)o / R) + 16u * ((uint32_t)o % R); };  // output o is bin t + joff_of(o)
  uint32_t keepmask = 0;
  if (HITS) {
#pragma unroll
    for (int o = 0; o < 16; o++) {
      const uint32_t j = t + joff_of(o);
      const uint32_t i = j ^ (N / 2);  // (j + N/2) % N, process.cpp:47
      const bool keep = !(j < args.dc_ignore || (N - j) < args.dc_ignore) && !(i < args.i_lo || i > args.i_hi);
      keepmask |= keep ? (1u << o) : 0u;
    }
  }

  for (; first < args.n_buffers; first += gridDim.x * SLOTS) {
    const uint32_t buf = first + slot;
    const bool valid = buf < args.n_buffers;
    // ---- the chunk into the raw tile, this thread's 16 samples out of it ----
#pragma unroll
    for (int a = 0; a < 16; a++) S::stage(smem_raw, tid, a, raw[a]);
    __syncthreads();  // barrier 1: raw tile complete
    typename L::raw_t mine[16];
#pragma unroll
    for (int a = 0; a < 16; a++) mine[a] = S::gather(smem_raw, slot, R * (uint32_t)a + t);
    const __amdgpu_buffer_rsrc_t rn = in_rsrc(first + gridDim.x * SLOTS);
    load_group(rn, 0, 8);
    int dc_re = 0, dc_im = 0;
    if (DC) {  // integer mean with the reference's int32 /= uint32 quirk (utility.cpp:77-78), summed over the R lanes of this buffer
      int sr = 0, si = 0;
#pragma unroll
      for (int a = 0; a < 16; a++) {
        int re, im;
        L::ints(mine[a], re, im);
        sr += re;
        si += im;
      }
#pragma unroll
      for (uint32_t off = R / 2; off > 0; off >>= 1) {
        sr += __shfl_xor(sr, (int)off, 64);
        si += __shfl_xor(si, (int)off, 64);
      }
      dc_re = (int)((uint32_t)sr / N);
      dc_im = (int)((uint32_t)si / N);
    }
    cf v[16];
#pragma unroll
    for (int a = 0; a < 16; a++) v[a] = L::conv(mine[a], dc_re, dc_im, 1.0f) * win[a];

    // ---- pass 1 ----
    fft16(v);
    cf x[16];  // output o = R m + r
    __syncthreads();  // barrier 2: every gather done, the region becomes the exchange (R = 1: the dB tile)
    if constexpr (R == 1) {
#pragma unroll
      for (int o = 0; o < 16; o++) x[o] = v[OUT16(o)];
      load_group(rn, 8, 16);
    } else {
#pragma unroll
      for (int p = 0; p < 16; p++) {
        cf y = v[OUT16(p)];
        if (p) y = cmul(y, tw1[p]);
        lds[p * P1 + t] = to_v2f(y);
      }
      __syncthreads();  // barrier 3
      // ---- pass 2: R-point DFTs over c, for p = t + R m ----
#pragma unroll
      for (int m = 0; m < 16 / R; m++)
#pragma unroll
        for (int c = 0; c < R; c++) x[m * R + c] = from_v2f(lds[(t + R * m) * P1 + c]);
      load_group(rn, 8, 16);
#pragma unroll
      for (int m = 0; m < 16 / R; m++) {
        if constexpr (R == 2) {
          const cf a = x[2 * m], b = x[2 * m + 1];
          x[2 * m] = a + b;
          x[2 * m + 1] = a - b;
        } else if constexpr (R == 4) {
          radix4(x[4 * m], x[4 * m + 1], x[4 * m + 2], x[4 * m + 3]);
        } else {
          cf z[8];
#pragma unroll
          for (int c = 0; c < 8; c++) z[c] = x[8 * m + c];
          fft8(z);
#pragma unroll
          for (int r = 0; r < 8; r++) x[8 * m + r] = z[OUT8(r)];
        }
      }
      if constexpr (SPEC) __syncthreads();  // barrier 4: every exchange read done, the region becomes the dB tile
    }
    // ---- K4, K5 (scn_small_db_store_record); the dB values into the tile, then out as whole lines ----
    float *const my_tile = tile_out + slot * PITCH + t;
    scn_small_db_store_record<R, SPEC, HITS>(
        [&](int o) -> float { return power_of(x[o]); }, joff_of,
        [&](int o, float d, bool pred) {
          if (pred) my_tile[joff_of(o)] = d;
        },
        args, buf, valid, keepmask, t, N);
    if constexpr (SPEC) {
      __syncthreads();  // barrier 5: dB tile complete
      const uint32_t nvalid = args.n_buffers - first < SLOTS ? args.n_buffers - first : SLOTS;
      __amdgpu_buffer_rsrc_t rout = make_rsrc(args.power_db + (size_t)first * N, args.power_db ? nvalid * 4u * N : 0u);
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const uint32_t e = 256u * (uint32_t)u + tid;  // bin e of the chunk: 256 consecutive bytes per wave instruction
        const float d = tile_out[(e / N) * PITCH + e % N];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d), rout, tid * 4u, 1024u * (uint32_t)u, AUX_ST);
      }
    }
    __syncthreads();  // barrier 6: the region is free for the next chunk
  }
}

// ------------------------------------------------------------------------------------
// 8192 points: ONE workgroup of 256 threads per buffer, 32 points per thread (scn_fft8k_kernel).
//
// The decomposition of the kernel above with M = 32 (n = 512a + 32b + c, k = p + 16q + 256r, passes 16 x 16 x 32), but every
// thread plays two of the 512 "virtual threads" of passes 1 and 2 and owns one whole 32-point DFT in pass 3 (in registers:
// two 16-point DFTs + one radix-2 step, no cross-lane exchange).
// Why: with 512 threads a workgroup is 8 waves, two workgroups per CU are 4 waves per SIMD = 128 VGPRs,
// which leaves no registers to prefetch the next buffer -- the stamp profile of that form showed the load
// latency fully exposed (19 % waiting for samples, 20 % at barrier 1, 13 % at barrier 4).  256 threads x
// 2 workgroups per CU (LDS: 2 x 70 KiB) are 2 waves per SIMD = 256 VGPRs: room for the next buffer's
// 32 samples per thread, fetched in three groups spread over the passes like the smaller sizes.
// LDS layouts: L1(p, tau) = 512 p + tau, L2(c, kl) = 257 c + kl (all four access patterns conflict-free).
// In pass 1 a thread plays the NEIGHBOURS tau = 2t and 2t + 1: their inputs 512a + 2t (+1) are adjacent in memory and
// their exchange-1 slots adjacent in LDS, so a buffer is fetched in 16 loads of two samples (16 / 8 / 4 bytes per lane)
// and exchange 1 is written in 16 ds_write_b128; in pass 2 it plays (p, c) and (p + 8, c).
// ------------------------------------------------------------------------------------
namespace {
struct Geo8k {
  static constexpr uint32_t N = 8192, T = 256, TV = 512, P1 = TV, P2 = 257;
  static constexpr uint32_t EXCH = (16u * P1 > 32u * P2) ? 16u * P1 : 32u * P2;  // slots
  static constexpr uint32_t LDS_BYTES = EXCH * 8u + TV * 8u + 16u * 4u + 2u * 4u + 8u;
  static constexpr uint32_t WG_PER_CU = 2;  // 70 KiB of LDS each
};
typedef float v32f __attribute__((ext_vector_type(32)));

template <int KIND, bool DC, bool HITS, bool SPEC>
__device__ __forceinline__ void scn_fft8k_body(const ScnFftArgs &args) {
  static_assert(HITS || SPEC, "a kernel that reports nothing");
  typedef Geo8k G;
  constexpr int AUX_LD = SCN_AUX_LD;
  constexpr int AUX_ST = SCN_AUX_ST;
  constexpr uint32_t N = G::N, T = G::T, TV = G::TV, P1 = G::P1, P2 = G::P2;
  constexpr bool DYN = scn_uses_queue(KIND, N);
  typedef RawLoader<KIND> L;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v2f *lds = reinterpret_cast<v2f *>(smem_raw);
  v2f *lds_tw2 = lds + G::EXCH;                              // [16][32]: W_512^(c q) at 32 q + c
  int *lds_cnt = reinterpret_cast<int *>(lds_tw2 + TV);     // [16] DC-sum scratch (re[8], im[8])
  int *lds_hits = lds_cnt + 16;                             // [2] hit counters, alternating per buffer
  uint32_t *lds_next = reinterpret_cast<uint32_t *>(lds_hits + 2);  // [1] the buffer this workgroup takes after the next one

  const uint32_t t = threadIdx.x;
  const uint32_t lane = t & 63, wave = t >> 6;
  const uint32_t c2 = t % 32u, p2 = t / 32u;   // pass-2 identities of this thread: (p2, c2) and (p2 + 8, c2), p2 < 8
  const uint32_t tau0 = 2u * t;                // pass-1 identities: tau0 and tau0 + 1

  // first buffer's samples first: raw[2a + h] = x[512 a + tau0 + h]
  typename L::raw_t raw[32];
  auto load_group = [&](const __amdgpu_buffer_rsrc_t &r, int a_lo, int a_hi) {  // inputs a_lo .. a_hi - 1 of both virtual threads
#pragma unroll
    for (int a = 0; a < 16; a++)
      if (a >= a_lo && a < a_hi) L::template load2<AUX_LD>(r, N, t, TV * a, raw[2 * a], raw[2 * a + 1]);
  };
  if (blockIdx.x < args.n_buffers) {
    __amdgpu_buffer_rsrc_t r0 = make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)blockIdx.x * L::kBufBytes(N), L::kBufBytes(N));
    load_group(r0, 0, 16);
  }
  // persistent constants: pass-1 twiddles of both virtual threads (table rows of TV), window taps
  cf tw1a[16], tw1b[16];
#pragma unroll
  for (int p = 1; p < 16; p++) {
    tw1a[p] = from_v2f(args.tw1_table[(p - 1) * TV + tau0]);
    tw1b[p] = from_v2f(args.tw1_table[(p - 1) * TV + tau0 + 1u]);
  }
  float win[32];
#pragma unroll
  for (int a = 0; a < 16; a++) {
    win[2 * a] = args.window[TV * a + tau0] * args.scale;
    win[2 * a + 1] = args.window[TV * a + tau0 + 1u] * args.scale;
  }
  // pass-2 twiddles W_512^(c q) = W_N^(16 c q), entry 32 q + c: this thread fills entries t (q = p2) and t + T (q = p2 + 8)
  lds_tw2[t] = args.twiddle[(16u * p2 * c2) & (N - 1)];
  lds_tw2[t + T] = args.twiddle[(16u * (p2 + 8u) * c2) & (N - 1)];
  SCN_WORK_QUEUE_SETUP();
  if (t == 0) {
    lds_hits[0] = lds_hits[1] = 0;
    lds_next[0] = DYN ? wq_buffer(wq_take()) : blockIdx.x + gridDim.x;
  }
  __syncthreads();

  v2f *w1 = lds + tau0;                    // + p*P1: slots tau0, tau0 + 1
  v2f *r1 = lds + p2 * P1 + c2;            // + 32 b (+ 8*P1 for the second)
  v2f *w2 = lds + c2 * P2 + p2;            // + 16*q (+ 8 for the second)
  v2f *r3 = lds + t;                       // + c*P2
  const v2f *tw2 = lds_tw2 + c2;           // + 32 q
  const uint32_t st_voff = t * 4u;         // output r of this thread is bin j = t + 256 r

  uint32_t keepmask = 0;  // K5 mask of this thread's 32 bins (process.cpp:46-52)
  if (HITS) {
#pragma unroll
    for (int r = 0; r < 32; r++) {
      const uint32_t j = t + 256u * r;
      const uint32_t i = j ^ (N / 2);  // (j + N/2) % N, process.cpp:47
      const bool keep = !(j < args.dc_ignore || (N - j) < args.dc_ignore) && !(i < args.i_lo || i > args.i_hi);
      keepmask |= keep ? (1u << r) : 0u;
    }
  }
  uint32_t par = 0;
  uint32_t prev = 0xffffffffu;  // the buffer whose recorders may still be running

  uint32_t buf = blockIdx.x;
  uint32_t nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_next[0]);
  while (buf < args.n_buffers) {
    const bool more = nxt < args.n_buffers;
    // ---- K1 + K2 ----
    int dc_re = 0, dc_im = 0;
    if (DC) {
      int sr = 0, si = 0;
#pragma unroll
      for (int a = 0; a < 32; a++) {
        int re, im;
        L::ints(raw[a], re, im);
        sr += re;
        si += im;
      }
      sr = wave_sum(sr);
      si = wave_sum(si);
      if (lane == 0) {
        lds_cnt[wave] = sr;
        lds_cnt[8 + wave] = si;
      }
      __syncthreads();
      sr = si = 0;
#pragma unroll
      for (uint32_t w = 0; w < T / 64u; w++) {
        sr += lds_cnt[w];
        si += lds_cnt[8 + w];
      }
      dc_re = (int)((uint32_t)sr / N);  // int32 /= uint32, utility.cpp:77-78
      dc_im = (int)((uint32_t)si / N);
    }
    cf va[16], vb[16];
#pragma unroll
    for (int a = 0; a < 16; a++) {
      va[a] = L::conv(raw[2 * a], dc_re, dc_im, 1.0f) * win[2 * a];
      vb[a] = L::conv(raw[2 * a + 1], dc_re, dc_im, 1.0f) * win[2 * a + 1];
    }
    // take the buffer after the next one (behind the convert, see scn_fft_kernel); needed at the end of the iteration
    uint32_t taken = 0;
    if (DYN && t == 0 && more) taken = wq_take();
    // next buffer of this workgroup, branch-free (zero records past the end), in three groups
    const __amdgpu_buffer_rsrc_t rn =
        make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)(more ? nxt : buf) * L::kBufBytes(N), more ? L::kBufBytes(N) : 0u);
    load_group(rn, 0, 6);

    // ---- pass 1: virtual threads tau0 and tau0 + 1 ----
    fft16(va);
    fft16(vb);
#pragma unroll
    for (int p = 0; p < 16; p++) {
      cf ya = va[OUT16(p)], yb = vb[OUT16(p)];
      if (p) {
        ya = cmul(ya, tw1a[p]);
        yb = cmul(yb, tw1b[p]);
      }
      typedef float v4f_t __attribute__((ext_vector_type(4)));
      *reinterpret_cast<v4f_t *>(w1 + p * P1) = v4f_t{ya.x, ya.y, yb.x, yb.y};  // slots tau0, tau0 + 1: 16-byte aligned (P1 even)
    }
    __syncthreads();  // barrier 1
    if (HITS) {
      if (t == 0 && prev != 0xffffffffu) {  // every wave is past the barrier: the previous buffer's recorders are done
        args.per_buffer_hits[prev] = (uint32_t)lds_hits[par ^ 1];
        if (args.host_hits) args.host_hits[prev] = (uint32_t)lds_hits[par ^ 1];
        lds_hits[par ^ 1] = 0;
      }
    }
    load_group(rn, 6, 11);

    // ---- pass 2: virtual threads (p2, c2) and (p2 + 8, c2) ----
#pragma unroll
    for (int b = 0; b < 16; b++) {
      va[b] = from_v2f(r1[b * 32]);
      vb[b] = from_v2f(r1[b * 32 + 8 * P1]);
    }
    fft16(va);
    fft16(vb);
#pragma unroll
    for (int q = 1; q < 16; q++) {
      const cf w = from_v2f(tw2[q * 32]);
      va[OUT16(q)] = cmul(va[OUT16(q)], w);
      vb[OUT16(q)] = cmul(vb[OUT16(q)], w);
    }
    __syncthreads();  // barrier 2: every exchange-1 read done before the area is re-used
    load_group(rn, 11, 16);
#pragma unroll
    for (int q = 0; q < 16; q++) {
      w2[q * 16] = to_v2f(va[OUT16(q)]);
      w2[q * 16 + 8] = to_v2f(vb[OUT16(q)]);
    }
    __syncthreads();  // barrier 3

    // ---- pass 3: one 32-point DFT over c per thread: even c -> va, odd c -> vb.  K4: the thread's 32 LINEAR powers stay in
    // `pw` (bin j = t + 256 r at index r) for the hit path; the spectrum gets the dB map of scn_device.h -- product form
    // inline, exact form stored over it for the strong bins in the waves that hold one (see scn_fft_kernel) ----
    v32f pw;
    float gmax[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // the largest power of each group of eight outputs (max ignores NaN)
    __amdgpu_buffer_rsrc_t rout = make_rsrc(args.power_db + (size_t)buf * N, (SPEC && args.power_db) ? 4u * N : 0u);
#pragma unroll
    for (int c = 0; c < 16; c++) {
      va[c] = from_v2f(r3[(2 * c) * P2]);
      vb[c] = from_v2f(r3[(2 * c + 1) * P2]);
    }
    fft16(va);
    fft16(vb);
#pragma unroll
    for (int r = 0; r < 16; r++) {  // X[r], X[r + 16] = E[r] +- W_32^r O[r]
      const cf ev = va[OUT16(r)];
      cf od = vb[OUT16(r)];
      if (r) {
        const float cr = (float)__builtin_cos(6.283185307179586476925286766559 * r / 32.0);
        const float sr = (float)__builtin_sin(6.283185307179586476925286766559 * r / 32.0);
        od = cmul(od, cf{cr, -sr});
      }
      const cf x0 = ev + od, x1 = ev - od;
      const float p0 = power_of(x0), p1 = power_of(x1);
      pw[r] = p0;
      pw[r + 16] = p1;
      gmax[r >> 3] = fmaxf(gmax[r >> 3], p0);
      gmax[2 + (r >> 3)] = fmaxf(gmax[2 + (r >> 3)], p1);
      if constexpr (SPEC) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, db_fast(p0)), rout, st_voff, 1024u * r, AUX_ST);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, db_fast(p1)), rout, st_voff, 1024u * (r + 16), AUX_ST);
      }
    }

    const float pmax = fmaxf(fmaxf(gmax[0], gmax[1]), fmaxf(gmax[2], gmax[3]));  // pre-filter of the hit path and of the exact half
    if constexpr (SPEC) {
      if (__ballot(pmax >= SCN_P_EXACT_FROM)) {
#pragma unroll
        for (int g = 0; g < 4; g++) {  // groups of eight outputs first, then the outputs of a group that holds a strong bin (see scn_fft_kernel)
          if (__ballot(gmax[g] >= SCN_P_EXACT_FROM)) {
#pragma unroll
            for (int r = 8 * g; r < 8 * g + 8; r++) {
              const float q = pw[r];
              if (__ballot(q >= SCN_P_EXACT_FROM)) {
                const float d = db_exact(q);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d), rout, q >= SCN_P_EXACT_FROM ? st_voff : 0x80000000u, 1024u * r, AUX_ST);
              }
            }
          }
        }
      }
    }
    if (t == 0) lds_next[0] = !more ? 0xffffffffu : DYN ? wq_buffer(taken) : nxt + gridDim.x;
    __syncthreads();  // barrier 4: exchange area free again; lds_next visible
    const uint32_t after = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_next[0]);
    if (HITS) {
      if (__ballot(pmax > args.p_lo))
        scn_record_hits_lanes<32, false, true>(pw, gmax, keepmask, args, &lds_hits[par], buf, lane, [&](int r) -> uint32_t { return (t + 256u * (uint32_t)r) ^ (N / 2); });
      prev = buf;
      par ^= 1;
    }
    buf = nxt;
    nxt = after;
  }
  if (HITS) {
    __syncthreads();  // last buffer's recorders done
    if (t == 0 && prev != 0xffffffffu) {
      args.per_buffer_hits[prev] = (uint32_t)lds_hits[par ^ 1];
      if (args.host_hits) args.host_hits[prev] = (uint32_t)lds_hits[par ^ 1];
    }
  }
}

}  // namespace

template <int KIND, bool DC, bool HITS, bool SPEC>
__global__ __launch_bounds__(256, 2) void scn_fft8k_kernel(ScnFftArgs args) {
  scn_fft8k_body<KIND, DC, HITS, SPEC>(args);
}


// ------------------------------------------------------------------------------------
// 16384 points as 32 x 16 x 32: 512 threads x 32 points, one workgroup per CU (scn_fft16k2_kernel).
//
//   n = 512 a + 32 b + c          k = p + 32 q + 512 r          a, p, c, r in [0, 32)   b, q in [0, 16)
//   pass 1  thread tau = 32 b + c:  32-pt DFT over a of x[512 a + tau] w[..] (two 16-pt DFTs + one radix-2 step, in
//           registers) -> * W_N^(tau p) -> LDS row p                                    L1(p, tau) = 512 p + tau
//   pass 2  threads (p, c) and (p + 16, c):  16-pt DFT over b -> * W_512^(c q) -> LDS    L2(c, kl) = 513 c + kl, kl = p + 32 q
//   pass 3  thread kl:  32-pt DFT over c -> X[kl + 512 r], r = 0 .. 31 -- a WHOLE DFT per thread
//
// The first form (16 x 16 x 64, rounds 1-2; profiles/r03_experiments.md section 2) shared its 64-point last pass between two
// lanes: a twiddle multiply of every value by W_64^r', a v_permlane32_swap per register and a combining step -- 280 of its
// ~1700 VALU operations per thread and buffer.  Here pass 1 takes the extra radix-2 level instead (one more twiddle-free step
// on values that are in registers anyway): ~150 operations fewer, no cross-lane traffic, every output bin of a lane 512 apart.  All four LDS access
// patterns are conflict-free in the 16-lane groups a ds_*_b64 is served in (row pitch 513 on the transposed side).
// Price: one virtual thread per lane in pass 1, so the samples arrive one per load (32 loads of 8 / 4 / 2 bytes per
// lane instead of 16 of twice that).
// Pass 3 runs in DOUBLE (decimation in frequency, 16 double values live at a time): the parity metric's tail at this size
// is the float rounding of a strong tone's partial sums in the LAST pass, which lands on the 31 other bins of the tone's
// column -- any float32 FFT shows it (scripts/emul_fused.py) -- and goes away 3x when the last five radix-2 levels are exact.
// v_add_f64 / v_fma_f64 issue at 0.85x / 0.7x the float rate (scripts/ubench/f64_rate.hip).  Measured on MI355X, 3072
// strong-tone buffers per run (scripts/acc16k.py) and us per 2048-buffer launch, float pass 3 -> double: parity metric max
// 4.9e-6 .. 7.5e-6 -> 2.9e-6 .. 4.0e-6, p99 4.0e-6 -> 2.2e-6, median 1.11e-6 -> 0.88e-6 (the float quantisation of the dB value
// itself); cfloat 93.5 -> 100.2 us, int16 75.6 -> 88.1.  Parity comes first, so double is the product.
// ------------------------------------------------------------------------------------
namespace {
struct Geo16k2 {
  static constexpr uint32_t N = 16384, T = 512, P1 = 512, P2 = 513;
  static constexpr uint32_t EXCH = 32u * P2;  // slots (exchange 1 needs 32 * 512)
  static constexpr uint32_t LDS_BYTES = EXCH * 8u + T * 8u + 16u * 4u + 2u * 4u + 8u;
  static constexpr uint32_t WG_PER_CU = 1;
};


template <int KIND, bool DC, bool HITS, bool SPEC>
__device__ __forceinline__ void scn_fft16k2_body(const ScnFftArgs &args) {
  static_assert(HITS || SPEC, "a kernel that reports nothing");
  typedef Geo16k2 G;
  constexpr int AUX_LD = SCN_AUX_LD;
  constexpr int AUX_ST = SCN_AUX_ST;
  constexpr uint32_t N = G::N, T = G::T, P1 = G::P1, P2 = G::P2;
  typedef RawLoader<KIND> L;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v2f *lds = reinterpret_cast<v2f *>(smem_raw);
  v2f *lds_tw2 = lds + G::EXCH;                             // [16][32]: W_512^(c q) at 32 q + c
  int *lds_cnt = reinterpret_cast<int *>(lds_tw2 + T);     // [16] DC-sum scratch (re[8], im[8])
  int *lds_hits = lds_cnt + 16;                            // [2] hit counters, alternating per buffer

  const uint32_t t = threadIdx.x;
  const uint32_t lane = t & 63, wave = t >> 6;
  const uint32_t c2 = t & 31u, p2 = t >> 5;  // pass 2: (p2, c2) and (p2 + 16, c2), p2 < 16
  // how many of a buffer's 32 loads are prefetched across the previous buffer's passes, in three groups (float input: the
  // registers; planar int16 is two loads and a pack per sample: beside the double-precision pass 3 only half)
  constexpr int PFN = (KIND == SCN_K_FLOAT_COMPLEX || KIND == SCN_K_SHORT) ? 16 : 32;
  constexpr int PF0 = (3 * PFN) / 8, PF1 = (11 * PFN) / 16;

  // first buffer's samples first: raw[a] = x[512 a + t]
  typename L::raw_t raw[32];
  auto load_group = [&](const __amdgpu_buffer_rsrc_t &r, int a_lo, int a_hi) {
#pragma unroll
    for (int a = 0; a < 32; a++)
      if (a >= a_lo && a < a_hi) raw[a] = L::template load<AUX_LD>(r, N, t, 512u * a);
  };
  if (blockIdx.x < args.n_buffers) {
    __amdgpu_buffer_rsrc_t r0 =
        make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)blockIdx.x * L::kBufBytes(N), L::kBufBytes(N));
    load_group(r0, 0, PFN);
  }
  // persistent constants: W_N^(t p) (table rows of 512, scn_tw1_layout), window taps.  Only p = 1 .. 16 are kept and
  // W_N^(t (p' + 16)) = W_N^(t p') W_N^(16 t) is formed per buffer: 15 more complex multiplies (+3 % of the kernel's
  // arithmetic) buy the 30 registers the double-precision values of pass 3 need -- spilled instead, they cost 30 .. 90 %
  // (scratch reloads at the top of every buffer).
  constexpr int TWN = 17;
  cf tw1[TWN];
#pragma unroll
  for (int p = 1; p < TWN; p++) tw1[p] = from_v2f(args.tw1_table[(p - 1) * 512 + t]);
  float win[32];
#pragma unroll
  for (int a = 0; a < 32; a++) win[a] = args.window[512u * a + t] * args.scale;
  lds_tw2[t] = args.twiddle[(32u * p2 * c2) & (N - 1)];  // W_512^(c q) = W_N^(32 c q), entry 32 q + c = t
  if (t == 0) lds_hits[0] = lds_hits[1] = 0;
  __syncthreads();

  v2f *w1 = lds + t;                       // + p*P1
  v2f *r1 = lds + p2 * P1 + c2;            // + 32 b (+ 16*P1 for the second virtual thread)
  v2f *w2 = lds + c2 * P2 + p2;            // + 32 q (+ 16 for the second)
  v2f *r3 = lds + t;                       // + c*P2
  const v2f *tw2 = lds_tw2 + c2;           // + 32 q
  const uint32_t st_voff = t * 4u;         // output o of this thread is bin j = t + 512 o

  uint32_t keepmask = 0;  // K5 mask of this thread's 32 bins (process.cpp:46-52)
  if (HITS) {
#pragma unroll
    for (int r = 0; r < 32; r++) {
      const uint32_t j = t + 512u * r;
      const uint32_t i = j ^ (N / 2);  // (j + N/2) % N, process.cpp:47
      const bool keep = !(j < args.dc_ignore || (N - j) < args.dc_ignore) && !(i < args.i_lo || i > args.i_hi);
      keepmask |= keep ? (1u << r) : 0u;
    }
  }
  uint32_t par = 0;
  uint32_t prev = 0xffffffffu;  // the buffer whose recorders may still be running

  for (uint32_t buf = blockIdx.x; buf < args.n_buffers; buf += gridDim.x) {  // (static assignment: one workgroup per CU)
    const uint32_t nxt = buf + gridDim.x;
    const bool more = nxt < args.n_buffers;
    if (PFN < 32) {  // the part of this buffer that was not prefetched
      const __amdgpu_buffer_rsrc_t rc = make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)buf * L::kBufBytes(N),
                                                  L::kBufBytes(N));
      load_group(rc, PFN, 32);
    }
    // ---- K1 + K2 ----
    int dc_re = 0, dc_im = 0;
    if (DC) {
      int sr = 0, si = 0;
#pragma unroll
      for (int a = 0; a < 32; a++) {
        int re, im;
        L::ints(raw[a], re, im);
        sr += re;
        si += im;
      }
      sr = wave_sum(sr);
      si = wave_sum(si);
      if (lane == 0) {
        lds_cnt[wave] = sr;
        lds_cnt[8 + wave] = si;
      }
      __syncthreads();
      sr = si = 0;
#pragma unroll
      for (uint32_t w = 0; w < T / 64u; w++) {
        sr += lds_cnt[w];
        si += lds_cnt[8 + w];
      }
      dc_re = (int)((uint32_t)sr / N);  // int32 /= uint32, utility.cpp:77-78
      dc_im = (int)((uint32_t)si / N);
    }
    cf va[16], vb[16];  // even / odd a
#pragma unroll
    for (int a = 0; a < 16; a++) {
      va[a] = L::conv(raw[2 * a], dc_re, dc_im, 1.0f) * win[2 * a];
      vb[a] = L::conv(raw[2 * a + 1], dc_re, dc_im, 1.0f) * win[2 * a + 1];
    }
    // next buffer of this workgroup, branch-free (zero records past the end), in three groups
    const __amdgpu_buffer_rsrc_t rn =
        make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)(more ? nxt : buf) * L::kBufBytes(N),
                  more ? L::kBufBytes(N) : 0u);
    load_group(rn, 0, PF0);

    // ---- pass 1: 32-point DFT over a: E, O = DFT16 of the even / odd samples, A[p'] = E + W_32^p' O, A[p' + 16] = E - W_32^p' O ----
    fft16(va);
    fft16(vb);
    cf tw16 = tw1[16];
    asm volatile("" : "+v"(tw16.x), "+v"(tw16.y));  // opaque per buffer: or hipcc hoists the 15 products out of the loop and keeps them in registers
#pragma unroll
    for (int p = 0; p < 16; p++) {
      const cf ev = va[OUT16(p)];
      cf od = vb[OUT16(p)];
      if (p) {
        const float cr = (float)__builtin_cos(6.283185307179586476925286766559 * p / 32.0);
        const float sr = (float)__builtin_sin(6.283185307179586476925286766559 * p / 32.0);
        od = cmul(od, cf{cr, -sr});
      }
      cf y0 = ev + od, y1 = ev - od;
      if (p) y0 = cmul(y0, tw1[p]);
      y1 = cmul(y1, p ? cmul(tw1[p], tw16) : tw16);
      w1[p * P1] = to_v2f(y0);
      w1[(p + 16) * P1] = to_v2f(y1);
    }
    __syncthreads();  // barrier 1
    if (HITS) {
      if (t == 0 && prev != 0xffffffffu) {  // every wave is past the barrier: the previous buffer's recorders are done
        args.per_buffer_hits[prev] = (uint32_t)lds_hits[par ^ 1];
        if (args.host_hits) args.host_hits[prev] = (uint32_t)lds_hits[par ^ 1];
        lds_hits[par ^ 1] = 0;
      }
    }
    load_group(rn, PF0, PF1);

    // ---- pass 2: virtual threads (p2, c2) and (p2 + 16, c2): DFT16 over b, twiddle W_512^(c q) ----
#pragma unroll
    for (int b = 0; b < 16; b++) {
      va[b] = from_v2f(r1[b * 32]);
      vb[b] = from_v2f(r1[b * 32 + 16 * P1]);
    }
    fft16(va);
    fft16(vb);
#pragma unroll
    for (int q = 1; q < 16; q++) {
      const cf w = from_v2f(tw2[q * 32]);
      va[OUT16(q)] = cmul(va[OUT16(q)], w);
      vb[OUT16(q)] = cmul(vb[OUT16(q)], w);
    }
    __syncthreads();  // barrier 2: every exchange-1 read done before the area is re-used
    load_group(rn, PF1, PFN);
#pragma unroll
    for (int q = 0; q < 16; q++) {
      w2[q * 32] = to_v2f(va[OUT16(q)]);
      w2[q * 32 + 16] = to_v2f(vb[OUT16(q)]);
    }
    __syncthreads();  // barrier 3

    // ---- pass 3: the 32-point DFT over c of column kl = t; K4 as in the other kernels (linear powers kept for the hit path,
    //      product-form dB stored inline, exact form stored over it for strong bins in the waves that hold one) ----
    v32f pw;
    float gmax[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t cand = 0;  // (HITS) this lane's candidate outputs, collected as they are produced
    __amdgpu_buffer_rsrc_t rout = make_rsrc(args.power_db + (size_t)buf * N, (SPEC && args.power_db) ? 4u * N : 0u);
    // decimation in frequency: A[c''] = x[c''] + x[c''+16] -> DFT16 -> X[2 rho];  B[c''] = (x[c''] - x[c''+16]) W_32^c'' -> DFT16 -> X[2 rho + 1]
#pragma unroll
    for (int h = 0; h < 2; h++) {
      cd vd[16];
#pragma unroll
      for (int c = 0; c < 16; c++) {
        const cd x0 = to_cd(r3[c * P2]), x1 = to_cd(r3[(c + 16) * P2]);
        if (h == 0) {
          vd[c] = x0 + x1;
        } else {
          const cd dlt = x0 - x1;
          const double cr = __builtin_cos(6.283185307179586476925286766559 * c / 32.0), sr = __builtin_sin(6.283185307179586476925286766559 * c / 32.0);
          vd[c] = c ? cmul_d(dlt, cr, -sr) : dlt;
        }
      }
      fft16_d(vd);
#pragma unroll
      for (int rho = 0; rho < 16; rho++) {
        const int r = 2 * rho + h;
        const cd x = vd[OUT16(rho)];
        const float q = (float)__builtin_fma(x.y, x.y, x.x * x.x);
        gmax[r >> 3] = fmaxf(gmax[r >> 3], q);
        if constexpr (SPEC) {
          const float d = db_fast(q);
          pw[r] = d;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d), rout, st_voff, 2048u * r, AUX_ST);
        } else {
          pw[r] = q;
          if constexpr (HITS) cand |= q > args.p_lo ? (1u << r) : 0u;
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the second half's loads and conversions out of the first half's live range
    }
    const float pmax = fmaxf(fmaxf(gmax[0], gmax[1]), fmaxf(gmax[2], gmax[3]));
    if constexpr (SPEC) {
      if (__ballot(pmax >= SCN_P_EXACT_FROM)) {
#pragma unroll
        for (int g = 0; g < 4; g++) {  // the exact dB value for the strong maximum of each group of eight outputs (see scn_fft_kernel)
          if (__ballot(gmax[g] >= SCN_P_EXACT_FROM)) {
            const float dg = db_fast(gmax[g]), ex = db_exact(gmax[g]);
#pragma unroll
            for (int r = 8 * g; r < 8 * g + 8; r++) {
              const float d = pw[r];
              const bool sel = gmax[g] >= SCN_P_EXACT_FROM && d == dg;
              if (__ballot(sel)) {  // (the maximum sits in ONE output index of a lane: the others skip the store instruction altogether)
                pw[r] = sel ? ex : d;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, ex), rout, sel ? st_voff : 0x80000000u, 2048u * r, AUX_ST);
              }
            }
          }
        }
      }
    }
    __syncthreads();  // barrier 4: exchange area free again
    if (HITS) {
      if (__ballot(pmax > args.p_lo))
        scn_record_hits_lanes<32, SPEC, false, !SPEC>(pw, gmax, keepmask, args, &lds_hits[par], buf, lane, [&](int r) -> uint32_t { return (t + 512u * (uint32_t)r) ^ (N / 2); }, cand);
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

}  // namespace

template <int KIND, bool DC, bool HITS, bool SPEC>
__global__ __launch_bounds__(512, 2) void scn_fft16k2_kernel(ScnFftArgs args) {
  scn_fft16k2_body<KIND, DC, HITS, SPEC>(args);
}

// ------------------------------------------------------------------------------------
// Time-domain mode (process.cpp:203-237): per buffer, max and min over the samples of
// 10*log2(|x|)/log2(10).  The dB map is monotone, so the kernel reduces max/min of |x|^2
// (computed exactly as the reference does: re*re + im*im in float, no fused multiply-add) and
// converts the two extremes once.  One 256-thread workgroup per buffer, any N; pure HBM
// streaming (8/4/2 B per sample in, 8 B per buffer out).
// ------------------------------------------------------------------------------------
template <int KIND, bool DC>
__global__ __launch_bounds__(256) void scn_time_domain_kernel(ScnTdArgs args) {
  typedef RawLoader<KIND> L;
  __shared__ float s_max[4], s_min[4];
  __shared__ int s_sum[8];
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint32_t N = args.n;
  for (uint32_t buf = blockIdx.x; buf < args.n_buffers; buf += gridDim.x) {
    __amdgpu_buffer_rsrc_t rin =
        make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)buf * L::kBufBytes(N), L::kBufBytes(N));
    int dc_re = 0, dc_im = 0;
    if (DC) {
      int sr = 0, si = 0;
      for (uint32_t i = t; i < N; i += 256) {
        int re, im;
        L::ints(L::template load<0>(rin, N, i, 0), re, im);
        sr += re;
        si += im;
      }
      sr = wave_sum(sr);
      si = wave_sum(si);
      if (lane == 0) {
        s_sum[wave] = sr;
        s_sum[4 + wave] = si;
      }
      __syncthreads();
      dc_re = (int)((uint32_t)(s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]) / N);  // utility.cpp:77-78 quirk
      dc_im = (int)((uint32_t)(s_sum[4] + s_sum[5] + s_sum[6] + s_sum[7]) / N);
    }
    float pmax = -1.0f, pmin = 3.40282347e+38f;  // |x|^2 >= 0, so -1 is "no sample yet"
    for (uint32_t i = t; i < N; i += 256) {
      cf x = L::conv(L::template load<0>(rin, N, i, 0), dc_re, dc_im, args.scale);
      float p = __fadd_rn(__fmul_rn(x.x, x.x), __fmul_rn(x.y, x.y));  // process.cpp:220, unfused
      pmax = fmaxf(pmax, p);
      pmin = fminf(pmin, p);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      pmax = fmaxf(pmax, __shfl_xor(pmax, off, 64));
      pmin = fminf(pmin, __shfl_xor(pmin, off, 64));
    }
    if (lane == 0) {
      s_max[wave] = pmax;
      s_min[wave] = pmin;
    }
    __syncthreads();
    if (t == 0) {
      pmax = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
      pmin = fminf(fminf(s_min[0], s_min[1]), fminf(s_min[2], s_min[3]));
      // 10*log2(sqrt(p))/log2(10); the reference's odd initial values (numeric_limits<float>::min()
      // is the smallest POSITIVE float, process.cpp:207-208) bound the results
      const float k = 3.01029995663981195214f;  // 10/log2(10)
      float dmax = k * __builtin_amdgcn_logf(__builtin_amdgcn_sqrtf(pmax));
      float dmin = k * __builtin_amdgcn_logf(__builtin_amdgcn_sqrtf(pmin));
      args.max_db[buf] = fmaxf(1.17549435e-38f, dmax);
      args.min_db[buf] = fminf(3.40282347e+38f, dmin);
    }
    __syncthreads();
  }
}

// Time-domain mode, streaming form: ONE WAVE per buffer, 16-byte loads (2 / 4 / 8 samples per lane per instruction
// for float / int16 / int8), four loads in flight per lane, wave-level reductions only -- no LDS, no barriers.  The
// per-sample form above (one workgroup per buffer, one sample per lane per load) left a pure streaming reduction at
// 46-51 us per 33.5 M int16 samples and 77-93 us at 1024 points; this is the product path whenever N is a multiple
// of 8 (every supported size).  Same arithmetic, sample by sample, through RawLoader<KIND>::ints / conv.
template <int KIND, bool DC>
__global__ __launch_bounds__(256) void scn_time_domain_wave_kernel(ScnTdArgs args) {
  typedef RawLoader<KIND> L;
  typedef int v4i_t __attribute__((__vector_size__(16)));
  constexpr bool PLANAR = KIND == SCN_K_SHORT;
  const uint32_t lane = threadIdx.x & 63u;
  // (readfirstlane: the wave index is wave-uniform, which hipcc cannot see -- without it every load of the buffer's descriptor
  //  sits in a one-trip waterfall loop)
  const uint32_t gw = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = gridDim.x * 4u;
  const uint32_t N = args.n;
  const uint32_t bytes = L::kBufBytes(N);
  const uint32_t stream_bytes = PLANAR ? 2u * N : bytes;  // planar: the I block, then the Q block
  const uint32_t chunks = stream_bytes / 16u;
  for (uint32_t buf = gw; buf < args.n_buffers; buf += nw) {
    const __amdgpu_buffer_rsrc_t rin =
        make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)buf * bytes, bytes);
    // calls f(raw sample) for every sample of the buffer this lane is responsible for
    auto for_each_sample = [&](auto &&f) {
      for (uint32_t c0 = 0; c0 < chunks; c0 += 256u) {
        v4i_t w[4], wq[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t c = c0 + 64u * u + lane;
          w[u] = __builtin_bit_cast(v4i_t, __builtin_amdgcn_raw_buffer_load_b128(rin, c * 16u, 0, SCN_AUX_LD));
          if (PLANAR) wq[u] = __builtin_bit_cast(v4i_t, __builtin_amdgcn_raw_buffer_load_b128(rin, c * 16u, stream_bytes, SCN_AUX_LD));
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (c0 + 64u * u + lane < chunks) {  // lanes past the end of the (I) stream hold nothing of this buffer
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const int d = w[u][k];
              if constexpr (KIND == SCN_K_FLOAT_COMPLEX) {
                if (k & 1) f(v2f{__builtin_bit_cast(float, (int)w[u][k - 1]), __builtin_bit_cast(float, d)});
              } else if constexpr (KIND == SCN_K_SHORT_COMPLEX) {
                f(d);
              } else if constexpr (KIND == SCN_K_BYTE_COMPLEX) {
                f(d & 0xffff);
                f((int)((uint32_t)d >> 16));
              } else {  // planar int16: two I values in d, the two Q values of the same samples in wq
                const int q = wq[u][k];
                f((d & 0xffff) | (q << 16));
                f((int)((uint32_t)d >> 16) | (q & (int)0xffff0000));
              }
            }
          }
        }
      }
    };
    int dc_re = 0, dc_im = 0;
    if (DC) {
      int sr = 0, si = 0;
      for_each_sample([&](typename L::raw_t r) {
        int re, im;
        L::ints(r, re, im);
        sr += re;
        si += im;
      });
      sr = wave_sum(sr);
      si = wave_sum(si);
      dc_re = (int)((uint32_t)sr / N);  // utility.cpp:77-78 quirk
      dc_im = (int)((uint32_t)si / N);
    }
    float pmax = -1.0f, pmin = 3.40282347e+38f;  // |x|^2 >= 0, so -1 is "no sample yet"
    for_each_sample([&](typename L::raw_t r) {
      const cf x = L::conv(r, dc_re, dc_im, args.scale);
      const float p = __fadd_rn(__fmul_rn(x.x, x.x), __fmul_rn(x.y, x.y));  // process.cpp:220, unfused
      pmax = fmaxf(pmax, p);
      pmin = fminf(pmin, p);
    });
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      pmax = fmaxf(pmax, __shfl_xor(pmax, off, 64));
      pmin = fminf(pmin, __shfl_xor(pmin, off, 64));
    }
    if (lane == 0) {
      // 10*log2(sqrt(p))/log2(10); the reference's odd initial values bound the results (process.cpp:207-208)
      const float k = 3.01029995663981195214f;  // 10/log2(10)
      const float dmax = k * __builtin_amdgcn_logf(__builtin_amdgcn_sqrtf(pmax));
      const float dmin = k * __builtin_amdgcn_logf(__builtin_amdgcn_sqrtf(pmin));
      args.max_db[buf] = fmaxf(1.17549435e-38f, dmax);
      args.min_db[buf] = fminf(3.40282347e+38f, dmin);
    }
  }
}

#if SCN_IN_TU(7)
hipError_t scn_launch_time_domain(int kind, bool dc, const ScnTdArgs &a, int num_cus, hipStream_t s) {
  if (a.n_buffers == 0) return hipSuccess;
  void (*k)(ScnTdArgs) = nullptr;
  switch (kind) {
    case SCN_K_FLOAT_COMPLEX: k = scn_time_domain_kernel<SCN_K_FLOAT_COMPLEX, false>; break;
    case SCN_K_SHORT_COMPLEX: k = dc ? scn_time_domain_kernel<SCN_K_SHORT_COMPLEX, true> : scn_time_domain_kernel<SCN_K_SHORT_COMPLEX, false>; break;
    case SCN_K_SHORT: k = dc ? scn_time_domain_kernel<SCN_K_SHORT, true> : scn_time_domain_kernel<SCN_K_SHORT, false>; break;
    case SCN_K_BYTE_COMPLEX: k = dc ? scn_time_domain_kernel<SCN_K_BYTE_COMPLEX, true> : scn_time_domain_kernel<SCN_K_BYTE_COMPLEX, false>; break;
    default: return hipErrorInvalidValue;
  }
  if (a.n % 8u == 0) {  // one wave per buffer, 16-byte loads
    switch (kind) {
      case SCN_K_FLOAT_COMPLEX: k = scn_time_domain_wave_kernel<SCN_K_FLOAT_COMPLEX, false>; break;
      case SCN_K_SHORT_COMPLEX: k = dc ? scn_time_domain_wave_kernel<SCN_K_SHORT_COMPLEX, true> : scn_time_domain_wave_kernel<SCN_K_SHORT_COMPLEX, false>; break;
      case SCN_K_SHORT: k = dc ? scn_time_domain_wave_kernel<SCN_K_SHORT, true> : scn_time_domain_wave_kernel<SCN_K_SHORT, false>; break;
      default: k = dc ? scn_time_domain_wave_kernel<SCN_K_BYTE_COMPLEX, true> : scn_time_domain_wave_kernel<SCN_K_BYTE_COMPLEX, false>; break;
    }
    uint32_t blocks = (a.n_buffers + 3u) / 4u;  // four waves (buffers) per block
    const uint32_t resident = (uint32_t)num_cus * 8u;
    if (blocks > resident) blocks = resident;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
  }
  int grid = num_cus * 8;
  if ((uint32_t)grid > a.n_buffers) grid = (int)a.n_buffers;
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}
#endif  // SCN_IN_TU(7)

// ------------------------------------------------------------------------------------
// K1 alone: raw wire format -> complex float (utility.cpp:9-84), for the triggered-capture path
// (messageQueue.h:98-139 writes fftwf_complex[N] records).  One 256-thread workgroup per buffer.
// ------------------------------------------------------------------------------------
template <int KIND, bool DC>
__global__ __launch_bounds__(256) void scn_convert_kernel(const void *raw, scn_v2f *out, uint32_t n, uint32_t n_buffers,
                                                          float scale) {
  typedef RawLoader<KIND> L;
  __shared__ int s_sum[8];
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (uint32_t buf = blockIdx.x; buf < n_buffers; buf += gridDim.x) {
    __amdgpu_buffer_rsrc_t rin = make_rsrc(reinterpret_cast<const char *>(raw) + (size_t)buf * L::kBufBytes(n), L::kBufBytes(n));
    int dc_re = 0, dc_im = 0;
    if (DC) {
      int sr = 0, si = 0;
      for (uint32_t i = t; i < n; i += 256) {
        int re, im;
        L::ints(L::template load<0>(rin, n, i, 0), re, im);
        sr += re;
        si += im;
      }
      sr = wave_sum(sr);
      si = wave_sum(si);
      if (lane == 0) {
        s_sum[wave] = sr;
        s_sum[4 + wave] = si;
      }
      __syncthreads();
      dc_re = (int)((uint32_t)(s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]) / n);  // utility.cpp:77-78 quirk
      dc_im = (int)((uint32_t)(s_sum[4] + s_sum[5] + s_sum[6] + s_sum[7]) / n);
    }
    for (uint32_t i = t; i < n; i += 256)
      out[(size_t)buf * n + i] = to_v2f(L::conv(L::template load<0>(rin, n, i, 0), dc_re, dc_im, scale));
    __syncthreads();
  }
}

#if SCN_IN_TU(7)
hipError_t scn_launch_convert(int kind, bool dc, const void *raw, scn_v2f *out, uint32_t n, uint32_t n_buffers, float scale,
                              hipStream_t s) {
  if (n_buffers == 0) return hipSuccess;
  void (*k)(const void *, scn_v2f *, uint32_t, uint32_t, float) = nullptr;
  switch (kind) {
    case SCN_K_FLOAT_COMPLEX: k = scn_convert_kernel<SCN_K_FLOAT_COMPLEX, false>; break;
    case SCN_K_SHORT_COMPLEX: k = dc ? scn_convert_kernel<SCN_K_SHORT_COMPLEX, true> : scn_convert_kernel<SCN_K_SHORT_COMPLEX, false>; break;
    case SCN_K_SHORT: k = dc ? scn_convert_kernel<SCN_K_SHORT, true> : scn_convert_kernel<SCN_K_SHORT, false>; break;
    case SCN_K_BYTE_COMPLEX: k = dc ? scn_convert_kernel<SCN_K_BYTE_COMPLEX, true> : scn_convert_kernel<SCN_K_BYTE_COMPLEX, false>; break;
    default: return hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(k, dim3(n_buffers < 2048 ? n_buffers : 2048), dim3(256), 0, s, raw, out, n, n_buffers, scale);
  return hipGetLastError();
}
#endif  // SCN_IN_TU(7)

// ------------------------------------------------------------------------------------
// host-side launcher
// ------------------------------------------------------------------------------------
// `stop` (may be null): an event completed by the kernel's own dispatch packet (hipExtLaunchKernel) instead of a marker
// packet of its own behind the kernel.
template <typename K>
static hipError_t launch_with_stop(K k, int grid, uint32_t threads, uint32_t lds, hipStream_t s, hipEvent_t stop, const ScnFftArgs &a) {
  if (stop) hipExtLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, s, nullptr, stop, 0, a);
  else hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, s, a);
  return hipGetLastError();
}

// Output mode of a launch: which of the three kernels of a (size, wire format, DC) combination runs
//   hits && spec: spectrum + hits;   !hits: spectrum only;   hits && !spec: hits only (no stores, no per-bin logarithm)
template <template <bool, bool, bool> class K>
static void (*pick_mode(bool dc, bool hits, bool spec))(ScnFftArgs) {
  if (!hits) return dc ? K<true, false, true>::fn : K<false, false, true>::fn;
  if (spec) return dc ? K<true, true, true>::fn : K<false, true, true>::fn;
  return dc ? K<true, true, false>::fn : K<false, true, false>::fn;
}
// the kernel families: FAMILY<KIND>::T<DC, HITS, SPEC>::fn, with the family's geometry (threads, LDS, workgroups per CU,
// buffers per workgroup iteration)
template <int M>
struct NarrowFamily {
  typedef Geo<M> G;
  static constexpr uint32_t THREADS = G::T, SLOTS = 1;
  template <int KIND>
  struct K {
    template <bool DC, bool HITS, bool SPEC>
    struct T {
      static constexpr void (*fn)(ScnFftArgs) = scn_fft_kernel<M, KIND, DC, HITS, SPEC>;
    };
  };
};
template <int M>
struct SmallFamily {
  typedef GeoSmall<M> G;
  static constexpr uint32_t THREADS = 256, SLOTS = G::SLOTS;
  template <int KIND>
  struct K {
    template <bool DC, bool HITS, bool SPEC>
    struct T {
      static constexpr void (*fn)(ScnFftArgs) = scn_fft_small_kernel<M, KIND, DC, HITS, SPEC>;
    };
  };
};
template <int R>
struct TinyFamily {
  typedef GeoTiny<R> G;
  static constexpr uint32_t THREADS = 256, SLOTS = G::SLOTS;
  template <int KIND>
  struct K {
    template <bool DC, bool HITS, bool SPEC>
    struct T {
      static constexpr void (*fn)(ScnFftArgs) = scn_fft_tiny_kernel<R, KIND, DC, HITS, SPEC>;
    };
  };
};
struct Family8k {
  typedef Geo8k G;
  static constexpr uint32_t THREADS = G::T, SLOTS = 1;
  template <int KIND>
  struct K {
    template <bool DC, bool HITS, bool SPEC>
    struct T {
      static constexpr void (*fn)(ScnFftArgs) = scn_fft8k_kernel<KIND, DC, HITS, SPEC>;
    };
  };
};
struct Family16k {
  typedef Geo16k2 G;
  static constexpr uint32_t THREADS = G::T, SLOTS = 1;
  template <int KIND>
  struct K {
    template <bool DC, bool HITS, bool SPEC>
    struct T {
      static constexpr void (*fn)(ScnFftArgs) = scn_fft16k2_kernel<KIND, DC, HITS, SPEC>;
    };
  };
};

template <class F, int KIND>
static hipError_t launch_kind(const ScnFftArgs &a, bool dc, bool hits, bool spec, int num_cus, hipStream_t s, hipEvent_t stop) {
  typedef typename F::G G;
  void (*k)(ScnFftArgs) = pick_mode<F::template K<KIND>::template T>(dc, hits, spec);
  if (G::LDS_BYTES > 65536u) {
    // > 64 KiB of dynamic LDS needs the opt-in; it is per function AND per device, and a process may
    // drive several GPUs (one plan per consumer thread), so it is simply set on every launch
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (e != hipSuccess) return e;
  }
  const uint32_t groups = (a.n_buffers + F::SLOTS - 1u) / F::SLOTS;
  int grid = num_cus * (int)G::WG_PER_CU;  // one resident wave of persistent workgroups
  if ((uint32_t)grid > groups) grid = (int)groups;
  return launch_with_stop(k, grid, F::THREADS, G::LDS_BYTES, s, stop, a);
}

template <class F>
static hipError_t launch_family(int kind, bool dc, bool hits, bool spec, const ScnFftArgs &args, int num_cus, hipStream_t stream, hipEvent_t stop) {
  switch (kind) {
    case SCN_K_FLOAT_COMPLEX: return launch_kind<F, SCN_K_FLOAT_COMPLEX>(args, false, hits, spec, num_cus, stream, stop);
    case SCN_K_SHORT_COMPLEX: return launch_kind<F, SCN_K_SHORT_COMPLEX>(args, dc, hits, spec, num_cus, stream, stop);
    case SCN_K_SHORT: return launch_kind<F, SCN_K_SHORT>(args, dc, hits, spec, num_cus, stream, stop);
    case SCN_K_BYTE_COMPLEX: return launch_kind<F, SCN_K_BYTE_COMPLEX>(args, dc, hits, spec, num_cus, stream, stop);
    default: return hipErrorInvalidValue;
  }
}

// one launcher per translation unit (the sizes it instantiates), and the dispatcher over them
#define SCN_FFT_LAUNCH_ARGS int kind, bool dc, bool hits, bool spec, const ScnFftArgs &args, int num_cus, hipStream_t stream, hipEvent_t stop
#define SCN_FFT_LAUNCH_PASS kind, dc, hits, spec, args, num_cus, stream, stop
hipError_t scn_launch_fft_tu0(uint32_t n, SCN_FFT_LAUNCH_ARGS);
hipError_t scn_launch_fft_tu1(uint32_t n, SCN_FFT_LAUNCH_ARGS);
hipError_t scn_launch_fft_tu2(uint32_t n, SCN_FFT_LAUNCH_ARGS);
hipError_t scn_launch_fft_tu3(uint32_t n, SCN_FFT_LAUNCH_ARGS);
hipError_t scn_launch_fft_tu4(uint32_t n, SCN_FFT_LAUNCH_ARGS);
hipError_t scn_launch_fft_tu5(uint32_t n, SCN_FFT_LAUNCH_ARGS);
hipError_t scn_launch_fft_tu6(uint32_t n, SCN_FFT_LAUNCH_ARGS);
#if SCN_IN_TU(0)
hipError_t scn_launch_fft_tu0(uint32_t n, SCN_FFT_LAUNCH_ARGS) {
  return n == 16 ? launch_family<TinyFamily<1>>(SCN_FFT_LAUNCH_PASS) : launch_family<TinyFamily<2>>(SCN_FFT_LAUNCH_PASS);
}
#endif
#if SCN_IN_TU(1)
hipError_t scn_launch_fft_tu1(uint32_t n, SCN_FFT_LAUNCH_ARGS) {
  return n == 64 ? launch_family<TinyFamily<4>>(SCN_FFT_LAUNCH_PASS) : launch_family<TinyFamily<8>>(SCN_FFT_LAUNCH_PASS);
}
#endif
#if SCN_IN_TU(2)
hipError_t scn_launch_fft_tu2(uint32_t n, SCN_FFT_LAUNCH_ARGS) {
  return n == 256 ? launch_family<SmallFamily<1>>(SCN_FFT_LAUNCH_PASS) : launch_family<SmallFamily<2>>(SCN_FFT_LAUNCH_PASS);
}
#endif
#if SCN_IN_TU(3)
hipError_t scn_launch_fft_tu3(uint32_t n, SCN_FFT_LAUNCH_ARGS) {
  return n == 1024 ? launch_family<NarrowFamily<4>>(SCN_FFT_LAUNCH_PASS) : launch_family<NarrowFamily<8>>(SCN_FFT_LAUNCH_PASS);
}
#endif
#if SCN_IN_TU(4)
hipError_t scn_launch_fft_tu4(uint32_t, SCN_FFT_LAUNCH_ARGS) { return launch_family<NarrowFamily<16>>(SCN_FFT_LAUNCH_PASS); }
#endif
#if SCN_IN_TU(5)
hipError_t scn_launch_fft_tu5(uint32_t, SCN_FFT_LAUNCH_ARGS) { return launch_family<Family8k>(SCN_FFT_LAUNCH_PASS); }
#endif
#if SCN_IN_TU(6)
hipError_t scn_launch_fft_tu6(uint32_t, SCN_FFT_LAUNCH_ARGS) { return launch_family<Family16k>(SCN_FFT_LAUNCH_PASS); }
#endif

#if SCN_IN_TU(7)
hipError_t scn_launch_fft(uint32_t n, int kind, bool dc, bool hits, bool spec, const ScnFftArgs &args, int num_cus,
                          hipStream_t stream, hipEvent_t stop) {
  if (!hits && !spec) return hipErrorInvalidValue;
  if (args.n_buffers == 0) return stop ? hipEventRecord(stop, stream) : hipSuccess;
  // (256 / 512 points: one descriptor spans a workgroup's SLOTS buffers only, but the per-lane offsets are 32-bit)
  if (n < 1024 && (uint64_t)args.n_buffers * n * 8u > 0xffffffffull) return hipErrorInvalidValue;
  switch (n) {
    case 16: case 32: return scn_launch_fft_tu0(n, SCN_FFT_LAUNCH_PASS);
    case 64: case 128: return scn_launch_fft_tu1(n, SCN_FFT_LAUNCH_PASS);
    case 256: case 512: return scn_launch_fft_tu2(n, SCN_FFT_LAUNCH_PASS);
    case 1024: case 2048: return scn_launch_fft_tu3(n, SCN_FFT_LAUNCH_PASS);
    case 4096: return scn_launch_fft_tu4(n, SCN_FFT_LAUNCH_PASS);
    case 8192: return scn_launch_fft_tu5(n, SCN_FFT_LAUNCH_PASS);
    case 16384: return scn_launch_fft_tu6(n, SCN_FFT_LAUNCH_PASS);
    default: return hipErrorInvalidValue;
  }
}

bool scn_fft_size_supported(uint32_t n) { return n >= 16 && n <= 16384 && (n & (n - 1u)) == 0u; }

// layout of ScnFftArgs::tw1_table for size n: `rows` rows of `threads` entries, row p-1 = W_n^(t p): 15 rows of n/16 for the
// 16 x 16 x M kernels, 31 rows of 512 for the 32 x 16 x 32 form of 16384 points
void scn_tw1_layout(uint32_t n, uint32_t *rows, uint32_t *threads) {
  *rows = n == 16384 ? 31u : 15u;
  *threads = n == 16384 ? 512u : n / 16u;
}
#endif  // SCN_IN_TU(7)
