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

#include "scn_kernels.h"


#include "scn_device.h"
// The hit path of the narrow / 8192 / 16384-point kernels: the per-lane form (scn_record_hits_lanes, product) or, with 0, the
// per-output-index form (scn_record_hits; with SCN_ONE_ATOMIC_* its two-pass variant) -- profiles/r03_experiments.md section 6
#ifndef SCN_HITS_LANES_N
#define SCN_HITS_LANES_N 1
#endif
#ifndef SCN_HITS_LANES_W
#define SCN_HITS_LANES_W 1
#endif
#ifndef SCN_HITS_LANES_S
#define SCN_HITS_LANES_S 1
#endif
#ifndef SCN_HITS_LANES_16K
#define SCN_HITS_LANES_16K 1
#endif
#ifndef SCN_HITS_MASK_16K
#define SCN_HITS_MASK_16K 1  // the 16384-point HITS-ONLY kernel collects the candidate bits while it produces the powers (with the
                             // spectrum's stores in the same loop that cost 1-3 us instead of saving 3: profiles/r03_experiments.md section 6)
#endif
#ifndef SCN_ONE_ATOMIC_N
#define SCN_ONE_ATOMIC_N false
#endif
#ifndef SCN_ONE_ATOMIC_W
#define SCN_ONE_ATOMIC_W false
#endif
#ifndef SCN_ONE_ATOMIC_16K
#define SCN_ONE_ATOMIC_16K false
#endif


namespace {

// ---- global memory access through buffer descriptors ---------------------------------
// A raw buffer resource (SGPR descriptor, wave-uniform base) + one per-lane VGPR offset
// + scalar/immediate offsets: the 16 strided accesses of a thread cost no address VGPRs.
// Timing-only experiments (never in the product build): a zero-record descriptor makes the range
// check drop every access through it while the instruction stream is unchanged.
#ifndef SCN_EXP_NO_LOADS
#define SCN_EXP_NO_LOADS 0
#endif
#ifndef SCN_EXP_NO_STORES
#define SCN_EXP_NO_STORES 0
#endif

// Tunables (compile-time; scripts/build_variants.py builds one library per setting).
// Cache-policy immediates of the buffer instructions on gfx950: bit0 = sc0, bit1 = nt, bit4 = sc1.
// Both streams are touched exactly once, so both are non-temporal: measured with inputs AND outputs
// rotated over 1.5 GiB (nothing can live in the 256 MiB Infinity Cache), a no-compute skeleton
// of this kernel's traffic moves 5.66 TB/s with the default policy and 6.40 TB/s with nt on both
// (scripts/membw.hip); the FFT kernel itself gains 4-5 %.
#ifndef SCN_AUX_LD
#define SCN_AUX_LD 2
#endif
#ifndef SCN_AUX_ST
#define SCN_AUX_ST 2
#endif
#ifndef SCN_PREFETCH
#define SCN_PREFETCH 1 // fetch the next buffer's raw samples into registers during the FFT passes: reads are
                       // latency/MLP-bound otherwise (~32 KiB in flight per CU).  Costs 32 VGPRs for float
                       // input (-> 3 waves per SIMD), 16 for the integer formats; worth 3-4 % on C2.
                       // (A "touch-ahead" of the next buffer into L2/Infinity Cache was measured too: -5 %.)
#endif
#ifndef SCN_PF32_INT
#define SCN_PF32_INT 0  // experiment: register prefetch in the 512-thread 8192-point form for the 4-byte and 2-byte formats (16 registers;
                       // 13 VGPRs spill at the 128 of four waves per SIMD).  C3 shape, us per step: wide kernel 65.8, 512-thread form 73.9,
                       // with this prefetch 77.8 (scripts/narrow8k_check.sh)
#endif
#ifndef SCN_WAVES_PER_SIMD_PF
#define SCN_WAVES_PER_SIMD_PF 3  // VGPR budget 168 with the prefetch registers
#endif

template <int KIND>
struct RawLoader;

// float I,Q interleaved: 8 B per sample
template <>
struct RawLoader<SCN_K_FLOAT_COMPLEX> {
  static constexpr uint32_t kBufBytes(uint32_t n) { return 8u * n; }
  typedef v2f raw_t;
  template <int AUX>
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t, uint32_t t, uint32_t idx0) {
    // (the builtin returns a GCC-style vector; bit_cast, never assign it to an ext_vector)
    return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, t * 8u, idx0 * 8u, AUX));
  }
  // two consecutive samples idx0 + 2t, idx0 + 2t + 1 in one 16-byte load (the wide 8192-point kernel's lane pairs)
  template <int AUX>
  static __device__ __forceinline__ void load2(__amdgpu_buffer_rsrc_t r, uint32_t, uint32_t t, uint32_t idx0, raw_t &r0, raw_t &r1) {
    typedef float v4f_t __attribute__((ext_vector_type(4)));
    const v4f_t v = __builtin_bit_cast(v4f_t, __builtin_amdgcn_raw_buffer_load_b128(r, t * 16u, idx0 * 8u, AUX));
    r0 = v2f{v.x, v.y};
    r1 = v2f{v.z, v.w};
  }
  static __device__ __forceinline__ void ints(raw_t, int &re, int &im) { re = im = 0; }
  static __device__ __forceinline__ cf conv(raw_t r, int, int, float) { return from_v2f(r); }
};

// int16 I,Q interleaved: 4 B per sample
template <>
struct RawLoader<SCN_K_SHORT_COMPLEX> {
  static constexpr uint32_t kBufBytes(uint32_t n) { return 4u * n; }
  typedef int raw_t;
  template <int AUX>
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t, uint32_t t, uint32_t idx0) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, t * 4u, idx0 * 4u, AUX);
  }
  template <int AUX>
  static __device__ __forceinline__ void load2(__amdgpu_buffer_rsrc_t r, uint32_t, uint32_t t, uint32_t idx0, raw_t &r0, raw_t &r1) {
    typedef int v2i_t __attribute__((ext_vector_type(2)));
    const v2i_t v = __builtin_bit_cast(v2i_t, __builtin_amdgcn_raw_buffer_load_b64(r, t * 8u, idx0 * 4u, AUX));
    r0 = v.x;
    r1 = v.y;
  }
  static __device__ __forceinline__ void ints(raw_t r, int &re, int &im) {
    re = (int)(short)(r & 0xffff);
    im = r >> 16;
  }
  static __device__ __forceinline__ cf conv(raw_t r, int dc_re, int dc_im, float scale) {
    int re, im;
    ints(r, re, im);
    // float(source - dc) * onebymax, utility.cpp:81-82 (wrapping int arithmetic)
    return cf{(float)(int)((uint32_t)re - (uint32_t)dc_re) * scale,
              (float)(int)((uint32_t)im - (uint32_t)dc_im) * scale};
  }
};

// int8 I,Q interleaved: 2 B per sample
template <>
struct RawLoader<SCN_K_BYTE_COMPLEX> {
  static constexpr uint32_t kBufBytes(uint32_t n) { return 2u * n; }
  typedef int raw_t;
  template <int AUX>
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t, uint32_t t, uint32_t idx0) {
    return (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, t * 2u, idx0 * 2u, AUX);
  }
  template <int AUX>
  static __device__ __forceinline__ void load2(__amdgpu_buffer_rsrc_t r, uint32_t, uint32_t t, uint32_t idx0, raw_t &r0, raw_t &r1) {
    const uint32_t v = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, t * 4u, idx0 * 2u, AUX);
    r0 = (int)(v & 0xffffu);
    r1 = (int)(v >> 16);
  }
  static __device__ __forceinline__ void ints(raw_t r, int &re, int &im) {
    re = (int)(signed char)(r & 0xff);
    im = (int)(signed char)((r >> 8) & 0xff);
  }
  static __device__ __forceinline__ cf conv(raw_t r, int dc_re, int dc_im, float scale) {
    int re, im;
    ints(r, re, im);
    return cf{(float)(int)((uint32_t)re - (uint32_t)dc_re) * scale,
              (float)(int)((uint32_t)im - (uint32_t)dc_im) * scale};
  }
};

// int16 planar: I[n] then Q[n] per buffer; packed into the SHORT_COMPLEX register form
template <>
struct RawLoader<SCN_K_SHORT> {
  static constexpr uint32_t kBufBytes(uint32_t n) { return 4u * n; }
  typedef int raw_t;
  template <int AUX>
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t n, uint32_t t, uint32_t idx0) {
    int re = (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, t * 2u, idx0 * 2u, AUX);
    int im = (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, t * 2u, (n + idx0) * 2u, AUX);
    return (re & 0xffff) | (im << 16);
  }
  template <int AUX>
  static __device__ __forceinline__ void load2(__amdgpu_buffer_rsrc_t r, uint32_t n, uint32_t t, uint32_t idx0, raw_t &r0, raw_t &r1) {
    const uint32_t re = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, t * 4u, idx0 * 2u, AUX);        // I[2t], I[2t+1]
    const uint32_t im = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, t * 4u, (n + idx0) * 2u, AUX);  // Q[2t], Q[2t+1]
    r0 = (int)((re & 0xffffu) | (im << 16));
    r1 = (int)((re >> 16) | (im & 0xffff0000u));
  }
  static __device__ __forceinline__ void ints(raw_t r, int &re, int &im) {
    RawLoader<SCN_K_SHORT_COMPLEX>::ints(r, re, im);
  }
  static __device__ __forceinline__ cf conv(raw_t r, int dc_re, int dc_im, float scale) {
    return RawLoader<SCN_K_SHORT_COMPLEX>::conv(r, dc_re, dc_im, scale);
  }
};

}  // namespace

// ------------------------------------------------------------------------------------
// Fused kernel for N = 256*M points, M in {4, 8, 16, 32, 64} (N = 1024 / 2048 / 4096 / 8192 / 16384; the product uses the
// wide kernel further down for 8192).
// One workgroup of T = 16*M threads per buffer, persistent over buffers, 16 points per thread.
//
//   n = T*a + M*b + c          k = p + 16q + 256r         a,b,p,q in [0,16), c,r in [0,M)
//   pass 1  thread t = M*b + c:  16-pt DFT over a of x[T*a + t]*w[T*a + t] -> *W_N^{t p} -> LDS row p
//   pass 2  thread (p, c):       16-pt DFT over b                          -> *W_{16M}^{c q} -> LDS
//   pass 3  M-pt DFT over c for each (p,q):
//       M <= 16: thread t does the 16/M butterflies kl = t + T*u, outputs k = kl + 256 r
//       M == 32: two lanes (l, l+32) share a butterfly: each does the 16-pt DFT over c = 2c'+e
//                of its parity e, the odd one applies W_32^{r'}, one cross-half exchange
//                (radix 2) finishes it; lane half e outputs r = r' + 16e
//       M == 64: FOUR lanes (l, l+16, l+32, l+48) share a butterfly: lane row e does the 16-pt DFT over
//                c = 4c'+e, applies W_64^{e r'}, and a radix-4 step made of two lane exchanges
//                (v_permlane32_swap across the halves, v_permlane16_swap across the rows of a half)
//                leaves output r = r' + 16 s in the lane with s = 2 (e & 1) + (e >> 1)
//
// LDS (complex = 8 B slots, every address = per-thread base + immediate, all four access patterns
// bank-conflict-free -- checked with SQ_LDS_BANK_CONFLICT and the bank model of the guide):
//   exchange 1   L1(p, col) = p*P1 + col,   P1 = T + (M < 32 ? M : 0)
//       write (pass 1): fixed p, lanes col = t                -> consecutive slots
//       read  (pass 2): thread (p, c), fixed b: p*P1 + b*M + c -> the 32/M rows a 32-lane group
//                       touches sit 8M bytes apart in bank space
//   exchange 2   L2(c, kl) = c*P2 + kl,     P2 = 256 + (M <= 16 ? 16/M : 1)
//       write (pass 2): thread (p, c), fixed q: c*P2 + p + 16q -> 16 lanes on 16 distinct bank pairs
//       read  (pass 3): fixed c, lanes kl consecutive
//   then the pass-2 twiddle table [q][c] (16*M entries), DC-sum scratch, the hit counter.
// ------------------------------------------------------------------------------------
// Experiment build (-DSCN_STAMPS=1, scripts/stamp_profile.py): every wave reads the shader clock at the
// phase boundaries of the buffer loop; wave 0 of each workgroup leaves the per-phase cycle totals in the
// first 12 floats of its first buffer's spectrum.  Never defined in the product build.
#ifndef SCN_STAMPS
#define SCN_STAMPS 0
#endif
// Where the next buffer's 16 loads are issued: 0 all before pass 1; 1 all after barrier 1; 2 8/8 over those
// two points; 3 4/4/4/4 over the four barrier-separated phases; 4 (product) 6/5/5 over the first three; 5 8/4/4.
// Measured (profiles/r01_floors.md): 0 -> 4 is -3..-5 us per launch; 3, 4, 5 are within noise of each other.
#ifndef SCN_PF_SPLIT
#define SCN_PF_SPLIT 4
#endif
// Work distribution over the persistent workgroups.  1 (product): every workgroup starts on buffer blockIdx.x and
// then takes buffers from a device-scope queue, one iteration ahead so that the prefetch knows its target.
// 0: static grid-stride assignment (buf = blockIdx.x + k*gridDim.x).  With the static form the workgroups of one
// launch finish far apart (dispatch order, 10 or 11 buffers each, uneven speeds: 36..55 us in a 55 us int16
// launch).  For the memory-bound float path that tail is not idle capacity (the remaining workgroups run faster:
// measured neutral); for the compute-bound integer formats and the 8192-point kernel (8 buffers per workgroup) it is.
// The queue is sharded 8 ways: one returning device-scope atomic saturates at ~88 per us on one word, so 8192
// dequeues on one head would take longer than the launch (measured: 74 -> 115 us).  Workgroups land on XCD
// blockIdx.x % 8, so a shard is pulled by one XCD; shard x owns the buffers b = 8 j + x.  The launcher never
// starts more workgroups than buffers, so workgroup g takes buffer g statically (shard g % 8, j = g / 8) and
// head x hands out j = j0, j0 + 1, ... with j0 = the number of workgroups in shard x.  Heads are never reset:
// work_base[x] is head x's value before the launch (host-tracked; a launch adds exactly the number of buffers of
// shard x, because every workgroup stops at its first index past the end).  (The one take whose answer is needed at
// once is the prologue's; giving every workgroup a second static buffer instead, so that no take is ever waited for,
// was built and measured: C3 shape 66.4 -> 67.2 us, 4096-pt int16 58.8 -> 58.4 -- nothing, so the simpler form stays.)
// Measured per wire format (one box, single stream, us per launch static -> queue): int16 4096-pt 64.1 -> 61.2,
// int8 63.9 -> 59.4, 8192-pt int16 98.5 -> 89.9, float 4096-pt 76.1 -> 77.0, 8192-pt float 83.8 -> 85.8: the
// queue is compiled in for the integer formats from 4096 points up (scn_uses_queue, scn_kernels.h).
#ifndef SCN_DYNAMIC_WORK
#define SCN_DYNAMIC_WORK 1
#endif
// The dequeue must stay ONE plain global_atomic_add whose result is collected an iteration later.  LLVM's AMDGPU
// atomic optimizer rewrites any atomic with a provably uniform address into a wave reduction followed at once
// by s_waitcnt vmcnt(0) + readfirstlane (wave 0 then sits out the atomic's ~3 us round trip on every buffer:
// 74 -> 116 us), so the address gets an opaque per-lane zero offset.  (Switching the pass off globally,
// -amdgpu-atomic-optimizer-strategy=None, is not an option: it is what aggregates the hit path's overflow
// atomics per wave -- without it the int16 C3-shaped launch went from 64 to 97 us.)
#define SCN_WORK_QUEUE_SETUP()                                                                         \
  const uint32_t wq_shard = blockIdx.x & 7u;                                                           \
  uint32_t *const wq_head = args.work_counter + 32u * wq_shard; /* one 128-byte line per head */       \
  const uint32_t wq_base = args.work_base[wq_shard];                                                   \
  const uint32_t wq_j0 = (gridDim.x - wq_shard + 7u) >> 3;                                             \
  uint32_t wq_zero;                                                                                    \
  asm("v_mov_b32 %0, 0" : "=v"(wq_zero));                                                              \
  auto wq_take = [&]() -> uint32_t { return atomicAdd(wq_head + wq_zero, 1u); };                       \
  auto wq_buffer = [&](uint32_t taken) -> uint32_t { return 8u * (wq_j0 + (taken - wq_base)) + wq_shard; }

#if SCN_STAMPS
#define SCN_STAMP(i)                                             \
  do {                                                           \
    __builtin_amdgcn_sched_barrier(0);                           \
    const uint32_t now_ = (uint32_t)__builtin_readcyclecounter(); \
    stamp_acc[i] += now_ - stamp_prev;                           \
    stamp_prev = now_;                                           \
    __builtin_amdgcn_sched_barrier(0);                           \
  } while (0)
#else
#define SCN_STAMP(i)
#endif

template <int M>
struct Geo {
  static constexpr uint32_t N = 256u * M;
  static constexpr uint32_t T = 16u * M;
  static constexpr uint32_t P1 = T + (M < 32 ? M : 0);
  static constexpr uint32_t P2 = 256u + (M <= 16 ? 16u / M : 1u);
  static constexpr uint32_t EXCH = (16u * P1 > M * P2) ? 16u * P1 : M * P2;  // slots
  static constexpr uint32_t LDS_BYTES = EXCH * 8u + T * 8u + 16u * 4u + 2u * 4u + 64u * 4u + 8u + (M == 64 ? 64u * 8u : 0u);
  static constexpr uint32_t WAVES = T >= 64 ? T / 64 : 1;
  // Register prefetch costs a wave per SIMD (4 -> 3).  At 8192 points a workgroup is 8 waves, so
  // 3 waves per SIMD would leave ONE workgroup per CU: no prefetch there, two workgroups instead.
  static constexpr bool PREFETCH = SCN_PREFETCH != 0 && M <= 16;
  static constexpr uint32_t WAVES_PER_SIMD = PREFETCH ? SCN_WAVES_PER_SIMD_PF : 4;
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
  constexpr bool PF = G::PREFETCH || (SCN_PF32_INT != 0 && M == 32 && KIND != SCN_K_FLOAT_COMPLEX);
  constexpr bool DYN = scn_uses_queue(KIND, G::N);
  constexpr uint32_t N = G::N, T = G::T, P1 = G::P1, P2 = G::P2;
  typedef RawLoader<KIND> L;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v2f *lds = reinterpret_cast<v2f *>(smem_raw);
  v2f *lds_tw2 = lds + G::EXCH;                              // [16][M]
  int *lds_cnt = reinterpret_cast<int *>(lds_tw2 + T);      // [32] DC-sum scratch (re[16], im[16]: up to 16 waves)
  int *lds_hits = lds_cnt + 32;                             // [2] hit counters, alternating per buffer
  uint32_t *lds_next = reinterpret_cast<uint32_t *>(lds_hits + 2);  // [1] the buffer this workgroup takes after the next one
  v2f *lds_w64 = reinterpret_cast<v2f *>(smem_raw + G::LDS_BYTES - 64u * 8u);  // M == 64 only: W_64^m, m < 64

#if SCN_STAMPS
  const uint32_t stamp_entry = (uint32_t)wall_clock64();  // first instruction of the workgroup
#endif
  const uint32_t t = threadIdx.x;
  const uint32_t p2 = t / M, c2 = t % M;  // pass-2 identity (p, c); also the (q, c) of the table entry below
  const uint32_t lane = t & 63, wave = t >> 6;
  // pass-3 identity for M == 32: parity e = lane half, butterfly kl;  for M == 64: e = lane row (16 lanes), 16 butterflies per wave
  const uint32_t e = (M == 64) ? (t >> 4) & 3u : (t >> 5) & 1u;
  const uint32_t kl32 = (M == 64) ? (t & 15u) + 16u * (t >> 6) : (t & 31u) + 32u * (t >> 6);
  // the output block this lane ends up holding: r = r' + 16 s
  const uint32_t s_out = (M == 64) ? 2u * (e & 1u) + (e >> 1) : e;

  // the first buffer's samples go out before anything else: their latency then overlaps the
  // ~31 table loads below instead of following them
  // (Sizes without the register prefetch load at the top of the iteration.  Letting the next buffer's loads leave behind this
  // buffer's stores, before the end-of-buffer barrier and the hit recording -- free in registers, and at 16384 points, one
  // workgroup per CU, everything after the stores is dead time for the memory system: 5.9 k cycles at barrier 4 + 2 k of
  // recording per 30 k-cycle buffer -- was measured twice and is worse both times: 8192 points, 512-thread form, 85.9 -> 89.8 us
  // (round 1); 16384 points 109.7 -> 126.2 us cfloat, 94.2 -> 102.2 int16 (round 2).)
  typename L::raw_t raw[16];
  if (PF && blockIdx.x < args.n_buffers) {
    __amdgpu_buffer_rsrc_t r0 =
        make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)blockIdx.x * L::kBufBytes(N), SCN_EXP_NO_LOADS ? 0u : L::kBufBytes(N));
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
  if (M == 64 && t < 64) lds_w64[t] = args.twiddle[t * (N / 64u)];
  SCN_WORK_QUEUE_SETUP();
  if (t == 0) {
    lds_hits[0] = lds_hits[1] = 0;
    lds_next[0] = DYN ? wq_buffer(wq_take()) : blockIdx.x + gridDim.x;
  }
  __syncthreads();

  v2f *w1 = lds + t;                      // + p*P1
  v2f *r1 = lds + p2 * P1 + c2;           // + b*M
  v2f *w2 = lds + c2 * P2 + p2;           // + 16*q
  v2f *r3 = (M >= 32) ? lds + e * P2 + kl32 : lds + t;  // + 2c'*P2 (M = 32), + 4c'*P2 (M = 64)   |   + c*P2 + T*u
  const v2f *tw2 = lds_tw2 + c2;          // + q*M
  // global store offset of output o: voffset (per lane) + scalar part
  const uint32_t st_voff = (M >= 32) ? (kl32 + 4096u * s_out) * 4u : t * 4u;

  // K5 mask of this thread's 16 output bins (process.cpp:46-52): depends on (t, o) only
  const uint32_t jbase = (M >= 32) ? kl32 + 4096u * s_out : t;
  uint32_t keepmask = 0;
  if (HITS) {
#pragma unroll
    for (int o = 0; o < 16; o++) {
      const uint32_t joff = (M >= 32) ? 256u * o : T * (o / M) + 256u * (o % M);
      const uint32_t j = jbase + joff;
      const uint32_t i = j ^ (N / 2);  // (j + N/2) % N, process.cpp:47
      const bool keep = !(j < args.dc_ignore || (N - j) < args.dc_ignore) && !(i < args.i_lo || i > args.i_hi);
      keepmask |= keep ? (1u << o) : 0u;
    }
  }
  uint32_t par = 0;             // which of the two LDS hit counters the buffer in flight uses
  uint32_t prev = 0xffffffffu;  // the buffer whose recorders may still be running (none yet)

  // output o of this thread is bin j = jbase + joff(o):  M <= 16: o = u*M + r, j = t + T*u + 256*r;  M == 32: o = r', j = kl + 4096*e + 256*r'
  auto joff_of = [](int o) -> uint32_t { return (M >= 32) ? 256u * o : T * ((uint32_t)o / M) + 256u * ((uint32_t)o % M); };
#if SCN_STAMPS
  uint32_t stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t stamp_hit_cyc = 0, stamp_hit_n = 0;  // cycles wave 0 spent in scn_record_hits, and how often it went in
  uint32_t stamp_prev = (uint32_t)__builtin_readcyclecounter();
  const uint32_t stamp_t0 = (uint32_t)wall_clock64();  // 100 MHz
  uint32_t stamp_hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(stamp_hw));
#endif
  uint32_t buf = blockIdx.x;
  uint32_t nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_next[0]);
  while (buf < args.n_buffers) {
    SCN_STAMP(0);  // previous buffer's hit recording + loop back
    const bool more = nxt < args.n_buffers;
    // ---- K1 + K2: load, convert, window ----
    if (!PF) {
      __amdgpu_buffer_rsrc_t rin =
          make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)buf * L::kBufBytes(N), SCN_EXP_NO_LOADS ? 0u : L::kBufBytes(N));
#pragma unroll
      for (int a = 0; a < 16; a++) raw[a] = L::template load<AUX_LD>(rin, N, t, T * a);
    }

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
    SCN_STAMP(1);  // wait for this buffer's samples (+ convert, window)
    // Take the buffer after the next one; the answer is only needed at the end of this iteration.  Issued here,
    // behind the convert: a grab ahead of it sits in a divergent branch and the merged wait count then makes
    // wave 0 sit out the atomic's round trip where it waits for the samples.
    uint32_t taken = 0;
    if (DYN && t == 0 && more) taken = wq_take();
    // The raw registers are free again: fetch the next buffer of this workgroup while this one is
    // transformed.  Branch-free (past the last buffer the descriptor has zero records: the loads return
    // zeros without touching memory), so every load sits in the same basic block as the butterflies and
    // can be issued between them: a burst of 16 loads stalls the wave at issue for ~1600 cycles when the
    // memory pipeline is backed up (profiles/r01_floors.md, stamp profile), spread out they do not.
    const __amdgpu_buffer_rsrc_t rn =
        make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)(more ? nxt : buf) * L::kBufBytes(N),
                  (more && !SCN_EXP_NO_LOADS) ? L::kBufBytes(N) : 0u);
    auto prefetch = [&](int a_lo, int a_hi) {
#pragma unroll
      for (int a = 0; a < 16; a++)
        if (a >= a_lo && a < a_hi) raw[a] = L::template load<AUX_LD>(rn, N, t, T * a);
    };
    // issue points: 0 before pass 1, 1 after barrier 1, 2 after barrier 2, 3 after barrier 3;
    // pf_cut[k] .. pf_cut[k+1] = the loads issued at point k
    constexpr int pf_cut[5] = {0,
                               SCN_PF_SPLIT == 0 ? 16 : SCN_PF_SPLIT == 1 ? 0 : SCN_PF_SPLIT == 2 ? 8 : SCN_PF_SPLIT == 3 ? 4 : SCN_PF_SPLIT == 4 ? 6 : 8,
                               SCN_PF_SPLIT <= 2 ? 16 : SCN_PF_SPLIT == 3 ? 8 : SCN_PF_SPLIT == 4 ? 11 : 12,
                               SCN_PF_SPLIT <= 2 ? 16 : SCN_PF_SPLIT == 3 ? 12 : 16,
                               16};
    if (PF) prefetch(pf_cut[0], pf_cut[1]);

    SCN_STAMP(2);  // issue of the next buffer's loads
    // ---- pass 1: DFT over a, twiddle W_N^(t p), scatter to row p ----
    fft16(v);
#pragma unroll
    for (int p = 0; p < 16; p++) {
      cf y = v[OUT16(p)];
      if (p) y = cmul(y, tw1[p]);
      w1[p * P1] = to_v2f(y);
    }
    SCN_STAMP(3);  // pass 1 + exchange-1 writes
    __syncthreads();
    SCN_STAMP(4);  // barrier 1
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
    if (PF) prefetch(pf_cut[1], pf_cut[2]);
    fft16(v);
#pragma unroll
    for (int q = 1; q < 16; q++) v[OUT16(q)] = cmul(v[OUT16(q)], from_v2f(tw2[q * M]));
    SCN_STAMP(5);  // exchange-1 reads + pass 2 + twiddles
    __syncthreads();  // every exchange-1 read done before the area is re-used
    SCN_STAMP(6);  // barrier 2
    if (PF) prefetch(pf_cut[2], pf_cut[3]);
#pragma unroll
    for (int q = 0; q < 16; q++) w2[q * 16] = to_v2f(v[OUT16(q)]);
    SCN_STAMP(7);  // exchange-2 writes
    __syncthreads();
    SCN_STAMP(8);  // barrier 3
    if (PF) prefetch(pf_cut[3], pf_cut[4]);

    // ---- pass 3: M-point DFT over c ----
    if constexpr (M == 64) {
#pragma unroll
      for (int c = 0; c < 16; c++) v[c] = from_v2f(r3[4 * c * P2]);
      fft16(v);
      // radix 4 across the four rows of the wave: T_e = W_64^(e r') Y_e[r'], then
      //   halves:  A_e0 = T_e0 + T_(e0+2) (rows 0, 1)     B_e0 = T_e0 - T_(e0+2) (rows 2, 3)
      //   rows:    X_0 = A_0 + A_1, X_2 = A_0 - A_1 (rows 0, 1)     X_1 = B_0 - i B_1, X_3 = B_0 + i B_1 (rows 2, 3)
      // Both exchanges use the swap instructions on two copies of the value, which leaves the pair's two values in both
      // lanes (see M == 32 below); the signs are per-lane constants, so every lane runs the same instructions.
      const bool upper = (e & 2u) != 0, odd = (e & 1u) != 0;
      const float sgn_h = upper ? -1.0f : 1.0f, sgn_r = odd ? -1.0f : 1.0f;
      const v2f *wtab = lds_w64;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        cf y = v[OUT16(r)];
        if (r) y = cmul(y, from_v2f(wtab[(e * (uint32_t)r) & 63u]));  // W_64^(e r); row 0 multiplies by 1
        auto hx = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, y.x), __builtin_bit_cast(unsigned, y.x), false, false);
        auto hy = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, y.y), __builtin_bit_cast(unsigned, y.y), false, false);
        const cf ab = cf{__builtin_fmaf(__builtin_bit_cast(float, (unsigned)hx[1]), sgn_h, __builtin_bit_cast(float, (unsigned)hx[0])),
                         __builtin_fmaf(__builtin_bit_cast(float, (unsigned)hy[1]), sgn_h, __builtin_bit_cast(float, (unsigned)hy[0]))};
        auto rx = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, ab.x), __builtin_bit_cast(unsigned, ab.x), false, false);
        auto ry = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, ab.y), __builtin_bit_cast(unsigned, ab.y), false, false);
        const cf s0 = cf{__builtin_bit_cast(float, (unsigned)rx[0]), __builtin_bit_cast(float, (unsigned)ry[0])};  // even row of the pair
        const cf s1 = cf{__builtin_bit_cast(float, (unsigned)rx[1]), __builtin_bit_cast(float, (unsigned)ry[1])};  // odd row
        const cf tt = upper ? cf{s1.y, -s1.x} : s1;  // -i B_1 in the upper half, A_1 in the lower
        v[OUT16(r)] = cf{__builtin_fmaf(tt.x, sgn_r, s0.x), __builtin_fmaf(tt.y, sgn_r, s0.y)};
      }
    } else if constexpr (M == 32) {
#pragma unroll
      for (int c = 0; c < 16; c++) v[c] = from_v2f(r3[2 * c * P2]);
      fft16(v);
      // radix 2 across the wave's two halves: X[r'] = Y0 + W32^r' Y1, X[r'+16] = Y0 - W32^r' Y1.
      // v_permlane32_swap exchanges the upper half of one register with the lower half of another:
      // applied to two copies of y it leaves (Y0, W Y1) in BOTH halves, so the butterfly is one
      // FMA with a per-half sign -- no LDS crossbar (ds_bpermute) traffic.
      const float sel = e ? 1.0f : 0.0f, sgn = e ? -1.0f : 1.0f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        cf y = v[OUT16(r)];
        if (r) {
          // W_32^r for the odd half, 1 for the even half (branch-free: both halves run the same code)
          const float cr = (float)__builtin_cos(6.283185307179586476925286766559 * r / 32.0);
          const float sr = (float)__builtin_sin(6.283185307179586476925286766559 * r / 32.0);
          cf w = cf{e ? cr : 1.0f, -sr * sel};
          y = cmul(y, w);
        }
        auto sx = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, y.x), __builtin_bit_cast(unsigned, y.x), false, false);
        auto sy = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, y.y), __builtin_bit_cast(unsigned, y.y), false, false);
        v[OUT16(r)] = cf{__builtin_fmaf(__builtin_bit_cast(float, (unsigned)sx[1]), sgn, __builtin_bit_cast(float, (unsigned)sx[0])),
                         __builtin_fmaf(__builtin_bit_cast(float, (unsigned)sy[1]), sgn, __builtin_bit_cast(float, (unsigned)sy[0]))};
      }
    } else {
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
    }

    // ---- K4 + K5 ----
    // The thread's 16 LINEAR powers stay in `pw` (a true vector: the recording path indexes it with a wave-uniform o,
    // s_set_gpr_idx) for the hit path; the spectrum gets the dB map of scn_device.h: its product form inline, and -- only in
    // waves that hold a bin from SCN_P_EXACT_FROM up -- the exact form stored over it for those bins (the other lanes' stores
    // go to an out-of-range offset and are dropped by the descriptor's range check: no branch per bin).
    v16f pw;
    float gmax[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // the largest power of each group of four outputs (max ignores NaN)
    __amdgpu_buffer_rsrc_t rout = make_rsrc(args.power_db + (size_t)buf * N, (SPEC && args.power_db && !SCN_EXP_NO_STORES) ? 4u * N : 0u);
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
    SCN_STAMP(9);  // exchange-2 reads + pass 3 + dB + stores issued
    if (t == 0) lds_next[0] = !more ? 0xffffffffu : DYN ? wq_buffer(taken) : nxt + gridDim.x;
    __syncthreads();  // exchange area free again; lds_next visible (it is rewritten three barriers from now)
    SCN_STAMP(10);  // barrier 4
    const uint32_t after = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_next[0]);
#if SCN_STAMPS
    stamp_acc[11] += 1;
#endif
    if (HITS) {
      // Recording runs AFTER the barrier: a wave that holds detections does not stall the other
      // waves of its workgroup, they go on to the next buffer and meet it at that buffer's first
      // barrier (its loads are in flight meanwhile).  Only such a wave evaluates the per-bin test.
      if (__ballot(pmax > args.p_lo)) {
#if SCN_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t h0_ = (uint32_t)__builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
#endif
#if SCN_HITS_LANES_N
        scn_record_hits_lanes<16, false, true>(pw, gmax, keepmask, args, &lds_hits[par], buf, lane, [&](int o) -> uint32_t { return (jbase + joff_of(o)) ^ (N / 2); });
#else
        scn_record_hits<16, false, true, SCN_ONE_ATOMIC_N>(pw, gmax, keepmask, args, &lds_hits[par], buf, lane, [&](int o) -> uint32_t { return (jbase + joff_of(o)) ^ (N / 2); });
#endif
#if SCN_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        stamp_hit_cyc += (uint32_t)__builtin_readcyclecounter() - h0_;
        stamp_hit_n += 1;
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
      prev = buf;
      par ^= 1;
    }
    buf = nxt;
    nxt = after;
  }
#if SCN_STAMPS
  if (t == 0 && blockIdx.x < args.n_buffers && args.power_db) {
    __builtin_amdgcn_s_waitcnt(0);
    for (int i = 0; i < 12; i++) args.power_db[(size_t)blockIdx.x * N + i] = (float)stamp_acc[i];
    // start / end of this workgroup on the 100 MHz wall clock (low 24 bits: exact in a float) and where it ran
    args.power_db[(size_t)blockIdx.x * N + 12] = (float)(stamp_t0 & 0xffffffu);
    args.power_db[(size_t)blockIdx.x * N + 13] = (float)((uint32_t)wall_clock64() & 0xffffffu);
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    args.power_db[(size_t)blockIdx.x * N + 14] = (float)(((xcc & 0xfu) << 8) | ((stamp_hw >> 8) & 0xffu));  // xcc, se/sh/cu
    args.power_db[(size_t)blockIdx.x * N + 15] = (float)(stamp_entry & 0xffffffu);
    args.power_db[(size_t)blockIdx.x * N + 16] = (float)stamp_hit_cyc;
    args.power_db[(size_t)blockIdx.x * N + 17] = (float)stamp_hit_n;
  }
#endif
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
//   pass 3  M = 2: eight radix-2 butterflies per thread (kl = t + 32 u -> bins kl, kl + 256);  M = 1: nothing left to transform,
//           the exchange only transposes (thread t stores bins t + 16 u)
// One buffer descriptor spans the workgroup's SLOTS consecutive buffers (a wave holds two or four of them, so a per-buffer
// descriptor would not be wave-uniform); slots past the end of the batch read zeros and store nothing (range check).
// Hits: a wave is no longer one buffer, so slots in a buffer's region come from one LDS atomic per HIT (rare) on the slot's
// counter instead of one per wave.  These sizes ran the staged double-precision path (scn_generic.hip) at 40 Gsamples/s.
// ------------------------------------------------------------------------------------
namespace {
template <int M>
struct GeoSmall {
  static constexpr uint32_t N = 256u * M, T = 16u * M, SLOTS = 256u / T;
  static constexpr uint32_t P1 = T + M, P2 = 256u + 16u / M;
  static constexpr uint32_t EXCH = (16u * P1 > M * P2) ? 16u * P1 : M * P2;  // slots per buffer
  static constexpr uint32_t LDS_BYTES = SLOTS * EXCH * 8u + T * 8u + SLOTS * 4u + 16u;
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
  int *lds_hits = reinterpret_cast<int *>(lds_tw2 + T);                              // [SLOTS]
  const uint32_t p2 = t / M, c2 = t % M;
  const uint32_t buf_bytes = L::kBufBytes(N);

  // the iteration's SLOTS consecutive buffers under one descriptor (zero records past the end of the batch)
  auto in_rsrc = [&](uint32_t first) {
    const bool ok = first < args.n_buffers;
    const uint32_t nb = ok ? (args.n_buffers - first < SLOTS ? args.n_buffers - first : SLOTS) : 0u;
    return make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)(ok ? first : 0u) * buf_bytes, SCN_EXP_NO_LOADS ? 0u : nb * buf_bytes);
  };
  // planar int16 reads I and Q of a buffer n samples apart: the loader takes the buffer's own sample count, the slot offset
  // goes into the index (in samples of the loader's element size)
  typename L::raw_t raw[16];
  auto load_group = [&](const __amdgpu_buffer_rsrc_t &r, int a_lo, int a_hi) {
#pragma unroll
    for (int a = 0; a < 16; a++)
      if (a >= a_lo && a < a_hi) raw[a] = L::template load<AUX_LD>(r, N, t, slot * (KIND == SCN_K_SHORT ? 2u * N : N) + T * a);
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
  if (tid < SLOTS) lds_hits[tid] = 0;
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
#pragma unroll
    for (int q = 1; q < 16; q++) v[OUT16(q)] = cmul(v[OUT16(q)], from_v2f(tw2[q * M]));
    __syncthreads();  // every exchange-1 read done before the area is re-used
    load_group(rn, 11, 16);
#pragma unroll
    for (int q = 0; q < 16; q++) w2[q * 16] = to_v2f(v[OUT16(q)]);
    __syncthreads();
    // ---- pass 3 ----
#pragma unroll
    for (int u = 0; u < 16 / M; u++)
#pragma unroll
      for (int c = 0; c < M; c++) v[u * M + c] = from_v2f(r3[c * P2 + T * u]);
    if constexpr (M == 2) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const cf a = v[2 * u], b = v[2 * u + 1];
        v[2 * u] = a + b;      // r = 0
        v[2 * u + 1] = a - b;  // r = 1
      }
    }
    // ---- K4 (see scn_fft_kernel) ----
    v16f pw;
    float gmax[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const uint32_t nvalid = args.n_buffers - first < SLOTS ? args.n_buffers - first : SLOTS;
    __amdgpu_buffer_rsrc_t rout = make_rsrc(args.power_db + (size_t)first * N, (SPEC && args.power_db && !SCN_EXP_NO_STORES) ? nvalid * 4u * N : 0u);
    const uint32_t st_voff = (slot * N + t) * 4u;
#pragma unroll
    for (int o = 0; o < 16; o++) {
      const float q = power_of(v[o]);
      pw[o] = q;
      gmax[o >> 2] = fmaxf(gmax[o >> 2], q);
      if constexpr (SPEC) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, db_fast(q)), rout, st_voff, 4u * joff_of(o), AUX_ST);
    }
    const float pmax = fmaxf(fmaxf(gmax[0], gmax[1]), fmaxf(gmax[2], gmax[3]));
    if constexpr (SPEC) {
      if (__ballot(pmax >= SCN_P_EXACT_FROM)) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
          if (__ballot(gmax[g] >= SCN_P_EXACT_FROM)) {
#pragma unroll
            for (int o = 4 * g; o < 4 * g + 4; o++) {
              const float q = pw[o];
              if (__ballot(q >= SCN_P_EXACT_FROM)) {
                const float d = db_exact(q);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d), rout, q >= SCN_P_EXACT_FROM ? st_voff : 0x80000000u, 4u * joff_of(o), AUX_ST);
              }
            }
          }
        }
      }
    }
    // ---- K5: a wave holds several buffers here, so every hit takes its slot of the region with its own LDS atomic ----
    if (HITS) {
      if (__ballot(valid && pmax > args.p_lo)) {
        uint32_t cand = 0;
#pragma unroll
        for (int o = 0; o < 16; o++) cand |= (pw[o] > args.p_lo) ? (1u << o) : 0u;
        cand &= valid ? keepmask : 0u;
        ScnDevHit *const region = args.hits + (size_t)buf * args.hit_region;
#if SCN_HITS_LANES_S
        while (__ballot(cand != 0u)) {  // one candidate per lane per trip (scn_record_hits_lanes; here a lane's counter is its slot's)
          const bool act = cand != 0u;
          const uint32_t o = act ? (uint32_t)__builtin_ctz(cand) : 0u;
          cand &= cand - 1u;
          const float q = scn_select_output<16>(pw, o);
          float d = db_fast(q);
          if (__ballot(act && q >= SCN_P_EXACT_FROM)) d = q >= SCN_P_EXACT_FROM ? db_exact(q) : d;
          if (act && d > args.threshold) {  // strict >, process.cpp:54
            const uint32_t pos = (uint32_t)atomicAdd(&lds_hits[slot], 1);
            if (pos < args.hit_region) region[pos] = ScnDevHit{(t + joff_of((int)o)) ^ (N / 2), d};
          }
        }
#else
        uint32_t wm = wave_or_u32(cand);
        while (wm) {
          const int o = __builtin_ctz(wm);  // wave-uniform
          wm &= wm - 1u;
          const float q = pw[o];
          float d = db_fast(q);
          if (__ballot(q >= SCN_P_EXACT_FROM)) d = q >= SCN_P_EXACT_FROM ? db_exact(q) : d;
          if (((cand >> o) & 1u) && d > args.threshold) {  // strict >, process.cpp:54
            const uint32_t pos = (uint32_t)atomicAdd(&lds_hits[slot], 1);
            if (pos < args.hit_region) region[pos] = ScnDevHit{(t + joff_of(o)) ^ (N / 2), d};
          }
        }
#endif
      }
    }
    __syncthreads();  // exchange areas free again; the slots' hit counters final
    if (HITS && t == 0) {
      if (valid) {
        args.per_buffer_hits[buf] = (uint32_t)lds_hits[slot];
        if (args.host_hits) args.host_hits[buf] = (uint32_t)lds_hits[slot];
      }
      lds_hits[slot] = 0;  // (the next recording is three barriers away)
    }
  }
}

// ------------------------------------------------------------------------------------
// 8192 points, wide form: ONE workgroup of 256 threads per buffer, 32 points per thread.
//
// Same decomposition as above (n = 512a + 32b + c, k = p + 16q + 256r, passes 16 x 16 x 32), but every
// thread plays two of the 512 "virtual threads" of passes 1 and 2 (tau and tau + 256) and owns one whole
// 32-point DFT in pass 3 (in registers: two 16-point DFTs + one radix-2 step, no cross-lane exchange).
// Why: with 512 threads a workgroup is 8 waves, two workgroups per CU are 4 waves per SIMD = 128 VGPRs,
// which leaves no registers to prefetch the next buffer -- the stamp profile of that form shows the load
// latency fully exposed (19 % waiting for samples, 20 % at barrier 1, 13 % at barrier 4).  256 threads x
// 2 workgroups per CU (LDS: 2 x 70 KiB) are 2 waves per SIMD = 256 VGPRs: room for the next buffer's
// 32 samples per thread, fetched in three groups spread over the passes like the smaller sizes.
// LDS layouts and bank behaviour are those of Geo<32> (P1 = 512, P2 = 257): each wave touches the same
// slots per instruction as a wave of the 512-thread form does.
// ------------------------------------------------------------------------------------
#ifndef SCN_WIDE_8192
#define SCN_WIDE_8192 1
#endif
#ifndef SCN_16K_FLOAT_PFN
#define SCN_16K_FLOAT_PFN 8
#endif
#ifndef SCN_8K_FLOAT_PFN
#define SCN_8K_FLOAT_PFN 16  // (8192-pt cfloat, 16 / 14 / 12 / 8 prefetched: 77.5 / 76.7 / 77.5 / 77.7 us -- flat)
#endif
#ifndef SCN_16K_INT_PFN
#define SCN_16K_INT_PFN 16
#endif
#ifndef SCN_WIDE_16384_FLOAT
#define SCN_WIDE_16384_FLOAT 1  // float input at 16384 points in the wide form with a partial prefetch (0: scn_fft_kernel<64>)
#endif
#ifndef SCN_WIDE_16384
#define SCN_WIDE_16384 1  // 16384 points: 512 threads x 32 points with a register prefetch instead of scn_fft_kernel<64>
#endif
#ifndef SCN_16K_P3_DOUBLE
// Pass 3 of the 16384-point kernel in double (scn_fft16k2_body).  Measured on MI355X, 3072 strong-tone buffers per run
// (scripts/acc16k.py) and us per 2048-buffer launch, float pass 3 -> double: parity metric max 4.9e-6 .. 7.5e-6 -> 2.9e-6 .. 4.0e-6,
// p99 4.0e-6 -> 2.2e-6, median 1.11e-6 -> 0.88e-6 (the float quantisation of the dB value itself); cfloat 93.5 -> 100.2 us,
// int16 75.6 -> 88.1 (+7 % / +16 %: 128 conversions and 1.2 .. 1.4x on a third of the arithmetic).  Round 2's kernel of this
// size (16 x 16 x 64 with a lane-pair last pass, float) took 100.6 / 95.7 us on the same box and reached 1.05e-5: the bar.
// Parity comes first, so double is the product; -DSCN_16K_P3_DOUBLE=0 is the float build.  (In the round-2 form of the
// kernel the same change cost +38 % and spilled: the lane exchange doubles, and its registers do not fit.)
#define SCN_16K_P3_DOUBLE 1
#endif
#ifndef SCN_8K_PAIR
#define SCN_8K_PAIR 1  // the wide kernel's threads play neighbouring virtual threads (two-sample loads, 16-byte exchange-1 writes)
#endif
// (16384 points run as scn_fft_kernel<64>: 1024 threads x 16 points, four waves per SIMD, lane-quad radix-4 in pass 3.
// The first form -- 256 threads x 64 points, one wave per SIMD, modelled on the wide kernel below -- was 1.5-1.8x slower:
// 195 / 191 Gsamples/s against 296 / 344 for cfloat / int16 at batch 2048.)
namespace {
// the wide form for N = 256*M2, M2 = 32 (8192 points: the kernel described above) or 64 (16384 points): 8*M2 threads play
// the 16*M2 virtual threads of passes 1 and 2, two each, and hold 32 points each; two waves per SIMD either way
template <int M2>
struct GeoWide {
  static constexpr uint32_t N = 256u * M2, T = 8u * M2, TV = 16u * M2, P1 = TV, P2 = 257;
  static constexpr uint32_t EXCH = (16u * P1 > M2 * P2) ? 16u * P1 : M2 * P2;  // slots
  static constexpr uint32_t LDS_BYTES = EXCH * 8u + TV * 8u + 16u * 4u + 2u * 4u + 8u;
  static constexpr uint32_t WG_PER_CU = M2 == 32 ? 2 : 1;  // 70 KiB / 140 KiB of LDS
};
typedef GeoWide<32> Geo8k;
typedef float v32f __attribute__((ext_vector_type(32)));

// 16384 points in the wide form (M2 = 64).  The 1024-thread form (scn_fft_kernel<64>: four waves per SIMD, 128 VGPRs, no room
// to prefetch) has ONE workgroup per CU, so nothing covers its loads: its stamp profile shows 13 k of a buffer's 30 k cycles
// waiting for the samples (and letting the next buffer's loads leave behind the stores makes it worse, see there).  512 threads
// x 32 points are two waves per SIMD = 256 VGPRs: the next buffer is prefetched in three groups spread over the passes, as
// at 8192 points.  Pass 3 is a 64-point DFT per kl shared by TWO lanes (l, l + 32): lane half e does the 16-point DFTs over
// c = 4c' + e and c = 4c' + e + 2, joins them in registers (U, V = Y_e +- W_32^r' Y_(e+2): the 8192-point kernel's last step),
// the odd half multiplies by W_64^r', and one v_permlane32_swap per register finishes the radix-4 step:
//   X[r']      = U_0 + W U_1 (lower half)      X[r' + 32] = U_0 - W U_1 (upper half)
//   X[r' + 16] = V_0 - i W V_1 (lower half)    X[r' + 48] = V_0 + i W V_1 (upper half)
template <int M2, int KIND, bool DC, bool HITS, bool SPEC>
__device__ __forceinline__ void scn_fft_wide_body(const ScnFftArgs &args) {
  static_assert(HITS || SPEC, "a kernel that reports nothing");
  typedef GeoWide<M2> G;
  constexpr bool P3D = M2 == 64 && SCN_16K_P3_DOUBLE != 0;  // pass 3 in double (see there)
  constexpr int AUX_LD = SCN_AUX_LD;
  constexpr int AUX_ST = SCN_AUX_ST;
  constexpr uint32_t N = G::N, T = G::T, TV = G::TV, P1 = G::P1, P2 = G::P2;
  constexpr bool DYN = scn_uses_queue(KIND, N);
  typedef RawLoader<KIND> L;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v2f *lds = reinterpret_cast<v2f *>(smem_raw);
  v2f *lds_tw2 = lds + G::EXCH;                              // [16][M2]: W_(16 M2)^(c q) at q*M2 + c
  int *lds_cnt = reinterpret_cast<int *>(lds_tw2 + TV);     // [16] DC-sum scratch (re[8], im[8])
  int *lds_hits = lds_cnt + 16;                             // [2] hit counters, alternating per buffer
  uint32_t *lds_next = reinterpret_cast<uint32_t *>(lds_hits + 2);  // [1] the buffer this workgroup takes after the next one

#if SCN_STAMPS
  const uint32_t stamp_entry = (uint32_t)wall_clock64();
#endif
  const uint32_t t = threadIdx.x;
  const uint32_t lane = t & 63, wave = t >> 6;
  const uint32_t c2 = t % M2, p2 = t / M2;   // pass-2 identities of this thread: (p2, c2) and (p2 + 8, c2), p2 < 8
  // pass-3 identity.  M2 = 32: one whole 32-point DFT, kl = t.  M2 = 64: half e of the 64-point DFT of kl (lanes l, l + 32)
  const uint32_t e = (M2 == 64) ? (t >> 5) & 1u : 0u;
  const uint32_t kl = (M2 == 64) ? (t & 31u) + 32u * (t >> 6) : t;
  // Which two of the 512 pass-1 virtual threads this thread plays.  PAIR (product): tau = 2t and 2t + 1 -- their inputs
  // 512a + 2t (+1) are neighbours in memory and their exchange-1 slots neighbours in LDS, so a buffer is fetched in 16
  // loads of two samples instead of 32 of one (16 / 8 / 4 bytes per lane) and exchange 1 is written in 16 ds_write_b128
  // instead of 32 ds_write_b64.  !PAIR: tau = t and t + 256 (the first form).
  constexpr bool PAIR = SCN_8K_PAIR != 0;
  // how many of a buffer's 16 two-sample loads are prefetched during the previous buffer's passes (in three groups); the rest
  // is fetched at the top of its own iteration.  16 everywhere except float input at 16384 points, whose 64 prefetch registers
  // do not fit beside the lane-pair step of pass 3 (22 VGPRs spilled, slower than the 1024-thread form): there 12 (48 registers)
  constexpr int PFN = M2 != 64 ? (KIND == SCN_K_FLOAT_COMPLEX ? SCN_8K_FLOAT_PFN : 16) : KIND == SCN_K_FLOAT_COMPLEX ? SCN_16K_FLOAT_PFN : SCN_16K_INT_PFN;
  constexpr int PF0 = PFN == 16 ? 6 : PFN / 3 + 1, PF1 = PFN == 16 ? 11 : 2 * PFN / 3 + 1;
  const uint32_t tau0 = PAIR ? 2u * t : t, tau1 = PAIR ? 2u * t + 1u : t + T;

  // first buffer's samples first: raw[2a + h] = x[512 a + tau_h]
  typename L::raw_t raw[32];
  auto load_group = [&](const __amdgpu_buffer_rsrc_t &r, int a_lo, int a_hi) {  // inputs a_lo .. a_hi - 1 of both virtual threads
#pragma unroll
    for (int a = 0; a < 16; a++)
      if (a >= a_lo && a < a_hi) {
        if (PAIR) {
          L::template load2<AUX_LD>(r, N, t, TV * a, raw[2 * a], raw[2 * a + 1]);
        } else {
          raw[2 * a] = L::template load<AUX_LD>(r, N, t, TV * a);
          raw[2 * a + 1] = L::template load<AUX_LD>(r, N, t, TV * a + T);
        }
      }
  };
  if (blockIdx.x < args.n_buffers) {
    __amdgpu_buffer_rsrc_t r0 =
        make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)blockIdx.x * L::kBufBytes(N), SCN_EXP_NO_LOADS ? 0u : L::kBufBytes(N));
    load_group(r0, 0, PFN);
  }
  // persistent constants: pass-1 twiddles of both virtual threads (table rows of TV), window taps
  // (16384 points with pass 3 in double: the second virtual thread's twiddles are NOT kept -- tau1 = tau0 + 1, so
  // W_N^(tau1 p) = W_N^(tau0 p) W_N^p with W_N^p a compile-time constant: one more complex multiply per p and buffer buys the
  // 30 registers pass 3 needs for its double-precision values)
  constexpr bool TW1B = !(P3D && PAIR);
  cf tw1a[16], tw1b[TW1B ? 16 : 1];
#pragma unroll
  for (int p = 1; p < 16; p++) {
    tw1a[p] = from_v2f(args.tw1_table[(p - 1) * TV + tau0]);
    if (TW1B) tw1b[p] = from_v2f(args.tw1_table[(p - 1) * TV + tau1]);
  }
  float win[32];
#pragma unroll
  for (int a = 0; a < 16; a++) {
    win[2 * a] = args.window[TV * a + tau0] * args.scale;
    win[2 * a + 1] = args.window[TV * a + tau1] * args.scale;
  }
  // pass-2 twiddles W_(16 M2)^(c q) = W_N^(16 c q), entry q*M2 + c: this thread fills entries t (q = p2) and t + T (q = p2 + 8)
  lds_tw2[t] = args.twiddle[(16u * p2 * c2) & (N - 1)];
  lds_tw2[t + T] = args.twiddle[(16u * (p2 + 8u) * c2) & (N - 1)];
  SCN_WORK_QUEUE_SETUP();
  if (t == 0) {
    lds_hits[0] = lds_hits[1] = 0;
    lds_next[0] = DYN ? wq_buffer(wq_take()) : blockIdx.x + gridDim.x;
  }
  __syncthreads();

  v2f *w1 = lds + tau0;                    // + p*P1 (the second virtual thread: + (tau1 - tau0))
  v2f *r1 = lds + p2 * P1 + c2;            // + b*M2 (+ 8*P1 for the second)
  v2f *w2 = lds + c2 * P2 + p2;            // + 16*q (+ 8 for the second)
  v2f *r3 = lds + e * P2 + kl;             // M2 = 32: + c*P2;  M2 = 64: + (4c')*P2 and + (4c' + 2)*P2
  const v2f *tw2 = lds_tw2 + c2;           // + q*M2
  // output o of this thread is bin j = jbase + 256 o  (M2 = 64: o = r' + 16 h is block r = o + 32 e)
  const uint32_t jbase = kl + 8192u * e;
  const uint32_t st_voff = jbase * 4u;

  uint32_t keepmask = 0;  // K5 mask of this thread's 32 bins (process.cpp:46-52)
  if (HITS) {
#pragma unroll
    for (int r = 0; r < 32; r++) {
      const uint32_t j = jbase + 256u * r;
      const uint32_t i = j ^ (N / 2);  // (j + N/2) % N, process.cpp:47
      const bool keep = !(j < args.dc_ignore || (N - j) < args.dc_ignore) && !(i < args.i_lo || i > args.i_hi);
      keepmask |= keep ? (1u << r) : 0u;
    }
  }
  uint32_t par = 0;
  uint32_t prev = 0xffffffffu;  // the buffer whose recorders may still be running

#if SCN_STAMPS
  uint32_t stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t stamp_hit_cyc = 0, stamp_hit_n = 0;  // cycles wave 0 spent in scn_record_hits, and how often it went in
  uint32_t stamp_prev = (uint32_t)__builtin_readcyclecounter();
  const uint32_t stamp_t0 = (uint32_t)wall_clock64();  // 100 MHz
  uint32_t stamp_hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(stamp_hw));
#endif
  uint32_t buf = blockIdx.x;
  uint32_t nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_next[0]);
  while (buf < args.n_buffers) {
    SCN_STAMP(0);  // previous buffer's hit recording + loop back
    const bool more = nxt < args.n_buffers;
    if (PFN < 16) {  // the part of this buffer that was not prefetched (float input at 16384 points: registers)
      const __amdgpu_buffer_rsrc_t rc = make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)buf * L::kBufBytes(N),
                                                  SCN_EXP_NO_LOADS ? 0u : L::kBufBytes(N));
      load_group(rc, PFN, 16);
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
    cf va[16], vb[16];
#pragma unroll
    for (int a = 0; a < 16; a++) {
      va[a] = L::conv(raw[2 * a], dc_re, dc_im, 1.0f) * win[2 * a];
      vb[a] = L::conv(raw[2 * a + 1], dc_re, dc_im, 1.0f) * win[2 * a + 1];
    }
    SCN_STAMP(1);  // wait for this buffer's samples (+ convert, window)
    // take the buffer after the next one (behind the convert, see scn_fft_kernel); needed at the end of the iteration
    uint32_t taken = 0;
    if (DYN && t == 0 && more) taken = wq_take();
    // next buffer of this workgroup, branch-free (zero records past the end), in three groups
    const __amdgpu_buffer_rsrc_t rn =
        make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)(more ? nxt : buf) * L::kBufBytes(N),
                  (more && !SCN_EXP_NO_LOADS) ? L::kBufBytes(N) : 0u);
    auto prefetch = [&](int a_lo, int a_hi) { load_group(rn, a_lo, a_hi); };  // groups of inputs a: two samples each
    prefetch(0, PF0);
    SCN_STAMP(2);  // issue of the first group of the next buffer's loads

    // ---- pass 1: virtual threads t and t + 256 ----
    if (PAIR) {
      fft16(va);
      fft16(vb);
#pragma unroll
      for (int p = 0; p < 16; p++) {
        cf ya = va[OUT16(p)], yb = vb[OUT16(p)];
        if (p) {
          ya = cmul(ya, tw1a[p]);
          if constexpr (TW1B) {
            yb = cmul(yb, tw1b[p]);
          } else {
            const float cr = (float)__builtin_cos(6.283185307179586476925286766559 * p / (double)N);
            const float sr = (float)__builtin_sin(6.283185307179586476925286766559 * p / (double)N);
            yb = cmul(yb, cmul(tw1a[p], cf{cr, -sr}));
          }
        }
        typedef float v4f_t __attribute__((ext_vector_type(4)));
        *reinterpret_cast<v4f_t *>(w1 + p * P1) = v4f_t{ya.x, ya.y, yb.x, yb.y};  // slots tau0, tau0 + 1: 16-byte aligned (P1 even)
      }
    } else {
      fft16(va);
#pragma unroll
      for (int p = 0; p < 16; p++) {
        cf y = va[OUT16(p)];
        if (p) y = cmul(y, tw1a[p]);
        w1[p * P1] = to_v2f(y);
      }
      fft16(vb);
#pragma unroll
      for (int p = 0; p < 16; p++) {
        cf y = vb[OUT16(p)];
        if (p) y = cmul(y, tw1b[p]);
        w1[p * P1 + T] = to_v2f(y);
      }
    }
    SCN_STAMP(3);  // pass 1 + exchange-1 writes
    __syncthreads();
    SCN_STAMP(4);  // barrier 1
    if (HITS) {
      if (t == 0 && prev != 0xffffffffu) {  // every wave is past the barrier: the previous buffer's recorders are done
        args.per_buffer_hits[prev] = (uint32_t)lds_hits[par ^ 1];
        if (args.host_hits) args.host_hits[prev] = (uint32_t)lds_hits[par ^ 1];
        lds_hits[par ^ 1] = 0;
      }
    }
    prefetch(PF0, PF1);

    // ---- pass 2: virtual threads (p2, c2) and (p2 + 8, c2) ----
#pragma unroll
    for (int b = 0; b < 16; b++) {
      va[b] = from_v2f(r1[b * M2]);
      vb[b] = from_v2f(r1[b * M2 + 8 * P1]);
    }
    fft16(va);
    fft16(vb);
#pragma unroll
    for (int q = 1; q < 16; q++) {
      const cf w = from_v2f(tw2[q * M2]);
      va[OUT16(q)] = cmul(va[OUT16(q)], w);
      vb[OUT16(q)] = cmul(vb[OUT16(q)], w);
    }
    SCN_STAMP(5);  // exchange-1 reads + pass 2 + twiddles (+ second load group)
    __syncthreads();  // every exchange-1 read done before the area is re-used
    SCN_STAMP(6);  // barrier 2
    prefetch(PF1, PFN);
#pragma unroll
    for (int q = 0; q < 16; q++) {
      w2[q * 16] = to_v2f(va[OUT16(q)]);
      w2[q * 16 + 8] = to_v2f(vb[OUT16(q)]);
    }
    SCN_STAMP(7);  // exchange-2 writes (+ third load group)
    __syncthreads();
    SCN_STAMP(8);  // barrier 3

    // K4: the thread's 32 LINEAR powers stay in `pw` (bin j = jbase + 256 r at index r) for the hit path; the spectrum gets the
    // dB map of scn_device.h -- product form inline, exact form stored over it for the strong bins in the waves that hold one
    // (see scn_fft_kernel)
    v32f pw;
    float gmax[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // the largest power of each group of eight outputs (max ignores NaN)
    __amdgpu_buffer_rsrc_t rout = make_rsrc(args.power_db + (size_t)buf * N, (SPEC && args.power_db && !SCN_EXP_NO_STORES) ? 4u * N : 0u);
    if constexpr (P3D) {
      // ---- pass 3 in DOUBLE (16384 points).  A strong tone's partial sums are largest in the last pass, and their float
      // rounding errors land on the 63 other bins of the tone's 64-point column (k = kl + 256 r): with peak / mean power of
      // 3e3 .. 7e3 that is the 0.7e-5 .. 1.05e-5 tail of the parity metric at this size -- ANY float32 FFT shows it
      // (scripts/emul_fused.py: this decomposition and pocketfft's have the same tail).  Evaluated in double from the float
      // exchange-2 values, the last six of the fourteen radix-2 levels stop contributing: the tail shrinks 3x.  v_add_f64 /
      // v_fma_f64 issue at 0.85x / 0.7x the float rate on gfx950 (scripts/ubench/f64_rate.hip), so the price is the
      // conversions and ~1.3x on a third of the arithmetic.
      // Lane half e holds the 32 values x[c''] = L2(c = 2c'' + e, kl).  Decimation in frequency, so that only 16 double
      // values are live at a time: A[c''] = x[c''] + x[c''+16] -> DFT16 -> Y_e[2 rho];  B[c''] = (x[c''] - x[c''+16]) W_32^c''
      // -> DFT16 -> Y_e[2 rho + 1].  Then X[r''] = Y_0[r''] + W_64^r'' Y_1[r''] (lower half), X[r'' + 32] = Y_0 - W Y_1 (upper):
      // only |X|^2 is needed, so BOTH halves rotate by half the angle -- Y_0 conj(W_128^r'') and Y_1 W_128^r'' (same cosine,
      // the sine's sign by lane half) -- and one v_permlane32_swap per 32-bit half of each component finishes it.
      // (every twiddle below is a wave-uniform compile-time constant -- scalar operands, no registers: the lane halves differ
      // by ONE sign flip per value instead.  |conj(h) Y_0 +- h Y_1| = |conj(h conj(Y_0)) +- h Y_1|: the lower half conjugates
      // its value, both halves multiply by h = W_128^r'', and the conjugation of the lower half's product is folded into the
      // signs of the last step.)
      const unsigned conj_lo = e ? 0u : 0x80000000u;  // xor mask of the imaginary part's high word: lower half conjugates
      const double sgn_x = e ? -1.0 : 1.0;             // upper half holds X[r'' + 32] = Y_0 - W Y_1
      const double sgn_y = -sgn_x;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        cd vd[16];
#pragma unroll
        for (int c = 0; c < 16; c++) {
          const cd x0 = to_cd(r3[(2 * c) * P2]), x1 = to_cd(r3[(2 * (c + 16)) * P2]);
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
          const int r = 2 * rho + h;  // r'': this lane's output block is r + 32 e
          cd y = vd[OUT16(rho)];
          unsigned long long by = __builtin_bit_cast(unsigned long long, y.y);
          if (r) {
            by ^= (unsigned long long)conj_lo << 32;
            y.y = __builtin_bit_cast(double, by);
            const double cr = __builtin_cos(6.283185307179586476925286766559 * r / 128.0), sr = __builtin_sin(6.283185307179586476925286766559 * r / 128.0);
            y = cmul_d(y, cr, -sr);
            by = __builtin_bit_cast(unsigned long long, y.y);
          }
          // both halves' values into both halves: [0] = the lower half's (A = h conj(Y_0); r'' = 0: Y_0 itself), [1] = the upper half's (B = h Y_1)
          const unsigned long long bx = __builtin_bit_cast(unsigned long long, y.x);
          auto sxl = __builtin_amdgcn_permlane32_swap((unsigned)bx, (unsigned)bx, false, false);
          auto sxh = __builtin_amdgcn_permlane32_swap((unsigned)(bx >> 32), (unsigned)(bx >> 32), false, false);
          auto syl = __builtin_amdgcn_permlane32_swap((unsigned)by, (unsigned)by, false, false);
          auto syh = __builtin_amdgcn_permlane32_swap((unsigned)(by >> 32), (unsigned)(by >> 32), false, false);
          const double ax = __builtin_bit_cast(double, ((unsigned long long)(unsigned)sxh[0] << 32) | (unsigned)sxl[0]);
          const double bxx = __builtin_bit_cast(double, ((unsigned long long)(unsigned)sxh[1] << 32) | (unsigned)sxl[1]);
          const double ay = __builtin_bit_cast(double, ((unsigned long long)(unsigned)syh[0] << 32) | (unsigned)syl[0]);
          const double byy = __builtin_bit_cast(double, ((unsigned long long)(unsigned)syh[1] << 32) | (unsigned)syl[1]);
          // X = conj(A) +- B (r'' = 0: A +- B): real parts A.x +- B.x; imaginary parts -A.y +- B.y, whose overall sign |X|^2 does not see
          const double xr = __builtin_fma(bxx, sgn_x, ax), xi = __builtin_fma(byy, r ? sgn_y : sgn_x, ay);
          const float q = (float)__builtin_fma(xi, xi, xr * xr);
          pw[r] = q;
          gmax[r >> 3] = fmaxf(gmax[r >> 3], q);
          if constexpr (SPEC) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, db_fast(q)), rout, st_voff, 1024u * r, AUX_ST);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the second half's loads and conversions out of the first half's live range
      }
    } else {
      // ---- pass 3.  M2 = 32: one 32-point DFT over c per thread: even c -> va, odd c -> vb.
      //             M2 = 64: this lane's half of a 64-point DFT: c = 4c' + e -> va, c = 4c' + e + 2 -> vb ----
#pragma unroll
      for (int c = 0; c < 16; c++) {
        va[c] = from_v2f(r3[(M2 == 64 ? 4 * c : 2 * c) * P2]);
        vb[c] = from_v2f(r3[(M2 == 64 ? 4 * c + 2 : 2 * c + 1) * P2]);
      }
      fft16(va);
      fft16(vb);
      // U, V = E[r'] +- W_32^r' O[r']: the outputs X[r'], X[r' + 16] themselves (M2 = 32), or this lane's share of them
      // (M2 = 64: one exchange across the wave's halves finishes the radix-4 step)
      const float sel = e ? 1.0f : 0.0f, sgn = e ? -1.0f : 1.0f;  // (M2 = 64)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const cf ev = va[OUT16(r)];
        cf od = vb[OUT16(r)];
        if (r) {
          const float cr = (float)__builtin_cos(6.283185307179586476925286766559 * r / 32.0);
          const float sr = (float)__builtin_sin(6.283185307179586476925286766559 * r / 32.0);
          od = cmul(od, cf{cr, -sr});
        }
        cf x0 = ev + od, x1 = ev - od;
        if constexpr (M2 == 64) {
          if (r) {  // W_64^r' in the odd half, 1 in the even half (branch-free: both halves run the same code)
            const float cr = (float)__builtin_cos(6.283185307179586476925286766559 * r / 64.0);
            const float sr = (float)__builtin_sin(6.283185307179586476925286766559 * r / 64.0);
            const cf w = cf{e ? cr : 1.0f, -sr * sel};
            x0 = cmul(x0, w);
            x1 = cmul(x1, w);
          }
          // v_permlane32_swap on two copies of a register leaves the lower half's value in [0] and the upper half's in [1], in BOTH halves
          auto ax = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x0.x), __builtin_bit_cast(unsigned, x0.x), false, false);
          auto ay = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x0.y), __builtin_bit_cast(unsigned, x0.y), false, false);
          auto bx = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x1.x), __builtin_bit_cast(unsigned, x1.x), false, false);
          auto by = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x1.y), __builtin_bit_cast(unsigned, x1.y), false, false);
          const cf u0 = cf{__builtin_bit_cast(float, (unsigned)ax[0]), __builtin_bit_cast(float, (unsigned)ay[0])};
          const cf u1 = cf{__builtin_bit_cast(float, (unsigned)ax[1]), __builtin_bit_cast(float, (unsigned)ay[1])};
          const cf v0 = cf{__builtin_bit_cast(float, (unsigned)bx[0]), __builtin_bit_cast(float, (unsigned)by[0])};
          const cf v1 = cf{__builtin_bit_cast(float, (unsigned)bx[1]), __builtin_bit_cast(float, (unsigned)by[1])};
          x0 = cf{__builtin_fmaf(u1.x, sgn, u0.x), __builtin_fmaf(u1.y, sgn, u0.y)};    // U_0 +- W U_1
          x1 = cf{__builtin_fmaf(v1.y, sgn, v0.x), __builtin_fmaf(v1.x, -sgn, v0.y)};   // V_0 -+ i W V_1
        }
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
    SCN_STAMP(9);  // exchange-2 reads + pass 3 + dB + stores issued
    if (t == 0) lds_next[0] = !more ? 0xffffffffu : DYN ? wq_buffer(taken) : nxt + gridDim.x;
    __syncthreads();  // exchange area free again; lds_next visible
    SCN_STAMP(10);  // barrier 4
#if SCN_STAMPS
    stamp_acc[11] += 1;
#endif
    const uint32_t after = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_next[0]);
    if (HITS) {
      if (__ballot(pmax > args.p_lo)) {
#if SCN_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t h0_ = (uint32_t)__builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
#endif
#if SCN_HITS_LANES_W
        scn_record_hits_lanes<32, false, true>(pw, gmax, keepmask, args, &lds_hits[par], buf, lane, [&](int r) -> uint32_t { return (jbase + 256u * (uint32_t)r) ^ (N / 2); });
#else
        scn_record_hits<32, false, true, SCN_ONE_ATOMIC_W>(pw, gmax, keepmask, args, &lds_hits[par], buf, lane, [&](int r) -> uint32_t { return (jbase + 256u * (uint32_t)r) ^ (N / 2); });
#endif
#if SCN_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        stamp_hit_cyc += (uint32_t)__builtin_readcyclecounter() - h0_;
        stamp_hit_n += 1;
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
      prev = buf;
      par ^= 1;
    }
    buf = nxt;
    nxt = after;
  }
#if SCN_STAMPS
  if (t == 0 && blockIdx.x < args.n_buffers && args.power_db) {
    __builtin_amdgcn_s_waitcnt(0);
    for (int i = 0; i < 12; i++) args.power_db[(size_t)blockIdx.x * N + i] = (float)stamp_acc[i];
    args.power_db[(size_t)blockIdx.x * N + 12] = (float)(stamp_t0 & 0xffffffu);
    args.power_db[(size_t)blockIdx.x * N + 13] = (float)((uint32_t)wall_clock64() & 0xffffffu);
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    args.power_db[(size_t)blockIdx.x * N + 14] = (float)(((xcc & 0xfu) << 8) | ((stamp_hw >> 8) & 0xffu));
    args.power_db[(size_t)blockIdx.x * N + 15] = (float)(stamp_entry & 0xffffffu);
    args.power_db[(size_t)blockIdx.x * N + 16] = (float)stamp_hit_cyc;
    args.power_db[(size_t)blockIdx.x * N + 17] = (float)stamp_hit_n;
  }
#endif
  if (HITS) {
    __syncthreads();  // last buffer's recorders done
    if (t == 0 && prev != 0xffffffffu) {
      args.per_buffer_hits[prev] = (uint32_t)lds_hits[par ^ 1];
      if (args.host_hits) args.host_hits[prev] = (uint32_t)lds_hits[par ^ 1];
    }
  }
}
}  // namespace

// ------------------------------------------------------------------------------------
// 16384 points as 32 x 16 x 32 (round 3): 512 threads x 32 points, one workgroup per CU.
//
//   n = 512 a + 32 b + c          k = p + 32 q + 512 r          a, p, c, r in [0, 32)   b, q in [0, 16)
//   pass 1  thread tau = 32 b + c:  32-pt DFT over a of x[512 a + tau] w[..] (two 16-pt DFTs + one radix-2 step, in
//           registers) -> * W_N^(tau p) -> LDS row p                                    L1(p, tau) = 512 p + tau
//   pass 2  threads (p, c) and (p + 16, c):  16-pt DFT over b -> * W_512^(c q) -> LDS    L2(c, kl) = 513 c + kl, kl = p + 32 q
//   pass 3  thread kl:  32-pt DFT over c -> X[kl + 512 r], r = 0 .. 31 -- a WHOLE DFT per thread
//
// The first form (scn_fft_wide_body<64>: 16 x 16 x 64) shares its 64-point last pass between two lanes: a twiddle multiply
// of every value by W_64^r', a v_permlane32_swap per register and a combining step -- 280 of its ~1700 VALU operations per
// thread and buffer.  Here pass 1 takes the extra radix-2 level instead (one more twiddle-free step on values that are in
// registers anyway): ~150 operations fewer, no cross-lane traffic, every output bin of a lane 512 apart.  All four LDS access
// patterns are conflict-free in the 16-lane groups a ds_*_b64 is served in (row pitch 513 on the transposed side).
// Price: one virtual thread per lane in pass 1, so the samples arrive one per load (32 loads of 8 / 4 / 2 bytes per
// lane instead of 16 of twice that).
// With SCN_16K_P3_DOUBLE pass 3 runs in double (decimation in frequency, 16 double values live at a time): the parity
// metric's tail at this size is the float rounding of a strong tone's partial sums in the LAST pass, which lands on the 31
// other bins of the tone's column -- any float32 FFT shows it (scripts/emul_fused.py) -- and goes away 3x when the last five
// radix-2 levels are exact.  v_add_f64 / v_fma_f64 issue at 0.85x / 0.7x the float rate (scripts/ubench/f64_rate.hip).
// ------------------------------------------------------------------------------------
#ifndef SCN_16K_V2
#define SCN_16K_V2 1
#endif
#ifndef SCN_16K2_FLOAT_PFN
#define SCN_16K2_FLOAT_PFN 16  // float input: how many of a buffer's 32 loads are prefetched across the previous buffer's passes
#endif
#ifndef SCN_16K2_INT_PFN
#define SCN_16K2_INT_PFN 32
#endif
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
  constexpr bool P3D = SCN_16K_P3_DOUBLE != 0;
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
  // loads prefetched, in three groups (planar int16 is two loads and a pack per sample: beside the double-precision pass 3 only half)
  constexpr int PFN = KIND == SCN_K_FLOAT_COMPLEX ? SCN_16K2_FLOAT_PFN : (P3D && KIND == SCN_K_SHORT) ? 16 : SCN_16K2_INT_PFN;
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
        make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)blockIdx.x * L::kBufBytes(N), SCN_EXP_NO_LOADS ? 0u : L::kBufBytes(N));
    load_group(r0, 0, PFN);
  }
  // persistent constants: W_N^(t p), p = 1 .. 31 (table rows of 512, scn_tw1_layout), window taps.  With pass 3 in double only
  // p = 1 .. 16 are kept and W_N^(t (p' + 16)) = W_N^(t p') W_N^(16 t) is formed per buffer: 15 more complex multiplies
  // (+3 % of the kernel's arithmetic) buy the 30 registers the double-precision values of pass 3 need -- spilled instead,
  // they cost 30 .. 90 % (scratch reloads at the top of every buffer).
  constexpr int TWN = P3D ? 17 : 32;
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
                                                  SCN_EXP_NO_LOADS ? 0u : L::kBufBytes(N));
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
                  (more && !SCN_EXP_NO_LOADS) ? L::kBufBytes(N) : 0u);
    load_group(rn, 0, PF0);

    // ---- pass 1: 32-point DFT over a: E, O = DFT16 of the even / odd samples, A[p'] = E + W_32^p' O, A[p' + 16] = E - W_32^p' O ----
    fft16(va);
    fft16(vb);
    cf tw16 = tw1[16];
    if constexpr (P3D) asm volatile("" : "+v"(tw16.x), "+v"(tw16.y));  // opaque per buffer: or hipcc hoists the 15 products out of the loop and keeps them in registers
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
      if constexpr (P3D) y1 = cmul(y1, p ? cmul(tw1[p], tw16) : tw16);
      else y1 = cmul(y1, tw1[p + 16]);
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
    __amdgpu_buffer_rsrc_t rout = make_rsrc(args.power_db + (size_t)buf * N, (SPEC && args.power_db && !SCN_EXP_NO_STORES) ? 4u * N : 0u);
    if constexpr (P3D) {
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
            if constexpr (HITS && SCN_HITS_MASK_16K) cand |= q > args.p_lo ? (1u << r) : 0u;
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the second half's loads and conversions out of the first half's live range
      }
    } else {
#pragma unroll
      for (int c = 0; c < 16; c++) {
        va[c] = from_v2f(r3[(2 * c) * P2]);
        vb[c] = from_v2f(r3[(2 * c + 1) * P2]);
      }
      fft16(va);
      fft16(vb);
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const cf ev = va[OUT16(r)];
        cf od = vb[OUT16(r)];
        if (r) {
          const float cr = (float)__builtin_cos(6.283185307179586476925286766559 * r / 32.0);
          const float sr = (float)__builtin_sin(6.283185307179586476925286766559 * r / 32.0);
          od = cmul(od, cf{cr, -sr});
        }
        const float p0 = power_of(ev + od), p1 = power_of(ev - od);
        gmax[r >> 3] = fmaxf(gmax[r >> 3], p0);
        gmax[2 + (r >> 3)] = fmaxf(gmax[2 + (r >> 3)], p1);
        if constexpr (SPEC) {
          const float d0 = db_fast(p0), d1 = db_fast(p1);
          pw[r] = d0;
          pw[r + 16] = d1;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d0), rout, st_voff, 2048u * r, AUX_ST);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d1), rout, st_voff, 2048u * (r + 16), AUX_ST);
        } else {
          pw[r] = p0;
          pw[r + 16] = p1;
          if constexpr (HITS && SCN_HITS_MASK_16K) cand |= (p0 > args.p_lo ? (1u << r) : 0u) | (p1 > args.p_lo ? (1u << (r + 16)) : 0u);
        }
      }
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
#if SCN_HITS_LANES_16K
        scn_record_hits_lanes<32, SPEC, false, !SPEC && SCN_HITS_MASK_16K != 0>(pw, gmax, keepmask, args, &lds_hits[par], buf, lane, [&](int r) -> uint32_t { return (t + 512u * (uint32_t)r) ^ (N / 2); }, cand);
#else
        scn_record_hits<32, SPEC, false, SCN_ONE_ATOMIC_16K>(pw, gmax, keepmask, args, &lds_hits[par], buf, lane, [&](int r) -> uint32_t { return (t + 512u * (uint32_t)r) ^ (N / 2); });
#endif
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

template <int KIND, bool DC, bool HITS, bool SPEC>
__global__ __launch_bounds__(256, 2) void scn_fft8k_kernel(ScnFftArgs args) {
  scn_fft_wide_body<32, KIND, DC, HITS, SPEC>(args);
}

template <int KIND, bool DC, bool HITS, bool SPEC>
__global__ __launch_bounds__(512, 2) void scn_fft16k_kernel(ScnFftArgs args) {
  scn_fft_wide_body<64, KIND, DC, HITS, SPEC>(args);
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
        make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)buf * L::kBufBytes(N), SCN_EXP_NO_LOADS ? 0u : L::kBufBytes(N));
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
#ifndef SCN_TD_WAVE
#define SCN_TD_WAVE 1
#endif
template <int KIND, bool DC>
__global__ __launch_bounds__(256) void scn_time_domain_wave_kernel(ScnTdArgs args) {
  typedef RawLoader<KIND> L;
  typedef int v4i_t __attribute__((__vector_size__(16)));
  constexpr bool PLANAR = KIND == SCN_K_SHORT;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t gw = blockIdx.x * 4u + (threadIdx.x >> 6), nw = gridDim.x * 4u;
  const uint32_t N = args.n;
  const uint32_t bytes = L::kBufBytes(N);
  const uint32_t stream_bytes = PLANAR ? 2u * N : bytes;  // planar: the I block, then the Q block
  const uint32_t chunks = stream_bytes / 16u;
  for (uint32_t buf = gw; buf < args.n_buffers; buf += nw) {
    const __amdgpu_buffer_rsrc_t rin =
        make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)buf * bytes, SCN_EXP_NO_LOADS ? 0u : bytes);
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
  if (SCN_TD_WAVE && a.n % 8u == 0) {  // one wave per buffer, 16-byte loads
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
template <int M, int KIND>
struct NarrowK {
  template <bool DC, bool HITS, bool SPEC>
  struct T {
    static constexpr void (*fn)(ScnFftArgs) = scn_fft_kernel<M, KIND, DC, HITS, SPEC>;
  };
};
template <int KIND>
struct Wide8K {
  template <bool DC, bool HITS, bool SPEC>
  struct T {
    static constexpr void (*fn)(ScnFftArgs) = scn_fft8k_kernel<KIND, DC, HITS, SPEC>;
  };
};
template <int KIND>
struct Wide16K {
  // (no hits-only specialisation at 16384 points: without the store instructions hipcc's allocation of this kernel's 256
  // registers falls apart -- 42 .. 75 VGPRs of persistent twiddles and window taps spilled and reloaded per buffer -- so a
  // hits-only plan runs the spectrum + hits kernel with a zero-record store descriptor, as all sizes did before round 3)
  template <bool DC, bool HITS, bool SPEC>
  struct T {
    static constexpr void (*fn)(ScnFftArgs) = scn_fft16k_kernel<KIND, DC, HITS, HITS ? true : SPEC>;
  };
};

template <int M, int KIND>
static hipError_t launch_kind(const ScnFftArgs &a, bool dc, bool hits, bool spec, int num_cus, hipStream_t s, hipEvent_t stop) {
  typedef Geo<M> G;
  void (*k)(ScnFftArgs) = pick_mode<NarrowK<M, KIND>::template T>(dc, hits, spec);
  if (G::LDS_BYTES > 65536u) {
    // > 64 KiB of dynamic LDS needs the opt-in; it is per function AND per device, and a process may
    // drive several GPUs (one plan per consumer thread), so it is simply set on every launch
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)G::LDS_BYTES);
    if (e != hipSuccess) return e;
  }
  int grid = num_cus * (int)G::WG_PER_CU;  // one resident wave of persistent workgroups
  if ((uint32_t)grid > a.n_buffers) grid = (int)a.n_buffers;
  return launch_with_stop(k, grid, G::T, G::LDS_BYTES, s, stop, a);
}

template <int M>
static hipError_t launch_size(int kind, bool dc, bool hits, bool spec, const ScnFftArgs &args, int num_cus, hipStream_t stream, hipEvent_t stop) {
  switch (kind) {
    case SCN_K_FLOAT_COMPLEX: return launch_kind<M, SCN_K_FLOAT_COMPLEX>(args, false, hits, spec, num_cus, stream, stop);
    case SCN_K_SHORT_COMPLEX: return launch_kind<M, SCN_K_SHORT_COMPLEX>(args, dc, hits, spec, num_cus, stream, stop);
    case SCN_K_SHORT: return launch_kind<M, SCN_K_SHORT>(args, dc, hits, spec, num_cus, stream, stop);
    case SCN_K_BYTE_COMPLEX: return launch_kind<M, SCN_K_BYTE_COMPLEX>(args, dc, hits, spec, num_cus, stream, stop);
    default: return hipErrorInvalidValue;
  }
}

template <int M2, void (*(*PICK)(bool, bool, bool))(ScnFftArgs)>
static hipError_t launch_wide(const ScnFftArgs &a, bool dc, bool hits, bool spec, int num_cus, hipStream_t s, hipEvent_t stop) {
  typedef GeoWide<M2> G;
  void (*k)(ScnFftArgs) = PICK(dc, hits, spec);
  // > 64 KiB of dynamic LDS: opt-in per function and per device, set on every launch (see launch_kind)
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
  if (e != hipSuccess) return e;
  int grid = num_cus * (int)G::WG_PER_CU;
  if ((uint32_t)grid > a.n_buffers) grid = (int)a.n_buffers;
  return launch_with_stop(k, grid, G::T, G::LDS_BYTES, s, stop, a);
}
template <int KIND>
static hipError_t launch_8k_kind(const ScnFftArgs &a, bool dc, bool hits, bool spec, int num_cus, hipStream_t s, hipEvent_t stop) {
  return launch_wide<32, pick_mode<Wide8K<KIND>::template T>>(a, dc, hits, spec, num_cus, s, stop);
}
template <int KIND>
struct Wide16K2 {
  template <bool DC, bool HITS, bool SPEC>
  struct T {
    static constexpr void (*fn)(ScnFftArgs) = scn_fft16k2_kernel<KIND, DC, HITS, SPEC>;
  };
};
template <int KIND>
static hipError_t launch_16k_kind(const ScnFftArgs &a, bool dc, bool hits, bool spec, int num_cus, hipStream_t s, hipEvent_t stop) {
#if SCN_16K_V2
  typedef Geo16k2 G;
  void (*k)(ScnFftArgs) = pick_mode<Wide16K2<KIND>::template T>(dc, hits, spec);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
  if (e != hipSuccess) return e;
  int grid = num_cus * (int)G::WG_PER_CU;
  if ((uint32_t)grid > a.n_buffers) grid = (int)a.n_buffers;
  return launch_with_stop(k, grid, G::T, G::LDS_BYTES, s, stop, a);
#else
  return launch_wide<64, pick_mode<Wide16K<KIND>::template T>>(a, dc, hits, spec, num_cus, s, stop);
#endif
}
static hipError_t launch_16k(int kind, bool dc, bool hits, bool spec, const ScnFftArgs &args, int num_cus, hipStream_t stream, hipEvent_t stop) {
  switch (kind) {
    // us per 2048-buffer launch, scn_fft_kernel<64> -> wide form: integer formats 94.8 -> 88.5 (scripts/wide16k_check.sh); float
    // input 111.5 -> 101.8 with HALF of a buffer prefetched (its 64 prefetch registers do not fit beside the lane-pair step:
    // 113 us with all 16 two-sample loads prefetched and 22 VGPRs spilled; 0 / 4 / 6 / 8 / 10 / 12 / 14 / 16 prefetched:
    // 110.9 / 107.3 / 105.9 / 101.8 / 105.7 / 108.5 / 107.1 / 113.1, scripts/wide16k_float_check.sh)
#if SCN_WIDE_16384_FLOAT
    case SCN_K_FLOAT_COMPLEX: return launch_16k_kind<SCN_K_FLOAT_COMPLEX>(args, false, hits, spec, num_cus, stream, stop);
#else
    case SCN_K_FLOAT_COMPLEX: return launch_size<64>(kind, false, hits, spec, args, num_cus, stream, stop);
#endif
    case SCN_K_SHORT_COMPLEX: return launch_16k_kind<SCN_K_SHORT_COMPLEX>(args, dc, hits, spec, num_cus, stream, stop);
    case SCN_K_SHORT: return launch_16k_kind<SCN_K_SHORT>(args, dc, hits, spec, num_cus, stream, stop);
    case SCN_K_BYTE_COMPLEX: return launch_16k_kind<SCN_K_BYTE_COMPLEX>(args, dc, hits, spec, num_cus, stream, stop);
    default: return hipErrorInvalidValue;
  }
}
static hipError_t launch_8k(int kind, bool dc, bool hits, bool spec, const ScnFftArgs &args, int num_cus, hipStream_t stream, hipEvent_t stop) {
  switch (kind) {
    case SCN_K_FLOAT_COMPLEX: return launch_8k_kind<SCN_K_FLOAT_COMPLEX>(args, false, hits, spec, num_cus, stream, stop);
    case SCN_K_SHORT_COMPLEX: return launch_8k_kind<SCN_K_SHORT_COMPLEX>(args, dc, hits, spec, num_cus, stream, stop);
    case SCN_K_SHORT: return launch_8k_kind<SCN_K_SHORT>(args, dc, hits, spec, num_cus, stream, stop);
    case SCN_K_BYTE_COMPLEX: return launch_8k_kind<SCN_K_BYTE_COMPLEX>(args, dc, hits, spec, num_cus, stream, stop);
    default: return hipErrorInvalidValue;
  }
}

template <int M, int KIND>
struct SmallK {
  template <bool DC, bool HITS, bool SPEC>
  struct T {
    static constexpr void (*fn)(ScnFftArgs) = scn_fft_small_kernel<M, KIND, DC, HITS, SPEC>;
  };
};
template <int M, int KIND>
static hipError_t launch_small_kind(const ScnFftArgs &a, bool dc, bool hits, bool spec, int num_cus, hipStream_t s, hipEvent_t stop) {
  typedef GeoSmall<M> G;
  void (*k)(ScnFftArgs) = pick_mode<SmallK<M, KIND>::template T>(dc, hits, spec);
  const uint32_t groups = (a.n_buffers + G::SLOTS - 1u) / G::SLOTS;
  int grid = num_cus * (int)G::WG_PER_CU;
  if ((uint32_t)grid > groups) grid = (int)groups;
  return launch_with_stop(k, grid, 256u, G::LDS_BYTES, s, stop, a);
}
template <int M>
static hipError_t launch_small(int kind, bool dc, bool hits, bool spec, const ScnFftArgs &args, int num_cus, hipStream_t stream, hipEvent_t stop) {
  if ((uint64_t)args.n_buffers * 256u * M * 8u > 0xffffffffull) return hipErrorInvalidValue;  // (one descriptor spans SLOTS buffers only: never near)
  switch (kind) {
    case SCN_K_FLOAT_COMPLEX: return launch_small_kind<M, SCN_K_FLOAT_COMPLEX>(args, false, hits, spec, num_cus, stream, stop);
    case SCN_K_SHORT_COMPLEX: return launch_small_kind<M, SCN_K_SHORT_COMPLEX>(args, dc, hits, spec, num_cus, stream, stop);
    case SCN_K_SHORT: return launch_small_kind<M, SCN_K_SHORT>(args, dc, hits, spec, num_cus, stream, stop);
    case SCN_K_BYTE_COMPLEX: return launch_small_kind<M, SCN_K_BYTE_COMPLEX>(args, dc, hits, spec, num_cus, stream, stop);
    default: return hipErrorInvalidValue;
  }
}

hipError_t scn_launch_fft(uint32_t n, int kind, bool dc, bool hits, bool spec, const ScnFftArgs &args, int num_cus,
                          hipStream_t stream, hipEvent_t stop) {
  if (!hits && !spec) return hipErrorInvalidValue;
  if (args.n_buffers == 0) return stop ? hipEventRecord(stop, stream) : hipSuccess;
  switch (n) {
    case 256: return launch_small<1>(kind, dc, hits, spec, args, num_cus, stream, stop);
    case 512: return launch_small<2>(kind, dc, hits, spec, args, num_cus, stream, stop);
    case 1024: return launch_size<4>(kind, dc, hits, spec, args, num_cus, stream, stop);
    case 2048: return launch_size<8>(kind, dc, hits, spec, args, num_cus, stream, stop);
    case 4096: return launch_size<16>(kind, dc, hits, spec, args, num_cus, stream, stop);
#if SCN_WIDE_8192
    case 8192: return launch_8k(kind, dc, hits, spec, args, num_cus, stream, stop);
#else
    case 8192: return launch_size<32>(kind, dc, hits, spec, args, num_cus, stream, stop);  // the 512-thread form (variant build)
#endif
#if SCN_WIDE_16384
    case 16384: return launch_16k(kind, dc, hits, spec, args, num_cus, stream, stop);
#else
    case 16384: return launch_size<64>(kind, dc, hits, spec, args, num_cus, stream, stop);  // the 1024-thread form (variant build)
#endif
    default: return hipErrorInvalidValue;
  }
}

bool scn_fft_size_supported(uint32_t n) { return n == 256 || n == 512 || n == 1024 || n == 2048 || n == 4096 || n == 8192 || n == 16384; }

// layout of ScnFftArgs::tw1_table for size n: `rows` rows of `threads` entries, row p-1 = W_n^(t p): 15 rows of n/16 for the
// 16 x 16 x M kernels, 31 rows of 512 for the 32 x 16 x 32 form of 16384 points
void scn_tw1_layout(uint32_t n, uint32_t *rows, uint32_t *threads) {
  const bool v2 = n == 16384 && SCN_16K_V2 != 0 && SCN_WIDE_16384 != 0;
  *rows = v2 ? 31u : 15u;
  *threads = v2 ? 512u : n / 16u;
}
