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

// ---- K4: the dB map, 10*log2(sqrt(P))/log2(10) == (5/log2(10)) * log2(P)  (utility.cpp:86-98) ---------------------------
// The reference evaluates it in double and rounds once (its <cmath> log2 is the double one), i.e. its output is the
// correctly rounded dB value of a float |X|.  v_log_f32 is good to ~1 ulp OF ITS RESULT, and log2(P) of a strong bin is a
// large number: the product form k * v_log_f32(P) measures up to 2.2 ulp of the dB value (scripts/ubench/log_acc.hip on
// MI355X) -- at 35 dB that alone is a 2.4e-6 relative power error, a quarter of the parity bar and the whole MEDIAN of the
// per-buffer maximum error of round 2 (1.76e-6 at every size; an exactly rounded map leaves 0.88e-6, the float quantisation
// of a 32..64 dB value).  The exact form splits the exponent off first (log2 of the mantissa is in [-1, 0): its ulp is
// 6e-8) and adds k * e in two exactly representable pieces: <= 1.0 ulp, rms 0.38 (correct rounding: 0.5 / 0.29).  It costs
// five more VALU operations per bin, so the map is DEFINED as: the product form for powers below SCN_P_EXACT_FROM = 10^3.2
// (16 dB: its error there is <= 2.2 ulp of a value below 16, i.e. 2e-6 dB or 1e-6 in relative power), the exact form from
// there up -- db_of_power, a pure function of the bin's power, the same in every kernel and output mode; the kernels
// evaluate the exact form only in waves that hold such a bin (strong signals), under wave-uniform branches.
// ONE exception, the 16384-point kernel: it keeps the dB values (not the powers) for its hit path -- at that size a
// threshold near the noise floor makes every wave a hit wave, and re-deriving the values cost 15 % -- and so has the power
// only of each thread-group's maximum at hand.  Its map, in all three output modes (identical thread layout): a bin of a
// group of eight outputs of a thread whose maximum power gm is at least SCN_P_EXACT_FROM gets db_exact(gm) iff its
// product-form value equals db_fast(gm) -- i.e. the group's strongest bin (a tone's strongest bins, one per thread), plus
// any bin of the group within the product form's resolution of it; every other bin keeps the product form's <= 2.2 ulp.
// Spectrum + hits and hits-only plans therefore report bit-identical records (tests/test_dispatch_gpu.py).
#define SCN_P_EXACT_FROM 1584.8932f
__device__ __forceinline__ float db_fast(float p) { return 1.50514997831990597607f * __builtin_amdgcn_logf(p); }
__device__ __forceinline__ float db_exact(float p) {
  const float m = __builtin_amdgcn_frexp_mantf(p);          // [1/2, 1); 0, inf and nan come back unchanged
  const float e = (float)__builtin_amdgcn_frexp_expf(p);
  const float KH = 1.505126953125f;                          // k to 12 significant bits: KH * e is exact (|e| < 256)
  const float KL = (float)(1.50514997831990597607 - 1.505126953125);
  return __builtin_fmaf(KL, e, __builtin_fmaf(1.50514997831990597607f, __builtin_amdgcn_logf(m), KH * e));
}
// the map itself, one bin
__device__ __forceinline__ float db_of_power(float p) {
  const float lo = db_fast(p), hi = db_exact(p);  // both evaluated, then ONE select: no per-bin branch (the callers sit behind a wave-uniform one)
  return p >= SCN_P_EXACT_FROM ? hi : lo;
}
__device__ __forceinline__ float power_of(cf x) { return __builtin_fmaf(x.y, x.y, x.x * x.x); }
__device__ __forceinline__ float power_db(cf x) { return db_of_power(power_of(x)); }

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

// ---- double-precision twins (the staged path of scn_generic.hip; pass 3 of the 16384-point kernel) --------------------
// v_add_f64 issues at 0.85x, v_mul_f64 / v_fma_f64 at 0.7x the rate of v_fma_f32 on gfx950 (scripts/ubench/f64_rate.hip):
// a double butterfly costs ~1.3x a float one, not 2x or 16x.
typedef double scn_v2d __attribute__((ext_vector_type(2)));
struct cd {
  double x, y;
};
__device__ __forceinline__ cd operator+(cd a, cd b) { return cd{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd operator-(cd a, cd b) { return cd{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd from_v2d(scn_v2d v) { return cd{v.x, v.y}; }
__device__ __forceinline__ scn_v2d to_v2d(cd c) { return scn_v2d{c.x, c.y}; }
__device__ __forceinline__ cd to_cd(v2f v) { return cd{(double)v.x, (double)v.y}; }
__device__ __forceinline__ cd cmul_d(cd a, scn_v2d w) { return cd{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
__device__ __forceinline__ cd cmul_d(cd a, double wx, double wy) {
  return cd{__builtin_fma(-a.y, wy, a.x * wx), __builtin_fma(a.y, wx, a.x * wy)};
}
// DFT4 with W4 = -i, results left in (x0, x1, x2, x3) = (X0, X1, X2, X3)
__device__ __forceinline__ void radix4_d(cd &x0, cd &x1, cd &x2, cd &x3) {
  const cd t0 = x0 + x2, t1 = x0 - x2, t2 = x1 + x3, t3 = x1 - x3;
  x0 = t0 + t2;
  x2 = t0 - t2;
  x1 = cd{t1.x + t3.y, t1.y - t3.x};  // t1 - i*t3
  x3 = cd{t1.x - t3.y, t1.y + t3.x};  // t1 + i*t3
}
// a * (1 - i) * h  and  a * (-1 - i) * h  (W16^2, W16^6 with h = sqrt(1/2))
__device__ __forceinline__ cd mul_w2_d(cd a, double h) { return cd{(a.x + a.y) * h, (a.y - a.x) * h}; }
__device__ __forceinline__ cd mul_w6_d(cd a, double h) { return cd{(a.y - a.x) * h, -(a.x + a.y) * h}; }
// In-register 16-point forward DFT (radix 4 x 4), in double: on return X[k] sits in v[OUT16(k)] like the float fft16.
__device__ __forceinline__ void fft16_d(cd v[16]) {
  const double C1 = 0.92387953251128675613, S1 = 0.38268343236508977173, H = 0.70710678118654752440;
#pragma unroll
  for (int n0 = 0; n0 < 4; n0++) radix4_d(v[n0], v[n0 + 4], v[n0 + 8], v[n0 + 12]);
  // v[n0 + 4 k0] *= W16^(n0 k0)
  v[5] = cmul_d(v[5], C1, -S1);                          // W^1
  v[9] = mul_w2_d(v[9], H);                              // W^2
  v[13] = cmul_d(v[13], S1, -C1);                        // W^3
  v[6] = mul_w2_d(v[6], H);                              // W^2
  v[10] = cd{v[10].y, -v[10].x};                         // W^4 = -i
  v[14] = mul_w6_d(v[14], H);                            // W^6
  v[7] = cmul_d(v[7], S1, -C1);                          // W^3
  v[11] = mul_w6_d(v[11], H);                            // W^6
  v[15] = cmul_d(v[15], -C1, S1);                        // W^9
#pragma unroll
  for (int k0 = 0; k0 < 4; k0++) radix4_d(v[4 * k0], v[4 * k0 + 1], v[4 * k0 + 2], v[4 * k0 + 3]);
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

// ---- K5, the recording half (process.cpp:54-57), shared by the fused kernels ------------------------------------------
// `pw` holds the LINEAR powers of this thread's NB bins (bin index i of output o from `bin_i`), `gmax` the largest power of
// each of its four groups of NB / 4 outputs.  Candidates are found in the linear domain against p_lo (a shade below
// 10^(threshold / 5), scn_hit_prefilter); the decision itself is magnitudes[j] > m_threshold on the dB value -- the map above,
// a pure function of the power, so spectrum + hits and hits-only plans decide identically and report the float the spectrum
// holds.  (`args`: anything with p_lo, threshold, hits, hit_region.)
//
// scn_record_hits_device_counter: the form for a hit counter in DEVICE memory (scn_big.hip: a buffer is spread over many
// workgroups), where a returning atomic costs microseconds, not ~100 cycles.  A tone's detections are neighbouring bins,
// i.e. ONE or two output indices of neighbouring lanes: the wave tests its four groups, then the outputs of a group that
// holds a candidate (one compare each, into a scalar mask), evaluates every output index in that mask once (values written
// back into `pw`, hit lanes and the count accumulated), takes its slots with ONE atomic and records in a second pass over
// the indices that had hits.
template <int NB, typename VEC, typename ARGS, typename BINI>
__device__ __forceinline__ void scn_record_hits_device_counter(VEC &pw, const float (&gmax)[4], uint32_t keepmask, const ARGS &args, int *count,
                                                               uint32_t buf, uint32_t lane, BINI bin_i) {
  constexpr int GS = NB / 4;
  uint32_t wmc = 0;  // output indices with a candidate somewhere in the wave (a scalar mask: no per-lane masks, no cross-lane reduction)
#pragma unroll
  for (int g = 0; g < 4; g++) {
    if (__ballot(gmax[g] > args.p_lo)) {
#pragma unroll
      for (int o = g * GS; o < (g + 1) * GS; o++) wmc |= __ballot(pw[o] > args.p_lo) ? (1u << o) : 0u;
    }
  }
  ScnDevHit *const region = args.hits + (size_t)buf * args.hit_region;
  uint32_t hm = 0, wm2 = 0, total = 0;  // this lane's hits, the indices with hits, the wave's count
  while (wmc) {  // a tone's main lobe: one or two indices
    const int o = __builtin_ctz(wmc);  // wave-uniform
    wmc &= wmc - 1u;
    const float p = pw[o];
    float d = db_fast(p);
    if (__ballot(p >= SCN_P_EXACT_FROM)) d = p >= SCN_P_EXACT_FROM ? db_exact(p) : d;  // = db_of_power(p)
    const bool hit = ((keepmask >> o) & 1u) && d > args.threshold && p > args.p_lo;  // strict >, process.cpp:54
    const unsigned long long m = __ballot(hit);
    pw[o] = d;
    hm |= hit ? (1u << o) : 0u;
    wm2 |= m ? (1u << o) : 0u;
    total += (uint32_t)__popcll(m);
  }
  if (!total) return;
  uint32_t base = 0;
  if (lane == 0) base = (uint32_t)atomicAdd(count, (int)total);
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
  while (wm2) {
    const int o = __builtin_ctz(wm2);  // wave-uniform
    wm2 &= wm2 - 1u;
    const bool hit = (hm >> o) & 1u;
    const unsigned long long m = __ballot(hit);
    if (hit) {
      const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if (pos < args.hit_region) region[pos] = ScnDevHit{bin_i(o), pw[o]};
    }
    base += (uint32_t)__popcll(m);
  }
}

// scn_record_hits_lanes: the form for a counter in LDS (the fused kernels: one workgroup per buffer).  Every lane collects ITS
// candidate outputs in a bit mask with straight-line VALU work (no wave-wide test per output index), then the wave takes one
// candidate per lane per trip -- the value comes out of the registers through a select tree, not a register-indexed move --
// until no lane has one left.  Trips = the largest number of candidates any one lane holds (a tone's main lobe sits in
// neighbouring lanes of one output index: 1; noise hits at the bench's threshold: 2..3).  (The form it replaced walked the
// output indices that held a candidate anywhere in the wave -- a dozen of 32 on a hit-dense 16384-point launch: 101 us
// against 86; profiles/r03_experiments.md section 6.)
// IS_DB: `pw` already holds the dB values (the 16384-point spectrum kernel); otherwise the powers, and the value is formed
// here: by db_of_power (PURE), or by the 16384-point kernel's rule -- the exact half of the map for the strong maximum of
// the thread's group of outputs only.
// (v_cndmask_b32 spelled out: written as C++ selects, hipcc recognises "element o of a vector" and lowers THAT as a chain of
// 32 compares + selects, several times over -- 1400 instructions per trip)
__device__ __forceinline__ float scn_cndmask(float a, float b, unsigned long long m) {  // lane in m ? b : a
  float r;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
  return r;
}
template <int NB, typename VEC>
__device__ __forceinline__ float scn_select_output(const VEC &pw, uint32_t o) {  // all lanes active
  float t[NB / 2];
  const unsigned long long m0 = __ballot(o & 1u);
#pragma unroll
  for (int i = 0; i < NB / 2; i++) t[i] = scn_cndmask(pw[2 * i], pw[2 * i + 1], m0);
#pragma unroll
  for (int w = NB / 4, bit = 1; w >= 1; w >>= 1, bit++) {
    const unsigned long long m = __ballot((o >> bit) & 1u);
#pragma unroll
    for (int i = 0; i < w; i++) t[i] = scn_cndmask(t[2 * i], t[2 * i + 1], m);
  }
  return t[0];
}

// scn_record_hits_segment: the form for the kernels in which a wave holds SEVERAL buffers (16 .. 512 points): the hits of one
// buffer sit in T consecutive lanes (T = 1 .. 32, a power of two, aligned).  `dbv` holds the dB values of the lane's NB outputs
// (db_of_power), `hm` its hit mask (dB > threshold, keep mask applied).  A record's slot in the buffer's region is the number
// of hits in the lanes before it (an inclusive prefix sum over the segment: log2 T cross-lane reads) plus its rank in its own
// lane -- no atomic, no counter in LDS; the segment's total is the buffer's count.  (Until round 4 every hit took its slot
// with its own LDS atomic on the buffer's counter -- the hits of one buffer on ONE address -- and evaluated its dB value in
// the trip: 30 of a 256-point launch's 82 us with the bench's eight hits per buffer.)  Region order does not matter: the
// compaction kernel ranks a buffer's records by their bins (scn_hits.hip).
template <int T, int NB, typename VEC, typename ARGS, typename BINI>
__device__ __forceinline__ uint32_t scn_record_hits_segment(const VEC &dbv, uint32_t hm, const ARGS &args, uint32_t buf, uint32_t t, BINI bin_i) {
  static_assert(T >= 1 && T <= 32 && (T & (T - 1)) == 0, "a segment is a power of two of lanes inside a wave");
  const uint32_t cnt = (uint32_t)__builtin_popcount(hm);
  uint32_t inc = cnt;
#pragma unroll
  for (int off = 1; off < T; off <<= 1) {
    const uint32_t v = (uint32_t)__shfl_up((int)inc, off, T);
    inc += t >= (uint32_t)off ? v : 0u;
  }
  const uint32_t total = T == 1 ? inc : (uint32_t)__shfl((int)inc, T - 1, T);
  uint32_t pos = inc - cnt;
  ScnDevHit *const region = args.hits + (size_t)buf * args.hit_region;
  while (__ballot(hm != 0u)) {  // one hit per lane per trip; the value comes out of the registers through the select tree
    const bool act = hm != 0u;
    const uint32_t o = act ? (uint32_t)__builtin_ctz(hm) : 0u;
    hm &= hm - 1u;
    const float d = scn_select_output<NB>(dbv, o);
    if (act && pos < args.hit_region) region[pos] = ScnDevHit{bin_i((int)o), d};
    pos += 1u;
  }
  return total;
}

// HAVE_MASK: the caller collected the candidate bits (`cand`, before the keep mask) while it produced the outputs.
template <int NB, bool IS_DB, bool PURE = false, bool HAVE_MASK = false, typename VEC, typename ARGS, typename BINI>
__device__ __forceinline__ void scn_record_hits_lanes(VEC &pw, const float (&gmax)[4], uint32_t keepmask, const ARGS &args, int *count, uint32_t buf,
                                                      uint32_t lane, BINI bin_i, uint32_t cand = 0) {
  constexpr int GS = NB / 4;
  uint32_t hm = cand;
  if constexpr (!HAVE_MASK) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
      if (__ballot(gmax[g] > args.p_lo)) {
#pragma unroll
        for (int o = g * GS; o < (g + 1) * GS; o++) hm |= (IS_DB ? pw[o] > args.threshold : pw[o] > args.p_lo) ? (1u << o) : 0u;
      }
    }
  }
  hm &= keepmask;
  ScnDevHit *const region = args.hits + (size_t)buf * args.hit_region;
  if constexpr (IS_DB) {
    // the outputs ARE the dB values: every candidate bit is a hit, the wave's count is known before the first record -- one
    // prefix sum over the lanes' counts, ONE counter atomic, then the trips carry no wave-wide step at all
    const uint32_t cnt = (uint32_t)__builtin_popcount(hm);
    uint32_t inc = cnt;
    SCN_DPP_STEP(+, inc, 0x111, 0xf);
    SCN_DPP_STEP(+, inc, 0x112, 0xf);
    SCN_DPP_STEP(+, inc, 0x114, 0xf);
    SCN_DPP_STEP(+, inc, 0x118, 0xf);
    SCN_DPP_STEP(+, inc, 0x142, 0xa);
    SCN_DPP_STEP(+, inc, 0x143, 0xc);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    if (!total) return;
    uint32_t base = 0;
    if (lane == 0) base = (uint32_t)atomicAdd(count, (int)total);
    uint32_t pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + inc - cnt;
    while (__ballot(hm != 0u)) {
      const bool act = hm != 0u;
      const uint32_t o = act ? (uint32_t)__builtin_ctz(hm) : 0u;
      hm &= hm - 1u;
      const float d = scn_select_output<NB>(pw, o);
      if (act && pos < args.hit_region) region[pos] = ScnDevHit{bin_i((int)o), d};
      pos += 1u;
    }
    return;
  }
  while (__ballot(hm != 0u)) {
    const bool act = hm != 0u;
    const uint32_t o = act ? (uint32_t)__builtin_ctz(hm) : 0u;
    hm &= hm - 1u;
    const float p = scn_select_output<NB>(pw, o);
    float d = p;
    if constexpr (!IS_DB) {
      d = db_fast(p);
      if constexpr (PURE) {
        const bool ex = p >= SCN_P_EXACT_FROM;
        if (__ballot(act && ex)) d = ex ? db_exact(p) : d;  // = db_of_power(p)
      } else {
        // the 16384-point kernel's map, EXACTLY as its spectrum kernel applies it (which holds dB values, not powers, at this
        // point): a bin whose product-form value equals that of its thread-group's strong maximum gets the exact form of that
        // maximum -- so the two output modes report the same float for every bin, also when two distinct powers of a group
        // share a product-form value
        // (through the select tree, not `gmax[o / GS]`: hipcc turns the C++ selects into an indexed read of a PRIVATE array --
        //  32 bytes of scratch, a scratch store per buffer and a scratch load per trip in every hits-only 16384-point kernel)
        const float gm = scn_select_output<4>(gmax, o / GS);
        const bool ex = gm >= SCN_P_EXACT_FROM && d == db_fast(gm);
        if (__ballot(act && ex)) d = ex ? db_exact(gm) : d;
      }
    }
    const bool hit = act && d > args.threshold && (IS_DB || p > args.p_lo);  // strict >, process.cpp:54
    const unsigned long long m = __ballot(hit);
    if (m) {
      const int first = __builtin_ctzll(m);
      uint32_t base = 0;
      if (lane == (uint32_t)first) base = (uint32_t)atomicAdd(count, (int)__popcll(m));
      base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
      if (hit) {
        const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (pos < args.hit_region) region[pos] = ScnDevHit{bin_i((int)o), d};
      }
    }
  }
}

// ---- global memory access through buffer descriptors ---------------------------------
// A raw buffer resource (SGPR descriptor, wave-uniform base) + one per-lane VGPR offset
// + scalar/immediate offsets: the 16 strided accesses of a thread cost no address VGPRs.
//
// Cache-policy immediates of the buffer instructions on gfx950: bit0 = sc0, bit1 = nt, bit4 = sc1.
// Both streams are touched exactly once, so both are non-temporal: measured with inputs AND outputs
// rotated over 1.5 GiB (nothing can live in the 256 MiB Infinity Cache), a no-compute skeleton
// of this kernel's traffic moves 5.66 TB/s with the default policy and 6.40 TB/s with nt on both
// (scripts/membw.hip); the FFT kernel itself gains 4-5 %.
constexpr int SCN_AUX_LD = 2;
constexpr int SCN_AUX_ST = 2;

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
    // (only bytes 0 and 1 of the register are ever read -- by the SDWA converts of conv() --, so the 16-bit load is taken as it
    // comes: widening it in C++ costs a v_and_b32 per sample that buffer_load_ushort has already done)
    int v;
    const unsigned short h = __builtin_amdgcn_raw_buffer_load_b16(r, t * 2u, idx0 * 2u, AUX);
    asm("" : "=v"(v) : "0"(h));
    return v;
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
