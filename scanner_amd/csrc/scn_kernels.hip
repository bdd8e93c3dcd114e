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
#include <stdint.h>
#include <stdlib.h>

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

// ---- global memory access through buffer descriptors ---------------------------------
// A raw buffer resource (SGPR descriptor, wave-uniform base) + one per-lane VGPR offset
// + scalar/immediate offsets: the 16 strided accesses of a thread cost no address VGPRs.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

// Tunables (compile-time; scripts/build_variants.py builds one library per setting).
// Cache-policy immediates of the buffer instructions on gfx950: bit0 = sc0, bit1 = nt, bit4 = sc1.
#ifndef SCN_AUX_LD
#define SCN_AUX_LD 0   // streaming loads: default policy (nt here loses when the stores are nt too)
#endif
#ifndef SCN_AUX_ST
#define SCN_AUX_ST 2   // streaming dB stores: non-temporal -- the output is never re-read by the
                       // kernel, keeping it out of L2/MALL is worth 15 % (87 -> 74 us on C2)
#endif
#ifndef SCN_PREFETCH
#define SCN_PREFETCH 0 // 1: fetch the next buffer's raw samples into registers during the FFT
                       //    (145 VGPRs -> 3 workgroups per CU)
#endif
#ifndef SCN_WG_PER_CU
#define SCN_WG_PER_CU (SCN_PREFETCH ? 3 : 4)
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
  static __device__ __forceinline__ void ints(raw_t r, int &re, int &im) {
    RawLoader<SCN_K_SHORT_COMPLEX>::ints(r, re, im);
  }
  static __device__ __forceinline__ cf conv(raw_t r, int dc_re, int dc_im, float scale) {
    return RawLoader<SCN_K_SHORT_COMPLEX>::conv(r, dc_re, dc_im, scale);
  }
};

// sum over the 64 lanes of a wave (result in every lane)
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace

// ------------------------------------------------------------------------------------
// 4096-point fused kernel: one 256-thread workgroup per buffer, persistent over buffers.
//
// LDS (all offsets in complex = 8 B units, every address = per-thread base + immediate):
//   exchange 1   L1(p, col)  = p*272 + col          16 rows of 256, row pitch padded by 16
//       write (pass 1): fixed p, lanes col = t            -> 64 consecutive slots
//       read  (pass 2): thread (p=hi, c=lo), fixed b      -> hi*272 + b*16 + lo; the two rows a
//                       32-lane group touches sit 128 B apart in bank space -> conflict-free
//   exchange 2   L2(c, q, p) = c*257 + q*16 + p      row pitch padded by 1
//       write (pass 2): thread (p=hi, c=lo), fixed q      -> lanes 8 B apart mod 256 B
//       read  (pass 3): thread (p=lo, q=hi), fixed c      -> 64 consecutive slots
//   then the pass-2 twiddle table [q][c] (2 KiB), 8 ints of per-wave scratch, the hit counter.
// ------------------------------------------------------------------------------------
#define SCN_L1_PITCH 272
#define SCN_L2_PITCH 257
#define SCN_LDS_EXCH (16 * SCN_L1_PITCH)  // complex slots (>= 16*257)
#define SCN_LDS_BYTES_4096 (SCN_LDS_EXCH * 8 + 256 * 8 + 48)

template <int KIND, bool DC, bool HITS>
__global__ __launch_bounds__(256, SCN_WG_PER_CU) void scn_fft4096_kernel(ScnFftArgs args) {
  constexpr int AUX_LD = SCN_AUX_LD;
  constexpr int AUX_ST = SCN_AUX_ST;
  constexpr bool PF = SCN_PREFETCH != 0;
  constexpr uint32_t N = 4096;
  typedef RawLoader<KIND> L;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v2f *lds = reinterpret_cast<v2f *>(smem_raw);
  v2f *lds_tw2 = lds + SCN_LDS_EXCH;                        // [16][16]
  int *lds_cnt = reinterpret_cast<int *>(lds_tw2 + 256);    // [8] DC-sum scratch
  int *lds_hits = lds_cnt + 8;                              // hits of the buffer in flight

  const uint32_t t = threadIdx.x;
  const uint32_t hi = t >> 4, lo = t & 15;
  const uint32_t wave = t >> 6;

  // persistent per-thread constants: pass-1 twiddles W_4096^(t*p) and window taps
  cf tw1[16];
#pragma unroll
  for (int p = 1; p < 16; p++) tw1[p] = from_v2f(args.twiddle[(t * p) & (N - 1)]);
  float win[16];
#pragma unroll
  for (int a = 0; a < 16; a++) win[a] = args.window[256 * a + t];
  // pass-2 twiddles W_256^(c*q), table [q][c] shared by the workgroup
  lds_tw2[t] = args.twiddle[(16 * hi * lo) & (N - 1)];
  if (t == 0) *lds_hits = 0;
  __syncthreads();

  v2f *w1 = lds + t;                                  // + p*272
  v2f *r1 = lds + hi * SCN_L1_PITCH + lo;             // + b*16
  v2f *w2 = lds + lo * SCN_L2_PITCH + hi;             // + q*16
  v2f *r3 = lds + t;                                  // + c*257
  const v2f *tw2 = lds_tw2 + lo;                      // + q*16

  typename L::raw_t raw[16];
  if (PF && blockIdx.x < args.n_buffers) {
    __amdgpu_buffer_rsrc_t r0 =
        make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)blockIdx.x * L::kBufBytes(N), L::kBufBytes(N));
#pragma unroll
    for (int a = 0; a < 16; a++) raw[a] = L::template load<AUX_LD>(r0, N, t, 256 * a);
  }

  for (uint32_t buf = blockIdx.x; buf < args.n_buffers; buf += gridDim.x) {
    // ---- K1 + K2: load, convert, window ----
    if (!PF) {
      __amdgpu_buffer_rsrc_t rin =
          make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)buf * L::kBufBytes(N), L::kBufBytes(N));
#pragma unroll
      for (int a = 0; a < 16; a++) raw[a] = L::template load<AUX_LD>(rin, N, t, 256 * a);
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
      if ((t & 63) == 0) {
        lds_cnt[wave] = sr;
        lds_cnt[4 + wave] = si;
      }
      __syncthreads();
      sr = lds_cnt[0] + lds_cnt[1] + lds_cnt[2] + lds_cnt[3];
      si = lds_cnt[4] + lds_cnt[5] + lds_cnt[6] + lds_cnt[7];
      dc_re = (int)((uint32_t)sr / N);
      dc_im = (int)((uint32_t)si / N);
    }

    cf v[16];
#pragma unroll
    for (int a = 0; a < 16; a++) v[a] = L::conv(raw[a], dc_re, dc_im, args.scale) * win[a];
    if (PF) {
      // the raw registers are free again: start fetching the next buffer of this workgroup now,
      // its latency hides behind the three FFT passes below
      const uint32_t nxt = buf + gridDim.x;
      if (nxt < args.n_buffers) {
        __amdgpu_buffer_rsrc_t rn =
            make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)nxt * L::kBufBytes(N), L::kBufBytes(N));
#pragma unroll
        for (int a = 0; a < 16; a++) raw[a] = L::template load<AUX_LD>(rn, N, t, 256 * a);
      }
    }

    // ---- pass 1: DFT over a, twiddle W_N^(t p), scatter to row p ----
    fft16(v);
#pragma unroll
    for (int p = 0; p < 16; p++) {
      cf y = v[OUT16(p)];
      if (p) y = cmul(y, tw1[p]);
      w1[p * SCN_L1_PITCH] = to_v2f(y);
    }
    __syncthreads();

    // ---- pass 2: thread (p=hi, c=lo): DFT over b, twiddle W_256^(c q) ----
#pragma unroll
    for (int b = 0; b < 16; b++) v[b] = from_v2f(r1[b * 16]);
    fft16(v);
#pragma unroll
    for (int q = 1; q < 16; q++) v[OUT16(q)] = cmul(v[OUT16(q)], from_v2f(tw2[q * 16]));
    __syncthreads();  // every exchange-1 read done before the area is re-used
#pragma unroll
    for (int q = 0; q < 16; q++) w2[q * 16] = to_v2f(v[OUT16(q)]);
    __syncthreads();

    // ---- pass 3: thread (p=lo, q=hi): DFT over c; outputs k = t + 256 r ----
#pragma unroll
    for (int c = 0; c < 16; c++) v[c] = from_v2f(r3[c * SCN_L2_PITCH]);
    fft16(v);

    // ---- K4 + K5 ----
    v16f db;  // a true vector: the slow path below indexes it with a wave-uniform r (s_set_gpr_idx)
    uint32_t hitmask = 0;
    __amdgpu_buffer_rsrc_t rout = make_rsrc(args.power_db + (size_t)buf * N, args.power_db ? 4u * N : 0u);
#pragma unroll
    for (int r = 0; r < 16; r++) {
      // NB: never __builtin_bit_cast a vector ELEMENT (db[r]): clang reads element 0 for every r
      const float d = power_db(v[OUT16(r)]);
      db[r] = d;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d), rout, t * 4u, 1024u * r, AUX_ST);
      if (HITS) {
        uint32_t j = t + 256 * r;
        uint32_t i = j ^ (N / 2);  // (j + N/2) % N, process.cpp:47
        bool keep = !(j < args.dc_ignore || (N - j) < args.dc_ignore) && !(i < args.i_lo || i > args.i_hi);
        hitmask |= (keep && (d > args.threshold)) ? (1u << r) : 0u;
      }
    }
    if (HITS) {
      if (__ballot(hitmask != 0)) {  // rare: some lane of this wave holds a detection
#pragma unroll 1
        for (int r = 0; r < 16; r++) {
          bool hit = (hitmask >> r) & 1u;
          unsigned long long m = __ballot(hit);
          if (!m) continue;
          // slot inside this buffer's region: one LDS atomic per wave and r
          uint32_t base = 0;
          if ((t & 63) == 0) base = (uint32_t)atomicAdd(lds_hits, (int)__popcll(m));
          base = __shfl(base, 0, 64);
          if (hit) {
            uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << (t & 63)) - 1ull));
            uint32_t j = t + 256 * r;
            ScnDevHit rec = ScnDevHit{buf, j ^ (N / 2), db[r], 0u};
            if (pos < args.hit_region) {
              args.hits[(size_t)buf * args.hit_region + pos] = rec;
            } else {  // region full: spill through the device-scope counter
              uint32_t opos = atomicAdd(args.ov_counter, 1u) - args.ov_base;
              if (opos < args.ov_cap) args.ov_hits[opos] = rec;
            }
          }
        }
      }
    }
    __syncthreads();  // exchange area free again; this buffer's hit count complete
    if (HITS) {
      if (t == 0) {
        args.per_buffer_hits[buf] = (uint32_t)*lds_hits;
        *lds_hits = 0;  // visible to the next buffer's recorders after its first barrier
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// host-side launcher
// ------------------------------------------------------------------------------------
template <int KIND>
static hipError_t launch4096_kind(const ScnFftArgs &a, bool dc, bool hits, int num_cus, hipStream_t s) {
  const size_t lds = SCN_LDS_BYTES_4096;
  void (*k)(ScnFftArgs) = nullptr;
  if (dc && hits) k = scn_fft4096_kernel<KIND, true, true>;
  else if (dc) k = scn_fft4096_kernel<KIND, true, false>;
  else if (hits) k = scn_fft4096_kernel<KIND, false, true>;
  else k = scn_fft4096_kernel<KIND, false, false>;
  int grid = num_cus * SCN_WG_PER_CU;  // one resident wave of persistent workgroups
  if ((uint32_t)grid > a.n_buffers) grid = (int)a.n_buffers;
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, a);
  return hipGetLastError();
}

hipError_t scn_launch_fft(uint32_t n, int kind, bool dc, bool hits, const ScnFftArgs &args, int num_cus,
                          hipStream_t stream) {
  if (args.n_buffers == 0) return hipSuccess;
  if (n != 4096) return hipErrorInvalidValue;
  switch (kind) {
    case SCN_K_FLOAT_COMPLEX: return launch4096_kind<SCN_K_FLOAT_COMPLEX>(args, false, hits, num_cus, stream);
    case SCN_K_SHORT_COMPLEX: return launch4096_kind<SCN_K_SHORT_COMPLEX>(args, dc, hits, num_cus, stream);
    case SCN_K_SHORT: return launch4096_kind<SCN_K_SHORT>(args, dc, hits, num_cus, stream);
    case SCN_K_BYTE_COMPLEX: return launch4096_kind<SCN_K_BYTE_COMPLEX>(args, dc, hits, num_cus, stream);
    default: return hipErrorInvalidValue;
  }
}

bool scn_fft_size_supported(uint32_t n) { return n == 4096; }
