// scn_hits.hip -- the batch's ordered hit list, built on the GPU.
//
// The reference prints its detections from a single loop over i (process.cpp:46-61), one buffer after the other:
// the list a caller sees is ordered by (buffer, i).  The fused FFT kernels record a buffer's hits in whatever order
// its waves find them (one LDS atomic per wave hands out slots of the buffer's region, scn_kernels.hip), so two small
// kernels restore the reference's order without the host ever touching an unordered record:
//
//   scn_hit_scan_kernel     exclusive prefix sum of the per-buffer counts -> offsets[b]; total at offsets[n_buffers]
//   scn_hit_compact_kernel  one WAVE per buffer that has hits: the rank of a record inside its buffer is the number
//                           of set bits below its bin in the buffer's hit bitmap, built in LDS from the records
//                           themselves (bins are distinct within a buffer) -- O(hits) work, no sort, no comparisons
//                           between records; record k of the batch is written completed to the public 24-byte form
//                           {seq_id, i, power_db, freq_hz} with the frequency in the reference's own arithmetic
//                           (process.cpp:38-39,55-57: uint32 bin_step, uint32 i*bin_step, double sum, cast to uint64).
//
// Both are byte work on a few KB..MB (HBM/L2-latency bound, no roofline of their own): the C2 batch has ~69 k hits
// = 0.5 MB of records in, 1.6 MB out.  The output window [first, first + out_cap) lets a caller with a small buffer
// walk an arbitrarily long list (scn_collect_more) -- nothing is ever dropped on the device.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/scanner_hip.h"
#include "scn_kernels.h"

namespace {

constexpr uint32_t kScanThreads = 1024;

__global__ __launch_bounds__(kScanThreads) void scn_hit_scan_kernel(ScnCompactArgs a) {
  __shared__ uint32_t s_wave[kScanThreads / 64];
  __shared__ uint32_t s_carry;
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  if (t == 0) s_carry = 0;
  __syncthreads();
  // chunks of 1024 consecutive buffers, one count per thread, carry from chunk to chunk
  for (uint32_t base = 0; base < a.n_buffers; base += kScanThreads) {
    const uint32_t b = base + t;
    const uint32_t c = b < a.n_buffers ? a.counts[b] : 0u;
    uint32_t incl = c;  // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t v = __shfl_up(incl, off, 64);
      if (lane >= (uint32_t)off) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = s_carry;  // everything in earlier chunks and earlier waves of this chunk
    for (uint32_t w = 0; w < wave; w++) before += s_wave[w];
    if (b < a.n_buffers) a.offsets[b] = before + incl - c;
    __syncthreads();
    if (t == kScanThreads - 1) s_carry = before + incl;
    __syncthreads();
  }
  if (t == 0) a.offsets[a.n_buffers] = s_carry;
}

// uint64_t(double) the way the reference's x86-64 build does it (process.cpp:57 casts a double that is negative for the
// lowest bins of a sweep starting at 0 Hz: cvttsd2si gives the two's complement of the magnitude, not 0)
__device__ __forceinline__ uint64_t to_u64_like_x86(double f) {
  return f < 0.0 ? (uint64_t)(int64_t)f : (uint64_t)f;
}

__global__ __launch_bounds__(256) void scn_hit_compact_kernel(ScnCompactArgs a) {
  extern __shared__ uint32_t smem[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t words = (a.n + 31u) / 32u;            // bitmap words per buffer
  const uint32_t per_lane = (words + 63u) / 64u;       // contiguous words a lane sums
  uint32_t *const bits = smem + (size_t)wave * 2u * words;
  uint32_t *const below = bits + words;                // set bits in the words before word w
  const uint32_t bin_step = a.sample_rate / a.n;       // process.cpp:39 (truncating)
  const uint32_t last = a.first + a.out_cap;           // exclusive (out_cap <= 2^31: no wrap, checked by the caller)
  scn_hit *const out = static_cast<scn_hit *>(a.out);
  for (uint32_t b = blockIdx.x * 4u + wave; b < a.n_buffers; b += gridDim.x * 4u) {
    const uint32_t c = a.counts[b];
    if (c == 0) continue;
    const uint32_t o0 = a.offsets[b];
    if (o0 >= last || o0 + c <= a.first) continue;     // nothing of this buffer inside the window
    const ScnDevHit *const region = a.regions + (size_t)b * a.hit_region;
    const uint32_t stored = c < a.hit_region ? c : a.hit_region;
    for (uint32_t w = lane; w < words; w += 64u) bits[w] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t k = lane; k < stored; k += 64u) {
      const uint32_t i = region[k].i;
      atomicOr(&bits[i >> 5], 1u << (i & 31u));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // below[w]: lane l sums its words [l*per_lane, (l+1)*per_lane), exclusive scan over the lanes, then fills in
    uint32_t mine = 0;
    for (uint32_t k = 0; k < per_lane; k++) {
      const uint32_t w = lane * per_lane + k;
      if (w < words) mine += (uint32_t)__popc(bits[w]);
    }
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t v = __shfl_up(incl, off, 64);
      if (lane >= (uint32_t)off) incl += v;
    }
    uint32_t run = incl - mine;
    for (uint32_t k = 0; k < per_lane; k++) {
      const uint32_t w = lane * per_lane + k;
      if (w < words) {
        below[w] = run;
        run += (uint32_t)__popc(bits[w]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const double start_frequency = a.center_freq[b] - (double)(a.sample_rate / 2u);  // process.cpp:38 (uint32 division)
    const uint64_t seq = a.seq_id[b];
    for (uint32_t k = lane; k < stored; k += 64u) {
      const ScnDevHit h = region[k];
      const uint32_t w = h.i >> 5;
      const uint32_t pos = o0 + below[w] + (uint32_t)__popc(bits[w] & ((1u << (h.i & 31u)) - 1u));
      if (pos >= a.first && pos < last) {
        const double frequency = start_frequency + (double)(uint32_t)(h.i * bin_step);  // process.cpp:55
        scn_hit r;
        r.seq_id = seq;
        r.i = h.i;
        r.power_db = h.power_db;
        r.freq_hz = to_u64_like_x86(frequency);  // process.cpp:57
        out[pos - a.first] = r;
      }
    }
    __builtin_amdgcn_wave_barrier();  // the bitmap is reused by this wave's next buffer
  }
}

}  // namespace

hipError_t scn_launch_hit_scan(const ScnCompactArgs &a, hipStream_t stream) {
  hipLaunchKernelGGL(scn_hit_scan_kernel, dim3(1), dim3(kScanThreads), 0, stream, a);
  return hipGetLastError();
}

hipError_t scn_launch_hit_compact(const ScnCompactArgs &a, hipStream_t stream) {
  if (a.n_buffers == 0 || a.out_cap == 0) return hipSuccess;
  const uint32_t words = (a.n + 31u) / 32u;
  const size_t lds = (size_t)4u * 2u * words * sizeof(uint32_t);  // 4 waves x (bitmap + prefix)
  uint32_t blocks = (a.n_buffers + 3u) / 4u;
  if (blocks > 8192u) blocks = 8192u;
  hipLaunchKernelGGL(scn_hit_compact_kernel, dim3(blocks), dim3(256), lds, stream, a);
  return hipGetLastError();
}
