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
// Both are byte work on a few KB..MB (L2-latency bound, no roofline of their own): the C2 batch has ~69 k hits
// = 0.5 MB of records in, 1.6 MB out.  The list is written to DEVICE memory (writing it straight into pinned host
// memory stalled the FFT launch running beside it: 76 -> 100 us) and a DMA of the predicted size follows it on the same
// stream (scn_api.hip, build_list / fetch_list); alone on the GPU the kernel takes ~4.5 us for a C2 batch, beside a
// running FFT launch ~30 us, in its shadow either way, so a collect call is an event wait plus a copy out of pinned
// memory.  The output window [first, first + out_cap) lets a caller with a small buffer walk an arbitrarily long list
// (scn_collect_more) -- nothing is ever dropped on the device.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/scanner_hip.h"
#include "scn_kernels.h"

namespace {

// Workgroups of 256 threads = one wave per SIMD: small enough to be dispatched NEXT TO the resident workgroups of
// the following FFT launch (a 4096-point float workgroup set leaves ~100 VGPRs per SIMD and 49 KiB of LDS free; a
// 1024-thread scan would need four waves per SIMD on one CU and therefore waits until that launch drains).
// Workgroup j owns the 2048 counts [2048 j, 2048 (j+1)), 8 consecutive ones per thread in registers.  Instead of a
// carry chain between workgroups (a grid-wide dependency), each workgroup sums the counts BEFORE its chunk itself:
// redundant reads of at most n_buffers * 4 bytes from L2, all workgroups independent and in flight together.  (The
// first version, one workgroup walking its chunks with a load-add-store loop, took 15-30 us per 8192 counts.)
constexpr uint32_t kScanThreads = 256;
constexpr uint32_t kScanChunk = kScanThreads * 8u;

__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t *s_part) {  // total over the 256 threads, in every thread
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();  // s_part free again
  if ((threadIdx.x & 63u) == 0) s_part[threadIdx.x >> 6] = v;
  __syncthreads();
  return s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

__global__ __launch_bounds__(kScanThreads) void scn_hit_scan_kernel(ScnCompactArgs a) {
  __shared__ uint32_t s_part[kScanThreads / 64];
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  // (no s_setprio here: these waves usually run beside the next FFT launch's, and raising them made the scan twice as
  // fast -- 25 -> 12 us -- but cost the FFT launch beside them ~20 us; nothing waits for the list except its reader)
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  const uint32_t chunk0 = blockIdx.x * kScanChunk;
  // this chunk first (8 consecutive counts per thread), so that its loads and the ones below are in flight together
  const uint32_t b0 = chunk0 + t * 8u;
  uint32_t c[8];
#pragma unroll
  for (uint32_t k = 0; k < 8u; k++) c[k] = b0 + k < a.n_buffers ? a.counts[b0 + k] : 0u;
  // everything before this chunk (chunk0 is a multiple of 2048; counts is 16-byte aligned)
  uint32_t before = 0;
  for (uint32_t b = t * 4u; b < chunk0; b += kScanThreads * 4u) {
    const v4u v = *reinterpret_cast<const v4u *>(a.counts + b);
    before += v.x + v.y + v.z + v.w;
  }
  before = block_sum(before, s_part);
  uint32_t sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < 8u; k++) sum += c[k];
  uint32_t incl = sum;  // inclusive scan of the per-thread sums inside the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t v = __shfl_up(incl, off, 64);
    if (lane >= (uint32_t)off) incl += v;
  }
  __syncthreads();
  if (lane == 63) s_part[wave] = incl;
  __syncthreads();
  uint32_t run = before + incl - sum;
  for (uint32_t w = 0; w < wave; w++) run += s_part[w];
#pragma unroll
  for (uint32_t k = 0; k < 8u; k++) {
    if (b0 + k < a.n_buffers) a.offsets[b0 + k] = run;
    run += c[k];
  }
  if (blockIdx.x == gridDim.x - 1 && t == kScanThreads - 1) a.offsets[a.n_buffers] = run;  // the total
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
    // everything this buffer needs from memory is requested at once: its slice of the offsets (the count is the
    // difference: no second array), its header fields, and -- speculatively -- the first 64 records of its region
    const ScnDevHit *const region = a.regions + (size_t)b * a.hit_region;
    const uint32_t o0 = a.offsets[b], o1 = a.offsets[b + 1u];
    const ScnDevHit first64 = region[lane < a.hit_region ? lane : 0u];
    const double fc = a.center_freq[a.table_count ? (uint32_t)(((uint64_t)a.table_first + b) % a.table_count) : b];
    const uint64_t seq = a.seq_id ? a.seq_id[b] : (uint64_t)b;  // (no ids given at submit: a buffer's id is its index, messageQueue.h:86 from zero)
    const uint32_t c = o1 - o0;
    if (c == 0 || o0 >= last || o1 <= a.first) continue;  // no hits, or nothing of this buffer inside the window
    const uint32_t stored = c < a.hit_region ? c : a.hit_region;
    for (uint32_t w = lane; w < words; w += 64u) bits[w] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t k = lane; k < stored; k += 64u) {
      const uint32_t i = k < 64u ? first64.i : region[k].i;
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
    const double start_frequency = fc - (double)(a.sample_rate / 2u);  // process.cpp:38 (uint32 division)
    for (uint32_t k = lane; k < stored; k += 64u) {
      const ScnDevHit h = k < 64u ? first64 : region[k];
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

// The batch's total and its trigger flags, for launches of very many small buffers: what the host needs of the per-buffer counts
// is their sum (n_hits) and one bit each (process_fft's return value, hits > trigger_count, process.cpp:62) -- the records are
// ranked on the GPU from the counts where they lie.  So instead of 4 bytes per buffer crossing PCIe after every launch (2 MB per
// launch of 524288 16-point buffers, 40 us of the bus for a 27 us kernel) and the host walking them, one 8-byte word and a bitmap
// of nb / 8 bytes (64 KB) do.  Partial sums meet in a device word; the last workgroup to arrive stores the total for the host and
// leaves both words zero for the next launch.  A lane handles four counts (one 16-byte load); the eight lanes of a bitmap word
// meet in three cross-lane steps.
// (scn_api.hip launches it from 2^18 buffers per launch, on the launch's own stream: below, the ~6 us it takes there cost more than the counts' DMA.)
constexpr uint32_t kTotalThreads = 256, kTotalBlocks = 128;

__global__ __launch_bounds__(kTotalThreads) void scn_hit_total_kernel(const uint32_t *counts, uint32_t nb, uint32_t trigger_count, unsigned long long *acc,
                                                                        unsigned long long *host_total, uint32_t *bits) {
  unsigned long long sum = 0;
  const uint32_t quads = nb / 4u, rest = nb - quads * 4u;  // (hipMalloc'd: 16-byte loads are aligned)
  const uint32_t quads_all = quads + (rest ? 1u : 0u), words = (nb + 31u) / 32u;
  const uint4 *const c4 = reinterpret_cast<const uint4 *>(counts);
  const uint32_t lane8 = threadIdx.x & 7u;
  // (the loop's bound is the same for the eight lanes of a word -- for a whole wave, in fact: they all take part in the shuffles)
  for (uint32_t base = blockIdx.x * kTotalThreads + (threadIdx.x & ~63u); base < quads_all; base += kTotalBlocks * kTotalThreads) {
    const uint32_t i = base + (threadIdx.x & 63u);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (i < quads) {
      v = c4[i];
    } else if (i == quads && rest) {  // the last, partial quad
      v.x = counts[4u * i];
      if (rest > 1u) v.y = counts[4u * i + 1u];
      if (rest > 2u) v.z = counts[4u * i + 2u];
    }
    sum += (unsigned long long)v.x + v.y + v.z + v.w;
    uint32_t b = (v.x > trigger_count ? 1u : 0u) | (v.y > trigger_count ? 2u : 0u) | (v.z > trigger_count ? 4u : 0u) | (v.w > trigger_count ? 8u : 0u);
    b <<= 4u * lane8;
    b |= (uint32_t)__shfl_xor((int)b, 1, 64);
    b |= (uint32_t)__shfl_xor((int)b, 2, 64);
    b |= (uint32_t)__shfl_xor((int)b, 4, 64);
    if (lane8 == 0u && i / 8u < words) bits[i / 8u] = b;
  }
#pragma unroll
  for (int off = 32; off; off >>= 1) sum += __shfl_xor(sum, off, 64);
  __shared__ unsigned long long part[kTotalThreads / 64u];
  if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long block = 0;
    for (uint32_t w = 0; w < kTotalThreads / 64u; w++) block += part[w];
    atomicAdd(&acc[0], block);
    __threadfence();
    if (atomicAdd(&acc[1], 1ull) == kTotalBlocks - 1u) {  // every other workgroup's sum is in acc[0]
      __threadfence();
      const unsigned long long total = atomicExch(&acc[0], 0ull);
      atomicExch(&acc[1], 0ull);
      __hip_atomic_store(host_total, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

}  // namespace

hipError_t scn_launch_hit_total(const uint32_t *counts, uint32_t n_buffers, uint32_t trigger_count, unsigned long long *acc, unsigned long long *host_total,
                                uint32_t *trigger_bits, hipStream_t stream) {
  hipLaunchKernelGGL(scn_hit_total_kernel, dim3(kTotalBlocks), dim3(kTotalThreads), 0, stream, counts, n_buffers, trigger_count, acc, host_total, trigger_bits);
  return hipGetLastError();
}

hipError_t scn_launch_hit_scan(const ScnCompactArgs &a, hipStream_t stream) {
  const uint32_t blocks = a.n_buffers ? (a.n_buffers + kScanChunk - 1u) / kScanChunk : 1u;
  hipLaunchKernelGGL(scn_hit_scan_kernel, dim3(blocks), dim3(kScanThreads), 0, stream, a);
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
