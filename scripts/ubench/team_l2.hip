// team_l2.hip -- the memory skeleton of a ONE-kernel 65536-point four-step transform whose work buffer stays in an XCD's L2,
// against the skeleton of the two-kernel pair that ships (scn_big.hip).  VERDICT r4 next 5: decide by measurement whether the
// team-per-buffer form is worth building.  No arithmetic: only the traffic pattern and, for the team form, the team barrier.
//
//   pair  kernel 1: workgroup (j, b) reads its 32 KiB of buffer b (8 B per lane per load, 16 loads) and writes the 16 blocks
//                   (q, j) of 2 KiB of Y[b] (the tiled work buffer of scn_big.hip); kernel 2: workgroup (i, b) reads the 32 KiB
//                   block row i of Y[b] and writes 16 KiB of output.  Y = n_buffers x 512 KiB in HBM.
//   team  one persistent kernel: 16 workgroups with equal blockIdx % 8 (one XCD under round-robin placement) form a team; a team
//         takes buffers T, T + teams, ...; member m does "columns" of tile m into the team's PRIVATE, double-buffered 512 KiB
//         Y slot (plain stores: they stay in the XCD's L2), every wave waits for its stores, the team meets at an agent-scope
//         atomic counter, member m then reads block row m of the slot with sc1 loads (L1 bypassed, L2-served) and writes the
//         output.  The XCC_ID of every member is recorded, and so is a checksum of what the readers saw (every reader must see
//         the writers' bytes of THIS buffer: the coherence check).  Every spin is capped.
// usage: team_l2 [n_buffers 512] [teams per XCD 4] [reps 20]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((__vector_size__(8)));
constexpr unsigned N = 65536;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

// "columns" of tile j of buffer b: 16 loads of 8 B per lane (rows of 128 B of the input, like scn_big_cols_kernel), 16 stores of
// one 2 KiB block each into Y.  The value written encodes (b, j, q, t) so that a reader can check what it sees.
__device__ __forceinline__ void cols_phase(const char *in, char *y, unsigned b, unsigned j, unsigned t, int st_aux) {
  const __amdgpu_buffer_rsrc_t r = rsrc(in + (size_t)b * N * 8u, N * 8u);
  const unsigned hi = t >> 4, lo = t & 15u, s0 = 256u * hi + 16u * j + lo;
  v2f v[16];
#pragma unroll
  for (int a = 0; a < 16; a++) v[a] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, s0 * 8u, a * (N / 2u), 2));
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 16; a++) acc += v[a].x + v[a].y;
  const __amdgpu_buffer_rsrc_t rw = rsrc(y + j * 2048u, N * 8u - j * 2048u);
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const v2f o{__builtin_bit_cast(float, (b << 12) | (j << 8) | (unsigned)q << 4 | (t & 15u)), acc * 0.f + (float)t};
    if (st_aux == 0) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, o), rw, t * 8u, q * (16u * 2048u), 0);
    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, o), rw, t * 8u, q * (16u * 2048u), 2);
  }
}
// "rows" of tile i: block row i of Y (32 KiB contiguous = blocks (i, 0 .. 15)), 16 KiB of output; returns the number of
// values that are not what member jj wrote for buffer b
template <int AUX>
__device__ __forceinline__ unsigned rows_phase(const char *y, float *out, unsigned b, unsigned i, unsigned t) {
  const __amdgpu_buffer_rsrc_t rw = rsrc(y + i * 32768u, N * 8u - i * 32768u);
  v2f v[16];
#pragma unroll
  for (int a = 0; a < 16; a++) v[a] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rw, t * 8u, a * 2048u, AUX));
  unsigned bad = 0;
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 16; a++) {  // block (i, a) was written by member j = a with q = i
    const unsigned want = (b << 12) | ((unsigned)a << 8) | (i << 4) | (t & 15u);
    bad += __builtin_bit_cast(unsigned, v[a].x) != want || v[a].y != (float)t;
    acc += v[a].y;
  }
  const __amdgpu_buffer_rsrc_t ro = rsrc(out + (size_t)b * N, N * 4u);
#pragma unroll
  for (int u = 0; u < 16; u++)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc), ro, (16u * i + (t & 15u) + 256u * ((t >> 4) + 16u * u)) * 4u, 0, 2);
  return bad;
}

__global__ __launch_bounds__(256) void pair_cols(const char *in, char *y, unsigned nb) {
  const unsigned j = blockIdx.x & 15u;
  for (unsigned b = blockIdx.x >> 4; b < nb; b += gridDim.x >> 4) cols_phase(in, y + (size_t)b * N * 8u, b, j, threadIdx.x, 0);
}
__global__ __launch_bounds__(256) void pair_rows(const char *y, float *out, unsigned nb, unsigned *bad) {
  const unsigned i = blockIdx.x & 15u, b = blockIdx.x >> 4;
  const unsigned n = rows_phase<2>(y + (size_t)b * N * 8u, out, b, i, threadIdx.x);
  if (n) atomicAdd(bad, n);
}

struct TeamCtl {
  unsigned arrive;
  unsigned xmin, xmax;
  unsigned pad[29];
};
__global__ __launch_bounds__(256) void team_kernel(const char *in, char *ywork, float *out, unsigned nb, TeamCtl *ctl, unsigned *bad, unsigned *status,
                                                    unsigned spin_cap) {
  __shared__ unsigned s_fail;
  const unsigned t = threadIdx.x;
  const unsigned x = blockIdx.x & 7u, idx = blockIdx.x >> 3, tl = idx >> 4, m = idx & 15u;
  const unsigned teams = gridDim.x >> 4, T = tl * 8u + x;
  TeamCtl *c = ctl + T;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 0xfu;
  if (t == 0) {
    s_fail = 0;
    atomicMin(&c->xmin, xcc);
    atomicMax(&c->xmax, xcc);
  }
  unsigned phase = 0, nbad = 0;
  auto team_barrier = [&]() {  // every wave has waited for its own stores before this
    __syncthreads();
    phase++;
    if (t == 0) {
      __hip_atomic_fetch_add(&c->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while (__hip_atomic_load(&c->arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 16u * phase) {
        if (++spins > spin_cap) {
          s_fail = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
    return s_fail == 0;
  };
  if (!team_barrier()) {
    if (t == 0) atomicOr(status, 1u);  // a member never arrived: not co-resident
    return;
  }
  const unsigned xmin = __hip_atomic_load(&c->xmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned xmax = __hip_atomic_load(&c->xmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (xmin != xmax) {  // the team spans XCDs: its L2 hand-off would be wrong -- say so and leave (all 16 members see the same words)
    if (t == 0 && m == 0) atomicOr(status, 2u);
    return;
  }
  unsigned k = 0;
  for (unsigned b = T; b < nb; b += teams, k++) {
    char *y = ywork + ((size_t)T * 2u + (k & 1u)) * N * 8u;
    cols_phase(in, y, b, m, t, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!team_barrier()) {
      if (t == 0) atomicOr(status, 4u);
      return;
    }
    nbad += rows_phase<16>(y, out, b, m, t);  // sc1: bypass this CU's L1, served by the XCD's L2
  }
  if (nbad) atomicAdd(bad, nbad);
}

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

int main(int argc, char **argv) {
  const unsigned nb = argc > 1 ? atoi(argv[1]) : 512, tpx = argc > 2 ? atoi(argv[2]) : 4, reps = argc > 3 ? atoi(argv[3]) : 20;
  const unsigned R = 4;  // rotating inputs / outputs: 4 x (268 + 134) MB, past the Infinity Cache
  char *in[R], *y, *ywork;
  float *out[R];
  for (unsigned r = 0; r < R; r++) {
    CK(hipMalloc(&in[r], (size_t)nb * N * 8));
    CK(hipMemset(in[r], 0, (size_t)nb * N * 8));
    CK(hipMalloc(&out[r], (size_t)nb * N * 4));
  }
  CK(hipMalloc(&y, (size_t)nb * N * 8));
  const unsigned teams = 8 * tpx;
  CK(hipMalloc(&ywork, (size_t)teams * 2 * N * 8));
  TeamCtl *ctl;
  unsigned *bad, *status;
  CK(hipMalloc(&ctl, sizeof(TeamCtl) * teams));
  CK(hipMalloc(&bad, 8));
  status = bad + 1;
  CK(hipMemset(bad, 0, 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms;
  // ---- the pair ----
  const unsigned G = 768 / 16;  // 3 workgroups per CU, as scn_launch_big sizes the column kernel
  for (unsigned k = 0; k < reps + 5; k++) {
    if (k == 5) CK(hipEventRecord(e0));
    hipLaunchKernelGGL(pair_cols, dim3(16 * (G < nb ? G : nb)), dim3(256), 0, 0, in[k % R], y, nb);
    hipLaunchKernelGGL(pair_rows, dim3(16 * nb), dim3(256), 0, 0, y, out[k % R], nb, bad);
  }
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned h[2];
  CK(hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost));
  printf("pair : %8.1f us per %u buffers (%.2f TB/s of its 28 B per sample)   wrong values %u\n", ms * 1e3 / reps, nb, 28.0 * nb * N / (ms * 1e-3 / reps) / 1e12, h[0]);
  // ---- the team kernel ----
  CK(hipMemset(bad, 0, 8));
  std::vector<TeamCtl> init(teams);
  for (auto &c : init) {
    c.arrive = 0;
    c.xmin = 0xffffffffu;
    c.xmax = 0;
  }
  for (unsigned k = 0; k < reps + 5; k++) {
    if (k == 5) CK(hipEventRecord(e0));
    CK(hipMemcpyAsync(ctl, init.data(), sizeof(TeamCtl) * teams, hipMemcpyHostToDevice, 0));
    hipLaunchKernelGGL(team_kernel, dim3(16 * teams), dim3(256), 0, 0, in[k % R], ywork, out[k % R], nb, ctl, bad, status, 2000000u);
  }
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost));
  printf("team : %8.1f us per %u buffers (%u teams per XCD, %u workgroups; %.2f TB/s of the algorithmic 12 B per sample)   wrong values %u   status %u "
         "(1: a member never arrived, 2: a team spans XCDs, 4: barrier timeout)\n",
         ms * 1e3 / reps, nb, tpx, 16 * teams, 12.0 * nb * N / (ms * 1e-3 / reps) / 1e12, h[0], h[1]);
  return 0;
}
