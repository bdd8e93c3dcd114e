// Does a stream of PCIe writes (device -> pinned host) stall an HBM-streaming kernel running beside it?
//   hipcc --offload-arch=gfx950 -O3 -o pcie_beside pcie_beside.hip && ./pcie_beside
// Stream A: a streaming kernel with the FFT launch's traffic (256 MiB in, 128 MiB out, non-temporal), back to back.
// Stream B: every `period` us 1.66 MB go to pinned host memory -- by hipMemcpyAsync (a blit kernel on this ROCm), or by a
// store kernel of G workgroups (16-byte stores, consecutive lanes), optionally paced with s_sleep between its stores.
// Printed: A's average duration alone and beside each kind of B, and the rate B reached.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream_kernel(const v4f *in, v4f *out, size_t n_out) {  // reads 2 x 16 B, writes 16 B per element
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n_out; i += gridDim.x * 256ull) {
    v4f a = __builtin_nontemporal_load(in + 2 * i), b = __builtin_nontemporal_load(in + 2 * i + 1);
    __builtin_nontemporal_store(a + b, out + i);
  }
}
__global__ __launch_bounds__(256) void host_store_kernel(const v4f *src, v4f *host, size_t n, int sleep) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
    host[i] = src[i];
    for (int k = 0; k < sleep; k += 4) __builtin_amdgcn_s_sleep(4);  // (the operand is an immediate: 4 x 64 cycles per round)
  }
}

// the same through a buffer descriptor with the cache-policy immediate AUX (bit0 sc0, bit1 nt, bit4 sc1)
template <int AUX>
__global__ __launch_bounds__(256) void host_store_aux_kernel(const v4f *src, void *host, uint32_t n16) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(host, 0, n16 * 16u, 0x00020000);
  typedef int v4i __attribute__((__vector_size__(16)));
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, src[i]), r, i * 16u, 0, AUX);
}

int main(int argc, char **argv) {
  const size_t in_bytes = 256u << 20, out_bytes = 128u << 20, rec_bytes = 1660000 / 16 * 16;
  const int R = 4;  // rotate A's buffers past the Infinity Cache
  std::vector<void *> din(R), dout(R);
  for (int r = 0; r < R; r++) { CK(hipMalloc(&din[r], in_bytes)); CK(hipMalloc(&dout[r], out_bytes)); CK(hipMemset(din[r], 1, in_bytes)); }
  void *dsrc, *hdst;
  CK(hipMalloc(&dsrc, rec_bytes)); CK(hipMemset(dsrc, 2, rec_bytes));
  CK(hipHostMalloc(&hdst, rec_bytes, hipHostMallocDefault));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t e0, e1, b0, b1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
  const int iters = 400;
  auto run = [&](const char *name, int mode, int G, int sleep) {  // mode 0: A alone; 1: hipMemcpyAsync; 2: store kernel; 100 + AUX: buffer stores with that policy
    for (int warm = 0; warm < 2; warm++) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, sa));
      if (mode) CK(hipEventRecord(b0, sb));
      for (int k = 0; k < iters; k++) {
        hipLaunchKernelGGL(stream_kernel, dim3(256 * 8), dim3(256), 0, sa, (const v4f *)din[k % R], (v4f *)dout[k % R], out_bytes / 16);
        if (mode == 1) CK(hipMemcpyAsync(hdst, dsrc, rec_bytes, hipMemcpyDeviceToHost, sb));
        if (mode == 2) hipLaunchKernelGGL(host_store_kernel, dim3(G), dim3(256), 0, sb, (const v4f *)dsrc, (v4f *)hdst, rec_bytes / 16, sleep);
#define AUXCASE(A) if (mode == 100 + A) hipLaunchKernelGGL(host_store_aux_kernel<A>, dim3(G), dim3(256), 0, sb, (const v4f *)dsrc, hdst, (uint32_t)(rec_bytes / 16));
        AUXCASE(0) AUXCASE(1) AUXCASE(2) AUXCASE(3) AUXCASE(16) AUXCASE(17) AUXCASE(18) AUXCASE(19)
      }
      CK(hipEventRecord(e1, sa));
      if (mode) CK(hipEventRecord(b1, sb));
      CK(hipDeviceSynchronize());
    }
    float ma = 0, mb = 0;
    CK(hipEventElapsedTime(&ma, e0, e1));
    if (mode) CK(hipEventElapsedTime(&mb, b0, b1));
    printf("%-44s A: %6.1f us per launch (%.2f TB/s)", name, ma * 1e3 / iters, (in_bytes + out_bytes) / (ma * 1e-3 / iters) / 1e12);
    if (mode) printf("   B: %6.1f us per 1.66 MB (%.1f GB/s)", mb * 1e3 / iters, rec_bytes / (mb * 1e-3 / iters) / 1e9);
    printf("\n");
  };
  run("A alone", 0, 0, 0);
  run("A beside hipMemcpyAsync D2H", 1, 0, 0);
  const int gs[] = {1, 2, 4, 8, 16, 64, 256};
  for (int g : gs) { char nm[64]; snprintf(nm, sizeof nm, "A beside store kernel, %d workgroups", g); run(nm, 2, g, 0); }
  for (int g : {4, 16}) for (int sl : {4, 16, 64}) { char nm[64]; snprintf(nm, sizeof nm, "A beside store kernel, %d wg, s_sleep %d", g, sl); run(nm, 2, g, sl); }
  for (int aux : {0, 1, 2, 3, 16, 17, 18, 19}) { char nm[64]; snprintf(nm, sizeof nm, "A beside buffer stores aux %d, 64 wg", aux); run(nm, 100 + aux, 64, 0); }
  // B alone
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(b0, sb));
  for (int k = 0; k < iters; k++) CK(hipMemcpyAsync(hdst, dsrc, rec_bytes, hipMemcpyDeviceToHost, sb));
  CK(hipEventRecord(b1, sb));
  CK(hipDeviceSynchronize());
  float mb; CK(hipEventElapsedTime(&mb, b0, b1));
  printf("hipMemcpyAsync D2H alone: %.1f us per 1.66 MB (%.1f GB/s)\n", mb * 1e3 / iters, rec_bytes / (mb * 1e-3 / iters) / 1e9);
  return 0;
}
