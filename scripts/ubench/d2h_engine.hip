// Which engine does hipMemcpyAsync(device -> pinned host) run on -- an SDMA engine or a blit kernel -- depending on what
// precedes it on its stream?  Run under `rocprofv3 --kernel-trace --memory-copy-trace`: every case does exactly 100
// copies of its own distinct size, so the traces tell which cases became __amd_rocclr_copyBuffer kernels.
//   case 0 (1000 KB): copies only                case 1 (1100 KB): hipStreamWaitEvent on another stream's kernel, then the copy
//   case 2 (1200 KB): a kernel on the same stream, then the copy       case 3 (1300 KB): case 1 on a high-priority stream
//   case 4 (32 KB): copies only, small           case 5 (1400 KB): case 2, the stream synchronized between kernel and copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void busy(float *p, int n) { for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < n; i += gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.0f; }
int main() {
  void *d, *h; float *w;
  CK(hipMalloc(&d, 2 << 20)); CK(hipHostMalloc(&h, 2 << 20, hipHostMallocDefault)); CK(hipMalloc((void **)&w, 64 << 20));
  hipStream_t sa, sb, sp; int lo, hi;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi)); CK(hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, hi));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (int c = 0; c < 6; c++) {
    const size_t bytes = c == 4 ? 32768 : (1000 + 100 * (c == 5 ? 4 : c)) * 1024;
    hipStream_t s = c == 3 ? sp : sb;
    for (int k = 0; k < 100; k++) {
      if (c == 1 || c == 3) { hipLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, sa, w, 16 << 20); CK(hipEventRecord(ev, sa)); CK(hipStreamWaitEvent(s, ev, 0)); }
      if (c == 2 || c == 5) hipLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, s, w, 16 << 20);
      if (c == 5) CK(hipStreamSynchronize(s));
      CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s));
    }
    CK(hipDeviceSynchronize());
  }
  printf("done\n");
  return 0;
}
