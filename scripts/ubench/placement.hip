// placement.hip -- where does the dispatcher put N x 256-thread workgroups? (HW_ID / XCC_ID per workgroup)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned *out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float a = threadIdx.x;
  for (int i = 0; i < spin; i++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
  if (a == 123.f) out[0] = 0;
}
int main() {
  unsigned *d; hipMalloc(&d, 8 * 4096);
  for (int w = 1; w <= 4; w++) {
    int blocks = 256 * w;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * blocks);
    hipMemcpy(h.data(), d, 8 * blocks, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu;
    for (int b = 0; b < blocks; b++) {
      unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
      unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
    }
    std::map<int, int> hist;
    for (auto &kv : per_cu) hist[kv.second]++;
    printf("%d workgroups: %zu distinct CUs; workgroups-per-CU histogram:", blocks, per_cu.size());
    for (auto &kv : hist) printf("  %d WG x %d CUs", kv.first, kv.second);
    printf("\n");
  }
}
