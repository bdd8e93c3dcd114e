// membw.hip -- what the memory system delivers for the FFT kernel's traffic shape with no FFT:
// per 4096-sample buffer a 256-thread workgroup reads 32 KiB (cfloat) and writes 16 KiB (float).
// Variants: load width 8/16 B per lane, store width 4/16 B, cache policies, occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int LW, int SW, int AUXL, int AUXS>
__global__ __launch_bounds__(256) void k(const char* in, char* out, unsigned nb) {
  extern __shared__ char lds[];
  const unsigned t = threadIdx.x;
  for (unsigned buf = blockIdx.x; buf < nb; buf += gridDim.x) {
    __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)(in + (size_t)buf * 32768), 0, 32768, 0x00020000);
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (size_t)buf * 16384), 0, 16384, 0x00020000);
    float acc[16];
    if (LW == 8) {
#pragma unroll
      for (int a = 0; a < 16; a++) {
        v2f x = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(ri, t * 8, a * 2048, AUXL));
        acc[a] = x.x * x.x + x.y * x.y;
      }
    } else {
#pragma unroll
      for (int a = 0; a < 8; a++) {
        v4f x = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(ri, t * 16, a * 4096, AUXL));
        acc[2 * a] = x.x * x.x + x.y * x.y;
        acc[2 * a + 1] = x.z * x.z + x.w * x.w;
      }
    }
    if (lds[t] == 77) acc[0] += 1;  // keep the LDS allocation alive
    if (SW == 4) {
#pragma unroll
      for (int r = 0; r < 16; r++)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc[r]), ro, t * 4, r * 1024, AUXS);
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        v4f o = v4f{acc[4 * r], acc[4 * r + 1], acc[4 * r + 2], acc[4 * r + 3]};
        typedef unsigned u4 __attribute__((__vector_size__(16)));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, o), ro, t * 16, r * 4096, AUXS);
      }
    }
  }
}

template <int LW, int SW, int AUXL, int AUXS>
void run(const char* name, std::vector<char*>& ins, std::vector<char*>& out, unsigned nb, int wg_per_cu, size_t lds) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)k<LW, SW, AUXL, AUXS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int grid = 256 * wg_per_cu;
  for (int i = 0; i < 4; i++) hipLaunchKernelGGL((k<LW, SW, AUXL, AUXS>), dim3(grid), dim3(256), lds, 0, ins[i % ins.size()], out[i % out.size()], nb);
  CK(hipDeviceSynchronize());
  const int K = 40;
  float best = 1e9;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < K; i++) hipLaunchKernelGGL((k<LW, SW, AUXL, AUXS>), dim3(grid), dim3(256), lds, 0, ins[i % ins.size()], out[i % out.size()], nb);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms / K < best) best = ms / K;
  }
  printf("%-34s wg/cu %d lds %6zu Rin=%zu Rout=%zu: %7.2f us  %5.2f TB/s\n", name, wg_per_cu, lds, ins.size(), out.size(), best * 1e3, nb * 49152.0 / (best * 1e-3) / 1e12);
}

int main() {
  const unsigned nb = 8192;
  std::vector<char*> ins4(4), ins1(1);
  for (auto& p : ins4) { CK(hipMalloc(&p, (size_t)nb * 32768)); CK(hipMemset(p, 1, (size_t)nb * 32768)); }
  ins1[0] = ins4[0];
  std::vector<char*> out4(4), out2(2);
  for (auto& p : out4) CK(hipMalloc(&p, (size_t)nb * 16384));
  out2[0] = out4[0]; out2[1] = out4[1];
  run<8, 4, 0, 0>("ld8 st4 default", ins4, out2, nb, 4, 36944);
  run<8, 4, 0, 0>("ld8 st4 default", ins4, out4, nb, 4, 36944);
  run<8, 4, 0, 2>("ld8 st4 nt-store", ins4, out4, nb, 4, 36944);
  run<8, 4, 2, 0>("ld8 st4 nt-load", ins4, out2, nb, 4, 36944);
  run<8, 4, 2, 0>("ld8 st4 nt-load", ins4, out4, nb, 4, 36944);
  run<8, 4, 2, 2>("ld8 st4 nt-both", ins4, out4, nb, 4, 36944);
  run<8, 4, 1, 0>("ld8 st4 sc0-load", ins4, out4, nb, 4, 36944);
  run<8, 4, 16, 0>("ld8 st4 sc1-load", ins4, out4, nb, 4, 36944);
  run<8, 4, 17, 0>("ld8 st4 sc0sc1-load", ins4, out4, nb, 4, 36944);
  run<8, 4, 18, 0>("ld8 st4 nt-sc1-load", ins4, out4, nb, 4, 36944);
  run<8, 4, 2, 16>("ld8 st4 nt-load sc1-store", ins4, out4, nb, 4, 36944);
  run<8, 4, 2, 17>("ld8 st4 nt-load sc0sc1-store", ins4, out4, nb, 4, 36944);
  run<16, 16, 2, 0>("ld16 st16 nt-load", ins4, out4, nb, 4, 36944);
  run<16, 16, 2, 2>("ld16 st16 nt-both", ins4, out4, nb, 4, 36944);
  run<8, 4, 2, 0>("ld8 st4 nt-load (8 wg/cu)", ins4, out4, nb, 8, 16384);
  run<8, 4, 2, 0>("ld8 st4 nt-load (2 wg/cu)", ins4, out4, nb, 2, 70000);
  return 0;
}
