#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <emmintrin.h>
#include <cstdint>
static void stream_copy(unsigned char *dst, const unsigned char *src, size_t bytes) {
  const size_t blocks = bytes / 64;
  for (size_t i = 0; i < blocks; i++) {
    const __m128i a = _mm_loadu_si128((const __m128i *)src + 4 * i), b = _mm_loadu_si128((const __m128i *)src + 4 * i + 1);
    const __m128i c = _mm_loadu_si128((const __m128i *)src + 4 * i + 2), d = _mm_loadu_si128((const __m128i *)src + 4 * i + 3);
    _mm_stream_si128((__m128i *)dst + 4 * i, a); _mm_stream_si128((__m128i *)dst + 4 * i + 1, b);
    _mm_stream_si128((__m128i *)dst + 4 * i + 2, c); _mm_stream_si128((__m128i *)dst + 4 * i + 3, d);
  }
  _mm_sfence();
}
int main() {
  const size_t msg = 32768, N = 6144; std::vector<unsigned char> dst(msg * N), src(64 * msg, 1);
  for (int mode = 0; mode < 2; mode++) for (int rep = 0; rep < 2; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 10; r++) for (size_t k = 0; k < N; k++) { if (mode) stream_copy(dst.data() + k * msg, src.data() + (k % 64) * msg, msg); else memcpy(dst.data() + k * msg, src.data() + (k % 64) * msg, msg); }
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%s: %.2f us per 32 KiB, %.1f GB/s\n", mode ? "stream" : "memcpy", s / (10.0 * N) * 1e6, 10.0 * N * msg / s / 1e9);
  }
}
