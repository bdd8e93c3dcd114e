#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "messageQueue.h"
int main(int argc, char **argv) {
  const uint32_t n = 8192, depth = 8192, batch = 2048, nslots = 3; const int staged = argc > 1 ? atoi(argv[1]) : 1;
  const size_t total = 300000;
  SampleQueue q(SampleQueue::ShortComplex, 12, n, depth, false, false);
  std::vector<std::vector<unsigned char>> slots(nslots, std::vector<unsigned char>((size_t)batch * n * 4));
  void *bases[nslots]; for (uint32_t i = 0; i < nslots; i++) bases[i] = slots[i].data();
  const int ring = staged ? q.AttachStaging(bases, nslots, batch) : -1;
  std::vector<int16_t> src(64 * n * 2, 3);
  auto t0 = std::chrono::steady_clock::now();
  std::thread prod([&] {
    for (size_t k = 0; k < total + 1; k++) q.AppendSamples(reinterpret_cast<int16_t(*)[2]>(src.data() + (k % 64) * n * 2), 1e6 * k, k < 2 ? 1 : 0);
    q.SetIsDone();
  });
  size_t got = 0; std::vector<SampleQueue::MessageType *> out; std::vector<unsigned char> stage((size_t)batch * n * 4);
  while (true) {
    out.clear(); int slot = -1; uint32_t c = 0;
    if (staged) c = q.TakeStagedBatch(ring, out, &slot, true);
    else { SampleQueue::MessageType *m = q.GetNextSamples(); while (m) { memcpy(stage.data() + (size_t)c * n * 4, m->GetRawData(), n * 4); out.push_back(m); c++; if (c >= batch) break; m = q.TryGetNextSamples(); } }
    if (!c) break;
    for (auto *m : out) q.MessageProcessed(m);
    if (staged) q.ReleaseStaging(ring, slot);
    got += c;
  }
  prod.join();
  double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("staged=%d: %zu buffers in %.3f s = %.2f us per buffer, %.1f GB/s, %.1f Msamples/s\n", staged, got, s, s / got * 1e6, got * n * 4.0 / s / 1e9, got * (double)n / s / 1e6);
}
