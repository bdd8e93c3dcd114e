// SampleBuffer -> HipStagingProcessInterface -> a plan's PINNED staging slot -> scn_submit (double-buffered over both
// slots) -> scn_collect: "sampleBuffer.cpp's host staging replaced by pinned double-buffered hipMemcpyAsync"
// (north star), driven through the kept C++ classes.  A producer thread appends float IQ buffers
// (sampleBuffer.h:28-45); the consumer visits each one straight into slot memory the GPU copies from, batches of
// `batch` buffers alternating between the two slots so that one slot fills while the other is in flight.  The spectra
// and the ordered hit list are written to a file for the Python side to hold against the oracle.
//   usage: test_staging_gpu <in.c64> <n> <n_buffers> <batch> <threshold> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/scanner_hip.h"
#include "buffer.h"
#include "sampleBuffer.h"

#define CHECK_SCN(call)                                                                       \
  do {                                                                                        \
    int st_ = (call);                                                                         \
    if (st_ != SCN_OK) {                                                                      \
      fprintf(stderr, "%s: %s: %s\n", #call, scn_error_name(st_), scn_last_error());          \
      return 1;                                                                               \
    }                                                                                         \
  } while (0)

int main(int argc, char **argv) {
  if (argc != 7) return 2;
  const uint32_t n = (uint32_t)atoi(argv[2]), total = (uint32_t)atoi(argv[3]), batch = (uint32_t)atoi(argv[4]);
  const float threshold = (float)atof(argv[5]);
  std::vector<float> iq((size_t)2 * n * total);
  FILE *f = fopen(argv[1], "rb");
  if (!f || fread(iq.data(), sizeof(float), iq.size(), f) != iq.size()) return 3;
  fclose(f);

  scn_plan_desc d;
  memset(&d, 0, sizeof(d));
  d.struct_size = sizeof(d);
  d.n = n;
  d.sample_rate = 8000000;
  d.sample_kind = SCN_KIND_FLOAT_COMPLEX;
  d.enob = 12;
  d.threshold = threshold;
  d.max_batch = batch;
  d.max_hits = batch * n;
  scn_plan *plan = nullptr;
  CHECK_SCN(scn_plan_create(&d, &plan));
  void *slotBase[SCN_NUM_SLOTS];
  size_t slotBytes = 0;
  for (int s = 0; s < SCN_NUM_SLOTS; s++) CHECK_SCN(scn_host_buffer(plan, s, &slotBase[s], &slotBytes));
  if (slotBytes != (size_t)8 * n * batch) return 4;

  SampleBuffer buffer(SampleBuffer::FloatComplex, 12, n);
  std::thread producer([&] {
    for (uint32_t b = 0; b < total; b++)
      buffer.AppendSamples(reinterpret_cast<fftwf_complex *>(iq.data() + (size_t)2 * n * b), 100e6 + 6e6 * b);
    buffer.SetIsDone();
  });

  FILE *out = fopen(argv[6], "wb");
  if (!out) return 5;
  std::vector<float> power((size_t)n * batch);
  std::vector<scn_hit> hits((size_t)n * batch);
  std::vector<uint8_t> trigger(batch);
  std::vector<double> fc[SCN_NUM_SLOTS];
  std::vector<uint64_t> seq[SCN_NUM_SLOTS];
  uint32_t inSlot[SCN_NUM_SLOTS] = {0, 0};
  bool pending[SCN_NUM_SLOTS] = {false, false};
  uint64_t nextSeq = 0;
  uint32_t allHits = 0;
  auto drain = [&](int s) -> int {
    uint32_t nHits = 0;
    CHECK_SCN(scn_collect(plan, s, power.data(), hits.data(), (uint32_t)hits.size(), &nHits, trigger.data()));
    fwrite(&inSlot[s], sizeof(uint32_t), 1, out);
    fwrite(&nHits, sizeof(uint32_t), 1, out);
    fwrite(power.data(), sizeof(float), (size_t)n * inSlot[s], out);
    fwrite(hits.data(), sizeof(scn_hit), nHits, out);
    fwrite(trigger.data(), 1, inSlot[s], out);
    allHits += nHits;
    pending[s] = false;
    return 0;
  };
  int slot = 0;
  bool more = true;
  while (more) {
    if (pending[slot] && drain(slot)) return 6;
    fc[slot].clear();
    seq[slot].clear();
    uint32_t k = 0;
    for (; k < batch; k++) {
      HipStagingProcessInterface visitor(slotBase[slot], n, k);  // buffer k of the pinned slot
      double centre = 0;
      if (!buffer.ProcessNext(&visitor, centre)) {
        more = false;
        break;
      }
      fc[slot].push_back(centre);
      seq[slot].push_back(nextSeq++);
    }
    inSlot[slot] = k;
    if (k) {
      CHECK_SCN(scn_submit(plan, slot, k, fc[slot].data(), seq[slot].data()));  // H2D on its own stream + kernel
      pending[slot] = true;
    }
    slot ^= 1;
  }
  for (int s = 0; s < SCN_NUM_SLOTS; s++) {
    const int which = (slot + s) & 1;  // older submit first
    if (pending[which] && drain(which)) return 6;
  }
  producer.join();
  fclose(out);
  scn_plan_destroy(plan);
  printf("staged %llu buffers in batches of %u over %d pinned slots, %u hits\n", (unsigned long long)nextSeq, batch, SCN_NUM_SLOTS, allHits);
  return nextSeq == total ? 0 : 7;
}
