// CPU-only checks of the kept C++ host classes (no GPU, no HIP calls): queue semantics the
// reference defines (messageQueue.h:65-91,239-273), raw wire formats, SampleBuffer visitors,
// FrequencyTable cursor, SyntheticSource determinism and the producer thread lifecycle.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <vector>

#include "buffer.h"
#include "frequencyTable.h"
#include "messageQueue.h"
#include "sampleBuffer.h"
#include "fileSource.h"
#include "syntheticSource.h"

static int g_fail = 0;
#define CHECK(c)                                                      \
  do {                                                                \
    if (!(c)) {                                                       \
      fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #c);    \
      g_fail++;                                                       \
    }                                                                 \
  } while (0)

static void test_queue_basic() {
  const uint32_t n = 8;
  SampleQueue q(SampleQueue::ShortComplex, 12, n, 4, false, false);
  int16_t buf[n][2];
  auto fill = [&](int tag) { for (uint32_t i = 0; i < n; i++) { buf[i][0] = (int16_t)(tag * 100 + i); buf[i][1] = (int16_t)-(tag * 100 + (int)i); } };
  // first sweep (time != 0 once) is discarded: messageQueue.h:67-72
  fill(1); q.AppendSamples(buf, 1e6, 111);
  fill(2); q.AppendSamples(buf, 2e6, 0);
  CHECK(q.TryGetNextSamples() == nullptr);
  fill(3); q.AppendSamples(buf, 3e6, 222);   // second scan start: from here on messages flow
  fill(4); q.AppendSamples(buf, 4e6, 0);
  SampleQueue::MessageType *m = q.GetNextSamples();
  CHECK(m && m->GetHeader().m_sequenceId == 0 && m->GetHeader().m_frequency == 3e6 && m->GetHeader().m_time == 222);
  CHECK(m->GetRawBytes() == n * 4 && ((int16_t *)m->GetRawData())[2] == 301 && ((int16_t *)m->GetRawData())[3] == -301);
  CHECK(!q.ReceivedAck());                    // ClearAck on append
  q.SendAck();
  CHECK(q.ReceivedAck());
  q.MessageProcessed(m);
  m = q.TryGetNextSamples();
  CHECK(m && m->GetHeader().m_sequenceId == 1 && m->GetHeader().m_time == 0);
  q.MessageProcessed(m);
  q.SetIsDone();
  CHECK(q.GetIsDone() && q.GetNextSamples() == nullptr);
}

static void test_queue_blocking_and_recycle() {
  const uint32_t n = 16, depth = 3, total = 200;
  SampleQueue q(SampleQueue::FloatComplex, 12, n, depth, false, false);
  std::atomic<uint32_t> produced(0);
  std::thread prod([&] {
    std::vector<float> x(2 * n);
    q.AppendSamples((fftwf_complex *)x.data(), 0, 1);  // warm-up sweep marker
    for (uint32_t k = 0; k < total; k++) {
      x[0] = (float)k;
      q.AppendSamples((fftwf_complex *)x.data(), 1000.0 + k, k == 0 ? 2 : 0);
      produced++;
    }
    q.SetIsDone();
  });
  std::this_thread::sleep_for(std::chrono::milliseconds(50));
  CHECK(produced <= depth + 1);               // producer is blocked on the full ring (messageQueue.h:82-84)
  uint32_t got = 0;
  while (SampleQueue::MessageType *m = q.GetNextSamples()) {
    CHECK(m->GetHeader().m_sequenceId == got && m->GetData()[0][0] == (float)got);
    got++;
    q.MessageProcessed(m);                    // recycles through the history ring; the pool never runs dry
  }
  prod.join();
  CHECK(got == total);
}

static void test_queue_kinds() {
  const uint32_t n = 4;
  {
    SampleQueue q(SampleQueue::Short, 12, n, 2, false, false);
    int16_t re[n] = {1, 2, 3, 4}, im[n] = {-1, -2, -3, -4};
    q.AppendSamples(re, im, 0, 1); q.AppendSamples(re, im, 5.0, 1);
    SampleQueue::MessageType *m = q.GetNextSamples();
    int16_t *r = (int16_t *)m->GetRawData();
    CHECK(r[0] == 1 && r[3] == 4 && r[4] == -1 && r[7] == -4);   // planar: I block then Q block
    q.MessageProcessed(m); q.SetIsDone();
  }
  {
    SampleQueue q(SampleQueue::ByteComplex, 8, n, 2, true, false);
    int8_t s[n][2] = {{1, -1}, {2, -2}, {127, -128}, {0, 5}};
    q.AppendSamples(s, 0, 1); q.AppendSamples(s, 7.0, 1);
    SampleQueue::MessageType *m = q.GetNextSamples();
    CHECK(m->GetRawBytes() == 8 && ((int8_t *)m->GetRawData())[5] == -128 && q.GetCorrectDCOffset());
    q.MessageProcessed(m); q.SetIsDone();
  }
}

static void test_sample_buffer() {
  const uint32_t n = 32;
  SampleBuffer sb(SampleBuffer::FloatComplex, 12, n);
  std::vector<float> x(2 * n), y(2 * n), stage(2 * n * 3, -1.0f);
  for (uint32_t i = 0; i < 2 * n; i++) x[i] = (float)i;
  sb.AppendSamples((fftwf_complex *)x.data(), 10.0);
  sb.AppendSamples((fftwf_complex *)x.data(), 20.0);
  double fc = 0;
  CHECK(sb.GetNextSamples((fftwf_complex *)y.data(), fc) && fc == 10.0 && y == x);
  HipStagingProcessInterface visitor(stage.data(), n, 2);       // third buffer of a (fake) pinned slot
  CHECK(sb.ProcessNext(&visitor, fc) && fc == 20.0);
  CHECK(stage[2 * n * 2] == 0.0f && stage[2 * n * 3 - 1] == (float)(2 * n - 1) && stage[0] == -1.0f);
  sb.SetIsDone();
  CHECK(!sb.GetNextSamples((fftwf_complex *)y.data(), fc));
  SampleBuffer si(SampleBuffer::ShortComplex, 12, n);
  std::vector<int16_t> s(2 * n, 7), r(2 * n);
  si.AppendSamples((int16_t(*)[2])s.data(), 30.0);
  CHECK(si.GetNextRaw(r.data(), fc) && fc == 30.0 && r == s && si.GetBufferBytes() == 4 * n);
}

static void test_frequency_table() {
  FrequencyTable t(8000000, 0.0, 16384 * 6e6, 0.75, 0.0, true);
  CHECK(t.GetFrequencyCount() == 16384 && t.GetCurrentFrequency() == 3e6 && t.GetIsScanStart());
  CHECK(t.GetNextFrequency() == 9e6 && !t.GetIsScanStart() && t.GetIterationCount() == 0);
  for (uint32_t i = 2; i < 16384; i++) t.GetNextFrequency();
  CHECK(t.GetCurrentFrequency() == t.GetStopFrequency());
  CHECK(t.GetNextFrequency() == 3e6 && t.GetIsScanStart() && t.GetIterationCount() == 1);
  FrequencyTable one(8000000, 100e6, 0.0, 0.75, 0.0, true);
  CHECK(one.GetFrequencyCount() == 1 && one.GetCurrentFrequency() == 103e6);
}

static void test_synthetic_source() {
  const uint32_t n = 1024, fs = 8000000;
  SyntheticSource src(fs, n, 88e6, 100e6, SampleQueue::ShortComplex, 5, 0.01);
  src.AddEmitter(92.0e6, 0.25);
  std::vector<int16_t> a(2 * n), b(2 * n), c(2 * n);
  src.Generate(91e6, 3, a.data());
  src.Generate(91e6, 3, b.data());
  src.Generate(91e6, 4, c.data());
  CHECK(a == b && a != c);                                  // pure function of (seed, index, fc)
  double e_in = 0, e_out = 0;
  for (uint32_t i = 0; i < 2 * n; i++) e_in += (double)a[i] * a[i];
  src.Generate(97e6, 3, b.data());                          // emitter 5 MHz away: outside +-fs/2
  for (uint32_t i = 0; i < 2 * n; i++) e_out += (double)b[i] * b[i];
  CHECK(e_in > 100 * e_out);
  // producer thread: 3 sweeps requested, the first is the queue's warm-up discard
  SampleQueue q(SampleQueue::ShortComplex, 12, n, 8, false, false);
  src.StartStreaming(3, q);
  uint32_t got = 0, starts = 0;
  while (SampleQueue::MessageType *m = q.GetNextSamples()) {
    if (m->GetHeader().m_time != 0) starts++;
    got++;
    q.MessageProcessed(m);
  }
  src.StopStreaming();
  CHECK(got == 2 * src.GetFrequencyCount() && starts == 2);
}

static void test_file_source() {
  // a SyntheticSource dump replayed through FileSource delivers the same bytes in the same order
  const uint32_t n = 256, fs = 8000000;
  const char *path = "/tmp/scn_filesource_test.bin";
  std::vector<std::vector<int16_t>> sent;
  {
    SyntheticSource src(fs, n, 88e6, 100e6, SampleQueue::ShortComplex, 9, 0.05);
    src.SetDumpFile(path);
    SampleQueue q(SampleQueue::ShortComplex, 12, n, 64, false, false);
    src.StartStreaming(3, q);
    while (SampleQueue::MessageType *m = q.GetNextSamples()) {
      const int16_t *r = (const int16_t *)m->GetRawData();
      sent.push_back(std::vector<int16_t>(r, r + 2 * n));
      q.MessageProcessed(m);
    }
    src.StopStreaming();
  }
  FileSource fsrc(path, fs, n, 88e6, 100e6, SampleQueue::ShortComplex);
  SampleQueue q(SampleQueue::ShortComplex, 12, n, 64, false, false);
  fsrc.StartStreaming(100, q);  // more sweeps than the file holds: end of file ends the stream
  size_t k = 0;
  double lastFc = 0;
  while (SampleQueue::MessageType *m = q.GetNextSamples()) {
    const int16_t *r = (const int16_t *)m->GetRawData();
    CHECK(k < sent.size() && std::vector<int16_t>(r, r + 2 * n) == sent[k]);
    lastFc = m->GetHeader().m_frequency;
    k++;
    q.MessageProcessed(m);
  }
  fsrc.StopStreaming();
  // the dump holds 3 sweeps (incl. the warm-up one the first queue discarded); replay discards its own first sweep
  CHECK(k == sent.size() && fsrc.GetBuffersRead() == 3 * fsrc.GetFrequencyCount() && lastFc > 88e6);
  remove(path);
}

// Triggered capture as a queue of jobs the writer owns: a second BeginWrite while the first capture is still
// draining (the converter is slow: a GPU round trip per record for the integer formats) must neither close the
// file in use nor lose the first capture's tail; both files end up complete, each closed by the writer.
static std::vector<float> read_floats(const char *path) {
  std::vector<float> v;
  if (FILE *f = fopen(path, "rb")) {
    float x;
    while (fread(&x, sizeof x, 1, f) == 1) v.push_back(x);
    fclose(f);
  }
  return v;
}

static void test_capture_retrigger_while_draining() {
  const uint32_t n = 8;
  const char *a = "/tmp/scn_capture_a.bin", *b = "/tmp/scn_capture_b.bin";
  {
    SampleQueue q(SampleQueue::ShortComplex, 12, n, 16, false, /*doWrite=*/true);
    std::atomic<uint32_t> converted(0);
    q.SetConverter([&](const void *raw, uint32_t nb, float *out) {
      std::this_thread::sleep_for(std::chrono::milliseconds(5));  // slower than the consumer below
      const int16_t *s = static_cast<const int16_t *>(raw);
      for (uint32_t i = 0; i < 2 * n * nb; i++) out[i] = (float)s[i];
      converted += nb;
    });
    int16_t buf[n][2];
    auto push = [&](int tag, time_t t) {
      for (uint32_t i = 0; i < n; i++) buf[i][0] = buf[i][1] = (int16_t)tag;
      q.AppendSamples(buf, 1e6, t);
    };
    push(-1, 1);  // warm-up sweep, discarded
    auto feed = [&](int tag, time_t t) {  // produce + consume one message (sequence id == tag)
      push(tag, t);
      SampleQueue::MessageType *m = q.GetNextSamples();
      CHECK(m && m->GetHeader().m_sequenceId == (uint64_t)tag);
      q.MessageProcessed(m);
    };
    feed(0, 2);
    feed(1, 0);
    q.BeginWrite(0, a);  // capture A: ids [0, 4)
    feed(2, 0);
    feed(3, 0);
    q.EndWrite(4);
    feed(4, 0);
    q.BeginWrite(3, b);  // re-trigger at once, pre-trigger reaches back into A: ids [3, 7); A is still draining
    CHECK(converted < 4);
    feed(5, 0);
    feed(6, 0);
    q.EndWrite(7);
    feed(7, 0);
    q.SetIsDone();
    CHECK(q.GetWriteErrorCount() == 0);
  }  // ~SampleQueue joins the writer
  std::vector<float> fa = read_floats(a), fb = read_floats(b);
  CHECK(fa.size() == 4 * 2 * n && fb.size() == 4 * 2 * n);
  for (size_t k = 0; k < fa.size() && fa.size() == 4 * 2 * n; k++) CHECK(fa[k] == (float)(k / (2 * n)));
  for (size_t k = 0; k < fb.size() && fb.size() == 4 * 2 * n; k++) CHECK(fb[k] == (float)(3 + k / (2 * n)));
  remove(a);
  remove(b);
}

// errors are reported, not exit(1): a library must leave that decision to its caller
static void test_failures_are_reported() {
  FileSource missing("/nonexistent/dir/iq.bin", 8000000, 64, 88e6, 100e6, SampleQueue::ShortComplex);
  SampleQueue q(SampleQueue::ShortComplex, 12, 64, 4, false, false);
  CHECK(!missing.Start() && !missing.StartStreaming(1, q));
  FileWriteProcessInterface w("/nonexistent/dir/out.bin");
  CHECK(w.Failed());
  SyntheticSource src(8000000, 64, 88e6, 100e6, SampleQueue::ShortComplex);
  CHECK(!src.SetDumpFile("/nonexistent/dir/dump.bin"));
  src.SetSweepFraming(1, 0);  // sweep framing needs byte IQ
  CHECK(!src.StartStreaming(1, q));
  bool threw = false;
  try {
    FrequencyTable bad(8000000, 0.0, 1e9, 0.0, 0.0, true);  // zero step
  } catch (const std::invalid_argument &) {
    threw = true;
  }
  CHECK(threw);
  q.SetIsDone();
}

// The staging rings of SampleQueue::AttachStaging, single-threaded and deterministic: what is queued before the first attach is handed
// out first (unstaged, with a slot reserved), then every append lands IN PLACE in a consumer's slot, the batches dealt to the rings in
// turn, each ring seeing increasing sequence ids; a sealed slot makes the producer move on; a detached ring takes its queued messages
// with it and the other ring goes on; with the last ring gone appends are pooled messages again.
static void test_staging_rings() {
  const uint32_t n = 8, cap = 3, slots = 2;
  const size_t bytes = n * 4;
  SampleQueue q(SampleQueue::ShortComplex, 12, n, 64, false, false);
  int16_t buf[n][2];
  uint32_t tag = 0;
  auto append = [&](time_t t = 0) {
    for (uint32_t i = 0; i < n; i++) buf[i][0] = buf[i][1] = (int16_t)tag;
    q.AppendSamples(buf, 1e6 * tag, t);
    tag++;
  };
  append(1);      // the discarded warm-up sweep
  append(2);      // seq 0: from here on messages flow
  append();       // seq 1 -- both queued BEFORE any consumer attaches
  std::vector<unsigned char> mem[2][slots];
  void *bases[2][slots];
  for (int r = 0; r < 2; r++)
    for (uint32_t k = 0; k < slots; k++) {
      mem[r][k].assign(bytes * cap, 0xee);
      bases[r][k] = mem[r][k].data();
    }
  const int r0 = q.AttachStaging(bases[0], slots, cap), r1 = q.AttachStaging(bases[1], slots, cap);
  CHECK(r0 == 0 && r1 == 1 && q.GetQueuedAtAttachCount() == 2);
  CHECK(q.AttachStaging(bases[0], slots, cap + 1) == -1);  // one batch size for all the consumers of a queue
  std::vector<SampleQueue::MessageType *> out;
  int slot = -1;
  // the unstaged ones first: whoever asks gets them with its next slot reserved, and copies them itself
  CHECK(q.TakeStagedBatch(r1, out, &slot, false) == 2 && slot == 0 && out[0]->GetStagingSlot() < 0 && out[0]->GetHeader().m_sequenceId == 0 &&
        out[1]->GetHeader().m_sequenceId == 1);
  for (auto *m : out) q.MessageProcessed(m);
  out.clear();
  CHECK(q.TakeStagedBatch(r0, out, &slot, false) == 0);
  // now every append lands in place; batches of `cap` go to the rings in turn -- a new batch starts in the ring AFTER the one filled
  // last: ring 1 first (its slot 0 is reserved: slot 1), then ring 0, then -- ring 1 having no usable slot -- ring 0 again
  for (int k = 0; k < 7; k++) append();  // seq 2 .. 8
  CHECK(q.GetStagedAppendCount() == 7 && q.GetCopiedAppendCount() == 2);
  CHECK(q.TakeStagedBatch(r1, out, &slot, false) == 3 && slot == 1);
  CHECK(out[0]->GetHeader().m_sequenceId == 2 && out[2]->GetHeader().m_sequenceId == 4 && out[1]->GetRawData() == mem[1][1].data() + bytes);
  CHECK(((int16_t *)mem[1][1].data())[0] == 3 && ((int16_t *)(mem[1][1].data() + 2 * bytes))[0] == 5);  // tags 3, 4, 5 = seq 2, 3, 4: IN the slot
  out.clear();
  CHECK(q.TakeStagedBatch(r0, out, &slot, false) == 3 && slot == 0 && out[0]->GetHeader().m_sequenceId == 5 && out[0]->GetRawData() == mem[0][0].data());
  out.clear();
  CHECK(q.TakeStagedBatch(r0, out, &slot, false) == 1 && slot == 1 && out[0]->GetHeader().m_sequenceId == 8);  // a partial batch, sealed by being taken
  out.clear();
  // ring 0: slot 0 in flight (never released), slot 1 just sealed; ring 1: slot 0 reserved, slot 1 in flight -> nothing usable: release one
  q.ReleaseStaging(r1, 0);
  append();  // seq 9 -> ring 1, slot 0
  append();  // seq 10: the batch in progress goes on in the same slot
  CHECK(q.TakeStagedBatch(r0, out, &slot, false) == 0);
  CHECK(q.TakeStagedBatch(r1, out, &slot, false) == 2 && slot == 0 && out[0]->GetHeader().m_sequenceId == 9 && out[1]->GetRawData() == mem[1][0].data() + bytes);
  out.clear();
  // a consumer that leaves takes its queued messages with it; the other ring goes on
  q.ReleaseStaging(r0, 0);
  append();  // seq 11 -> ring 0, slot 0
  q.DetachStaging(r0);
  CHECK(q.TakeStagedBatch(r0, out, &slot, false) == 0);
  q.ReleaseStaging(r1, 1);
  append();  // seq 12 -> ring 1 (the only one left), slot 1
  CHECK(q.TakeStagedBatch(r1, out, &slot, false) == 1 && slot == 1 && out[0]->GetHeader().m_sequenceId == 12);
  out.clear();
  // the last ring gone: appends are pooled messages again, taken the reference's way
  q.DetachStaging(r1);
  append();  // seq 13
  SampleQueue::MessageType *m = q.TryGetNextSamples();
  CHECK(m && m->GetStagingSlot() < 0 && m->GetHeader().m_sequenceId == 13 && ((int16_t *)m->GetRawData())[0] == (int16_t)(tag - 1));
  q.MessageProcessed(m);
  q.SetIsDone();
  CHECK(q.GetNextSamples() == nullptr);
}

int main() {
  test_staging_rings();
  test_capture_retrigger_while_draining();
  test_failures_are_reported();
  test_queue_basic();
  test_queue_blocking_and_recycle();
  test_queue_kinds();
  test_sample_buffer();
  test_frequency_table();
  test_synthetic_source();
  test_file_source();
  if (g_fail) {
    fprintf(stderr, "%d check(s) failed\n", g_fail);
    return 1;
  }
  printf("host cpu tests ok\n");
  return 0;
}
