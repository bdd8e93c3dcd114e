#include "process.h"

#include <algorithm>
#include <cassert>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>

#include "../../include/scanner_hip.h"

// The reference's device layers print and exit(1) on any library error (hackRFSource.cpp:19-30).  This is a library:
// a failed C-ABI call is logged, remembered (GetLastError) and makes StartProcessing / Run return false.
bool ProcessSamples::Ok(int st, const char *what) {
  if (st == SCN_OK) return true;
  char text[600];
  snprintf(text, sizeof(text), "%s failed: %s: %s", what, scn_error_name(st), scn_last_error());
  fprintf(stderr, "%s\n", text);
  std::lock_guard<std::mutex> g(m_errorMutex);
  if (m_error.empty()) m_error = text;
  m_failed = true;
  return false;
}

bool ProcessSamples::Fail(const std::string &text) {
  fprintf(stderr, "%s\n", text.c_str());
  std::lock_guard<std::mutex> g(m_errorMutex);
  if (m_error.empty()) m_error = text;
  m_failed = true;
  return false;
}

std::string ProcessSamples::GetLastError() {
  std::lock_guard<std::mutex> g(m_errorMutex);
  return m_error;
}

ProcessSamples::ProcessSamples(uint32_t numSamples, uint32_t sampleRate, uint32_t enob, float threshold,
                               gr::fft::window::win_type windowType, Mode mode, uint32_t threadCount,
                               std::string fileNameBase, double useBandWidth, double dcIgnoreWidth,
                               uint32_t preTrigger, uint32_t postTrigger)
    : m_writeData(false), m_sampleCount(numSamples), m_sampleRate(sampleRate), m_enob(enob), m_fileCounter(0),
      m_preTrigger(preTrigger), m_postTrigger(postTrigger), m_endSequenceId(0), m_writing(false), m_mode(mode),
      m_fileNameBase(fileNameBase), m_threshold(threshold), m_useBandWidth(useBandWidth), m_windowType(windowType),
      m_sampleQueue(nullptr), m_threadCount(threadCount), m_maxBatch(1024), m_pipeDepth(4), m_firstDevice(0), m_hitCount(0),
      m_bufferCount(0), m_stagedWorkers(0), m_indexedSubmits(0), m_tWait(0), m_tSubmit(0), m_tCollect(0), m_tReport(0), m_failed(false) {
  (void)dcIgnoreWidth;  // the reference ignores it too and hard-codes 4 bins (process.cpp:86-88)
  assert(mode > Illegal && mode <= FrequencyDomain);  // process.cpp:99
  assert(threadCount <= MAX_THREADS);                 // process.cpp:100
}

ProcessSamples::~ProcessSamples() {}

void ProcessSamples::TimeToString(time_t time, char *buffer, uint32_t length) {  // process.cpp:146-158
  struct tm parts;
  if (!localtime_r(&time, &parts) || strftime(buffer, length, "%Y%m%d-%T", &parts) == 0)
    snprintf(buffer, length, "%lld", (long long)time);  // (the reference exits; a timestamp is not worth a process)
}

std::string ProcessSamples::GenerateFileName(std::string base, time_t startTime, double_t fc) {  // :160-171
  char timeBuffer[64], tail[64];
  TimeToString(startTime, timeBuffer, sizeof(timeBuffer));
  snprintf(tail, sizeof(tail), "-%.0f-%u", fc, ++m_fileCounter);
  return base + timeBuffer + tail;
}

void ProcessSamples::WriteSamplesToFile(uint64_t sequenceId, double fc) {  // process.cpp:173-181
  assert(m_sampleQueue != nullptr);
  std::string fileName = GenerateFileName(m_fileNameBase, time(NULL), fc);
  uint64_t decrement = std::min<uint64_t>(sequenceId, m_preTrigger);
  m_sampleQueue->BeginWrite(sequenceId - decrement, fileName);
}

void ProcessSamples::UpdateEndSequenceId(uint64_t newEnd) {  // process.cpp:239-248
  uint64_t cur = m_endSequenceId;
  while (cur < newEnd && !m_endSequenceId.compare_exchange_weak(cur, newEnd)) {
  }
}

void ProcessSamples::ProcessWrite(bool doWrite, double fc, uint64_t sequenceId) {  // process.cpp:250-270
  if (m_writing) {
    if (doWrite) {
      UpdateEndSequenceId(sequenceId + m_postTrigger + 1);
    } else if (sequenceId == m_endSequenceId) {
      m_sampleQueue->EndWrite(sequenceId);
      m_writing = false;
    }
  } else if (doWrite) {
    if (m_fileNameBase != "") {
      WriteSamplesToFile(sequenceId, fc);
      m_writing = true;
      UpdateEndSequenceId(sequenceId + m_postTrigger + 1);
    }
  }
}

static uint32_t planKind(SampleQueue::SampleKind k) { return (uint32_t)k; }  // same numbering by construction

void ProcessSamples::ThreadWorker(uint32_t threadId) {
  SampleQueue &q = *m_sampleQueue;
  const bool timeDomain = m_mode == TimeDomain;
  scn_plan_desc d;
  memset(&d, 0, sizeof(d));
  d.struct_size = sizeof(d);
  d.n = m_sampleCount;
  d.sample_rate = m_sampleRate;
  d.sample_kind = planKind(q.m_kind);
  d.enob = m_enob;
  d.correct_dc = q.GetCorrectDCOffset();
  d.window_type = m_windowType == gr::fft::window::WIN_HAMMING ? (uint32_t)SCN_WIN_HAMMING : (uint32_t)m_windowType;  // (0 is "the default" in the descriptor)
  d.mode = timeDomain ? SCN_MODE_TIME_DOMAIN : SCN_MODE_FREQUENCY_DOMAIN;
  d.threshold = m_threshold;
  d.use_bandwidth = m_useBandWidth;
  d.max_batch = std::min<uint32_t>(m_maxBatch, std::max<uint32_t>(1u, q.GetBufferCount()));
  d.max_hits = d.max_batch * 64u;
  d.flags = SCN_OUT_HITS;  // the reference reports hits only; the spectra never leave the GPU
  // a worker that cannot go on still has to empty the queue, or the producer blocks on a full one forever
  int ring = -1;  // this consumer's staging ring in the queue (SampleQueue::AttachStaging); -1: the copying path
  auto abandon = [&](scn_plan *plan) {
    q.DetachStaging(ring);  // (the producer may be waiting for a slot of this plan)
    if (plan) scn_plan_destroy(plan);
    while (SampleQueue::MessageType *m = q.GetNextSamples()) q.MessageProcessed(m);
  };
  int nDevices = 1;
  if (!Ok(scn_device_count(&nDevices), "scn_device_count")) return abandon(nullptr);
  d.device_id = (m_firstDevice + (int)threadId) % nDevices;  // consumer threads spread over the node's GPUs
  scn_plan *plan = nullptr;
  if (!Ok(scn_plan_create(&d, &plan), "scn_plan_create")) return abandon(nullptr);

  // Batches in flight.  This consumer prints every record of every batch (process.cpp:57), so a submit's chain is kernel ->
  // list kernels -> DMA, longer than one kernel: three in flight keep the GPU fed where two left it idle between launches
  // (HISTORY.md section 8, the records pipeline); a batch's lines still appear as soon as the queue runs empty, and always in
  // submit order.
  const int kPipe = (int)std::min<uint32_t>(std::max<uint32_t>(m_pipeDepth, 1u), (uint32_t)SCN_NUM_SLOTS);
  unsigned char *stage[SCN_NUM_SLOTS];
  size_t stageBytes = 0, bufBytes = 0;
  for (int s = 0; s < kPipe; s++)
    if (!Ok(scn_host_buffer(plan, s, (void **)&stage[s], &stageBytes), "scn_host_buffer")) return abandon(plan);
  if (!Ok(scn_buffer_bytes(plan, &bufBytes), "scn_buffer_bytes")) return abandon(plan);
  if (bufBytes != q.GetBufferBytes()) {  // the queue was built for another sample kind / count than this ProcessSamples
    Fail("ProcessSamples: the queue's buffers are " + std::to_string(q.GetBufferBytes()) + " bytes, the plan's " + std::to_string(bufBytes) +
         " (sample kind or count mismatch)");
    return abandon(plan);
  }
  // The queue writes the producer's buffers straight into the plan's pinned slots and a slot is submitted as it is -- no copy in
  // this thread (the reference's ThreadWorker copies every buffer, process.cpp:293; so did rounds 1-3 here).  Every consumer thread
  // lends the queue a ring of its own (scan.cpp:217 runs two: the producer deals its batches to them in turn); only a capturing
  // queue, whose history ring needs storage of its own, keeps the copying path.
  // (What the producer queued while the threads were creating their plans comes first, unstaged, with a slot reserved for it.)
  if (kPipe >= 2) ring = q.AttachStaging((void *const *)stage, (uint32_t)kPipe, d.max_batch);
  const bool staged = ring >= 0;
  if (staged) m_stagedWorkers++;

  // the sweep's frequency table on the GPU (SetFrequencyTable): a batch that is a consecutive, wrapping run of it names its first entry
  const std::vector<double> &table = m_tableCentres;
  const bool useTable = !table.empty() && !timeDomain && Ok(scn_plan_set_table(plan, table.data(), (uint32_t)table.size()), "scn_plan_set_table");
  uint32_t tableNext = 0;  // where the run is expected to go on
  auto tableRun = [&](const std::vector<double> &f, uint32_t n, uint32_t *first) -> bool {
    const uint32_t count = (uint32_t)table.size();
    uint32_t at = tableNext;
    if (table[at] != f[0]) {  // (another consumer took the batches in between, or the stream started mid-table: look it up)
      at = count;
      for (uint32_t i = 0; i < count; i++)
        if (table[i] == f[0]) {
          at = i;
          break;
        }
      if (at == count) return false;
    }
    for (uint32_t b = 0; b < n; b++)
      if (table[(at + b) % count] != f[b]) return false;
    *first = at;
    tableNext = (at + n) % count;
    return true;
  };

  std::vector<double> fc(d.max_batch);
  std::vector<uint64_t> seq(d.max_batch);
  std::vector<uint8_t> trig(d.max_batch);
  std::vector<float> tdMax(d.max_batch), tdMin(d.max_batch);
  std::vector<scn_hit> window;  // records beyond the plan's pinned list (a wideband burst), fetched window by window
  std::vector<SampleQueue::MessageType *> inflight[SCN_NUM_SLOTS];
  bool pending[SCN_NUM_SLOTS] = {};
  uint64_t lastSequenceId = 0;
  double lastFrequency = 0;

  auto nowNs = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  auto drain = [&](int s) {
    const uint64_t tDrain0 = nowNs();
    uint32_t nHits = 0, have = 0;
    const scn_hit *hits = nullptr;  // the batch's ordered records, read IN PLACE from the plan's pinned list (scn_hits_view)
    int st = SCN_OK;
    if (timeDomain) {
      st = scn_collect_time_domain(plan, s, tdMax.data(), tdMin.data(), trig.data());
    } else {
      st = scn_collect(plan, s, nullptr, nullptr, 0, &nHits, trig.data());
      if (st == SCN_OK && nHits) st = scn_hits_view(plan, s, &hits, &have);
    }
    // More detections than the plan's pinned list holds (a wideband burst: every triggered buffer alone has > 1047): the
    // list on the GPU is complete and ordered, so the rest is fetched window by window below -- the reference prints every line.
    bool failed = !Ok(st, timeDomain ? "scn_collect_time_domain" : "scn_collect / scn_hits_view");
    const uint64_t tDrain1 = nowNs();
    m_tCollect += tDrain1 - tDrain0;
    uint32_t first = 0;  // index, in the batch's ordered hit list, of hits[0]
    size_t k = 0;
    for (size_t b = 0; b < inflight[s].size(); b++) {
      SampleQueue::MessageType *m = inflight[s][b];
      SampleQueue::MessageHeader &h = m->GetHeader();
      if (failed) {  // nothing to report for this batch; hand the message back so the producer can go on
        q.MessageProcessed(m);
        continue;
      }
      if (h.m_time != 0) {  // process.cpp:280-287
        char timeBuffer[64];
        TimeToString(h.m_time, timeBuffer, sizeof(timeBuffer));
        printf("Start scan at %s\n", timeBuffer);
        fflush(stdout);
      }
      if (timeDomain && trig[b]) {  // process.cpp:226-233
        printf("Sequence[%llu]: ", (unsigned long long)h.m_sequenceId);
        printf("Max signal %f above threshold %f frequency %.0f, min %f\n", tdMax[b], m_threshold, h.m_frequency,
               tdMin[b]);
        nHits++;
      }
      while (!timeDomain && first + k < nHits) {  // hits arrive ordered by (buffer, i)
        if (k == have) {                          // window exhausted: fetch the next one
          first += have;
          window.resize(std::min<size_t>(nHits - first, (size_t)d.max_batch * 64u));
          if (!Ok(scn_collect_more(plan, s, first, window.data(), (uint32_t)window.size(), &have), "scn_collect_more") || !have) {
            nHits = first;  // give up on the rest of this batch's lines
            break;
          }
          hits = window.data();
          k = 0;
        }
        if (hits[k].seq_id != h.m_sequenceId) break;
        printf("freq %lu power_db %f\n", (unsigned long)hits[k].freq_hz, hits[k].power_db);  // process.cpp:57
        k++;
      }
      bool doWrite = trig[b] != 0;
      if (doWrite) fflush(stdout); else q.SendAck();          // process.cpp:303-307
      ProcessWrite(doWrite, h.m_frequency, h.m_sequenceId);   // process.cpp:308
      lastSequenceId = h.m_sequenceId;
      lastFrequency = h.m_frequency;
      q.MessageProcessed(m);                                  // process.cpp:309
    }
    if (!failed) m_hitCount += nHits;
    m_bufferCount += inflight[s].size();
    inflight[s].clear();
    pending[s] = false;
    if (staged) q.ReleaseStaging(ring, s);  // the slot goes back to the producer
    m_tReport += nowNs() - tDrain1;
  };

  // A ring of kPipe slots: `head` is filled next, the oldest submit in flight is `inFlight` behind it.  Results are
  // reported in submit order (drain always takes the oldest), as a single reference thread prints them.
  int head = 0, inFlight = 0;
  bool more = true;
  auto oldest = [&] { return (head + kPipe - inFlight) % kPipe; };
  while (more || inFlight) {
    if (pending[head]) {  // the ring is full: the slot to refill is the oldest one
      drain(head);
      inFlight--;
    }
    uint32_t n = 0;
    const uint64_t tTake0 = nowNs();
    if (more && staged) {
      // every queued message of the queue's oldest slot -- which is `head`: the queue fills its slots in ring order and a
      // slot is sealed by being taken.  Block only while nothing is in flight.
      int slot = -1;
      n = q.TakeStagedBatch(ring, inflight[head], &slot, inFlight == 0, 40);
      if (!n && !inFlight) more = false;
      // (The queue fills -- and, for messages queued before the attach, reserves -- its slots in ring order, the order this
      //  thread submits them in; anything else means the slot about to be submitted is not the memory the samples are in.)
      if (n && slot != head) {
        Fail("ProcessSamples: the queue handed out staging slot " + std::to_string(slot) + ", the worker expected " + std::to_string(head));
        for (SampleQueue::MessageType *m : inflight[head]) q.MessageProcessed(m);
        inflight[head].clear();
        while (inFlight) {
          drain(oldest());
          inFlight--;
        }
        return abandon(plan);
      }
      for (uint32_t b = 0; b < n; b++) {
        SampleQueue::MessageType *const m = inflight[head][b];
        if (m->GetStagingSlot() < 0) memcpy(stage[head] + (size_t)b * bufBytes, m->GetRawData(), bufBytes);  // queued before the attach
        fc[b] = m->GetHeader().m_frequency;
        seq[b] = m->GetHeader().m_sequenceId;
      }
    } else if (more) {
      // block for the first message only while nothing is in flight; then take what is queued
      SampleQueue::MessageType *m = inFlight ? q.TryGetNextSamples() : q.GetNextSamples();
      if (!m && !inFlight) more = false;
      while (m) {
        memcpy(stage[head] + (size_t)n * bufBytes, m->GetRawData(), bufBytes);
        fc[n] = m->GetHeader().m_frequency;
        seq[n] = m->GetHeader().m_sequenceId;
        inflight[head].push_back(m);
        n++;
        if (n >= d.max_batch) break;
        m = q.TryGetNextSamples();
      }
    }
    const uint64_t tTake1 = nowNs();
    m_tWait += tTake1 - tTake0;
    if (n) {
      uint32_t first = 0;
      const bool indexed = useTable && tableRun(fc, n, &first);
      const int stSubmit = indexed ? scn_submit_indexed(plan, head, n, first, seq.data()) : scn_submit(plan, head, n, fc.data(), seq.data());
      if (indexed) m_indexedSubmits++;
      m_tSubmit += nowNs() - tTake1;
      if (Ok(stSubmit, "scn_submit")) {
        pending[head] = true;
        inFlight++;
        head = (head + 1) % kPipe;
      } else {  // the GPU path is gone: hand everything back and stop consuming
        for (SampleQueue::MessageType *m : inflight[head]) q.MessageProcessed(m);
        inflight[head].clear();
        while (inFlight) {
          drain(oldest());
          inFlight--;
        }
        return abandon(plan);
      }
    } else if (inFlight) {  // nothing queued right now: report the oldest batch instead of spinning
      drain(oldest());
      inFlight--;
    }
  }
  // Shutdown writing gracefully (process.cpp:311-313).
  UpdateEndSequenceId(lastSequenceId);
  ProcessWrite(false, lastFrequency, lastSequenceId);
  q.DetachStaging(ring);
  scn_plan_destroy(plan);
}

bool ProcessSamples::StartProcessing(SampleQueue &sampleQueue) {  // process.cpp:316-331
  m_sampleQueue = &sampleQueue;
  // (a start that fails still empties the queue, or the producer blocks on a full one forever)
  auto refuse = [&] {
    while (SampleQueue::MessageType *m = sampleQueue.GetNextSamples()) sampleQueue.MessageProcessed(m);
    return false;
  };
  if (scn_abi_version() != SCN_ABI_VERSION) {  // a libscanner_hip.so built from another header: slot count and entry points differ
    Fail("ProcessSamples: libscanner_hip.so has ABI version " + std::to_string(scn_abi_version()) + ", this code was built against " +
         std::to_string((unsigned)SCN_ABI_VERSION));
    return refuse();
  }
  // Triggered capture writes converted samples (fftwf_complex records); the queue holds raw ones, so give it K1
  // on the GPU through a small dedicated plan.  The converter is installed once, before any thread can call it,
  // and OWNS what it uses (plan and lock are shared_ptr captures): the queue's write thread may still be
  // converting its last records after this function has returned and after this object is gone.
  if (m_fileNameBase != "" && sampleQueue.m_kind != SampleQueue::FloatComplex) {
    scn_plan_desc d;
    memset(&d, 0, sizeof(d));
    d.struct_size = sizeof(d);
    d.n = m_sampleCount;
    d.sample_rate = m_sampleRate;
    d.sample_kind = planKind(sampleQueue.m_kind);
    d.enob = m_enob;
    d.correct_dc = sampleQueue.GetCorrectDCOffset();
    d.mode = SCN_MODE_TIME_DOMAIN;  // no FFT-size restriction; only scn_convert_raw is used
    d.threshold = m_threshold;
    d.max_batch = 1;
    d.device_id = m_firstDevice;
    scn_plan *raw = nullptr;
    if (!Ok(scn_plan_create(&d, &raw), "scn_plan_create")) return refuse();
    std::shared_ptr<scn_plan> plan(raw, [](scn_plan *p) { scn_plan_destroy(p); });
    std::shared_ptr<std::mutex> lock = std::make_shared<std::mutex>();
    const uint32_t n = m_sampleCount;
    sampleQueue.SetConverter([plan, lock, n](const void *in, uint32_t nBuffers, float *out) {
      std::lock_guard<std::mutex> g(*lock);
      if (scn_convert_raw(plan.get(), in, nBuffers, out) != SCN_OK) {
        fprintf(stderr, "scn_convert_raw failed: %s\n", scn_last_error());
        memset(out, 0, sizeof(float) * 2 * (size_t)n * nBuffers);  // keep the record layout of the capture file
      }
    });
  }
  std::vector<std::thread> threads;
  for (uint32_t t = 0; t < m_threadCount; t++) {
    printf("Starting process thread %u\n", t);
    threads.emplace_back(&ProcessSamples::ThreadWorker, this, t);
  }
  for (uint32_t t = 0; t < m_threadCount; t++) {
    threads[t].join();
    printf("Stopped process thread %u\n", t);
  }
  return !m_failed;
}

bool ProcessSamples::Run(int16_t sample_buffer[][2], uint32_t centerFrequency) {
  if (m_mode != FrequencyDomain) return true;  // process.cpp:138
  scn_plan_desc d;
  memset(&d, 0, sizeof(d));
  d.struct_size = sizeof(d);
  d.n = m_sampleCount;
  d.sample_rate = m_sampleRate;
  d.sample_kind = SCN_KIND_SHORT_COMPLEX;
  d.enob = m_enob;
  d.correct_dc = 0;  // m_correctDCOffset(false), process.cpp:84
  d.window_type = m_windowType == gr::fft::window::WIN_HAMMING ? (uint32_t)SCN_WIN_HAMMING : (uint32_t)m_windowType;  // (0 is "the default" in the descriptor)
  d.threshold = m_threshold;
  d.use_bandwidth = m_useBandWidth;
  d.max_batch = 1;
  d.max_hits = m_sampleCount;
  d.flags = SCN_OUT_HITS;
  d.device_id = m_firstDevice;
  scn_plan *plan = nullptr;
  if (!Ok(scn_plan_create(&d, &plan), "scn_plan_create")) return false;
  void *stage = nullptr;
  size_t bytes = 0;
  std::vector<scn_hit> hits(m_sampleCount);
  uint32_t nHits = 0;
  double fc = centerFrequency;
  bool ok = Ok(scn_host_buffer(plan, 0, &stage, &bytes), "scn_host_buffer");
  if (ok) {
    memcpy(stage, sample_buffer, sizeof(int16_t) * 2 * m_sampleCount);
    ok = Ok(scn_submit(plan, 0, 1, &fc, nullptr), "scn_submit") &&
         Ok(scn_collect(plan, 0, nullptr, hits.data(), (uint32_t)hits.size(), &nHits, nullptr), "scn_collect");
  }
  if (ok) {
    for (uint32_t k = 0; k < nHits; k++)
      printf("freq %lu power_db %f\n", (unsigned long)hits[k].freq_hz, hits[k].power_db);
    m_hitCount += nHits;
    m_bufferCount += 1;
  }
  scn_plan_destroy(plan);
  return ok;
}
