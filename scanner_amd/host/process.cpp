#include "process.h"

#include <algorithm>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>

#include "../../include/scanner_hip.h"

namespace {
void check(int st, const char *what) {
  // the reference's device layers print and exit(1) on any library error (hackRFSource.cpp:19-30)
  if (st != SCN_OK) {
    fprintf(stderr, "%s failed: %s: %s\n", what, scn_error_name(st), scn_last_error());
    exit(1);
  }
}
}  // namespace

ProcessSamples::ProcessSamples(uint32_t numSamples, uint32_t sampleRate, uint32_t enob, float threshold,
                               gr::fft::window::win_type windowType, Mode mode, uint32_t threadCount,
                               std::string fileNameBase, double useBandWidth, double dcIgnoreWidth,
                               uint32_t preTrigger, uint32_t postTrigger)
    : m_writeData(false), m_sampleCount(numSamples), m_sampleRate(sampleRate), m_enob(enob), m_fileCounter(0),
      m_preTrigger(preTrigger), m_postTrigger(postTrigger), m_endSequenceId(0), m_writing(false), m_mode(mode),
      m_fileNameBase(fileNameBase), m_threshold(threshold), m_useBandWidth(useBandWidth), m_windowType(windowType),
      m_sampleQueue(nullptr), m_threadCount(threadCount), m_maxBatch(1024), m_firstDevice(0), m_hitCount(0),
      m_bufferCount(0), m_convertPlan(nullptr) {
  (void)dcIgnoreWidth;  // the reference ignores it too and hard-codes 4 bins (process.cpp:86-88)
  assert(mode > Illegal && mode <= FrequencyDomain);  // process.cpp:99
  assert(threadCount <= MAX_THREADS);                 // process.cpp:100
}

ProcessSamples::~ProcessSamples() {
  if (m_convertPlan) scn_plan_destroy(static_cast<scn_plan *>(m_convertPlan));
}

void ProcessSamples::TimeToString(time_t time, char *buffer, uint32_t length) {  // process.cpp:146-158
  struct tm *t = localtime(&time);
  if (!t) {
    perror("localtime");
    exit(1);
  }
  if (strftime(buffer, length, "%Y%m%d-%T", t) == 0) {
    fprintf(stderr, "strftime returned 0");
    exit(1);
  }
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
  d.window_type = (uint32_t)m_windowType;
  d.mode = timeDomain ? SCN_MODE_TIME_DOMAIN : SCN_MODE_FREQUENCY_DOMAIN;
  d.threshold = m_threshold;
  d.use_bandwidth = m_useBandWidth;
  d.max_batch = std::min<uint32_t>(m_maxBatch, std::max<uint32_t>(1u, q.GetBufferCount()));
  d.max_hits = d.max_batch * 64u;
  d.flags = SCN_OUT_HITS;  // the reference reports hits only; the spectra never leave the GPU
  int nDevices = 1;
  check(scn_device_count(&nDevices), "scn_device_count");
  d.device_id = (m_firstDevice + (int)threadId) % nDevices;  // consumer threads spread over the node's GPUs
  scn_plan *plan = nullptr;
  check(scn_plan_create(&d, &plan), "scn_plan_create");

  unsigned char *stage[SCN_NUM_SLOTS];
  size_t stageBytes = 0, bufBytes = 0;
  for (int s = 0; s < SCN_NUM_SLOTS; s++) check(scn_host_buffer(plan, s, (void **)&stage[s], &stageBytes), "scn_host_buffer");
  check(scn_buffer_bytes(plan, &bufBytes), "scn_buffer_bytes");
  assert(bufBytes == q.GetBufferBytes());

  std::vector<double> fc(d.max_batch);
  std::vector<uint64_t> seq(d.max_batch);
  std::vector<uint8_t> trig(d.max_batch);
  std::vector<float> tdMax(d.max_batch), tdMin(d.max_batch);
  std::vector<scn_hit> hits((size_t)d.max_batch * 64u);
  std::vector<SampleQueue::MessageType *> inflight[SCN_NUM_SLOTS];
  bool pending[SCN_NUM_SLOTS] = {false, false};
  uint64_t lastSequenceId = 0;
  double lastFrequency = 0;

  auto drain = [&](int s) {
    uint32_t nHits = 0;
    int st = SCN_OK;
    if (timeDomain)
      st = scn_collect_time_domain(plan, s, tdMax.data(), tdMin.data(), trig.data());
    else
      st = scn_collect(plan, s, nullptr, hits.data(), (uint32_t)hits.size(), &nHits, trig.data());
    if (st == SCN_E_TRUNCATED) {  // more detections than this worker buffers: report what was kept
      fprintf(stderr, "ProcessSamples: %u hits in one batch, reporting the first %zu\n", nHits, hits.size());
      nHits = (uint32_t)std::min<size_t>(nHits, hits.size());
    } else {
      check(st, "scn_collect");
    }
    size_t k = 0;
    for (size_t b = 0; b < inflight[s].size(); b++) {
      SampleQueue::MessageType *m = inflight[s][b];
      SampleQueue::MessageHeader &h = m->GetHeader();
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
      while (!timeDomain && k < nHits && hits[k].seq_id == h.m_sequenceId) {  // hits arrive ordered by (buffer, i)
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
    m_hitCount += nHits;
    m_bufferCount += inflight[s].size();
    inflight[s].clear();
    pending[s] = false;
  };

  int slot = 0;
  bool more = true;
  while (more || pending[0] || pending[1]) {
    if (pending[slot]) drain(slot);
    uint32_t n = 0;
    if (more) {
      // block for the first message only while nothing is in flight; then take what is queued
      SampleQueue::MessageType *m = (pending[slot ^ 1]) ? q.TryGetNextSamples() : q.GetNextSamples();
      if (!m && !pending[slot ^ 1]) more = false;
      while (m) {
        memcpy(stage[slot] + (size_t)n * bufBytes, m->GetRawData(), bufBytes);
        fc[n] = m->GetHeader().m_frequency;
        seq[n] = m->GetHeader().m_sequenceId;
        inflight[slot].push_back(m);
        n++;
        if (n >= d.max_batch) break;
        m = q.TryGetNextSamples();
      }
    }
    if (n) {
      check(scn_submit(plan, slot, n, fc.data(), seq.data()), "scn_submit");
      pending[slot] = true;
    }
    slot ^= 1;
  }
  // Shutdown writing gracefully (process.cpp:311-313).
  UpdateEndSequenceId(lastSequenceId);
  ProcessWrite(false, lastFrequency, lastSequenceId);
  scn_plan_destroy(plan);
}

bool ProcessSamples::StartProcessing(SampleQueue &sampleQueue) {  // process.cpp:316-331
  m_sampleQueue = &sampleQueue;
  // Triggered capture writes converted samples (fftwf_complex records); the queue holds raw ones,
  // so give it K1 on the GPU through a small dedicated plan.
  scn_plan *convertPlan = nullptr;
  std::mutex convertMutex;
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
    check(scn_plan_create(&d, &convertPlan), "scn_plan_create");
    sampleQueue.SetConverter([convertPlan, &convertMutex](const void *raw, uint32_t n, float *out) {
      std::lock_guard<std::mutex> g(convertMutex);
      check(scn_convert_raw(convertPlan, raw, n, out), "scn_convert_raw");
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
  if (convertPlan) {
    // the write thread may still be converting its last records: the queue's destructor joins it
    // before the plan could be needed again, so hand the plan's lifetime to the converter
    sampleQueue.SetConverter([convertPlan](const void *raw, uint32_t n, float *out) {
      check(scn_convert_raw(convertPlan, raw, n, out), "scn_convert_raw");
    });
    m_convertPlan = convertPlan;
  }
  return true;
}

void ProcessSamples::Run(int16_t sample_buffer[][2], uint32_t centerFrequency) {
  if (m_mode != FrequencyDomain) return;  // process.cpp:138
  scn_plan_desc d;
  memset(&d, 0, sizeof(d));
  d.struct_size = sizeof(d);
  d.n = m_sampleCount;
  d.sample_rate = m_sampleRate;
  d.sample_kind = SCN_KIND_SHORT_COMPLEX;
  d.enob = m_enob;
  d.correct_dc = 0;  // m_correctDCOffset(false), process.cpp:84
  d.window_type = (uint32_t)m_windowType;
  d.threshold = m_threshold;
  d.use_bandwidth = m_useBandWidth;
  d.max_batch = 1;
  d.max_hits = m_sampleCount;
  d.flags = SCN_OUT_HITS;
  d.device_id = m_firstDevice;
  scn_plan *plan = nullptr;
  check(scn_plan_create(&d, &plan), "scn_plan_create");
  void *stage = nullptr;
  size_t bytes = 0;
  check(scn_host_buffer(plan, 0, &stage, &bytes), "scn_host_buffer");
  memcpy(stage, sample_buffer, sizeof(int16_t) * 2 * m_sampleCount);
  double fc = centerFrequency;
  check(scn_submit(plan, 0, 1, &fc, nullptr), "scn_submit");
  std::vector<scn_hit> hits(m_sampleCount);
  uint32_t nHits = 0;
  check(scn_collect(plan, 0, nullptr, hits.data(), (uint32_t)hits.size(), &nHits, nullptr), "scn_collect");
  for (uint32_t k = 0; k < nHits; k++)
    printf("freq %lu power_db %f\n", (unsigned long)hits[k].freq_hz, hits[k].power_db);
  m_hitCount += nHits;
  m_bufferCount += 1;
  scn_plan_destroy(plan);
}
