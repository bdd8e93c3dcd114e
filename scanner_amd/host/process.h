// process.h -- ProcessSamples: the consumer side.  Same constructor, modes and entry points
// as the reference (process.h:24-93); the per-buffer CPU work of ThreadWorker
// (process.cpp:293-299: memcpy, FFTWindow::apply, FFT::process, process_fft) is replaced by
// one scn_plan per consumer thread: each thread submits up to a batch of queued messages at a time from a
// pinned staging slot and, while that runs on the GPU, the next slots fill.  The queue writes the producer's buffers
// straight into those slots -- a staging ring per consumer thread (SampleQueue::AttachStaging: no copy in the worker).
// The stdout protocol ("Start scan at", "freq %lu power_db %f", thread start/stop lines) and
// the ack / trigger bookkeeping per message follow the reference.
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "messageQueue.h"
#include "scannerCompat.h"

class SampleBuffer;
class SignalSource;
struct scn_hit;

class ProcessSamples {
 public:
  enum Mode { Illegal, TimeDomain, FrequencyDomain };  // process.h:27-31

  ProcessSamples(uint32_t numSamples, uint32_t sampleRate, uint32_t enob, float threshold,
                 gr::fft::window::win_type windowType, Mode mode, uint32_t threadCount = 1,
                 std::string fileNameBase = "", double useBandWidth = 0.75, double dcIgnoreWidth = 0.0,
                 uint32_t preTrigger = 2, uint32_t postTrigger = 4);
  ~ProcessSamples();

  // One-shot on a single int16 buffer (process.cpp:131-144).  Unlike the reference -- where
  // this dereferences a null header -- it reports against the given centre frequency.
  // Returns false when the GPU path failed (the reference's void return is still source compatible at its call sites).
  bool Run(int16_t sample_buffer[][2], uint32_t centerFrequency);
  // Blocks until the queue is done and drained; false if a worker had to give up (GetLastError says why).
  bool StartProcessing(SampleQueue &sampleQueue);
  std::string GetLastError();

  // knobs the reference hard-codes; set before StartProcessing
  void SetMaxBatch(uint32_t maxBatch) { m_maxBatch = maxBatch; }
  void SetDevice(int firstDevice) { m_firstDevice = firstDevice; }
  // Submits a consumer thread keeps in flight (1 .. SCN_NUM_SLOTS, default 4: a hits-only plan's 55 us kernel is shorter than what its record list needs on the D2H stream,
  // so the fourth slot still pays -- 270 -> 326 Gsamples/s with every record read, profiles/r05_experiments.md).  Each slot in use holds its pinned staging
  // (max_batch buffers), two generations of hit regions (8 B x evaluated bins x max_batch each: 2 x 201 MB for a 8192 x
  // 4096-point batch) and its record lists: a host short of GPU memory trades depth for footprint here.
  void SetPipelineDepth(uint32_t depth) { m_pipeDepth = depth; }
  // The sweep's centre frequencies, in table order (what FrequencyTable(sampleRate, start, stop, useBandWidth, dcIgnoreWidth) holds,
  // frequencyTable.cpp:9-37): every consumer thread then uploads them to its plan ONCE (scn_plan_set_table), and a batch whose
  // buffers carry a consecutive, wrapping run of the table -- what a source that retunes through it delivers, frequencyTable.cpp:38-46 --
  // is submitted by naming its first entry (scn_submit_indexed): no centre frequency per buffer crosses the boundary.  Any other
  // batch (HackRF sweep headers, a source with a table of its own) goes out with its centres as before; the records are the same.
  void SetFrequencyTable(const std::vector<double> &centres) { m_tableCentres = centres; }
  uint64_t GetIndexedSubmitCount() const { return m_indexedSubmits; }  // batches that went out as a run of the table
  uint64_t GetHitCount() const { return m_hitCount; }
  uint64_t GetBufferCount() const { return m_bufferCount; }
  // consumer threads that run the zero-copy path (the queue writes into their plan's pinned slots, SampleQueue::AttachStaging);
  // the others copy every message into a slot themselves
  uint32_t GetStagedWorkerCount() const { return m_stagedWorkers; }
  // where the consumer threads' time went, summed over threads (seconds): waiting for the producer, in scn_submit, in
  // scn_collect / scn_hits_view (waiting for the GPU), reporting (printf, acks, recycling)
  struct WorkerTimes {
    double waitProducer, submit, collect, report;
  };
  WorkerTimes GetWorkerTimes() const {
    return WorkerTimes{m_tWait.load() * 1e-9, m_tSubmit.load() * 1e-9, m_tCollect.load() * 1e-9, m_tReport.load() * 1e-9};
  }

  bool m_writeData;

 private:
  void ThreadWorker(uint32_t threadId);
  void TimeToString(time_t time, char *buffer, uint32_t length);
  std::string GenerateFileName(std::string fileNameBase, time_t startTime, double_t centerFrequency);
  void WriteSamplesToFile(uint64_t sequenceId, double centerFrequency);
  void UpdateEndSequenceId(uint64_t newEndSequenceId);
  void ProcessWrite(bool doWrite, double centerFrequency, uint64_t sequenceId);
  bool Ok(int status, const char *what);  // logs and remembers the first failed C-ABI call
  bool Fail(const std::string &text);     // the same for an error found on this side of the C-ABI; returns false

  static const uint32_t MAX_THREADS = 8;  // process.h:49
  uint32_t m_sampleCount, m_sampleRate, m_enob;
  uint32_t m_fileCounter, m_preTrigger, m_postTrigger;
  std::atomic<uint64_t> m_endSequenceId;
  std::atomic<bool> m_writing;
  Mode m_mode;
  std::string m_fileNameBase;
  float m_threshold;
  double m_useBandWidth;
  gr::fft::window::win_type m_windowType;
  SampleQueue *m_sampleQueue;
  uint32_t m_threadCount;
  uint32_t m_maxBatch, m_pipeDepth;
  int m_firstDevice;
  std::atomic<uint64_t> m_hitCount, m_bufferCount;
  std::atomic<uint32_t> m_stagedWorkers;
  std::vector<double> m_tableCentres;
  std::atomic<uint64_t> m_indexedSubmits;
  std::atomic<uint64_t> m_tWait, m_tSubmit, m_tCollect, m_tReport;  // nanoseconds, see GetWorkerTimes
  std::atomic<bool> m_failed;
  std::mutex m_errorMutex;
  std::string m_error;
};
