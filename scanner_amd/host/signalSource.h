// signalSource.h -- the producer side of a scan: an abstract receiver front-end.
//
// Drop-in for the reference's SignalSource (signalSource.h:11-67) at BOTH of its boundaries:
//   * the wiring: scan.cpp:234-238 does
//       source->Start(); source->StartStreaming(n, queue); process.StartProcessing(queue); source->StopStreaming();
//   * the device subclasses, which reach into the base's protected state: bladerfSource.cpp:91-99 walks
//     `this->m_frequencyTable`, every front-end reads `this->m_sampleCount` / `this->m_sampleRate` / `this->m_sampleQueue`
//     and loops on GetIsDone().  Those names and their visibility are therefore part of the interface and are kept
//     (tests/cpp/test_subclass_compat.cpp compiles a subclass written the way the reference's are).
// What is NOT part of the interface is how the base does its own bookkeeping: stop flags are atomics (the
// reference's plain bools are read by one thread and written by another, signalSource.cpp:85-86), and the optional
// retune / receive timing (only the B210 front-end ever calls it, b210Source.cpp:85-92) is a small CallTimer
// object on a monotonic clock instead of loose timespec members.
#pragma once
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <memory>
#include <thread>
#include <vector>

#include "frequencyTable.h"
#include "messageQueue.h"

// Milliseconds spent in the two calls a front-end may time (Retune, the receive call), one pair per tune.
class CallTimer {
 public:
  explicit CallTimer(bool enabled, size_t capacity);
  bool Enabled() const { return m_enabled; }
  void Begin();                 // start of a timed call
  void End();                   // end of it: the elapsed time becomes the "last" value
  void KeepAsRetune();          // file the last value in the retune column ...
  void KeepAsReceive();         // ... or in the receive column
  bool Full() const { return m_retuneMs.size() >= m_capacity; }
  // one "retune, receive" line per tune (the layout signalSource.cpp:164-176 writes to timings.txt); disables the timer
  bool Dump(const char *path);

 private:
  bool m_enabled;
  size_t m_capacity;
  std::chrono::steady_clock::time_point m_begin;
  double m_lastMs;
  std::vector<double> m_retuneMs, m_receiveMs;
};

class SignalSource {
 public:
  SignalSource(uint32_t sampleRate, uint32_t sampleCount, double startFrequency, double stopFrequency,
               double useBandWidth = 0.75, double dcIgnoreWidth = 0.0, bool doTiming = false);
  virtual ~SignalSource();

  // ---- what a front-end implements (signalSource.h:52-57) ----------------------------------------
  virtual bool Start();                                                                  // device bring-up; default: nothing
  virtual bool GetNextSamples(SampleQueue *sampleQueue, double_t &centerFrequency) = 0;  // one synchronous buffer
  virtual bool StartStreaming(uint32_t numIterations, SampleQueue &sampleQueue) = 0;     // spawn the producer
  virtual void ThreadWorker() = 0;                                                       // the producer's body
  virtual bool Stop();
  virtual double Retune(double frequency) = 0;                                           // returns the frequency actually set

  // ---- what the wiring and the workers call (signalSource.h:58-66) --------------------------------
  bool DoRetune();               // synchronous mode: only after the consumer acknowledged the last buffer
  uint32_t GetFrequencyCount();  // table size
  bool GetIsScanStart();         // the current table entry opens a sweep
  void StopStreaming();          // mark done, join the producer
  void StartTimer() { m_timer.Begin(); }
  void StopTimer() { m_timer.End(); }
  void AddRetuneTime() { m_timer.KeepAsRetune(); }
  void AddGetSamplesTime() { m_timer.KeepAsReceive(); }
  void WriteTimingData();        // once the timer is full: timings.txt, then timing is switched off

 protected:
  // ---- state the reference's subclasses use by name (signalSource.h:11-43): keep protected --------
  uint32_t m_sampleRate;
  uint32_t m_sampleCount;
  double m_startFrequency;
  double m_stopFrequency;
  FrequencyTable m_frequencyTable;  // bladerfSource.cpp:91-99 attaches per-frequency quick-tune data to it
  uint32_t m_iterationLimit;
  SampleQueue *m_sampleQueue;       // the running stream's queue; null when no stream is running
  std::atomic<bool> m_isDone;       // set to end the stream
  std::atomic<bool> m_finished;     // set by StopThread: the worker should leave its loop
  bool m_synchronousMode;           // retune only after the consumer's ack (never enabled by the reference either)
  std::unique_ptr<std::thread> m_thread;

  void SetIsDone();
  bool StopThread();
  bool StartThread(uint32_t numIterations, SampleQueue &sampleQueue);
  void ThreadWorkerHelper();  // ThreadWorker, then tells the queue that no more samples will come
  uint32_t GetIterationCount();
  double GetCurrentFrequency(void **pinfo = nullptr);
  double GetNextFrequency(void **pinfo = nullptr);  // advances the table
  double GetStartFrequency();
  double GetStopFrequency();
  bool GetIsDone();  // iteration limit reached or SetIsDone called

 private:
  CallTimer m_timer;
};
