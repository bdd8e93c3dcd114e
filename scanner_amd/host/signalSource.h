// signalSource.h -- the producer side of a scan: an abstract receiver front-end.
//
// Call-site compatible with the reference's SignalSource (signalSource.h:11-67): scan.cpp:234-238 does
//     source->Start(); source->StartStreaming(n, queue); process.StartProcessing(queue); source->StopStreaming();
// and the device classes override Retune / GetNextSamples / StartStreaming / ThreadWorker.  The device
// subclasses (HackRF, bladeRF, Airspy, SDRplay, B210) are out of scope here; SyntheticSource and FileSource are
// the in-tree implementations.  A source owns the sweep's FrequencyTable and one producer thread that pushes a
// fixed-size raw IQ buffer per tune into the SampleQueue it was started with.
#pragma once
#include <cmath>
#include <cstdint>
#include <ctime>
#include <memory>
#include <thread>
#include <vector>

#include "frequencyTable.h"
#include "messageQueue.h"

class SignalSource {
 public:
  SignalSource(uint32_t sampleRate, uint32_t sampleCount, double startFrequency, double stopFrequency,
               double useBandWidth = 0.75, double dcIgnoreWidth = 0.0, bool doTiming = false);
  virtual ~SignalSource();

  // ---- what a front-end implements --------------------------------------------------------------
  virtual double Retune(double frequency) = 0;                                                // returns the frequency actually set
  virtual bool GetNextSamples(SampleQueue *sampleQueue, double_t &centerFrequency) = 0;       // one synchronous buffer
  virtual bool StartStreaming(uint32_t numIterations, SampleQueue &sampleQueue) = 0;          // spawn the producer
  virtual void ThreadWorker() = 0;                                                            // the producer's body
  virtual bool Start();                                                                       // device bring-up; default: nothing
  virtual bool Stop();

  // ---- what the wiring and the workers call -------------------------------------------------------
  void StopStreaming();           // mark done, join the producer
  uint32_t GetFrequencyCount();   // table size
  bool GetIsScanStart();          // the current table entry opens a sweep
  bool DoRetune();                // synchronous mode: only after the consumer acknowledged the last buffer

  // optional retune / receive timing (doTiming): elapsed milliseconds kept per call, dumped once full
  void StartTimer();
  void StopTimer();
  void AddRetuneTime();
  void AddGetSamplesTime();
  void WriteTimingData();

 protected:
  // sweep parameters and the queue of the running stream, for the subclasses
  uint32_t m_sampleRate;
  uint32_t m_sampleCount;
  SampleQueue *m_sampleQueue;
  bool m_finished;  // set by StopThread: the worker should leave its loop

  bool StartThread(uint32_t numIterations, SampleQueue &sampleQueue);
  bool StopThread();
  void SetIsDone();
  bool GetIsDone();  // iteration limit reached or SetIsDone called
  uint32_t GetIterationCount();
  double GetCurrentFrequency(void **pinfo = nullptr);
  double GetNextFrequency(void **pinfo = nullptr);  // advances the table
  double GetStartFrequency();
  double GetStopFrequency();

 private:
  void ThreadWorkerHelper();  // ThreadWorker, then tells the queue that no more samples will come

  FrequencyTable m_frequencyTable;
  uint32_t m_iterationLimit;
  bool m_isDone;
  bool m_synchronousMode;
  std::unique_ptr<std::thread> m_thread;

  static const uint32_t s_maxIndex = 10000;  // timing samples kept
  bool m_doTiming;
  struct timespec m_start, m_stop;
  double m_elapsedTime;
  uint32_t m_retuneTimeIndex, m_getSamplesTimeIndex;
  std::vector<double> m_retuneTime, m_getSamplesTime;
};
