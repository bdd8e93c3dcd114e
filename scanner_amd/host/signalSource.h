// signalSource.h -- abstract device front-end; same surface as the reference's SignalSource
// (signalSource.h:11-67): owns the frequency table and the producer thread, pushes one
// fixed-size raw IQ buffer per tune into a SampleQueue.  Device subclasses (HackRF, bladeRF,
// ...) are out of scope; SyntheticSource (syntheticSource.h) is the in-tree implementation.
#pragma once
#include <cmath>
#include <cstdint>
#include <memory>
#include <thread>
#include <vector>

#include "frequencyTable.h"
#include "messageQueue.h"

class SignalSource {
 protected:
  bool m_doTiming;
  struct timespec m_start, m_stop;
  double m_elapsedTime;
  uint32_t m_retuneTimeIndex;
  uint32_t m_getSamplesTimeIndex;
  bool m_isDone;
  bool m_finished;
  bool m_synchronousMode;
  std::unique_ptr<std::thread> m_thread;
  std::vector<double> m_retuneTime;
  std::vector<double> m_getSamplesTime;
  static const uint32_t s_maxIndex = 10000;

  uint32_t m_sampleRate;
  uint32_t m_sampleCount;
  double m_startFrequency;
  double m_stopFrequency;
  FrequencyTable m_frequencyTable;
  uint32_t m_iterationLimit;
  SampleQueue *m_sampleQueue;
  void SetIsDone();
  bool StopThread();
  bool StartThread(uint32_t numIterations, SampleQueue &sampleQueue);
  void ThreadWorkerHelper();
  uint32_t GetIterationCount();
  double GetCurrentFrequency(void **pinfo = nullptr);
  double GetNextFrequency(void **pinfo = nullptr);
  double GetStartFrequency();
  double GetStopFrequency();
  bool GetIsDone();

 public:
  SignalSource(uint32_t sampleRate, uint32_t sampleCount, double startFrequency, double stopFrequency,
               double useBandWidth = 0.75, double dcIgnoreWidth = 0.0, bool doTiming = false);
  virtual ~SignalSource();
  virtual bool Start();
  virtual bool GetNextSamples(SampleQueue *sampleQueue, double_t &centerFrequency) = 0;
  virtual bool StartStreaming(uint32_t numIterations, SampleQueue &sampleQueue) = 0;
  virtual void ThreadWorker() = 0;
  virtual bool Stop();
  virtual double Retune(double frequency) = 0;
  bool DoRetune();
  uint32_t GetFrequencyCount();
  bool GetIsScanStart();
  void StopStreaming();
  void StartTimer();
  void StopTimer();
  void AddRetuneTime();
  void AddGetSamplesTime();
  void WriteTimingData();
};
