#include "signalSource.h"

#include <cstdio>
#include <fstream>
#include <iomanip>

// ---- CallTimer ---------------------------------------------------------------------------------------------
CallTimer::CallTimer(bool enabled, size_t capacity) : m_enabled(enabled), m_capacity(capacity), m_lastMs(0.0) {
  if (enabled) {
    m_retuneMs.reserve(capacity);
    m_receiveMs.reserve(capacity);
  }
}

void CallTimer::Begin() {
  if (m_enabled) m_begin = std::chrono::steady_clock::now();
}

void CallTimer::End() {
  if (m_enabled) m_lastMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - m_begin).count();
}

void CallTimer::KeepAsRetune() {
  if (m_enabled && m_retuneMs.size() < m_capacity) m_retuneMs.push_back(m_lastMs);
}

void CallTimer::KeepAsReceive() {
  if (m_enabled && m_receiveMs.size() < m_capacity) m_receiveMs.push_back(m_lastMs);
}

bool CallTimer::Dump(const char *path) {
  std::ofstream out(path);
  if (!out) return false;
  out << std::fixed << std::setprecision(6);  // what "%f, %f\n" prints
  for (size_t k = 0; k < m_retuneMs.size(); k++) out << m_retuneMs[k] << ", " << (k < m_receiveMs.size() ? m_receiveMs[k] : 0.0) << "\n";
  m_enabled = false;
  return bool(out);
}

// ---- SignalSource ------------------------------------------------------------------------------------------
SignalSource::SignalSource(uint32_t sampleRate, uint32_t sampleCount, double startFrequency, double stopFrequency,
                           double useBandWidth, double dcIgnoreWidth, bool doTiming)
    : m_sampleRate(sampleRate),
      m_sampleCount(sampleCount),
      m_startFrequency(startFrequency),
      m_stopFrequency(stopFrequency),
      m_frequencyTable(sampleRate, startFrequency, stopFrequency, useBandWidth, dcIgnoreWidth),
      m_iterationLimit(0),
      m_sampleQueue(nullptr),
      m_isDone(false),
      m_finished(false),
      m_synchronousMode(false),
      m_timer(doTiming, 10000) {}  // 10 000 tunes are kept (s_maxIndex, signalSource.h:23)

SignalSource::~SignalSource() {
  // Last resort only: by now the derived part of the object is gone, so a producer still inside the derived
  // ThreadWorker is already running on dead members.  Every front-end calls StopThread() first thing in its own
  // destructor (SyntheticSource, FileSource do); this join merely keeps the std::thread from terminating the process.
  m_finished = true;
  if (m_thread && m_thread->joinable()) m_thread->join();
}

bool SignalSource::Start() { return true; }
bool SignalSource::Stop() { return true; }

// the sweep position lives in the table (frequencyTable.cpp:39-70)
uint32_t SignalSource::GetFrequencyCount() { return m_frequencyTable.GetFrequencyCount(); }
bool SignalSource::GetIsScanStart() { return m_frequencyTable.GetIsScanStart(); }
uint32_t SignalSource::GetIterationCount() { return m_frequencyTable.GetIterationCount(); }
double SignalSource::GetCurrentFrequency(void **pinfo) { return m_frequencyTable.GetCurrentFrequency(pinfo); }
double SignalSource::GetNextFrequency(void **pinfo) { return m_frequencyTable.GetNextFrequency(pinfo); }
double SignalSource::GetStartFrequency() { return m_frequencyTable.GetStartFrequency(); }
double SignalSource::GetStopFrequency() { return m_frequencyTable.GetStopFrequency(); }

bool SignalSource::DoRetune() {
  // signalSource.cpp:75-81: in synchronous mode the next tune waits for the consumer's acknowledgement
  SampleQueue *queue = m_sampleQueue;
  return !(m_synchronousMode && queue) || queue->ReceivedAck();
}

void SignalSource::SetIsDone() { m_isDone = true; }

bool SignalSource::GetIsDone() { return m_isDone || GetIterationCount() >= m_iterationLimit; }

bool SignalSource::StartThread(uint32_t numIterations, SampleQueue &sampleQueue) {
  if (m_thread && m_thread->joinable()) return false;  // one stream at a time (the reference would leak the thread)
  m_iterationLimit = numIterations;
  m_sampleQueue = &sampleQueue;
  m_finished = false;
  printf("Starting source thread...\n");  // stdout protocol, signalSource.cpp:85
  m_thread.reset(new std::thread([this] { ThreadWorkerHelper(); }));
  return true;
}

void SignalSource::ThreadWorkerHelper() {
  ThreadWorker();
  // the consumers drain what is queued and then see "done" (messageQueue.h:290-310)
  SampleQueue *queue = m_sampleQueue;
  m_sampleQueue = nullptr;
  if (queue) queue->SetIsDone();
}

bool SignalSource::StopThread() {
  if (!m_thread) return true;
  printf("Stopping source thread...\n");  // signalSource.cpp:100
  m_finished = true;
  if (m_thread->joinable()) m_thread->join();
  m_thread.reset();
  return true;
}

void SignalSource::StopStreaming() {
  SetIsDone();
  StopThread();
}

void SignalSource::WriteTimingData() {
  if (m_timer.Enabled() && m_timer.Full()) m_timer.Dump("timings.txt");
}
