#include "signalSource.h"

#include <cstdio>
#include <ctime>

SignalSource::SignalSource(uint32_t sampleRate, uint32_t sampleCount, double startFrequency, double stopFrequency,
                           double useBandWidth, double dcIgnoreWidth, bool doTiming)
    : m_sampleRate(sampleRate), m_sampleCount(sampleCount), m_sampleQueue(nullptr), m_finished(false),
      m_frequencyTable(sampleRate, startFrequency, stopFrequency, useBandWidth, dcIgnoreWidth), m_iterationLimit(0),
      m_isDone(false), m_synchronousMode(false), m_thread(nullptr), m_doTiming(doTiming), m_elapsedTime(0),
      m_retuneTimeIndex(0), m_getSamplesTimeIndex(0), m_retuneTime(s_maxIndex), m_getSamplesTime(s_maxIndex) {}

SignalSource::~SignalSource() {
  if (m_thread && m_thread->joinable()) m_thread->join();
}

bool SignalSource::Start() { return true; }
bool SignalSource::Stop() { return true; }

double SignalSource::GetNextFrequency(void **pinfo) { return m_frequencyTable.GetNextFrequency(pinfo); }
double SignalSource::GetCurrentFrequency(void **pinfo) { return m_frequencyTable.GetCurrentFrequency(pinfo); }
double SignalSource::GetStartFrequency() { return m_frequencyTable.GetStartFrequency(); }
double SignalSource::GetStopFrequency() { return m_frequencyTable.GetStopFrequency(); }
uint32_t SignalSource::GetFrequencyCount() { return m_frequencyTable.GetFrequencyCount(); }
bool SignalSource::GetIsScanStart() { return m_frequencyTable.GetIsScanStart(); }
uint32_t SignalSource::GetIterationCount() { return m_frequencyTable.GetIterationCount(); }

bool SignalSource::DoRetune() {  // signalSource.cpp:75-81
  if (m_synchronousMode && m_sampleQueue != nullptr) return m_sampleQueue->ReceivedAck();
  return true;
}

bool SignalSource::StartThread(uint32_t numIterations, SampleQueue &sampleQueue) {  // signalSource.cpp:83-94
  printf("Starting source thread...\n");
  m_iterationLimit = numIterations;
  m_sampleQueue = &sampleQueue;
  m_finished = false;
  m_thread.reset(new std::thread(&SignalSource::ThreadWorkerHelper, this));
  return true;
}

bool SignalSource::StopThread() {
  if (m_thread != nullptr) {
    printf("Stopping source thread...\n");
    m_finished = true;
    if (m_thread->joinable()) m_thread->join();
  }
  return true;
}

bool SignalSource::GetIsDone() { return GetIterationCount() >= m_iterationLimit || m_isDone; }
void SignalSource::SetIsDone() { m_isDone = true; }

void SignalSource::ThreadWorkerHelper() {  // signalSource.cpp:120-125
  ThreadWorker();
  m_sampleQueue->SetIsDone();
  m_sampleQueue = nullptr;
}

void SignalSource::StopStreaming() {
  SetIsDone();
  StopThread();
}

void SignalSource::StartTimer() {
  if (m_doTiming) clock_gettime(CLOCK_REALTIME, &m_start);
}

void SignalSource::StopTimer() {
  if (m_doTiming) {
    clock_gettime(CLOCK_REALTIME, &m_stop);
    m_elapsedTime = (m_stop.tv_sec * 1000.0 + m_stop.tv_nsec / 1e6) - (m_start.tv_sec * 1000.0 + m_start.tv_nsec / 1e6);
  }
}

void SignalSource::AddRetuneTime() {
  if (m_doTiming && m_retuneTimeIndex < s_maxIndex) m_retuneTime[m_retuneTimeIndex++] = m_elapsedTime;
}

void SignalSource::AddGetSamplesTime() {
  if (m_doTiming && m_getSamplesTimeIndex < s_maxIndex) m_getSamplesTime[m_getSamplesTimeIndex++] = m_elapsedTime;
}

void SignalSource::WriteTimingData() {  // signalSource.cpp:164-176
  if (m_doTiming && m_retuneTimeIndex >= s_maxIndex) {
    if (FILE *f = fopen("timings.txt", "w")) {
      for (uint32_t i = 0; i < m_retuneTimeIndex; i++) fprintf(f, "%f, %f\n", m_retuneTime[i], m_getSamplesTime[i]);
      fclose(f);
    }
    m_doTiming = false;
  }
}
