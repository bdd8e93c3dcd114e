// A front-end written the way the reference's device classes are written, against the kept SignalSource
// base: it reaches into the base's PROTECTED state by name (bladerfSource.cpp:91-99 walks
// this->m_frequencyTable and hangs a per-frequency record on every entry; every front-end reads
// this->m_sampleCount / m_sampleRate / m_sampleQueue; b210Source.cpp:85-92,121-136 times its calls).  If the base
// hides any of that, this file does not compile -- that is the test.  It also runs (CPU only, no HIP call):
// the stream is a counting pattern whose frequency and per-frequency record can be checked at the consumer.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "signalSource.h"

struct QuickTune {  // what a device library would hand back per frequency
  uint32_t index;
  double frequency;
};

class QuickTuneSource : public SignalSource {
 public:
  QuickTuneSource(uint32_t sampleRate, uint32_t sampleCount, double start, double stop)
      : SignalSource(sampleRate, sampleCount, start, stop, 0.75, 0.0, /*doTiming=*/true), m_quickTunes(nullptr), m_retunes(0) {}
  ~QuickTuneSource() override { delete[] m_quickTunes; }

  bool Start() override {
    // the bladeRF pattern: one record per table entry, attached to the table
    uint32_t count = this->m_frequencyTable.GetFrequencyCount();
    this->m_quickTunes = new QuickTune[count];
    for (uint32_t i = 0; i < count; i++) {
      this->m_quickTunes[i].index = i;
      this->m_quickTunes[i].frequency = this->m_frequencyTable.GetFrequencyFromIndex(i);
      this->m_frequencyTable.SetFrequencyInfoForIndex(i, &this->m_quickTunes[i]);
    }
    return count == this->GetFrequencyCount() && this->m_startFrequency < this->m_stopFrequency;
  }
  double Retune(double frequency) override {
    this->StartTimer();
    m_retunes++;
    this->StopTimer();
    this->AddRetuneTime();
    return frequency;
  }
  bool GetNextSamples(SampleQueue *queue, double_t &centerFrequency) override {
    QuickTune *qt = nullptr;
    centerFrequency = this->GetCurrentFrequency(reinterpret_cast<void **>(&qt));
    if (!qt || qt->frequency != centerFrequency) return false;
    std::vector<int16_t> iq(2 * (size_t)this->m_sampleCount);
    for (uint32_t i = 0; i < this->m_sampleCount; i++) {
      iq[2 * i] = (int16_t)qt->index;
      iq[2 * i + 1] = (int16_t)(this->m_sampleRate / 1000000u);
    }
    this->StartTimer();
    queue->AppendSamples(reinterpret_cast<int16_t(*)[2]>(iq.data()), centerFrequency, this->GetIsScanStart() ? 1 : 0);
    this->StopTimer();
    this->AddGetSamplesTime();
    this->Retune(this->GetNextFrequency());
    return true;
  }
  bool StartStreaming(uint32_t numIterations, SampleQueue &queue) override { return this->StartThread(numIterations, queue); }
  void ThreadWorker() override {
    double_t fc;
    while (!this->GetIsDone() && !this->m_finished) {
      if (!this->DoRetune() || !this->GetNextSamples(this->m_sampleQueue, fc)) break;
    }
    this->WriteTimingData();  // not full after a few tunes: must not write
  }
  uint32_t Retunes() const { return m_retunes; }
  uint32_t IterationLimit() const { return this->m_iterationLimit; }
  bool SynchronousMode() const { return this->m_synchronousMode; }
  bool HasThread() const { return bool(this->m_thread); }

 private:
  QuickTune *m_quickTunes;
  uint32_t m_retunes;
};

int main() {
  const uint32_t fs = 8000000, n = 64, sweeps = 3;
  QuickTuneSource src(fs, n, 100e6, 130e6);  // 5 centres, 6 MHz apart
  const uint32_t count = src.GetFrequencyCount();
  SampleQueue queue(SampleQueue::ShortComplex, 12, n, 8, false, false);
  if (!src.Start() || !src.StartStreaming(sweeps, queue)) return 1;
  uint32_t got = 0, bad = 0;
  while (SampleQueue::MessageType *m = queue.GetNextSamples()) {
    const int16_t *raw = static_cast<const int16_t *>(m->GetRawData());
    const uint32_t idx = got % count;  // the warm-up sweep was discarded whole, so messages start at index 0
    const double want = 100e6 + 0.75 / 2 * fs + idx * 0.75 * fs;  // frequencyTable.cpp:17-33
    if (m->GetHeader().m_frequency != want || raw[0] != (int16_t)idx || raw[1] != 8 || m->GetHeader().m_sequenceId != got) bad++;
    got++;
    queue.MessageProcessed(m);
  }
  src.StopStreaming();
  src.Stop();
  // `sweeps` iterations were produced, the first of them is the queue's warm-up discard (messageQueue.h:67-72)
  const bool ok = bad == 0 && got == (sweeps - 1) * count && src.Retunes() == sweeps * count && src.IterationLimit() == sweeps &&
                  !src.SynchronousMode() && !src.HasThread() && fopen("timings.txt", "r") == nullptr;
  printf("subclass compat: %u messages, %u bad, %u retunes -> %s\n", got, bad, src.Retunes(), ok ? "ok" : "FAILED");
  return ok ? 0 : 1;
}
