// fileSource.h -- a SignalSource that replays raw IQ from a file: buffers of
// sampleCount samples in the wire format of the chosen SampleKind, back to back, one per tune,
// e.g. a SyntheticSource dump or a recording made with a vendor tool.  Sweeps wrap around the
// frequency table exactly like the hardware sources; the stream ends with the file or the
// iteration limit, whichever comes first.
#pragma once
#include <cstdio>
#include <string>
#include <vector>

#include "signalSource.h"

class FileSource : public SignalSource {
 public:
  FileSource(const std::string &path, uint32_t sampleRate, uint32_t sampleCount, double startFrequency,
             double stopFrequency, SampleQueue::SampleKind kind, double useBandWidth = 0.75, double dcIgnoreWidth = 0.0);
  ~FileSource() override;
  bool Start() override;  // false when the file could not be opened
  bool GetNextSamples(SampleQueue *sampleQueue, double_t &centerFrequency) override;
  bool StartStreaming(uint32_t numIterations, SampleQueue &sampleQueue) override;
  void ThreadWorker() override;
  double Retune(double frequency) override { return frequency; }
  uint64_t GetBuffersRead() const { return m_read; }

 private:
  bool ReadOne(std::vector<unsigned char> &raw);
  void Push(SampleQueue *q, void *raw, double fc, time_t t);
  SampleQueue::SampleKind m_kind;
  size_t m_bufferBytes;
  FILE *m_file;
  uint64_t m_read;
};
