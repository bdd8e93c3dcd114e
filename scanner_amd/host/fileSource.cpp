#include "fileSource.h"

#include <cassert>
#include <cstdlib>

FileSource::FileSource(const std::string &path, uint32_t sampleRate, uint32_t sampleCount, double startFrequency,
                       double stopFrequency, SampleQueue::SampleKind kind, double useBandWidth, double dcIgnoreWidth)
    : SignalSource(sampleRate, sampleCount, startFrequency, stopFrequency, useBandWidth, dcIgnoreWidth), m_kind(kind),
      m_file(nullptr), m_read(0) {
  size_t per = kind == SampleQueue::FloatComplex ? 8 : kind == SampleQueue::ByteComplex ? 2 : 4;
  m_bufferBytes = per * sampleCount;
  m_file = fopen(path.c_str(), "rb");
  // (the device sources exit(1) on open failure, e.g. hackRFSource.cpp:19-30; a library reports it: Start() and
  // StartStreaming() return false)
  if (!m_file) fprintf(stderr, "FileSource: cannot open '%s'\n", path.c_str());
}

bool FileSource::Start() { return m_file != nullptr; }

FileSource::~FileSource() {
  StopThread();  // the producer may be inside ReadOne: join it BEFORE the file is closed
  if (m_file) fclose(m_file);
}

bool FileSource::ReadOne(std::vector<unsigned char> &raw) {
  raw.resize(m_bufferBytes);
  if (!m_file || fread(raw.data(), 1, m_bufferBytes, m_file) != m_bufferBytes) return false;
  m_read++;
  return true;
}

void FileSource::Push(SampleQueue *q, void *raw, double fc, time_t t) {
  switch (m_kind) {
    case SampleQueue::FloatComplex: q->AppendSamples(static_cast<fftwf_complex *>(raw), fc, t); break;
    case SampleQueue::ShortComplex: q->AppendSamples(static_cast<int16_t(*)[2]>(raw), fc, t); break;
    case SampleQueue::Short:
      q->AppendSamples(static_cast<int16_t *>(raw), static_cast<int16_t *>(raw) + m_sampleCount, fc, t);
      break;
    case SampleQueue::ByteComplex: q->AppendSamples(static_cast<int8_t(*)[2]>(raw), fc, t); break;
    default: assert(false);
  }
}

bool FileSource::GetNextSamples(SampleQueue *q, double_t &centerFrequency) {
  std::vector<unsigned char> raw;
  if (!ReadOne(raw)) return false;
  centerFrequency = GetCurrentFrequency();
  Push(q, raw.data(), centerFrequency, 0);
  return true;
}

bool FileSource::StartStreaming(uint32_t numIterations, SampleQueue &sampleQueue) {
  return m_file != nullptr && StartThread(numIterations, sampleQueue);
}

void FileSource::ThreadWorker() {
  std::vector<unsigned char> raw;
  while (!GetIsDone() && !m_finished) {
    double centerFrequency = GetCurrentFrequency();
    bool isScanStart = GetIsScanStart();
    if (!ReadOne(raw)) break;  // end of file ends the stream
    time_t startTime = (time_t)(86400 + GetIterationCount());
    GetNextFrequency();
    Push(m_sampleQueue, raw.data(), centerFrequency, isScanStart ? startTime : 0);
  }
}
