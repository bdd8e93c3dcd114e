// sampleBuffer.h -- SampleBuffer: the reference's alternative staging class
// (sampleBuffer.h:8-46; compiled there, never wired into scan.cpp).  Same public surface;
// internally a bounded FIFO of raw buffers.  As with SampleQueue the integer formats are kept
// raw (the convert runs on the GPU), so the float GetNextSamples() serves FloatComplex
// buffers and the raw/visitor forms serve every kind.
#pragma once
#include <condition_variable>
#include <deque>
#include <list>
#include <mutex>
#include <string>
#include <vector>

#include "buffer.h"

class SampleBuffer {
 public:
  enum SampleKind { Illegal = 0, Short, ShortComplex, FloatComplex } m_kind;  // sampleBuffer.h:28-33

  SampleBuffer(SampleKind kind, uint32_t enob, uint32_t count);
  void AppendSamples(int16_t *realSamples, int16_t *imagSamples, double centerFrequency);
  void AppendSamples(int16_t shortComplexSamples[][2], double centerFrequency);
  void AppendSamples(fftwf_complex *floatComplexSamples, double centerFrequency);
  // FloatComplex only: copy the next buffer out.  false once done and empty.
  bool GetNextSamples(fftwf_complex *outputBuffer, double &centerFrequency);
  // Any kind: hand the next raw buffer out (GetBufferBytes() bytes).
  bool GetNextRaw(void *outputBuffer, double &centerFrequency);
  // FloatComplex only: visit the next buffer (Begin / Process / End), e.g. a HipStagingProcessInterface.
  bool ProcessNext(ProcessInterface<fftwf_complex> *process, double &centerFrequency);
  void WriteSamplesToFile(std::string fileName, uint32_t count);
  void SetIsDone();
  bool GetIsDone();
  size_t GetBufferBytes() const { return m_bufferBytes; }
  uint32_t GetEnob() const { return m_enob; }

 private:
  void Append(const void *a, size_t aBytes, const void *b, size_t bBytes, double fc);
  bool Pop(std::vector<unsigned char> &raw, double &fc);
  uint32_t m_sampleCount, m_enob;
  size_t m_bufferBytes, m_capacity;
  std::deque<std::pair<double, std::vector<unsigned char>>> m_queue;
  uint64_t m_nextOutSequenceId;
  std::mutex m_mutex;
  std::condition_variable m_conditionEmpty, m_conditionFull;
  bool m_done;
};
