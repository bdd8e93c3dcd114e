#include "sampleBuffer.h"

#include <cassert>
#include <cstring>

SampleBuffer::SampleBuffer(SampleKind kind, uint32_t enob, uint32_t sampleCount)
    : m_kind(kind), m_sampleCount(sampleCount), m_enob(enob), m_nextOutSequenceId(0), m_done(false) {
  assert(kind > Illegal && kind <= FloatComplex);  // sampleBuffer.cpp:16
  m_bufferBytes = (kind == FloatComplex ? 8 : 4) * (size_t)sampleCount;
  // the reference ring holds 16 blocks of 8192*16 complex samples (sampleBuffer.h:11, sampleBuffer.cpp:13)
  m_capacity = (size_t)16 * 8192 * 16 / sampleCount;
  if (m_capacity < 2) m_capacity = 2;
}

void SampleBuffer::Append(const void *a, size_t aBytes, const void *b, size_t bBytes, double fc) {
  std::vector<unsigned char> raw(m_bufferBytes);
  assert(aBytes + bBytes == m_bufferBytes);
  memcpy(raw.data(), a, aBytes);
  if (bBytes) memcpy(raw.data() + aBytes, b, bBytes);
  std::unique_lock<std::mutex> lock(m_mutex);  // sampleBuffer.cpp:76-88
  while (m_queue.size() >= m_capacity) m_conditionFull.wait(lock);
  bool wake = m_queue.empty();
  m_queue.emplace_back(fc, std::move(raw));
  if (wake) m_conditionEmpty.notify_one();
}

void SampleBuffer::AppendSamples(int16_t *re, int16_t *im, double fc) {
  assert(m_kind == Short);
  Append(re, 2 * (size_t)m_sampleCount, im, 2 * (size_t)m_sampleCount, fc);
}
void SampleBuffer::AppendSamples(int16_t s[][2], double fc) {
  assert(m_kind == ShortComplex);
  Append(s, m_bufferBytes, nullptr, 0, fc);
}
void SampleBuffer::AppendSamples(fftwf_complex *s, double fc) {
  assert(m_kind == FloatComplex);
  Append(s, m_bufferBytes, nullptr, 0, fc);
}

bool SampleBuffer::Pop(std::vector<unsigned char> &raw, double &fc) {
  std::unique_lock<std::mutex> lock(m_mutex);  // sampleBuffer.cpp:127-152
  while (!m_done && m_queue.empty()) m_conditionEmpty.wait(lock);
  if (m_queue.empty()) return false;
  bool wake = m_queue.size() >= m_capacity;
  fc = m_queue.front().first;
  raw.swap(m_queue.front().second);
  m_queue.pop_front();
  m_nextOutSequenceId += m_sampleCount;
  if (wake) m_conditionFull.notify_one();
  return true;
}

bool SampleBuffer::GetNextRaw(void *out, double &fc) {
  std::vector<unsigned char> raw;
  if (!Pop(raw, fc)) return false;
  memcpy(out, raw.data(), m_bufferBytes);
  return true;
}

bool SampleBuffer::GetNextSamples(fftwf_complex *out, double &fc) {
  assert(m_kind == FloatComplex && "integer formats are converted on the GPU: use GetNextRaw");
  CopyBufferProcessInterface copy(out);
  return ProcessNext(&copy, fc);
}

bool SampleBuffer::ProcessNext(ProcessInterface<fftwf_complex> *process, double &fc) {
  assert(m_kind == FloatComplex);
  std::vector<unsigned char> raw;
  uint64_t seq = m_nextOutSequenceId;
  if (!Pop(raw, fc)) return false;
  process->Begin(seq, m_sampleCount);
  process->Process(reinterpret_cast<const fftwf_complex *>(raw.data()), m_sampleCount);
  process->End();
  return true;
}

void SampleBuffer::WriteSamplesToFile(std::string fileName, uint32_t count) {
  assert(m_kind == FloatComplex);
  fprintf(stderr, "Writing to file %s\n", fileName.c_str());  // sampleBuffer.cpp:30
  FileWriteProcessInterface writer(fileName.c_str());
  std::unique_lock<std::mutex> lock(m_mutex);
  uint32_t left = count;
  for (auto &e : m_queue) {
    if (!left) break;
    uint32_t c = left < m_sampleCount ? left : m_sampleCount;
    writer.Begin(0, c);
    writer.Process(reinterpret_cast<const fftwf_complex *>(e.second.data()), c);
    writer.End();
    left -= c;
  }
}

void SampleBuffer::SetIsDone() {
  std::unique_lock<std::mutex> lock(m_mutex);
  assert(!m_done);
  m_done = true;
  m_conditionEmpty.notify_all();
}

bool SampleBuffer::GetIsDone() {
  std::unique_lock<std::mutex> lock(m_mutex);
  return m_done;
}
