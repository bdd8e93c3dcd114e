#include "messageQueue.h"

#include <cassert>
#include <cstdio>
#include <limits>

static size_t bytesPerSample(SampleQueue::SampleKind k) {
  switch (k) {
    case SampleQueue::ByteComplex: return 2;
    case SampleQueue::Short:
    case SampleQueue::ShortComplex: return 4;
    case SampleQueue::FloatComplex: return 8;
    default: return 0;
  }
}

SampleQueue::SampleQueue(SampleKind kind, uint32_t enob, uint32_t sampleCount, uint32_t bufferCount,
                         bool correctDCOffset, bool doWrite)
    : m_kind(kind), m_enob(enob), m_sampleCount(sampleCount), m_bufferCount(bufferCount),
      m_correctDCOffset(correctDCOffset), m_doWrite(doWrite), m_bufferBytes(bytesPerSample(kind) * sampleCount),
      m_historyCapacity(bufferCount / 10), m_poolSize(uint32_t(bufferCount * 1.1)), m_nextSequenceId(0),
      m_iterationCount(0), m_done(false), m_acknowledged(true), m_writeStart(0), m_writeEnd(0) {
  assert(kind > Illegal && kind <= FloatComplex);  // messageQueue.h:163
  assert(bufferCount > 0);
  if (m_poolSize <= bufferCount) m_poolSize = bufferCount + 1;
  for (uint32_t i = 0; i < m_poolSize; i++) m_free.push_back(new MessageType(m_bufferBytes));
}

SampleQueue::~SampleQueue() {
  assert(m_buffer.empty());  // messageQueue.h:173
  for (MessageType *m : m_history) delete m;
  for (MessageType *m : m_buffer) delete m;
  for (MessageType *m : m_free) delete m;
}

SampleQueue::MessageType *SampleQueue::Allocate() {  // memoryPool.h:60-68
  std::unique_lock<std::mutex> lock(m_poolMutex);
  while (m_free.empty()) m_poolNotEmpty.wait(lock);
  MessageType *m = m_free.front();
  m_free.pop_front();
  return m;
}

void SampleQueue::Free(MessageType *m) {  // memoryPool.h:69-76
  std::unique_lock<std::mutex> lock(m_poolMutex);
  bool wasEmpty = m_free.empty();
  m_free.push_back(m);
  if (wasEmpty) m_poolNotEmpty.notify_one();
}

void SampleQueue::SynchronizedAppend(const void *a, size_t aBytes, const void *b, size_t bBytes,
                                     double centerFrequency, time_t time) {
  // messageQueue.h:65-91
  if (time) m_iterationCount++;
  if (m_iterationCount < 2) return;  // the first (warm-up) sweep is discarded
  MessageType *message = Allocate();
  assert(aBytes + bBytes == m_bufferBytes);
  memcpy(message->GetRawData(), a, aBytes);
  if (bBytes) memcpy(static_cast<unsigned char *>(message->GetRawData()) + aBytes, b, bBytes);
  MessageHeader &h = message->GetHeader();
  h.m_time = time;
  h.m_frequency = centerFrequency;
  h.m_kind = MessageHeader::ProcessData;
  h.m_referenceCount = 0;
  std::unique_lock<std::mutex> lock(m_mutex);
  h.m_sequenceId = m_nextSequenceId++;
  while (m_buffer.size() >= m_bufferCount) m_notFull.wait(lock);
  bool wake = m_buffer.empty();
  m_buffer.push_front(message);
  if (wake) m_notEmpty.notify_one();
  ClearAck();
}

void SampleQueue::AppendSamples(int16_t *re, int16_t *im, double fc, time_t time) {
  assert(m_kind == Short);  // planar: I block then Q block (SCN_KIND_SHORT layout)
  SynchronizedAppend(re, 2 * (size_t)m_sampleCount, im, 2 * (size_t)m_sampleCount, fc, time);
}
void SampleQueue::AppendSamples(int16_t s[][2], double fc, time_t time) {
  assert(m_kind == ShortComplex);
  SynchronizedAppend(s, m_bufferBytes, nullptr, 0, fc, time);
}
void SampleQueue::AppendSamples(int8_t (*s)[2], double fc, time_t time) {
  assert(m_kind == ByteComplex);
  SynchronizedAppend(s, m_bufferBytes, nullptr, 0, fc, time);
}
void SampleQueue::AppendSamples(fftwf_complex *s, double fc, time_t time) {
  assert(m_kind == FloatComplex);
  SynchronizedAppend(s, m_bufferBytes, nullptr, 0, fc, time);
}

SampleQueue::MessageType *SampleQueue::GetNextSamples() {
  std::unique_lock<std::mutex> lock(m_mutex);
  while (!m_done && m_buffer.empty()) m_notEmpty.wait(lock);
  if (m_buffer.empty()) return nullptr;  // done and drained
  bool wake = m_buffer.size() >= m_bufferCount;
  MessageType *m = m_buffer.back();
  m_buffer.pop_back();
  if (wake) m_notFull.notify_one();
  return m;
}

SampleQueue::MessageType *SampleQueue::TryGetNextSamples() {
  std::unique_lock<std::mutex> lock(m_mutex);
  if (m_buffer.empty()) return nullptr;
  bool wake = m_buffer.size() >= m_bufferCount;
  MessageType *m = m_buffer.back();
  m_buffer.pop_back();
  if (wake) m_notFull.notify_one();
  return m;
}

void SampleQueue::MessageProcessed(MessageType *message) {
  assert(message->GetHeader().m_kind != MessageHeader::Illegal);
  std::unique_lock<std::mutex> lock(m_historyMutex);
  if (m_historyCapacity == 0) {  // (the reference's zero-capacity ring would misbehave; recycle at once)
    Free(message);
    return;
  }
  if (m_history.size() >= m_historyCapacity) {
    MessageType *old = m_history.back();
    m_history.pop_back();
    Free(old);
  }
  m_history.push_front(message);
}

void SampleQueue::BeginWrite(uint64_t startSequenceId, std::string fileName) {
  (void)fileName;
  std::unique_lock<std::mutex> lock(m_historyMutex);
  m_writeStart = startSequenceId;
  m_writeEnd = std::numeric_limits<uint64_t>::max();
}

void SampleQueue::EndWrite(uint64_t sequenceId) {
  std::unique_lock<std::mutex> lock(m_historyMutex);
  m_writeEnd = sequenceId;
}

void SampleQueue::SetIsDone() {
  std::unique_lock<std::mutex> lock(m_mutex);
  assert(!m_done);
  m_done = true;
  m_notEmpty.notify_all();
}

bool SampleQueue::GetIsDone() {
  std::unique_lock<std::mutex> lock(m_mutex);
  return m_done;
}

bool SampleQueue::ReceivedAck() { return m_acknowledged; }
void SampleQueue::SendAck() {
  bool expected = false;
  m_acknowledged.compare_exchange_strong(expected, true);
}
void SampleQueue::ClearAck() {
  bool expected = true;
  m_acknowledged.compare_exchange_strong(expected, false);
}
