#include "messageQueue.h"

#include <cassert>
#include <chrono>
#include <cstdio>
#include <limits>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

// Copy into a pinned submit slot.  The destination is read next by the GPU's DMA engine, never by this core, so the
// stores bypass the cache (no read-for-ownership of lines that are about to be overwritten whole, and the source
// stays cached): on one core of the test host a 32 KiB buffer takes 1.6 us this way against 3.0 us with memcpy.
static void stream_copy(unsigned char *dst, const unsigned char *src, size_t bytes) {
#if defined(__SSE2__)
  if (((uintptr_t)dst & 15u) == 0 && bytes >= 256) {
    const size_t blocks = bytes / 64;
    for (size_t i = 0; i < blocks; i++) {
      const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src) + 4 * i);
      const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src) + 4 * i + 1);
      const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src) + 4 * i + 2);
      const __m128i d = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src) + 4 * i + 3);
      _mm_stream_si128(reinterpret_cast<__m128i *>(dst) + 4 * i, a);
      _mm_stream_si128(reinterpret_cast<__m128i *>(dst) + 4 * i + 1, b);
      _mm_stream_si128(reinterpret_cast<__m128i *>(dst) + 4 * i + 2, c);
      _mm_stream_si128(reinterpret_cast<__m128i *>(dst) + 4 * i + 3, d);
    }
    _mm_sfence();  // the stores are globally visible before the message is queued
    if (bytes % 64) memcpy(dst + blocks * 64, src + blocks * 64, bytes % 64);
    return;
  }
#endif
  memcpy(dst, src, bytes);
}

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
      m_historyCapacity(bufferCount / 10), m_poolSize(uint32_t(bufferCount * 1.1)), m_attachedRings(0), m_fillRing(0), m_stagedQueued(0), m_stagingCapacity(0), m_unstagedQueued(0),
      m_nextSequenceId(0), m_iterationCount(0), m_done(false), m_acknowledged(true), m_writeStart(0), m_writeEnd(0),
      m_writeShutdown(false), m_writeErrors(0), m_producerWaitNs(0), m_stagedAppends(0), m_copiedAppends(0), m_queuedAtAttach(0) {
  assert(kind > Illegal && kind <= FloatComplex);  // messageQueue.h:163
  assert(bufferCount > 0);
  if (m_poolSize <= bufferCount) m_poolSize = bufferCount + 1;
  if (m_doWrite && m_historyCapacity < 8) {  // capture needs some history to find the pre-trigger buffers in
    m_historyCapacity = 8;
    m_poolSize += 8;
  }
  for (uint32_t i = 0; i < m_poolSize; i++) m_free.push_back(new MessageType(m_bufferBytes));
  if (m_doWrite) {
    printf("Starting write thread...\n");  // messageQueue.h:166
    m_writeThread.reset(new std::thread(&SampleQueue::WriteThreadWorker, this));
  }
}

SampleQueue::~SampleQueue() {
  assert(m_buffer.empty() && m_stagedQueued == 0);  // messageQueue.h:173
  if (m_doWrite && m_writeThread) {
    printf("Stopping write thread...\n");  // messageQueue.h:176
    {
      std::unique_lock<std::mutex> lock(m_historyMutex);
      m_writeShutdown = true;
      m_writeWake.notify_all();
    }
    m_writeThread->join();
  }
  for (CaptureJob &job : m_captures)
    if (job.file) fclose(job.file);  // only without a write thread
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
  assert(aBytes + bBytes == m_bufferBytes);
  MessageType *message = nullptr;
  std::unique_lock<std::mutex> lock(m_mutex);
  auto timedWait = [this](std::unique_lock<std::mutex> &l) {
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    m_notFull.wait(l);
    m_producerWaitNs += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  };
  // a place in a consumer's pinned slot: wait for room in the queue AND in a slot.  The batch being filled goes on while its slot
  // is open and has room; a new batch starts in the NEXT attached ring (round-robin over the consumers) that has a usable slot --
  // the ring's fill slot if it is free, or the one after it once that one is full and still queued.  False once no ring is attached
  // (never attached, or detached while waiting).
  auto acquirePlace = [&]() -> bool {
    while (m_attachedRings) {
      if (m_unstagedQueued || m_buffer.size() + m_stagedQueued >= m_bufferCount) {  // (messages queued before the attach go first, through slots the consumers are given)
        timedWait(lock);
        continue;
      }
      const int R = (int)m_rings.size();
      int ring = -1;
      {
        StagingRing &cur = m_rings[m_fillRing];
        if (cur.attached) {
          StagingSlot &fs = cur.slots[cur.fillSlot];
          if (fs.state == StagingSlot::Open && fs.fill < m_stagingCapacity) ring = m_fillRing;  // the batch in progress
        }
      }
      for (int k = 1; ring < 0 && k <= R; k++) {  // a new batch: the next consumer's turn
        const int r = (m_fillRing + k) % R;
        if (m_rings[r].attached && UsableSlot(m_rings[r])) ring = r;
      }
      if (ring < 0) {
        timedWait(lock);  // (every ring's next slot is still in flight; ReleaseStaging and DetachStaging signal this too)
        continue;
      }
      m_fillRing = ring;
      StagingRing &g = m_rings[ring];
      StagingSlot &fs = g.slots[g.fillSlot];
      if (fs.state == StagingSlot::Free) {
        fs.state = StagingSlot::Open;
        fs.fill = 0;
      }
      // the message object is the slot's own (one per buffer place): a staged buffer costs no pool round trip
      message = g.messages[(size_t)g.fillSlot * m_stagingCapacity + fs.fill].get();
      message->m_ring = ring;
      message->m_slot = g.fillSlot;
      message->m_staged = fs.base + (size_t)fs.fill * m_bufferBytes;
      fs.fill++;
      return true;
    }
    return false;
  };
  const bool staged = acquirePlace();
  if (staged) {
    // The copy runs under the queue's lock: the consumer seals a slot by what is QUEUED, so a buffer must not be half-way
    // into a slot when that happens (1.6 us for a 32 KiB buffer; the consumer takes a whole batch per lock round trip).
    stream_copy(message->m_staged, static_cast<const unsigned char *>(a), aBytes);
    if (bBytes) stream_copy(message->m_staged + aBytes, static_cast<const unsigned char *>(b), bBytes);
  } else {
    lock.unlock();
    message = Allocate();
    message->m_staged = nullptr;
    message->m_slot = -1;
    message->m_ring = -1;
    memcpy(message->GetRawData(), a, aBytes);
    if (bBytes) memcpy(static_cast<unsigned char *>(message->GetRawData()) + aBytes, b, bBytes);
    lock.lock();
    // The consumer may have attached its slots while this buffer was being copied outside the lock: it then takes a place in a
    // slot like any later one (behind whatever was queued at the attach), so that from the attach on every NEW message is a
    // staged one and the consumer never has to fit a copied message between slots the producer is writing to.
    MessageType *const pooled = message;
    while (!m_attachedRings && m_buffer.size() >= m_bufferCount) timedWait(lock);  // (room in the queue first: an attach may come while waiting)
    if (acquirePlace()) {
      memcpy(message->m_staged, pooled->GetRawData(), m_bufferBytes);
      m_stagedAppends++;
      Free(pooled);
    } else {
      message = pooled;
      m_copiedAppends++;
    }
  }
  if (staged) m_stagedAppends++;
  MessageHeader &h = message->GetHeader();
  h.m_time = time;
  h.m_frequency = centerFrequency;
  h.m_kind = MessageHeader::ProcessData;
  h.m_referenceCount = 0;
  h.m_sequenceId = m_nextSequenceId++;
  if (message->m_ring >= 0) {  // (room in the queue was waited for when the place was taken)
    StagingRing &g = m_rings[message->m_ring];
    const bool wake = g.queued.empty();
    g.queued.push_back(message);
    m_stagedQueued++;
    if (wake) m_notEmpty.notify_all();  // (every consumer waits on the one condition, each for its own ring)
  } else {
    while (m_buffer.size() + m_stagedQueued >= m_bufferCount) timedWait(lock);
    const bool wake = m_buffer.empty();
    m_buffer.push_front(message);
    if (wake) m_notEmpty.notify_all();
  }
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
  bool wake = m_buffer.size() + m_stagedQueued >= m_bufferCount;
  MessageType *m = m_buffer.back();
  m_buffer.pop_back();
  if (m_unstagedQueued) m_unstagedQueued--;  // (a consumer without a ring taking what was queued before the first attach)
  if (wake || m_attachedRings) m_notFull.notify_all();
  return m;
}

SampleQueue::MessageType *SampleQueue::TryGetNextSamples() {
  std::unique_lock<std::mutex> lock(m_mutex);
  if (m_buffer.empty()) return nullptr;
  bool wake = m_buffer.size() + m_stagedQueued >= m_bufferCount;
  MessageType *m = m_buffer.back();
  m_buffer.pop_back();
  if (m_unstagedQueued) m_unstagedQueued--;
  if (wake || m_attachedRings) m_notFull.notify_all();
  return m;
}

bool SampleQueue::UsableSlot(StagingRing &g) {
  const int n = (int)g.slots.size();
  StagingSlot &fs = g.slots[g.fillSlot];
  if (fs.state == StagingSlot::Free || (fs.state == StagingSlot::Open && fs.fill < m_stagingCapacity)) return true;
  if (fs.state == StagingSlot::Open) {  // full and still queued: on to the next slot of the ring, once the consumer has released it
    const int next = (g.fillSlot + 1) % n;
    if (g.slots[next].state == StagingSlot::Free) {
      g.fillSlot = next;
      return true;
    }
  }
  return false;  // (the ring's next slot in submit order is still in flight)
}

int SampleQueue::AttachStaging(void *const *slotBases, uint32_t nSlots, uint32_t buffersPerSlot) {
  if (m_doWrite || nSlots < 2 || buffersPerSlot == 0) return -1;  // (the capture writer's history needs storage of its own)
  std::unique_lock<std::mutex> lock(m_mutex);
  if (m_attachedRings && buffersPerSlot != m_stagingCapacity) return -1;  // one batch size for all the consumers of a queue
  // The documented start order is Start, StartStreaming, THEN StartProcessing (signalSource.h:5, scan.cpp:234-238), and a plan
  // takes hundreds of milliseconds to create: the producer has almost always queued buffers by now, often a full queue of
  // them.  (Until round 5 a non-empty queue refused the attach -- silently, so the zero-copy path all but never engaged.)
  // Those messages stay what they are, pooled and unstaged; the consumers take them first, a slot's worth at a time, into
  // slots the queue reserves for them in ring order (TakeStagedBatch), and the producer opens no slot before the last of them
  // is gone: at no time do a consumer's copy and the producer's append write to the same slot, and nothing is reordered.
  m_rings.emplace_back();
  StagingRing &g = m_rings.back();
  for (uint32_t i = 0; i < nSlots; i++) g.slots.push_back(StagingSlot{static_cast<unsigned char *>(slotBases[i]), 0, StagingSlot::Free});
  for (size_t i = 0; i < (size_t)nSlots * buffersPerSlot; i++) g.messages.emplace_back(new MessageType(0));  // headers only
  g.fillSlot = 0;
  g.attached = true;
  if (!m_attachedRings) {  // the first consumer: from here on every NEW message is a staged one
    m_stagingCapacity = buffersPerSlot;
    m_unstagedQueued = m_buffer.size();
    m_queuedAtAttach = m_buffer.size();
    m_fillRing = (int)m_rings.size() - 1;
  }
  m_attachedRings++;
  m_notFull.notify_all();  // (a producer waiting for a slot has one more ring to look at)
  return (int)m_rings.size() - 1;
}

void SampleQueue::DetachStaging(int ring) {
  std::unique_lock<std::mutex> lock(m_mutex);
  if (ring < 0 || (size_t)ring >= m_rings.size() || !m_rings[ring].attached) return;
  StagingRing &g = m_rings[ring];
  g.attached = false;
  m_stagedQueued -= g.queued.size();  // (a consumer that gives up: what it had queued lives in its plan's slots and goes with them)
  for (MessageType *m : g.queued) m->m_header.m_kind = MessageHeader::Free;
  g.queued.clear();
  m_attachedRings--;
  if (!m_attachedRings) m_unstagedQueued = 0;
  m_notFull.notify_all();  // a producer waiting for a slot goes on with another ring, or with the messages' own storage
  m_notEmpty.notify_all();
}

uint32_t SampleQueue::TakeStagedBatch(int ring, std::vector<MessageType *> &out, int *slot, bool block, uint32_t lingerMicros) {
  std::unique_lock<std::mutex> lock(m_mutex);
  if (ring < 0 || (size_t)ring >= m_rings.size() || !m_rings[ring].attached) return 0;
  StagingRing &g = m_rings[ring];
  auto nothing = [&] { return g.queued.empty() && !(m_unstagedQueued && !m_buffer.empty()); };
  while (block && !m_done && nothing()) m_notEmpty.wait(lock);
  // An idle consumer woken by the first buffer of a burst lingers a moment before it seals the slot: taking that one
  // buffer at once would make the producer pay a futex wake for EVERY append (it signals whenever it finds the ring
  // empty: ~1 us per 32 KiB buffer, as much as the copy) and the GPU a launch per buffer.  A front-end that delivers a
  // buffer every 80 us is not held up: the wait ends after lingerMicros at the latest.
  if (block && lingerMicros && !m_done && !m_unstagedQueued && !g.queued.empty() && g.queued.size() < m_stagingCapacity)
    m_notEmpty.wait_until(lock, std::chrono::system_clock::now() + std::chrono::microseconds(lingerMicros));  // (system clock -> pthread_cond_timedwait: gcc 11's TSan does not know the steady-clock wait)
  if (nothing()) return 0;
  if (m_unstagedQueued) {
    // messages queued before the attach: up to a slot's worth of them, oldest first, and this ring's next slot to copy
    // them into -- reserved (in flight) from here on; the producer opens no slot while any of them is left
    const int rs = g.fillSlot;
    uint32_t n = 0;
    while (m_unstagedQueued && n < m_stagingCapacity) {
      out.push_back(m_buffer.back());
      m_buffer.pop_back();
      m_unstagedQueued--;
      n++;
    }
    g.slots[rs].state = StagingSlot::InFlight;
    g.slots[rs].fill = 0;
    g.fillSlot = (rs + 1) % (int)g.slots.size();
    *slot = rs;
    m_notFull.notify_all();
    return n;
  }
  const int s = g.queued.front()->m_slot;
  uint32_t n = 0;
  while (!g.queued.empty() && g.queued.front()->m_slot == s) {
    out.push_back(g.queued.front());
    g.queued.pop_front();
    n++;
  }
  m_stagedQueued -= n;
  g.slots[s].state = StagingSlot::InFlight;  // sealed: the producer moves on -- within this ring strictly in ring order, which is
  if (s == g.fillSlot) g.fillSlot = (s + 1) % (int)g.slots.size();  // the order the consumer submits its slots in
  *slot = s;
  m_notFull.notify_all();
  return n;
}

void SampleQueue::ReleaseStaging(int ring, int slot) {
  std::unique_lock<std::mutex> lock(m_mutex);
  if (ring >= 0 && (size_t)ring < m_rings.size() && slot >= 0 && (size_t)slot < m_rings[ring].slots.size())
    m_rings[ring].slots[slot].state = StagingSlot::Free;
  m_notFull.notify_all();
}

void SampleQueue::MessageProcessed(MessageType *message) {
  assert(message->GetHeader().m_kind != MessageHeader::Illegal);
  if (message->m_staged) {  // its samples live in the consumer's slot, which is about to be refilled, and so does the
    message->m_header.m_kind = MessageHeader::Free;  // message object itself: nothing to keep, nothing to recycle
    return;
  }
  std::unique_lock<std::mutex> lock(m_historyMutex);
  if (m_historyCapacity == 0) {  // (the reference's zero-capacity ring would misbehave; recycle at once)
    Free(message);
    return;
  }
  if (m_history.size() >= m_historyCapacity) {
    // Back-pressure the reference lacks: the oldest message is not recycled while the capture
    // writer still has to dump it (the reference silently loses records when its ring laps the writer).
    while (!m_history.empty() && WriterNeeds(m_history.back()->GetHeader().m_sequenceId)) {
      m_writeWake.notify_one();
      m_writeDrained.wait(lock);
    }
    MessageType *old = m_history.back();
    m_history.pop_back();
    Free(old);
  }
  m_history.push_front(message);
  if (!m_captures.empty()) m_writeWake.notify_one();  // messageQueue.h:272
}

bool SampleQueue::WriterNeeds(uint64_t sequenceId) const {
  for (const CaptureJob &job : m_captures)
    if (sequenceId >= job.next && sequenceId < job.end) return true;
  return false;
}

void SampleQueue::SetConverter(Converter c) {
  std::unique_lock<std::mutex> lock(m_historyMutex);
  m_converter = c;
}

void SampleQueue::BeginWrite(uint64_t startSequenceId, std::string fileName) {
  printf("BeginWrite %s: %lu\n", fileName.c_str(), (unsigned long)startSequenceId);  // messageQueue.h:276
  FILE *file = nullptr;
  if (m_doWrite) {
    file = fopen(fileName.c_str(), "w");
    if (!file) {
      fprintf(stderr, "Failed to open file '%s'\n", fileName.c_str());
      m_writeErrors++;
    }
  }
  std::unique_lock<std::mutex> lock(m_historyMutex);
  m_writeStart = startSequenceId;
  m_writeEnd = std::numeric_limits<uint64_t>::max();
  // a capture still waiting for its EndWrite when the next trigger begins ends where the new one starts
  if (!m_captures.empty() && m_captures.back().open) {
    m_captures.back().end = startSequenceId;
    m_captures.back().open = false;
  }
  if (file) {
    m_captures.push_back(CaptureJob{file, startSequenceId, m_writeEnd, true});
    m_writeWake.notify_one();
  }
}

void SampleQueue::EndWrite(uint64_t sequenceId) {
  printf("EndWrite %lu\n", (unsigned long)sequenceId);  // messageQueue.h:285
  std::unique_lock<std::mutex> lock(m_historyMutex);
  m_writeEnd = sequenceId;
  // only the job the matching BeginWrite queued gets its end: if that BeginWrite queued nothing (writing switched off, or
  // its file could not be opened) the newest job belongs to an EARLIER trigger whose end is already fixed -- leave it alone
  if (!m_captures.empty() && m_captures.back().open) {
    m_captures.back().end = sequenceId;
    m_captures.back().open = false;
  }
  m_writeWake.notify_one();
}

// messageQueue.h:98-139 restated as a job loop: for the oldest queued capture write, in sequence order, every
// processed message with an id in [next, end) as it reaches the history ring; the job is finished -- and its file
// closed, by this thread only -- once a message at or past its end id (or the end of the stream) shows up.
void SampleQueue::WriteThreadWorker() {
  std::vector<unsigned char> raw(m_bufferBytes);
  std::vector<float> converted(2 * (size_t)m_sampleCount);
  std::unique_lock<std::mutex> lock(m_historyMutex);
  while (true) {
    // (the reference's writer gives up as soon as the PRODUCER is done, messageQueue.h:100-121, losing
    // whatever the consumers had not processed yet; this one runs until the queue is torn down)
    while (!m_writeShutdown && m_captures.empty()) m_writeWake.wait(lock);
    if (m_captures.empty()) break;  // shutting down, nothing being written
    CaptureJob &job = m_captures.front();
    // oldest -> newest: find the first message this capture still wants
    MessageType *next = nullptr;
    bool pastEnd = false;
    for (auto it = m_history.rbegin(); it != m_history.rend(); ++it) {
      const uint64_t seq = (*it)->GetHeader().m_sequenceId;
      if (seq < job.next) continue;
      if (seq >= job.end) {
        pastEnd = true;
      } else {
        next = *it;
      }
      break;
    }
    if (next) {
      const uint64_t seq = next->GetHeader().m_sequenceId;
      memcpy(raw.data(), next->GetRawData(), m_bufferBytes);
      job.next = seq + 1;  // the ring may recycle the message from here on
      FILE *const file = job.file;  // stays open: only this thread closes it, and only after this record
      const Converter convert = m_converter;
      lock.unlock();
      printf("Writing %lu\n", (unsigned long)seq);  // messageQueue.h:125
      const void *data = raw.data();
      size_t bytes = m_bufferBytes;
      if (m_kind != FloatComplex) {
        if (convert) {
          convert(raw.data(), 1, converted.data());
          data = converted.data();
          bytes = sizeof(fftwf_complex) * m_sampleCount;
        } else {
          fprintf(stderr, "SampleQueue: no converter installed, writing raw samples\n");
        }
      }
      if (fwrite(data, 1, bytes, file) != bytes) m_writeErrors++;
      lock.lock();
      m_writeDrained.notify_all();
      continue;
    }
    if (pastEnd || m_writeShutdown) {  // capture complete (or the queue is going away): close its file
      fclose(job.file);
      m_captures.pop_front();
      m_writeDrained.notify_all();
      continue;
    }
    m_writeWake.wait(lock);
  }
}

void SampleQueue::SetIsDone() {
  {
    std::unique_lock<std::mutex> lock(m_mutex);
    assert(!m_done);
    m_done = true;
    m_notEmpty.notify_all();
  }
  std::unique_lock<std::mutex> lock(m_historyMutex);  // messageQueue.h:299-303: wake the writer too
  m_writeWake.notify_all();
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
