// messageQueue.h -- SampleQueue: the bounded producer/consumer queue between a SignalSource
// and ProcessSamples.  Same public surface as the reference's MessageQueue<fftwf_complex>
// (messageQueue.h:141-324) with one deliberate change of representation: a message holds the
// RAW device-format samples (int8 / int16 / float IQ), not converted floats -- the
// int->float convert (utility.cpp:9-84) runs on the GPU inside the fused kernel, so the
// producer thread no longer does arithmetic (the reference converts in AppendSamples,
// messageQueue.h:190-229).  Sequence ids, the discarded warm-up sweep, blocking behaviour,
// the ack bit and the recycle ring follow the reference.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <functional>
#include <memory>
#include <thread>
#include <cstdint>
#include <cstring>
#include <ctime>
#include <deque>
#include <list>
#include <mutex>
#include <string>
#include <vector>

#include "scannerCompat.h"

class SampleQueue {
 public:
  struct MessageHeader {  // messageQueue.h:15-28
    enum MessageKind { Illegal = 0, ProcessData, WriteData, WriteDataAndStop, Free } m_kind;
    uint32_t m_referenceCount;
    double m_frequency;
    uint64_t m_sequenceId;
    time_t m_time;
  };
  enum SampleKind { Illegal = 0, ByteComplex, Short, ShortComplex, FloatComplex };  // messageQueue.h:31-37

  class MessageType {  // Buffer<MessageHeader, T> of memoryPool.h:7-30
   public:
    MessageHeader m_header;
    explicit MessageType(size_t bytes) : m_header(), m_raw(bytes), m_staged(nullptr), m_slot(-1), m_ring(-1) {}
    MessageHeader &GetHeader() { return m_header; }
    void *GetRawData() { return m_staged ? static_cast<void *>(m_staged) : static_cast<void *>(m_raw.data()); }
    size_t GetRawBytes() const { return m_raw.size(); }
    // Valid for FloatComplex queues only (raw == converted); integer kinds are converted on the GPU.
    fftwf_complex *GetData() { return reinterpret_cast<fftwf_complex *>(GetRawData()); }
    int GetStagingSlot() const { return m_slot; }  // -1: the samples live in the message itself

   private:
    friend class SampleQueue;
    std::vector<unsigned char> m_raw;
    unsigned char *m_staged;  // staged queues: the samples' place in the consumer's pinned submit slot (AttachStaging)
    int m_slot, m_ring;
  };

  SampleKind m_kind;

  SampleQueue(SampleKind kind, uint32_t enob, uint32_t sampleCount, uint32_t bufferCount, bool correctDCOffset,
              bool doWrite);
  ~SampleQueue();

  // messageQueue.h:190-237 (the four wire formats of the device front-ends)
  void AppendSamples(int16_t *realSamples, int16_t *imagSamples, double centerFrequency, time_t time);
  void AppendSamples(int16_t shortComplexSamples[][2], double centerFrequency, time_t time);
  void AppendSamples(int8_t (*byteComplexSamples)[2], double centerFrequency, time_t time);
  void AppendSamples(fftwf_complex *floatComplexSamples, double centerFrequency, time_t time);

  MessageType *GetNextSamples();     // blocking; nullptr once done and empty (messageQueue.h:239-257)
  MessageType *TryGetNextSamples();  // non-blocking variant used to fill a batch; nullptr if empty
  void MessageProcessed(MessageType *message);  // messageQueue.h:259-273

  // Triggered capture (messageQueue.h:98-139, 275-288): with doWrite a write thread dumps the
  // processed messages with sequence ids in [start, end) from the history ring to `fileName`
  // as raw little-endian fftwf_complex[sampleCount] records -- the reference's file format.
  // Messages hold raw wire-format samples here, so integer kinds go through the converter
  // (K1 on the GPU, scn_convert_raw) that the consumer installs with SetConverter.
  void BeginWrite(uint64_t startSequenceId, std::string fileName);
  void EndWrite(uint64_t sequenceId);
  // Captures are jobs the write thread owns: BeginWrite opens the file and queues {file, [start, end)}, EndWrite
  // fixes the end of the newest job, and the writer alone writes to and closes each file once it has dumped the
  // job's last record -- so a re-trigger while the previous capture is still draining (a record costs a GPU
  // round trip for the integer formats) starts a second job instead of closing a file in use.
  typedef std::function<void(const void *raw, uint32_t nBuffers, float *out)> Converter;
  void SetConverter(Converter c);  // install before the consumers start; the writer calls a copy taken under the lock

  // Zero-copy staging (sampleBuffer.cpp's host staging "replaced by pinned double-buffered hipMemcpyAsync", BASELINE.json): every
  // consumer thread lends the queue the pinned submit slots of its scn_plan (scn_host_buffer) as a RING of its own.  AppendSamples
  // then writes a buffer straight into the next free place of the slot being filled -- the producer's one copy lands in pinned
  // memory -- and the consumer submits a slot as it is, without copying: TakeStagedBatch hands out every queued message of that
  // consumer's oldest slot (buffers 0 .. n-1 of it, in order) and seals the slot, so that the producer moves on; the slot returns
  // to the producer with ReleaseStaging, after the consumer has collected its results.  With several consumers (scan.cpp:217 runs
  // two) the producer deals its batches round-robin: when the slot it is filling is full or has been sealed it goes on to the
  // NEXT consumer's ring that has a usable slot.  Sequence ids are the queue's own, in append order, so every consumer sees
  // increasing ids and the consumers' outputs interleave as the reference's threads' do (messageQueue.h:239-257 hands a message
  // to whichever thread asks first).  The discarded warm-up sweep (messageQueue.h:67-72), the queue's bound and its blocking
  // behaviour are unchanged; a staged message is recycled as soon as it is processed (the history ring of the capture writer
  // needs storage of its own: a queue built with doWrite refuses staging and keeps the copying path).  Messages that were queued
  // before the first attach are handed out first, unstaged, a slot's worth at a time together with a slot of the asking
  // consumer's ring reserved for them: the consumer copies those itself.
  // AttachStaging returns the consumer's ring id (>= 0), or -1 when staging is refused.
  int AttachStaging(void *const *slotBases, uint32_t nSlots, uint32_t buffersPerSlot);
  void DetachStaging(int ring);  // the consumer is leaving; with the last ring gone appends fall back to the messages' own storage
  uint32_t TakeStagedBatch(int ring, std::vector<MessageType *> &out, int *slot, bool block, uint32_t lingerMicros = 0);
  void ReleaseStaging(int ring, int slot);

  void SetIsDone();
  bool GetIsDone();
  bool ReceivedAck();
  void SendAck();
  void ClearAck();

  // what a consumer needs to create its scn_plan
  uint32_t GetEnob() const { return m_enob; }
  uint32_t GetSampleCount() const { return m_sampleCount; }
  bool GetCorrectDCOffset() const { return m_correctDCOffset; }
  size_t GetBufferBytes() const { return m_bufferBytes; }
  uint32_t GetBufferCount() const { return m_bufferCount; }
  uint64_t GetWriteStartSequenceId() const { return m_writeStart; }
  uint64_t GetWriteEndSequenceId() const { return m_writeEnd; }
  uint32_t GetWriteErrorCount() const { return m_writeErrors; }  // files that could not be opened / short writes
  // seconds the producer has spent blocked in AppendSamples (queue full / no free staging slot): where a slow pipeline shows
  double GetProducerWaitSeconds() const { return m_producerWaitNs.load() * 1e-9; }
  // which path the appends took: written straight into a consumer's pinned slot / copied into a pooled message; and how many
  // messages were already queued when the consumer attached (the producer runs while the consumer is still creating its plan)
  uint64_t GetStagedAppendCount() const { return m_stagedAppends.load(); }
  uint64_t GetCopiedAppendCount() const { return m_copiedAppends.load(); }
  uint64_t GetQueuedAtAttachCount() const { return m_queuedAtAttach.load(); }

 private:
  void SynchronizedAppend(const void *a, size_t aBytes, const void *b, size_t bBytes, double centerFrequency,
                          time_t time);
  MessageType *Allocate();
  void Free(MessageType *m);

  uint32_t m_enob, m_sampleCount, m_bufferCount;
  bool m_correctDCOffset, m_doWrite;
  size_t m_bufferBytes;
  std::deque<MessageType *> m_buffer;  // front = newest (push_front / pop_back like the reference)
  std::deque<MessageType *> m_history; // processed messages kept for capture (bufferCount/10)
  size_t m_historyCapacity;
  std::list<MessageType *> m_free;     // MemoryPool (memoryPool.h:32-77), 1.1 x depth
  uint32_t m_poolSize;
  std::mutex m_mutex, m_poolMutex, m_historyMutex;
  std::condition_variable m_notEmpty, m_notFull, m_poolNotEmpty;
  // staging (guarded by m_mutex).  A slot: Free -> Open (being filled / holding queued messages) -> InFlight (taken) -> Free
  struct StagingSlot {
    unsigned char *base;
    uint32_t fill;
    enum { Free, Open, InFlight } state;
  };
  struct StagingRing {  // one per consumer
    std::vector<StagingSlot> slots;
    std::vector<std::unique_ptr<MessageType>> messages;  // [slot][place]: the message objects of the staged buffers
    std::deque<MessageType *> queued;                     // appended and not yet taken; front = oldest
    int fillSlot;                                         // the slot the producer fills next in this ring (ring order = the consumer's submit order)
    bool attached;
  };
  std::deque<StagingRing> m_rings;   // index = the id AttachStaging returned; entries stay (detached) until the queue goes (a deque: rings are never moved)
  uint32_t m_attachedRings;
  int m_fillRing;                    // the ring whose slot the producer is filling
  size_t m_stagedQueued;             // messages queued in the rings (they count towards the queue's bound)
  uint32_t m_stagingCapacity;        // buffers per slot (the same for every ring: ProcessSamples' max batch)
  size_t m_unstagedQueued;  // messages queued before the first AttachStaging that no consumer has taken yet (the oldest in the queue)
  bool UsableSlot(StagingRing &g);   // positions g.fillSlot on a slot the producer may write to, if the ring has one
  uint64_t m_nextSequenceId;
  uint32_t m_iterationCount;
  std::atomic<bool> m_done;
  std::atomic<bool> m_acknowledged;
  uint64_t m_writeStart, m_writeEnd;  // bounds of the newest capture (the reference's two fields, messageQueue.h:52-53)
  // write thread state (guarded by m_historyMutex)
  struct CaptureJob {
    FILE *file;
    uint64_t next, end;  // the writer still owes the records with ids in [next, end)
    bool open;           // its EndWrite has not come yet: `end` is still the placeholder
  };
  void WriteThreadWorker();
  bool WriterNeeds(uint64_t sequenceId) const;  // some queued capture has not dumped this record yet
  std::unique_ptr<std::thread> m_writeThread;
  std::condition_variable m_writeWake, m_writeDrained;
  std::deque<CaptureJob> m_captures;  // front = the one being written
  bool m_writeShutdown;  // set by the destructor: the consumers are gone, drain what is there and stop
  std::atomic<uint32_t> m_writeErrors;
  std::atomic<uint64_t> m_producerWaitNs;
  std::atomic<uint64_t> m_stagedAppends, m_copiedAppends, m_queuedAtAttach;
  Converter m_converter;
};
