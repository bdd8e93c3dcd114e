// buffer.h -- the ProcessInterface visitor of the reference (buffer.h:9-24) and the visitors
// this build provides.  The reference's CircularBuffer ring internals (buffer.cpp) are host
// bookkeeping outside the hot path and are not rebuilt; SampleBuffer (sampleBuffer.h) keeps
// its public surface on top of a plain deque of raw buffers.
#pragma once
#include <cstdint>
#include <cstdio>

#include "scannerCompat.h"

template <typename ElementType>
class ProcessInterface {
  bool m_doMergeRequests;

 public:
  explicit ProcessInterface(bool doMergeRequests) : m_doMergeRequests(doMergeRequests) {}
  virtual ~ProcessInterface() {}
  bool GetDoMergeRequests() { return m_doMergeRequests; }
  virtual void Begin(uint64_t sequenceId, uint32_t totalItemCount) = 0;
  virtual void Process(const ElementType *items, uint32_t count) = 0;  // once per contiguous segment
  virtual void End() = 0;
};

// processInterface.cpp:63-90: gathers the segments into one caller-owned buffer.
class CopyBufferProcessInterface : public ProcessInterface<fftwf_complex> {
  uint32_t m_count, m_expectedCount;
  fftwf_complex *m_outputBuffer;

 public:
  explicit CopyBufferProcessInterface(fftwf_complex *outputBuffer);
  void Begin(uint64_t sequenceId, uint32_t totalItemCount) override;
  void Process(const fftwf_complex *items, uint32_t count) override;
  void End() override;
};

// processInterface.cpp:9-59: raw little-endian fftwf_complex dump.
class FileWriteProcessInterface : public ProcessInterface<fftwf_complex> {
  uint32_t m_count, m_expectedCount;
  FILE *m_outFile;
  bool m_failed;

 public:
  explicit FileWriteProcessInterface(const char *outFileName);
  bool Failed() const { return m_failed; }  // the file could not be created, or a write came up short
  ~FileWriteProcessInterface() override;
  void Begin(uint64_t sequenceId, uint32_t totalItemCount) override;
  void Process(const fftwf_complex *items, uint32_t count) override;
  void End() override;
};

// New here: copies segments of float IQ straight into a plan's pinned staging slot
// (scn_host_buffer), buffer index `slotIndex` -- the visitor form of "sampleBuffer.cpp's host
// staging replaced by pinned double-buffered hipMemcpyAsync".
class HipStagingProcessInterface : public ProcessInterface<fftwf_complex> {
  uint32_t m_count, m_expectedCount;
  fftwf_complex *m_slot;

 public:
  HipStagingProcessInterface(void *pinnedSlotBase, uint32_t samplesPerBuffer, uint32_t bufferIndex);
  void Begin(uint64_t sequenceId, uint32_t totalItemCount) override;
  void Process(const fftwf_complex *items, uint32_t count) override;
  void End() override;
};
