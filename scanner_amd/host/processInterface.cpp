#include <cassert>
#include <cstdlib>
#include <cstring>

#include "buffer.h"

CopyBufferProcessInterface::CopyBufferProcessInterface(fftwf_complex *outputBuffer)
    : ProcessInterface(false), m_count(0), m_expectedCount(0), m_outputBuffer(outputBuffer) {}
void CopyBufferProcessInterface::Begin(uint64_t, uint32_t total) { m_expectedCount = total; }
void CopyBufferProcessInterface::Process(const fftwf_complex *items, uint32_t count) {
  memcpy(m_outputBuffer + m_count, items, sizeof(fftwf_complex) * count);
  m_count += count;
}
void CopyBufferProcessInterface::End() {
  assert(m_expectedCount == m_count);
  m_count = m_expectedCount = 0;
}

FileWriteProcessInterface::FileWriteProcessInterface(const char *name)
    : ProcessInterface(true), m_count(0), m_expectedCount(0), m_outFile(nullptr), m_failed(false) {
  if (name) {
    m_outFile = fopen(name, "w");
    if (!m_outFile) {  // (processInterface.cpp:20-23 exits; here the visitor reports it through Failed())
      fprintf(stderr, "FileWriteProcessInterface: cannot create '%s'\n", name);
      m_failed = true;
    }
  }
}
FileWriteProcessInterface::~FileWriteProcessInterface() {
  if (m_outFile) fclose(m_outFile);
}
void FileWriteProcessInterface::Begin(uint64_t, uint32_t total) { m_expectedCount = total; }
void FileWriteProcessInterface::Process(const fftwf_complex *items, uint32_t count) {
  if (m_outFile && fwrite(items, sizeof(fftwf_complex), count, m_outFile) != count) m_failed = true;  // short write
  m_count += count;
}
void FileWriteProcessInterface::End() {
  assert(m_expectedCount == m_count);
  m_count = m_expectedCount = 0;
}

HipStagingProcessInterface::HipStagingProcessInterface(void *base, uint32_t samplesPerBuffer, uint32_t bufferIndex)
    : ProcessInterface(false), m_count(0), m_expectedCount(0),
      m_slot(static_cast<fftwf_complex *>(base) + (size_t)samplesPerBuffer * bufferIndex) {}
void HipStagingProcessInterface::Begin(uint64_t, uint32_t total) { m_expectedCount = total; }
void HipStagingProcessInterface::Process(const fftwf_complex *items, uint32_t count) {
  memcpy(m_slot + m_count, items, sizeof(fftwf_complex) * count);
  m_count += count;
}
void HipStagingProcessInterface::End() {
  assert(m_expectedCount == m_count);
  m_count = m_expectedCount = 0;
}
