#include "frequencyTable.h"

#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "../../include/scanner_hip.h"

FrequencyTable::FrequencyTable(uint32_t sampleRate, double startFrequency, double stopFrequency,
                               double useBandWidth, double dcIgnoreWidth, bool quiet)
    : m_index(0), m_iterations(0) {
  uint32_t count = 0, first = 0;
  int st = scn_frequency_table(sampleRate, startFrequency, stopFrequency, useBandWidth, dcIgnoreWidth, 0, 1, nullptr,
                               0, &count, &first);
  // the reference asserts count > 0 (frequencyTable.cpp:29); a constructor of a library class throws instead
  if (st != SCN_OK) throw std::invalid_argument(std::string("FrequencyTable: ") + scn_last_error());
  std::vector<double> f(count);
  scn_frequency_table(sampleRate, startFrequency, stopFrequency, useBandWidth, dcIgnoreWidth, 0, 1, f.data(), count,
                      &count, &first);
  m_table.resize(count);
  for (uint32_t i = 0; i < count; i++) {
    if (!quiet) printf("Frequency %d: %.0f\n", i, f[i]);  // frequencyTable.cpp:34
    m_table[i] = Entry{f[i], nullptr};
  }
}

double FrequencyTable::GetNextFrequency(void **pinfo) {
  m_index++;
  if (m_index >= m_table.size()) {
    m_index = 0;
    m_iterations++;
  }
  return GetCurrentFrequency(pinfo);
}

double FrequencyTable::GetCurrentFrequency(void **pinfo) {
  Entry &e = m_table[m_index];
  if (pinfo) *pinfo = e.info;
  return e.frequency;
}

uint32_t FrequencyTable::GetFrequencyCount() { return (uint32_t)m_table.size(); }

double FrequencyTable::GetFrequencyFromIndex(uint32_t index) {
  assert(index < m_table.size());
  return m_table[index].frequency;
}

void FrequencyTable::SetFrequencyInfoForIndex(uint32_t index, void *info) {
  assert(index < m_table.size());
  m_table[index].info = info;
}

uint32_t FrequencyTable::GetIterationCount() { return m_iterations; }
bool FrequencyTable::GetIsScanStart() { return m_index == 0; }
double FrequencyTable::GetStartFrequency() { return m_table.front().frequency; }
double FrequencyTable::GetStopFrequency() { return m_table.back().frequency; }
