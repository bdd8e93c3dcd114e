// frequencyTable.h -- centre-frequency table of a sweep; same public surface as the
// reference's FrequencyTable (frequencyTable.h:3-29), built through scn_frequency_table.
#pragma once
#include <cstdint>
#include <vector>

class FrequencyTable {
 public:
  // frequencyTable.cpp:9-37.  Prints "Frequency %d: %.0f" per entry like the reference unless quiet.
  FrequencyTable(uint32_t sampleRate, double startFrequency, double stopFrequency, double useBandWidth,
                 double dcIgnoreWidth, bool quiet = false);
  double GetNextFrequency(void **pinfo = nullptr);     // advance cursor (wraps, counts sweeps)
  double GetCurrentFrequency(void **pinfo = nullptr);
  uint32_t GetFrequencyCount();
  double GetFrequencyFromIndex(uint32_t index);
  void SetFrequencyInfoForIndex(uint32_t index, void *info);
  uint32_t GetIterationCount();
  bool GetIsScanStart();
  double GetStartFrequency();
  double GetStopFrequency();

 private:
  struct Entry {
    double frequency;
    void *info;
  };
  std::vector<Entry> m_table;
  uint32_t m_index;
  uint32_t m_iterations;
};
