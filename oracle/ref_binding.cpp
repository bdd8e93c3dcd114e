// ref_binding.cpp -- C binding around the ONE reference source that builds in this image.
//
// TEST INFRASTRUCTURE ONLY (see scn_oracle.h).  /root/reference/frequencyTable.cpp needs nothing but the
// standard library, so it is compiled from where it lies (oracle/Makefile, target `ref`) together with this
// binding into oracle/_ref/libref_frequency_table.so and used to pin scn_oracle_frequency_table, the product's
// scn_frequency_table and the fixtures under tests/golden/.  Every other hot-path source of the reference
// includes <fftw3.h>, <volk/volk.h>, <gnuradio/fft/window.h> or Boost (utility.cpp through "fft.h"), none of
// which exist here, and stand-in headers are not allowed -- those stay unbuildable.
// No reference source is copied: this file only calls the class the reference header declares.
#include <cstdint>
#include <cstdio>
#include <vector>

#include "frequencyTable.h"  // -I/root/reference

extern "C" {

// FrequencyTable's constructor (frequencyTable.cpp:9-37): returns the count, fills at most cap entries
uint32_t ref_frequency_table(uint32_t sample_rate, double start, double stop, double use_bandwidth,
                             double dc_ignore_width, double *out, uint32_t cap) {
  FrequencyTable t(sample_rate, start, stop, use_bandwidth, dc_ignore_width);
  const uint32_t n = t.GetFrequencyCount();
  for (uint32_t i = 0; i < n && i < cap; i++) out[i] = t.GetFrequencyFromIndex(i);
  return n;
}

// The walk the producers do (GetCurrentFrequency / GetIsScanStart / GetNextFrequency, frequencyTable.cpp:39-110):
// `steps` tunes from a fresh table; per step the frequency tuned, the iteration count and the scan-start flag
// BEFORE advancing.
void ref_frequency_walk(uint32_t sample_rate, double start, double stop, double use_bandwidth, double dc_ignore_width,
                        uint32_t steps, double *frequency, uint32_t *iteration, uint8_t *scan_start) {
  FrequencyTable t(sample_rate, start, stop, use_bandwidth, dc_ignore_width);
  for (uint32_t k = 0; k < steps; k++) {
    frequency[k] = t.GetCurrentFrequency();
    iteration[k] = t.GetIterationCount();
    scan_start[k] = t.GetIsScanStart() ? 1 : 0;
    t.GetNextFrequency();
  }
}

double ref_frequency_start(uint32_t sample_rate, double start, double stop, double use_bandwidth, double dc_ignore_width) {
  FrequencyTable t(sample_rate, start, stop, use_bandwidth, dc_ignore_width);
  return t.GetStartFrequency();
}
double ref_frequency_stop(uint32_t sample_rate, double start, double stop, double use_bandwidth, double dc_ignore_width) {
  FrequencyTable t(sample_rate, start, stop, use_bandwidth, dc_ignore_width);
  return t.GetStopFrequency();
}

}  // extern "C"
