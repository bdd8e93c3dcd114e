// syntheticSource.h -- a SignalSource that needs no USB hardware (the reference has none:
// every source wraps a vendor library).  Generates, deterministically from a
// seed, what a receiver tuned to each centre frequency would deliver: complex Gaussian
// noise plus every configured emitter that falls inside the tuned band, in the wire format
// of the chosen SampleKind (float IQ, int16 interleaved/planar as a 12-bit ADC, int8).
#pragma once
#include <cstdio>
#include <string>
#include <vector>

#include "signalSource.h"

class SyntheticSource : public SignalSource {
 public:
  struct Emitter {
    double frequency;  // absolute, Hz
    double amplitude;  // relative to full scale (1.0)
  };

  SyntheticSource(uint32_t sampleRate, uint32_t sampleCount, double startFrequency, double stopFrequency,
                  SampleQueue::SampleKind kind, uint64_t seed = 1, double noiseSigma = 0.01,
                  double useBandWidth = 0.75, double dcIgnoreWidth = 0.0);
  ~SyntheticSource() override;

  void AddEmitter(double frequency, double amplitude) { m_emitters.push_back(Emitter{frequency, amplitude}); }
  // Wideband burst: buffers with generation index in [first, last] get their noise scaled by `gain`
  // (enough bins above threshold to make process_fft return true, process.cpp:62 -> triggered capture).
  void SetBurst(uint64_t first, uint64_t last, double gain) { m_burstFirst = first; m_burstLast = last; m_burstGain = gain; }
  // Also append every generated raw buffer (queue order, including the discarded warm-up
  // sweep) to this file, so a test can replay the exact bytes through the CPU oracle.
  bool SetDumpFile(const std::string &path);  // false if the file cannot be created
  // HackRF sweep-mode framing (hackRFSource.cpp:186-270), ByteComplex only: every tune delivers ONE
  // transfer of blocksPerTransfer x 8192 samples whose blocks start with the firmware's in-band header
  // (0x7F 0x7F + tuned frequency, little-endian u64); the worker runs scn_hackrf_sweep_fixup on it and
  // appends transfer/sampleCount buffers at the frequency the header carried (+ scanOffset), exactly
  // as hackRF_rx_callback does.
  void SetSweepFraming(uint32_t blocksPerTransfer, uint32_t scanOffsetHz) {
    m_sweepBlocks = blocksPerTransfer;
    m_scanOffset = scanOffsetHz;
  }

  // Throughput runs: generate only the first `distinct` buffers (Box-Muller in double costs ~10 ns per sample, two orders of
  // magnitude more than the pipeline behind it) and hand them out again in turn -- buffer k is buffer k % distinct.
  void SetReplay(uint32_t distinct) { m_replay = distinct; }

  bool GetNextSamples(SampleQueue *sampleQueue, double_t &centerFrequency) override;
  bool StartStreaming(uint32_t numIterations, SampleQueue &sampleQueue) override;
  void ThreadWorker() override;
  double Retune(double frequency) override;

  // Fill `raw` (sampleCount samples in the kind's wire format) for one tune.
  void Generate(double centerFrequency, uint64_t bufferIndex, void *raw) {
    GenerateN(centerFrequency, bufferIndex, raw, m_sampleCount);
  }
  void GenerateN(double centerFrequency, uint64_t bufferIndex, void *raw, uint32_t n);
  size_t GetBufferBytes() const { return m_bufferBytes; }

 private:
  void Push(SampleQueue *q, void *raw, double fc, time_t t);
  void SweepWorker();
  SampleQueue::SampleKind m_kind;
  uint64_t m_seed;
  double m_sigma;
  std::vector<Emitter> m_emitters;
  size_t m_bufferBytes;
  uint64_t m_bufferIndex;
  double m_tuned;
  FILE *m_dump;
  uint64_t m_burstFirst = 1, m_burstLast = 0;
  double m_burstGain = 1.0;
  uint32_t m_sweepBlocks = 0, m_scanOffset = 0;
  uint32_t m_replay = 0;
  std::vector<std::vector<unsigned char> > m_replayCache;
};
