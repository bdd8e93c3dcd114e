#include "syntheticSource.h"

#include "../../include/scanner_hip.h"

#include <cassert>
#include <cmath>
#include <cstring>

namespace {
inline uint64_t splitmix64(uint64_t &s) {
  uint64_t z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
inline double uniform01(uint64_t &s) { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0) + 1e-300; }
}  // namespace

SyntheticSource::SyntheticSource(uint32_t sampleRate, uint32_t sampleCount, double startFrequency,
                                 double stopFrequency, SampleQueue::SampleKind kind, uint64_t seed, double noiseSigma,
                                 double useBandWidth, double dcIgnoreWidth)
    : SignalSource(sampleRate, sampleCount, startFrequency, stopFrequency, useBandWidth, dcIgnoreWidth),
      m_kind(kind), m_seed(seed), m_sigma(noiseSigma), m_bufferIndex(0), m_tuned(0), m_dump(nullptr) {
  size_t per = kind == SampleQueue::FloatComplex ? 8 : kind == SampleQueue::ByteComplex ? 2 : 4;
  m_bufferBytes = per * sampleCount;
}

SyntheticSource::~SyntheticSource() {
  StopThread();  // the producer may be inside Generate / the dump fwrite: join it BEFORE the members it uses go away
  if (m_dump) fclose(m_dump);
}

bool SyntheticSource::SetDumpFile(const std::string &path) {
  if (m_dump) fclose(m_dump);
  m_dump = fopen(path.c_str(), "wb");
  if (!m_dump) fprintf(stderr, "SyntheticSource: cannot open dump file '%s'\n", path.c_str());
  return m_dump != nullptr;
}

void SyntheticSource::GenerateN(double fc, uint64_t bufferIndex, void *raw, const uint32_t n) {
  const double twoPi = 6.283185307179586476925286766559;
  std::vector<double> re(n), im(n);
  uint64_t s = m_seed * 0x100000001b3ull + bufferIndex * 0x9e3779b97f4a7c15ull + 0x1234567;
  const double sigma = m_sigma * ((bufferIndex >= m_burstFirst && bufferIndex <= m_burstLast) ? m_burstGain : 1.0);
  for (uint32_t k = 0; k < n; k++) {  // Box-Muller
    double r = std::sqrt(-2.0 * std::log(uniform01(s))) * sigma;
    double a = twoPi * uniform01(s);
    re[k] = r * std::cos(a);
    im[k] = r * std::sin(a);
  }
  for (const Emitter &e : m_emitters) {
    double off = e.frequency - fc;
    if (std::fabs(off) >= m_sampleRate / 2.0) continue;  // outside the tuned band
    double w = twoPi * off / (double)m_sampleRate;
    double ph = twoPi * uniform01(s);
    for (uint32_t k = 0; k < n; k++) {
      re[k] += e.amplitude * std::cos(w * k + ph);
      im[k] += e.amplitude * std::sin(w * k + ph);
    }
  }
  auto clampRound = [](double v, double lo, double hi) {
    double r = std::nearbyint(v);
    return r < lo ? lo : r > hi ? hi : r;
  };
  switch (m_kind) {
    case SampleQueue::FloatComplex: {
      float *o = static_cast<float *>(raw);
      for (uint32_t k = 0; k < n; k++) {
        o[2 * k] = (float)re[k];
        o[2 * k + 1] = (float)im[k];
      }
      break;
    }
    case SampleQueue::ShortComplex: {
      int16_t *o = static_cast<int16_t *>(raw);
      for (uint32_t k = 0; k < n; k++) {
        o[2 * k] = (int16_t)clampRound(re[k] * 2047.0, -2048, 2047);
        o[2 * k + 1] = (int16_t)clampRound(im[k] * 2047.0, -2048, 2047);
      }
      break;
    }
    case SampleQueue::Short: {  // planar: I[n] then Q[n]
      int16_t *o = static_cast<int16_t *>(raw);
      for (uint32_t k = 0; k < n; k++) {
        o[k] = (int16_t)clampRound(re[k] * 2047.0, -2048, 2047);
        o[n + k] = (int16_t)clampRound(im[k] * 2047.0, -2048, 2047);
      }
      break;
    }
    case SampleQueue::ByteComplex: {
      int8_t *o = static_cast<int8_t *>(raw);
      for (uint32_t k = 0; k < n; k++) {
        o[2 * k] = (int8_t)clampRound(re[k] * 127.0, -128, 127);
        o[2 * k + 1] = (int8_t)clampRound(im[k] * 127.0, -128, 127);
      }
      break;
    }
    default: assert(false);
  }
}

void SyntheticSource::Push(SampleQueue *q, void *raw, double fc, time_t t) {
  if (m_dump) fwrite(raw, 1, m_bufferBytes, m_dump);
  switch (m_kind) {
    case SampleQueue::FloatComplex: q->AppendSamples(static_cast<fftwf_complex *>(raw), fc, t); break;
    case SampleQueue::ShortComplex: q->AppendSamples(static_cast<int16_t(*)[2]>(raw), fc, t); break;
    case SampleQueue::Short:
      q->AppendSamples(static_cast<int16_t *>(raw), static_cast<int16_t *>(raw) + m_sampleCount, fc, t);
      break;
    case SampleQueue::ByteComplex: q->AppendSamples(static_cast<int8_t(*)[2]>(raw), fc, t); break;
    default: assert(false);
  }
}

double SyntheticSource::Retune(double frequency) {
  m_tuned = frequency;
  return frequency;
}

bool SyntheticSource::GetNextSamples(SampleQueue *q, double_t &centerFrequency) {
  std::vector<unsigned char> raw(m_bufferBytes);
  centerFrequency = GetCurrentFrequency();
  Generate(centerFrequency, m_bufferIndex++, raw.data());
  Push(q, raw.data(), centerFrequency, 0);
  return true;
}

bool SyntheticSource::StartStreaming(uint32_t numIterations, SampleQueue &sampleQueue) {
  if (m_sweepBlocks) {  // hackRFSource.cpp:254: a transfer is a whole number of buffers
    const uint32_t count = m_sweepBlocks * 8192u;
    if (m_kind != SampleQueue::ByteComplex || count < m_sampleCount || count % m_sampleCount) {
      fprintf(stderr, "SyntheticSource: sweep framing needs byte IQ and a transfer that is a multiple of the buffer\n");
      return false;
    }
  }
  return StartThread(numIterations, sampleQueue);
}

// Same shape as the device workers (e.g. bladerfSource.cpp:244-302): tune, receive one
// buffer, advance the table, append with a scan-start timestamp on the first entry.
void SyntheticSource::ThreadWorker() {
  if (m_sweepBlocks) {
    SweepWorker();
    return;
  }
  std::vector<unsigned char> scratch(m_bufferBytes);
  if (m_replay) m_replayCache.resize(m_replay);
  Retune(GetCurrentFrequency());
  while (!GetIsDone() && !m_finished) {
    double centerFrequency = GetCurrentFrequency();
    bool isScanStart = GetIsScanStart();
    std::vector<unsigned char> &raw = m_replay ? m_replayCache[m_bufferIndex % m_replay] : scratch;
    if (raw.empty() || !m_replay) {
      raw.resize(m_bufferBytes);
      Generate(centerFrequency, m_bufferIndex, raw.data());
    }
    m_bufferIndex++;
    // a deterministic stand-in for time(NULL): one "second" per sweep, never 0
    time_t startTime = (time_t)(86400 + GetIterationCount());
    double next = GetNextFrequency();
    if (GetFrequencyCount() > 1) Retune(next);
    Push(m_sampleQueue, raw.data(), centerFrequency, isScanStart ? startTime : 0);
  }
}

// The shape of HackRFSource::hackRF_rx_callback (hackRFSource.cpp:224-270): one transfer per tune,
// in-band header -> centre frequency, then one AppendSamples per sampleCount samples of the transfer,
// every one of them carrying the scan-start time when the transfer opened a sweep.
void SyntheticSource::SweepWorker() {
  const uint32_t count = m_sweepBlocks * 8192u;
  std::vector<uint8_t> transfer(2u * (size_t)count);
  Retune(GetCurrentFrequency());
  while (!GetIsDone() && !m_finished) {
    const double tuned = GetCurrentFrequency();
    const bool isScanStart = GetIsScanStart();
    GenerateN(tuned, m_bufferIndex++, transfer.data(), count);
    const uint64_t hz = (uint64_t)tuned - m_scanOffset;  // what the firmware writes: tune minus the offset
    for (uint32_t b = 0; b < m_sweepBlocks; b++) {
      uint8_t *h = transfer.data() + (size_t)b * 16384u;
      h[0] = h[1] = 0x7F;
      for (int k = 0; k < 8; k++) h[2 + k] = (uint8_t)(hz >> (8 * k));
    }
    double centerFrequency = 0;
    int st = scn_hackrf_sweep_fixup(transfer.data(), 2u * count, m_scanOffset, &centerFrequency, nullptr);
    if (st != SCN_OK) {  // cannot happen for a well-formed transfer; end the stream rather than the process
      fprintf(stderr, "scn_hackrf_sweep_fixup: %s: %s\n", scn_error_name(st), scn_last_error());
      SetIsDone();
      break;
    }
    time_t startTime = (time_t)(86400 + GetIterationCount());
    double next = GetNextFrequency();
    if (GetFrequencyCount() > 1) Retune(next);
    for (uint32_t i = 0; i < count; i += m_sampleCount)
      Push(m_sampleQueue, transfer.data() + 2u * (size_t)i, centerFrequency, isScanStart ? startTime : 0);
  }
}
