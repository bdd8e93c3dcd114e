// scan_synth.cpp -- wiring of the kept classes exactly as scan.cpp:211-239 does it, with the
// SyntheticSource in place of a USB front-end.  Not the reference's CLI (out of scope): a
// small driver for tests, demos and the integration transcript.
//
//   scan_synth --kind short_complex --n 4096 --fs 8000000 --start 88e6 --stop 108e6
//              --niterations 3 --threshold 10 --emitter 98.5e6:0.2 --emitter 101.1e6:0.05 [--dump raw.bin]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "frequencyTable.h"
#include "process.h"
#include "syntheticSource.h"

int main(int argc, char **argv) {
  uint32_t n = 4096, fs = 8000000, iterations = 2, enob = 12, threads = 1, batch = 1024, depth = 1024;
  double start = 88e6, stop = 108e6, sigma = 0.01;
  float threshold = 10.0f;
  uint64_t seed = 1;
  bool correctDC = false, timeDomain = false, useTable = false;
  std::string kindName = "short_complex", dump, outFile;
  uint32_t pre = 0, post = 0;
  unsigned long burstFirst = 1, burstLast = 0;
  uint32_t sweepBlocks = 0, scanOffset = 0, replay = 0;
  double burstGain = 1.0;
  std::vector<SyntheticSource::Emitter> emitters;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto val = [&]() -> const char * {
      if (i + 1 >= argc) {
        fprintf(stderr, "missing value for %s\n", a.c_str());
        exit(2);
      }
      return argv[++i];
    };
    if (a == "--n") n = (uint32_t)atol(val());
    else if (a == "--fs") fs = (uint32_t)atof(val());
    else if (a == "--start") start = atof(val());
    else if (a == "--stop") stop = atof(val());
    else if (a == "--niterations") iterations = (uint32_t)atol(val());
    else if (a == "--threshold") threshold = (float)atof(val());
    else if (a == "--enob") enob = (uint32_t)atol(val());
    else if (a == "--threads") threads = (uint32_t)atol(val());
    else if (a == "--batch") batch = (uint32_t)atol(val());
    else if (a == "--depth") depth = (uint32_t)atol(val());
    else if (a == "--sigma") sigma = atof(val());
    else if (a == "--seed") seed = (uint64_t)atoll(val());
    else if (a == "--kind") kindName = val();
    else if (a == "--correct-dc") correctDC = true;
    else if (a == "--table") useTable = true;  // hand the worker the sweep's frequency table (scn_plan_set_table / scn_submit_indexed)
    else if (a == "--mode") timeDomain = std::string(val()) == "time";
    else if (a == "--dump") dump = val();
    else if (a == "--outfile") outFile = val();
    else if (a == "--pre") pre = (uint32_t)atol(val());
    else if (a == "--post") post = (uint32_t)atol(val());
    else if (a == "--sweep-blocks") sweepBlocks = (uint32_t)atol(val());
    else if (a == "--scan-offset") scanOffset = (uint32_t)atof(val());
    else if (a == "--replay") replay = (uint32_t)atol(val());
    else if (a == "--burst") {  // first:last:gain
      const char *v = val();
      if (sscanf(v, "%lu:%lu:%lf", &burstFirst, &burstLast, &burstGain) != 3) { fprintf(stderr, "--burst wants first:last:gain\n"); return 2; }
    } else if (a == "--emitter") {
      const char *v = val();
      const char *c = strchr(v, ':');
      if (!c) { fprintf(stderr, "--emitter wants freq:amplitude\n"); return 2; }
      emitters.push_back(SyntheticSource::Emitter{atof(v), atof(c + 1)});
    } else {
      fprintf(stderr, "unknown option %s\n", a.c_str());
      return 2;
    }
  }
  SampleQueue::SampleKind kind;
  if (kindName == "float") kind = SampleQueue::FloatComplex;
  else if (kindName == "short_complex") kind = SampleQueue::ShortComplex;
  else if (kindName == "short") kind = SampleQueue::Short;
  else if (kindName == "byte") kind = SampleQueue::ByteComplex;
  else { fprintf(stderr, "unknown kind %s\n", kindName.c_str()); return 2; }

  SyntheticSource source(fs, n, start, stop, kind, seed, sigma);
  for (auto &e : emitters) source.AddEmitter(e.frequency, e.amplitude);
  if (!dump.empty() && !source.SetDumpFile(dump)) return 1;
  source.SetBurst(burstFirst, burstLast, burstGain);
  if (sweepBlocks) source.SetSweepFraming(sweepBlocks, scanOffset);  // HackRF sweep-mode transfers
  if (replay) source.SetReplay(replay);  // throughput runs: the generator hands out its first `replay` buffers again and again

  // scan.cpp:211-223
  ProcessSamples process(n, fs, enob, threshold, gr::fft::window::WIN_BLACKMAN_HARRIS, timeDomain ? ProcessSamples::TimeDomain : ProcessSamples::FrequencyDomain,
                         threads, outFile, 0.75, 0.0, pre, post);
  process.SetMaxBatch(batch);
  if (useTable) {  // the table the source retunes through, built the way the reference builds it (frequencyTable.cpp:9-37)
    FrequencyTable table(fs, start, stop, 0.75, 0.0, true);
    std::vector<double> centres;
    for (uint32_t i = 0; i < table.GetFrequencyCount(); i++) centres.push_back(table.GetFrequencyFromIndex(i));
    process.SetFrequencyTable(centres);
  }
  SampleQueue sampleQueue(kind, enob, n, depth, correctDC, outFile != "");  // scan.cpp:223

  // scan.cpp:234-238 (the source delivers numIterations+1 sweeps: the first one is the queue's warm-up discard)
  const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  if (!source.Start() || !source.StartStreaming(iterations + 1, sampleQueue)) {
    sampleQueue.SetIsDone();
    return 1;
  }
  const bool ok = process.StartProcessing(sampleQueue);
  source.StopStreaming();
  if (!ok) fprintf(stderr, "scan_synth: %s\n", process.GetLastError().c_str());
  const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  fprintf(stderr, "buffers %lu hits %lu\n", (unsigned long)process.GetBufferCount(), (unsigned long)process.GetHitCount());
  if (useTable) fprintf(stderr, "table: %lu batch(es) submitted as a run of the GPU-resident frequency table\n", (unsigned long)process.GetIndexedSubmitCount());
  // (the clock starts before the discarded warm-up sweep and includes plan creation: a lower bound on the pipeline's rate)
  fprintf(stderr, "seconds %.3f Msamples/s %.1f\n", seconds, (double)process.GetBufferCount() * n / seconds / 1e6);
  const ProcessSamples::WorkerTimes wt = process.GetWorkerTimes();
  fprintf(stderr, "producer blocked %.3f s; consumer: waiting for samples %.3f s, scn_submit %.3f s, scn_collect %.3f s, reporting %.3f s\n",
          sampleQueue.GetProducerWaitSeconds(), wt.waitProducer, wt.submit, wt.collect, wt.report);
  // which path the samples took: a silent fall-back to the copying worker is a different measurement
  fprintf(stderr, "staging: %u of %u consumer thread(s) zero-copy; appends written into a pinned slot %lu, copied into a pooled message %lu, queued before the attach %lu\n",
          process.GetStagedWorkerCount(), threads, (unsigned long)sampleQueue.GetStagedAppendCount(), (unsigned long)sampleQueue.GetCopiedAppendCount(),
          (unsigned long)sampleQueue.GetQueuedAtAttachCount());
  return ok ? 0 : 1;
}
