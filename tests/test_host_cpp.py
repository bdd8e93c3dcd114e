"""The kept C++ host API (SignalSource / SampleQueue / SampleBuffer / ProcessInterface /
FrequencyTable / ProcessSamples) re-written on top of the C-ABI.
CPU: class semantics via a C++ test executable.  GPU: the scan_synth driver's stdout
transcript against the oracle fed the very bytes the SyntheticSource produced."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "scanner_amd", "host")


@pytest.fixture(scope="module")
def host_build(built_lib):
    from scanner_amd import build

    return build.build_host()


def test_host_classes_cpu(host_build, tmp_path):
    exe = tmp_path / "test_host_cpu"
    subprocess.check_call(["g++", "-std=gnu++11", "-O1", "-g", "-Wall", "-pthread", "-I", HOST,
                           os.path.join(ROOT, "tests", "cpp", "test_host_cpu.cpp"), "-o", str(exe),
                           "-L" + os.path.join(ROOT, "scanner_amd"), "-lscanner_host", "-lscanner_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "scanner_amd")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "host cpu tests ok" in out.stdout


def test_reference_style_subclass_compiles_and_runs(host_build, tmp_path):
    """SignalSource keeps the protected state the reference's device classes use by name (signalSource.h:11-43;
    bladerfSource.cpp:91-99 walks this->m_frequencyTable): a front-end written that way builds and streams."""
    exe = tmp_path / "subclass_compat"
    subprocess.check_call(["g++", "-std=gnu++11", "-O1", "-g", "-Wall", "-Werror", "-pthread", "-I", HOST,
                           os.path.join(ROOT, "tests", "cpp", "test_subclass_compat.cpp"), "-o", str(exe),
                           "-L" + os.path.join(ROOT, "scanner_amd"), "-lscanner_host", "-lscanner_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "scanner_amd")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120, cwd=tmp_path)
    assert out.returncode == 0 and "-> ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_host_classes_under_sanitizers(host_build, tmp_path, san):
    """CPU-only sanitizer runs of the queue / buffer / source classes (the reference ships none and has a
    known data race on its shared FFT plan).  The host sources are compiled INTO the test
    binary so they are instrumented; libscanner_hip is only linked for scn_frequency_table."""
    exe = tmp_path / "test_host_san"
    srcs = [os.path.join(HOST, f) for f in ("frequencyTable.cpp", "messageQueue.cpp", "signalSource.cpp",
                                            "syntheticSource.cpp", "fileSource.cpp", "processInterface.cpp", "sampleBuffer.cpp")]
    cmd = ["g++", "-std=gnu++11", "-O1", "-g", "-fno-omit-frame-pointer", f"-fsanitize={san}", "-pthread", "-I", HOST,
           os.path.join(ROOT, "tests", "cpp", "test_host_cpu.cpp")] + srcs + [
           "-o", str(exe), "-L" + os.path.join(ROOT, "scanner_amd"), "-lscanner_hip",
           "-Wl,-rpath," + os.path.join(ROOT, "scanner_amd")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not available: " + r.stderr[:200])
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=0",
               UBSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=env)
    if out.returncode != 0 and "unexpected memory mapping" in out.stderr:
        # ThreadSanitizer cannot lay out its shadow under this kernel's ASLR entropy: retry with ASLR off
        import shutil
        if not shutil.which("setarch"):
            pytest.skip("ThreadSanitizer cannot run under this kernel's ASLR settings")
        out = subprocess.run(["setarch", os.uname().machine, "-R", str(exe)], capture_output=True, text=True,
                             timeout=300, env=env)
        if out.returncode != 0 and ("unexpected memory mapping" in out.stderr or "setarch" in out.stderr):
            pytest.skip("ThreadSanitizer cannot run under this kernel's ASLR settings")
    assert out.returncode == 0 and "host cpu tests ok" in out.stdout, out.stderr[-3000:]


@pytest.mark.parametrize("san", ["", "thread", "address,undefined"])
def test_worker_slot_ring_against_a_fake_plan(host_build, tmp_path, san):
    """ProcessSamples::ThreadWorker keeps a ring of three submit slots in flight (process.cpp).  Against a fake of the C-ABI
    calls it makes (tests/cpp/test_worker_ring.cpp: protocol checks instead of DSP) on CPU: slots are only submitted when free,
    collected oldest first, every record is printed in submit order -- with the ring full, with the queue running empty, with
    more records than one collect window, and when a submit fails half way.  Plain, and under TSan / ASan+UBSan."""
    exe = tmp_path / "test_worker_ring"
    srcs = [os.path.join(HOST, f) for f in ("frequencyTable.cpp", "messageQueue.cpp", "signalSource.cpp", "syntheticSource.cpp",
                                            "processInterface.cpp", "sampleBuffer.cpp", "process.cpp")]
    cmd = ["g++", "-std=gnu++11", "-O1", "-g", "-fno-omit-frame-pointer", "-pthread", "-I", HOST] + (
        [f"-fsanitize={san}"] if san else []) + [os.path.join(ROOT, "tests", "cpp", "test_worker_ring.cpp")] + srcs + [
        "-o", str(exe), "-L" + os.path.join(ROOT, "scanner_amd"), "-lscanner_hip", "-Wl,-rpath," + os.path.join(ROOT, "scanner_amd")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and san and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not available: " + r.stderr[:200])
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)
    if out.returncode != 0 and "FATAL: ThreadSanitizer" in out.stderr and "unexpected memory mapping" in out.stderr:
        pytest.skip("TSan cannot run under this kernel's address-space layout")
    assert out.returncode == 0, out.stderr[-3000:] + out.stdout[-500:]
    assert "worker ring tests ok" in out.stdout


def test_abi_bench_refuses_what_it_does_not_understand(host_build):
    """bench.py's C++ child (the records legs): an unknown option or a depth outside the plan's slots is an error before
    anything touches the GPU, and without a GPU it says so instead of printing a line."""
    exe = os.path.join(HOST, "abi_bench")
    assert os.path.exists(exe)
    r = subprocess.run([exe, "--no-such-option", "1"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "unknown argument" in r.stderr and not r.stdout.strip()
    r = subprocess.run([exe, "--depth", "9"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "--depth" in r.stderr
    import torch

    if not torch.cuda.is_available():
        r = subprocess.run([exe, "--steps", "1"], capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_abi_bench_gather_loop(host_build):
    """abi_bench --mode gather: the steady-state gather loop from a C++ process on the system HIP runtime (a one-rank communicator:
    RCCL loaded with dlopen outside a torch process too) -- every launch's list is posted and waited for, none is lost."""
    import json

    exe = os.path.join(HOST, "abi_bench")
    r = subprocess.run([exe, "--batch", "512", "--steps", "40", "--warmup", "5", "--depth", "4", "--mode", "gather", "--flags", "4"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["mode"] == "gather" and d["lists_gathered"] == 40 and d["records_per_step"] > 100 and d["lag"] == 2
    plain = subprocess.run([exe, "--batch", "512", "--steps", "40", "--warmup", "5", "--depth", "4", "--mode", "counts", "--lag", "2", "--flags", "4"],
                           capture_output=True, text=True, timeout=300)
    assert plain.returncode == 0 and json.loads([ln for ln in plain.stdout.splitlines() if ln.startswith("{")][-1])["lists_gathered"] == 0


def test_reference_call_sites_compile(host_build, tmp_path):
    """The wiring of scan.cpp:211-239 written against the reference's names compiles unchanged."""
    src = tmp_path / "wiring.cpp"
    src.write_text('''
#include "process.h"
#include "syntheticSource.h"
#include "sampleBuffer.h"
int main(int argc, char**) {
  if (argc < 99) return 0;   // compile/link check only
  uint32_t sampleCount = 8192, sample_rate = 8000000, enob = 12; float threshold = 10.0;
  ProcessSamples::Mode mode = ProcessSamples::FrequencyDomain;
  ProcessSamples process(sampleCount, sample_rate, enob, threshold, gr::fft::window::WIN_BLACKMAN_HARRIS, mode, 2,
                         "", 0.75, 0.0, 2, 4);
  SampleQueue sampleQueue(SampleQueue::ShortComplex, enob, sampleCount, 1024, true, false);
  SignalSource * source = new SyntheticSource(sample_rate, sampleCount, 88e6, 108e6, SampleQueue::ShortComplex);
  source->Start();
  source->StartStreaming(10, sampleQueue);
  process.StartProcessing(sampleQueue);
  int16_t one[8192][2]; process.Run(one, 100000000);
  SampleBuffer sb(SampleBuffer::ShortComplex, enob, sampleCount); sb.AppendSamples(one, 1e8);
  return 0;
}''')
    subprocess.check_call(["g++", "-std=gnu++11", "-Wall", "-pthread", "-I", HOST, str(src), "-o",
                           str(tmp_path / "wiring"), "-L" + os.path.join(ROOT, "scanner_amd"), "-lscanner_host",
                           "-lscanner_hip", "-Wl,-rpath," + os.path.join(ROOT, "scanner_amd")])


KINDS = {"float": 4, "short_complex": 3, "short": 2, "byte": 1}


@pytest.mark.gpu
@pytest.mark.parametrize("kind,enob,dc,fs,table", [("short_complex", 12, False, 8000000, False), ("float", 12, False, 8000000, False), ("byte", 8, False, 8000000, False),
                                                   ("short", 12, True, 8000000, False),
                                                   # the HackRF's other rates (hackRFSource.cpp:156-161): fs / N does not divide, bin_step truncates
                                                   ("short_complex", 12, False, 20000000, False), ("byte", 8, False, 12500000, False),
                                                   # the worker with the sweep's frequency table on the GPU (ProcessSamples::SetFrequencyTable):
                                                   # batches of 5 over 7 / 4 centres, so the runs wrap; the same transcript
                                                   ("short_complex", 12, False, 8000000, True), ("float", 12, False, 12500000, True)])
def test_scan_synth_transcript_matches_oracle(host_build, oracle_mod, tmp_path, kind, enob, dc, fs, table):
    _, demo = host_build
    n, iters = 4096, 2
    dump = tmp_path / "raw.bin"
    cmd = [demo, "--kind", kind, "--n", str(n), "--fs", str(fs), "--start", "88e6", "--stop", "130e6",
           "--niterations", str(iters), "--threshold", "10", "--enob", str(enob), "--sigma", "0.02", "--batch", "5",
           "--depth", "16", "--dump", str(dump), "--emitter", "98.5e6:0.2", "--emitter", "101.1e6:0.05",
           "--emitter", "119.3e6:0.4", "--emitter", "127.0e6:0.1"] + (["--correct-dc"] if dc else []) + (["--table"] if table else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    if table:   # every batch went out as a run of the GPU-resident table (scn_submit_indexed)
        m = re.search(r"table: (\d+) batch", out.stderr)
        assert m and int(m.group(1)) >= 2
    got = [l for l in out.stdout.splitlines() if l.startswith("freq ")]
    starts = [l for l in out.stdout.splitlines() if l.startswith("Start scan at ")]

    # replay the exact bytes through the oracle: the first sweep is the queue's warm-up discard
    from scanner_amd import capi

    centres = capi.frequency_table(fs, 88e6, 130e6)[1]
    per = capi.BYTES_PER_SAMPLE[KINDS[kind]] * n
    raw = np.fromfile(dump, np.uint8)
    assert raw.size == per * len(centres) * (iters + 1)
    raw = raw[per * len(centres):]
    dt = {"float": np.complex64, "short_complex": np.int16, "short": np.int16, "byte": np.int8}[kind]
    fc = np.tile(centres, iters)
    o = oracle_mod.Oracle(n, fs, 10.0, kind=KINDS[kind], enob=enob, correct_dc=dc)
    p_ref, h_ref, _ = o.run(raw.view(dt), fc, np.arange(len(fc), dtype=np.uint64))
    from tests import tolerances as tol

    near = np.abs(p_ref[:, tol.evaluated_mask(n)] - 10.0) < tol.GUARD_DB
    assert not near.any(), "pick other emitters: a bin sits on the threshold"
    want = ["freq %d power_db" % h["freq_hz"] for h in h_ref]
    assert len(got) == len(want) > 20
    assert [re.match(r"(freq \d+ power_db)", l).group(1) for l in got] == want       # order + frequencies exact
    vals = np.array([float(l.split()[-1]) for l in got])
    assert np.abs(vals - h_ref["power_db"]).max() < 2e-3
    assert len(starts) == iters
    assert "Starting process thread 0" in out.stdout and "Stopped process thread 0" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("n,blocks", [(8192, 4), (4096, 2), (8192, 1)])
def test_scan_synth_hackrf_sweep_framing(host_build, oracle_mod, tmp_path, n, blocks):
    """SURVEY 8(f) row 4 end to end: HackRF sweep-mode transfers (in-band 0x7F7F header per 8192-sample
    block) -> scn_hackrf_sweep_fixup -> several int8 buffers per transfer at the header's frequency
    (hackRFSource.cpp:186-270) -> the HIP path; transcript against the oracle on the dumped bytes."""
    from scanner_amd import capi

    _, demo = host_build
    fs, iters, offset = 8000000, 2, 3000000
    dump = tmp_path / "raw.bin"
    cmd = [demo, "--kind", "byte", "--n", str(n), "--fs", str(fs), "--start", "2400e6", "--stop", "2440e6",
           "--niterations", str(iters), "--threshold", "12", "--enob", "8", "--sigma", "0.03", "--batch", "7",
           "--depth", "32", "--dump", str(dump), "--sweep-blocks", str(blocks), "--scan-offset", str(offset),
           "--emitter", "2412.3e6:0.3", "--emitter", "2437.1e6:0.2", "--emitter", "2426.6e6:0.1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = [l for l in out.stdout.splitlines() if l.startswith("freq ")]
    starts = [l for l in out.stdout.splitlines() if l.startswith("Start scan at ")]

    centres = capi.frequency_table(fs, 2400e6, 2440e6)[1]
    k = blocks * 8192 // n                      # buffers per transfer
    raw = np.fromfile(dump, np.int8).reshape(-1, 2 * n)
    assert raw.shape[0] == k * len(centres) * (iters + 1)
    fc_all = np.repeat(np.tile(centres, iters + 1), k)
    timed = np.repeat(np.tile(np.arange(len(centres)) == 0, iters + 1), k)   # buffers carrying a scan-start time
    # the fixup ran: the first five samples of every transfer equal the sixth, and the later blocks'
    # headers stay in the sample stream (the reference only ever looks at block 0)
    heads = raw[::k]
    assert (heads[:, :10].reshape(-1, 5, 2) == heads[:, None, 10:12]).all()
    if blocks > 1:
        block1 = raw[8192 // n]                 # the buffer that starts at sample 8192 of the first transfer
        assert block1[0] == 0x7F and block1[1] == 0x7F
    # SynchronizedAppend (messageQueue.h:65-72): the iteration count steps on EVERY timed buffer and
    # everything from its second step on is kept
    first_kept = int(np.flatnonzero(np.cumsum(timed) >= 2)[0])
    raw, fc, timed = raw[first_kept:], fc_all[first_kept:], timed[first_kept:]
    o = oracle_mod.Oracle(n, fs, 12.0, kind=KINDS["byte"], enob=8)
    p_ref, h_ref, _ = o.run(raw, fc, np.arange(len(fc), dtype=np.uint64))
    from tests import tolerances as tol

    near = np.abs(p_ref[:, tol.evaluated_mask(n)] - 12.0) < tol.GUARD_DB
    assert not near.any(), "pick other emitters: a bin sits on the threshold"
    want = ["freq %d power_db" % h["freq_hz"] for h in h_ref]
    assert len(got) == len(want) > 20
    assert [re.match(r"(freq \d+ power_db)", l).group(1) for l in got] == want
    vals = np.array([float(l.split()[-1]) for l in got])
    assert np.abs(vals - h_ref["power_db"]).max() < 2e-3
    assert len(starts) == int(timed.sum())


@pytest.mark.gpu
def test_scan_synth_replay_mode_repeats_its_first_buffers(host_build, tmp_path):
    """--replay R (the throughput runs of scripts/host_path_rate.sh): buffer k of the stream is buffer k % R, so with one centre
    frequency and R = 2 the hit lines of sweep j + 2 repeat those of sweep j exactly, and the run reports its rate."""
    _, demo = host_build
    r = subprocess.run([demo, "--kind", "short_complex", "--n", "4096", "--start", "433e6", "--stop", "0", "--niterations", "6",
                        "--threshold", "11", "--sigma", "0.02", "--batch", "2", "--depth", "4", "--emitter", "433.92e6:0.3",
                        "--replay", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert re.search(r"seconds [\d.]+ Msamples/s [\d.]+", r.stderr)
    sweeps, cur = [], None
    for l in r.stdout.splitlines():
        if l.startswith("Start scan"):
            cur = []
            sweeps.append(cur)
        elif l.startswith("freq ") and cur is not None:
            cur.append(l)
    assert len(sweeps) == 6 and len(sweeps[0]) > 3
    # the generator's buffer 0 is the discarded warm-up sweep, so processed sweep j carries generated buffer (j + 1) % 2
    assert sweeps[0] == sweeps[2] == sweeps[4] and sweeps[1] == sweeps[3] == sweeps[5] and sweeps[0] != sweeps[1]


@pytest.mark.gpu
def test_scan_synth_two_consumer_threads_and_time_domain(host_build, tmp_path):
    """scan.cpp:217 runs 2 consumer threads: two plans on their own streams share the queue.  Line
    order across threads is nondeterministic (as in the reference), the multiset of lines is not."""
    _, demo = host_build
    base = [demo, "--kind", "short_complex", "--n", "8192", "--start", "400e6", "--stop", "460e6", "--niterations", "3",
            "--threshold", "11", "--sigma", "0.02", "--batch", "3", "--depth", "8", "--emitter", "433.92e6:0.3",
            "--emitter", "446.0e6:0.08"]
    one = subprocess.run(base + ["--threads", "1"], capture_output=True, text=True, timeout=300)
    two = subprocess.run(base + ["--threads", "2"], capture_output=True, text=True, timeout=300)
    assert one.returncode == 0 and two.returncode == 0, one.stderr + two.stderr
    f1 = sorted(l for l in one.stdout.splitlines() if l.startswith("freq "))
    f2 = sorted(l for l in two.stdout.splitlines() if l.startswith("freq "))
    assert len(f1) > 10 and f1 == f2
    assert "Starting process thread 1" in two.stdout and "Stopped process thread 1" in two.stdout
    # both consumers run the zero-copy path: a staging ring each in the queue (the reference's shipped setting, scan.cpp:217)
    assert "staging: 2 of 2 consumer thread(s) zero-copy" in two.stderr and "staging: 1 of 1 consumer thread(s) zero-copy" in one.stderr
    # time-domain mode, the reference CLI's default (scan.cpp:87): one line per buffer whose peak is above threshold
    td = subprocess.run([demo, "--kind", "float", "--n", "8192", "--start", "400e6", "--stop", "460e6",
                         "--niterations", "2", "--mode", "time", "--threshold", "-7", "--sigma", "0.02",
                         "--emitter", "433.92e6:0.3"], capture_output=True, text=True, timeout=300)
    assert td.returncode == 0, td.stderr
    lines = [l for l in td.stdout.splitlines() if "Max signal" in l]
    # Reference quirk, reproduced on purpose (process.cpp:207): the running maximum starts at
    # numeric_limits<float>::min(), a tiny POSITIVE number, so for normalised samples (every dB value
    # negative) it never moves: all 10 tuned bands x 2 sweeps "exceed" a negative threshold and print 0.000000.
    assert len(lines) == 20
    m = [re.match(r"Sequence\[(\d+)\]: Max signal 0.000000 above threshold -7.000000 frequency (\d+), min (-?[\d.]+)", l)
         for l in lines]
    assert all(m) and [int(x.group(1)) for x in m] == list(range(20))
    mins = {int(x.group(2)): float(x.group(3)) for x in m[:10]}
    assert mins[433000000] > -10 and all(v < -30 for f, v in mins.items() if f != 433000000)   # the emitter's band
    # ... and with the CLI's default positive threshold (scan.cpp:94) nothing can ever print
    td2 = subprocess.run([demo, "--kind", "float", "--n", "8192", "--start", "400e6", "--stop", "460e6",
                          "--niterations", "2", "--mode", "time", "--threshold", "10", "--emitter", "433.92e6:0.3"],
                         capture_output=True, text=True, timeout=300)
    assert td2.returncode == 0 and "Max signal" not in td2.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("kind,enob", [("short_complex", 12), ("float", 12)])
def test_triggered_capture_file(host_build, oracle_mod, tmp_path, kind, enob):
    """Triggered capture (process.cpp:160-181,250-270; messageQueue.h:98-139,259-288): a wideband burst makes
    process_fft return true for three buffers; with pre=2 / post=3 the writer must dump exactly the buffers
    [first-2, last+3] as raw fftwf_complex[N] records (K1 for the integer formats runs on the GPU)."""
    _, demo = host_build
    n = 4096
    dump = tmp_path / "raw.bin"
    base = str(tmp_path / "cap_")
    cmd = [demo, "--kind", kind, "--n", str(n), "--start", "100e6", "--stop", "0", "--niterations", "40", "--threshold", "2",
           "--enob", str(enob), "--sigma", "0.01", "--batch", "4", "--depth", "32", "--burst", "20:22:10", "--dump", str(dump),
           "--outfile", base, "--pre", "2", "--post", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    # generation index g -> sequence id g-1 (the first sweep is the queue's warm-up discard)
    assert re.search(r"BeginWrite \S+: 17\n", out.stdout), out.stdout[-2000:]
    assert "EndWrite 25" in out.stdout
    assert [int(x) for x in re.findall(r"Writing (\d+)", out.stdout)] == list(range(17, 25))
    files = [f for f in os.listdir(tmp_path) if f.startswith("cap_")]
    assert len(files) == 1 and re.match(r"cap_\d{8}-\d\d:\d\d:\d\d-103000000-1$", files[0]), files   # process.cpp:160-171
    got = np.fromfile(tmp_path / files[0], np.complex64).reshape(-1, n)
    assert got.shape[0] == 8
    per = {"short_complex": 4, "float": 8}[kind] * n
    raw = np.fromfile(dump, np.uint8).reshape(-1, per)[18:26]          # generation indices 18..25
    o = oracle_mod.Oracle(n, kind={"short_complex": 3, "float": 4}[kind], enob=enob)
    dt = {"short_complex": np.int16, "float": np.complex64}[kind]
    want = np.stack([o.convert(r.view(dt)) for r in raw])
    assert np.array_equal(got, want)                                     # bit-exact K1 on the GPU


@pytest.mark.gpu
@pytest.mark.parametrize("n,total,batch", [(4096, 37, 8), (1024, 100, 33)])
def test_sample_buffer_through_pinned_staging_slots(host_build, oracle_mod, tmp_path, n, total, batch):
    """SampleBuffer -> HipStagingProcessInterface -> the plan's pinned slots -> scn_submit, double-buffered (the north
    star's replacement for sampleBuffer.cpp's host staging), in C++ on the GPU; spectra and hit list against the oracle."""
    from scanner_amd import capi, synth
    from tests import tolerances as tol

    exe = tmp_path / "test_staging_gpu"
    subprocess.check_call(["g++", "-std=gnu++11", "-O1", "-g", "-Wall", "-pthread", "-I", HOST,
                           os.path.join(ROOT, "tests", "cpp", "test_staging_gpu.cpp"), "-o", str(exe),
                           "-L" + os.path.join(ROOT, "scanner_amd"), "-lscanner_host", "-lscanner_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "scanner_amd")])
    x = synth.cfloat_batch(n, total, seed=77)
    fc = 100e6 + 6e6 * np.arange(total)
    p_ref, _, _ = oracle_mod.Oracle(n, 8000000, 1e9).run(x)
    thr = tol.pick_threshold(p_ref, n, start=9.0)
    p_ref, h_ref, t_ref = oracle_mod.Oracle(n, 8000000, thr).run(x, fc, np.arange(total, dtype=np.uint64))
    (tmp_path / "in.c64").write_bytes(x.tobytes())
    out = subprocess.run([str(exe), str(tmp_path / "in.c64"), str(n), str(total), str(batch), repr(thr), str(tmp_path / "out.bin")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    raw = (tmp_path / "out.bin").read_bytes()
    pos, powers, hits, trig, batches = 0, [], [], [], 0
    while pos < len(raw):
        nb, nh = np.frombuffer(raw, np.uint32, 2, pos)
        pos += 8
        powers.append(np.frombuffer(raw, np.float32, n * nb, pos).reshape(nb, n)); pos += 4 * n * nb
        hits.append(np.frombuffer(raw, capi.HIT_DTYPE, nh, pos)); pos += 24 * nh
        trig.append(np.frombuffer(raw, np.uint8, nb, pos)); pos += nb
        batches += 1
    assert batches == -(-total // batch) and batches > 2                   # both slots were used more than once
    tol.compare_spectra(np.concatenate(powers), p_ref)
    h = np.concatenate(hits)
    assert len(h) == len(h_ref) > 10
    for f in ("seq_id", "i", "freq_hz"):
        assert np.array_equal(h[f], h_ref[f]), f
    assert np.array_equal(np.concatenate(trig), t_ref)


@pytest.mark.gpu
def test_scan_synth_prints_every_hit_of_a_wideband_burst(host_build, oracle_mod, tmp_path):
    """A wideband burst makes every buffer it covers report > 1047 hits (process.cpp:62) -- far more than the 64 records
    per buffer the worker's hit buffer holds.  The reference prints every `freq ... power_db ...` line, so must the
    batched worker: it walks the rest of the GPU's ordered list with scn_collect_more (no silent truncation)."""
    from scanner_amd import capi

    _, demo = host_build
    n, fs, iters = 4096, 8000000, 30
    dump = tmp_path / "raw.bin"
    cmd = [demo, "--kind", "short_complex", "--n", str(n), "--start", "100e6", "--stop", "0", "--niterations", str(iters),
           "--threshold", "2", "--enob", "12", "--sigma", "0.01", "--batch", "4", "--depth", "32", "--burst", "10:14:10",
           "--dump", str(dump)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "reporting the first" not in out.stderr
    got = [l for l in out.stdout.splitlines() if l.startswith("freq ")]
    raw = np.fromfile(dump, np.int16).reshape(-1, 2 * n)[1:]          # generation 0 is the queue's warm-up discard
    fc = np.full(len(raw), capi.frequency_table(fs, 100e6, 0.0)[1][0])
    o = oracle_mod.Oracle(n, fs, 2.0, kind=KINDS["short_complex"], enob=12)
    _, h_ref, t_ref = o.run(raw, fc, np.arange(len(raw), dtype=np.uint64))
    assert t_ref.sum() == 5 and len(h_ref) > 5 * 1047 > 4 * 64                       # the burst buffers trigger
    assert len(got) == len(h_ref)
    assert [l.split()[1] for l in got] == [str(f) for f in h_ref["freq_hz"]]
