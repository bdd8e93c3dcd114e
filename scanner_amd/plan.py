"""`Plan`: thin object wrapper over one scn_plan handle.

Mirrors what one consumer thread of the reference owns (ProcessSamples + SampleQueue ctor
arguments: process.h:74-85, messageQueue.h:141-146).  Device memory handed to
``submit_device`` is a torch tensor (plumbing only); results come back as numpy arrays.
"""
import ctypes as C

import numpy as np

from . import capi


class Plan:
    def __init__(self, n, sample_rate=8000000, threshold=10.0, kind=capi.KIND_FLOAT_COMPLEX, enob=12,
                 correct_dc=False, max_batch=1, use_bandwidth=0.75, dc_ignore_bins=4, trigger_count=1047,
                 max_hits=0, flags=capi.OUT_SPECTRUM | capi.OUT_HITS, device_id=0,
                 window_type=capi.WIN_BLACKMAN_HARRIS, mode=capi.MODE_FREQUENCY_DOMAIN):
        self._L = capi.lib()
        d = capi.PlanDesc()
        d.struct_size = C.sizeof(capi.PlanDesc)
        d.n = n
        d.sample_rate = int(sample_rate)
        d.sample_kind = kind
        d.enob = enob
        d.correct_dc = int(bool(correct_dc))
        d.window_type = window_type
        d.mode = mode
        d.threshold = threshold
        d.dc_ignore_bins = capi.DC_IGNORE_NONE if dc_ignore_bins == 0 else dc_ignore_bins
        d.use_bandwidth = use_bandwidth
        d.trigger_count = trigger_count
        d.max_batch = max_batch
        d.max_hits = max_hits
        d.flags = flags
        d.device_id = device_id
        self.n, self.kind, self.max_batch, self.flags, self.device_id = n, kind, max_batch, flags, device_id
        self.sample_rate = int(sample_rate)
        self.use_bandwidth = use_bandwidth
        self.buffer_bytes = capi.BYTES_PER_SAMPLE[kind] * n
        self._h = C.c_void_p()
        capi.check(self._L.scn_plan_create(C.byref(d), C.byref(self._h)), "scn_plan_create")
        self._nb = [0] * capi.NUM_SLOTS
        self._keep = [None] * capi.NUM_SLOTS
        self._submit_device, self._collect, self._n_hits = self._L.scn_submit_device, self._L.scn_collect, C.c_uint32()
        self._submit_device_indexed = self._L.scn_submit_device_indexed

    # -- lifetime -----------------------------------------------------------
    @property
    def handle(self):
        """the scn_plan* (for C-ABI calls that take a plan: scn_gather_hits_device)"""
        return self._h

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.scn_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- staging ------------------------------------------------------------
    def host_buffer(self, slot):
        """The pinned staging slot as a writable uint8 numpy view (max_batch raw buffers)."""
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        capi.check(self._L.scn_host_buffer(self._h, slot, C.byref(ptr), C.byref(nbytes)), "scn_host_buffer")
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(nbytes.value,))

    @staticmethod
    def _meta(nb, center_freqs, seq_ids):
        fc = np.zeros(nb, np.float64) if center_freqs is None else np.ascontiguousarray(center_freqs, np.float64)
        seq = None if seq_ids is None else np.ascontiguousarray(seq_ids, np.uint64)
        assert fc.size == nb and (seq is None or seq.size == nb)
        return fc, seq

    def set_table(self, center_freqs):
        """scn_plan_set_table: the frequency table the source retunes through (frequencyTable.cpp:31-36), uploaded once; a
        submit with `first_index` then tags buffer b with entry (first_index + b) % len(table) and sends no per-buffer centres."""
        fc = np.ascontiguousarray(center_freqs, np.float64).reshape(-1)
        capi.check(self._L.scn_plan_set_table(self._h, fc.ctypes.data_as(C.c_void_p), fc.size), "scn_plan_set_table")

    def submit(self, slot, n_buffers, center_freqs=None, seq_ids=None, first_index=None):
        """Process the first n_buffers raw buffers of the pinned slot (async)."""
        if first_index is not None:
            assert center_freqs is None, "a submit names its centres or a run of the plan's table, not both"
            seq = None if seq_ids is None else np.ascontiguousarray(seq_ids, np.uint64)
            capi.check(self._L.scn_submit_indexed(self._h, slot, n_buffers, int(first_index),
                                                  None if seq is None else seq.ctypes.data_as(C.c_void_p)), "scn_submit_indexed")
            self._nb[slot] = n_buffers
            return
        fc, seq = self._meta(n_buffers, center_freqs, seq_ids)
        capi.check(self._L.scn_submit(self._h, slot, n_buffers, fc.ctypes.data_as(C.c_void_p),
                                      None if seq is None else seq.ctypes.data_as(C.c_void_p)), "scn_submit")
        self._nb[slot] = n_buffers

    def submit_device(self, slot, d_raw, n_buffers=None, center_freqs=None, seq_ids=None, d_power_db=None,
                      sync_producer=True, first_index=None):
        """Process raw IQ already in device memory (a torch tensor or an int address).

        The plan runs on its own non-blocking HIP stream.  With sync_producer (default) the
        torch stream that produced `d_raw` is drained first, so a tensor that was just
        written (`.cuda()`, a generator kernel) is complete before the plan reads it; pass
        False when the caller has already ordered the two streams."""
        if hasattr(d_raw, "data_ptr"):
            if sync_producer:
                import torch

                torch.cuda.current_stream(d_raw.device).synchronize()
            nbytes = d_raw.numel() * d_raw.element_size()
            if n_buffers is None:
                n_buffers = nbytes // self.buffer_bytes
            assert d_raw.is_contiguous() and nbytes >= n_buffers * self.buffer_bytes
            ptr = d_raw.data_ptr()
        else:
            ptr = int(d_raw)
        out_ptr = None
        if d_power_db is not None:
            assert d_power_db.is_contiguous() and d_power_db.numel() >= n_buffers * self.n
            out_ptr = C.c_void_p(d_power_db.data_ptr())
        if first_index is not None:  # a run of the plan's frequency table (set_table)
            assert center_freqs is None, "a submit names its centres or a run of the plan's table, not both"
            seq = None if seq_ids is None else np.ascontiguousarray(seq_ids, np.uint64)
            capi.check(self._L.scn_submit_device_indexed(self._h, slot, C.c_void_p(ptr), n_buffers, int(first_index),
                                                         None if seq is None else seq.ctypes.data_as(C.c_void_p), out_ptr),
                       "scn_submit_device_indexed")
        else:
            fc, seq = self._meta(n_buffers, center_freqs, seq_ids)
            capi.check(self._L.scn_submit_device(self._h, slot, C.c_void_p(ptr), n_buffers,
                                                 fc.ctypes.data_as(C.c_void_p),
                                                 None if seq is None else seq.ctypes.data_as(C.c_void_p), out_ptr),
                       "scn_submit_device")
        self._nb[slot] = n_buffers
        self._keep[slot] = (d_raw, d_power_db)  # keep the tensors alive until collected

    # -- the same two calls without the conveniences: for loops whose step is tens of microseconds ---------------
    def submit_prepared(self, slot, raw_ptr, n_buffers, fc_ptr, seq_ptr, out_ptr):
        """scn_submit_device with everything already a C value (device addresses as ints / c_void_p, the header arrays as
        pointers the caller keeps alive): one ctypes call, no array conversion, nothing retained.  A 2048-buffer launch takes
        24 us on the GPU; submit_device + collect cost more than that in Python alone."""
        st = self._submit_device(self._h, slot, raw_ptr, n_buffers, fc_ptr, seq_ptr, out_ptr)
        if st:
            capi.check(st, "scn_submit_device")
        self._nb[slot] = n_buffers

    def submit_prepared_indexed(self, slot, raw_ptr, n_buffers, first_index, seq_ptr, out_ptr):
        """scn_submit_device_indexed the same way: the centres are a run of the plan's table (set_table)."""
        st = self._submit_device_indexed(self._h, slot, raw_ptr, n_buffers, first_index, seq_ptr, out_ptr)
        if st:
            capi.check(st, "scn_submit_device_indexed")
        self._nb[slot] = n_buffers

    def collect_counts(self, slot, trigger_ptr=None):
        """scn_collect for the per-buffer counts / trigger flags only; returns the batch's total number of hits."""
        st = self._collect(self._h, slot, None, None, 0, C.byref(self._n_hits), trigger_ptr)
        if st:
            capi.check(st, "scn_collect")
        self.last_n_hits = self._n_hits.value
        return self.last_n_hits

    def wait(self, slot):
        capi.check(self._L.scn_wait(self._h, slot), "scn_wait")

    def collect(self, slot, want_power=True, want_hits=True, hit_cap=None, hits_out=None):
        """Block on the slot and return (power_db [B,n] | None, hits (HIT_DTYPE) | None, trigger uint8[B] | None).
        The hit list arrives ordered by (buffer, i) and complete from the GPU.  With hit_cap=None every hit is returned
        however many there are (the part beyond the plan's pinned list is fetched with scn_collect_more); with an
        explicit hit_cap the C-ABI's own behaviour shows: ScannerError(E_TRUNCATED) when more hits exist.
        hits_out: a caller-owned HIT_DTYPE array to receive the records (its length is the capacity): a loop that
        collects every step should not pay for a fresh 12 MB allocation and its page faults each time."""
        nb = self._nb[slot]
        have_hits = bool(self.flags & capi.OUT_HITS)
        power = np.empty((nb, self.n), np.float32) if (want_power and self.flags & capi.OUT_SPECTRUM) else None
        want_hits = want_hits and have_hits
        cap = (nb * 64 + 1024) if hit_cap is None else hit_cap
        if hits_out is not None:
            assert hits_out.dtype == capi.HIT_DTYPE and hits_out.flags.c_contiguous
            cap, hit_cap = len(hits_out), len(hits_out)
        hits = (hits_out if hits_out is not None else np.empty(cap, capi.HIT_DTYPE)) if want_hits else None
        trig = np.zeros(nb, np.uint8) if have_hits else None
        n_hits = C.c_uint32()
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
        st = self._L.scn_collect(self._h, slot, vp(power), vp(hits), cap if want_hits else 0, C.byref(n_hits),
                                 vp(trig))
        self._keep[slot] = None
        self.last_n_hits = n_hits.value
        if st == capi.E_TRUNCATED and want_hits and hit_cap is None:
            hits = self.collect_more(slot, 0, n_hits.value)
            st = capi.OK
        capi.check(st, "scn_collect")
        if want_hits:
            hits = hits[: min(n_hits.value, len(hits))]
        return power, hits, trig

    def collect_more(self, slot, first, count):
        """Records [first, first+count) of the ordered hit list of the slot's last collected submit (scn_collect_more)."""
        out = np.empty(count, capi.HIT_DTYPE)
        done = 0
        while done < count:
            got = C.c_uint32()
            capi.check(self._L.scn_collect_more(self._h, slot, first + done, out[done:].ctypes.data_as(C.c_void_p),
                                                count - done, C.byref(got)), "scn_collect_more")
            if not got.value:
                break
            done += got.value
        return out[:done]

    def hits_view(self, slot):
        """Zero-copy numpy view of the plan's pinned ordered hit list (valid until the slot's next submit)."""
        ptr, n = C.c_void_p(), C.c_uint32()
        capi.check(self._L.scn_hits_view(self._h, slot, C.byref(ptr), C.byref(n)), "scn_hits_view")
        if not n.value:
            return np.zeros(0, capi.HIT_DTYPE)
        raw = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n.value * capi.HIT_DTYPE.itemsize,))
        return raw.view(capi.HIT_DTYPE)

    def collect_time_domain(self, slot):
        """Time-domain plans: (max_db float32[B], min_db float32[B], above uint8[B]) -- process.cpp:203-237."""
        nb = self._nb[slot]
        mx, mn, ab = np.empty(nb, np.float32), np.empty(nb, np.float32), np.zeros(nb, np.uint8)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        capi.check(self._L.scn_collect_time_domain(self._h, slot, vp(mx), vp(mn), vp(ab)), "scn_collect_time_domain")
        self._keep[slot] = None
        return mx, mn, ab

    def convert_raw(self, raw):
        """K1 alone on the GPU: raw buffers (numpy, the plan's wire format) -> complex64 [B, n]."""
        raw = np.ascontiguousarray(raw)
        nb = raw.nbytes // self.buffer_bytes
        out = np.empty((nb, self.n), np.complex64)
        capi.check(self._L.scn_convert_raw(self._h, raw.ctypes.data_as(C.c_void_p), nb, out.ctypes.data_as(C.c_void_p)),
                   "scn_convert_raw")
        return out

    # -- plumbing -----------------------------------------------------------
    @property
    def stream_handle(self):
        s = C.c_void_p()
        capi.check(self._L.scn_plan_stream(self._h, C.byref(s)), "scn_plan_stream")
        return s.value or 0

    def slot_stream_handle(self, slot):
        """The stream slot's kernels run on (differs per slot only with capi.PLAN_OVERLAP_SLOTS)."""
        s = C.c_void_p()
        capi.check(self._L.scn_slot_stream(self._h, slot, C.byref(s)), "scn_slot_stream")
        return s.value or 0

    def window(self):
        w = np.empty(self.n, np.float32)
        capi.check(self._L.scn_plan_window(self._h, w.ctypes.data_as(C.c_void_p), self.n), "scn_plan_window")
        return w


class WelchPlan:
    """Streaming 65 536-pt, 50 %-overlap Welch PSD (BASELINE config C5) -- wrapper over scn_welch."""

    def __init__(self, n=65536, segments_per_psd=16, max_psd=1, device_id=0, window_type=capi.WIN_BLACKMAN_HARRIS,
                 kind=capi.KIND_FLOAT_COMPLEX, enob=0, correct_dc=False):
        self._L = capi.lib()
        d = capi.WelchDesc()
        d.struct_size = C.sizeof(capi.WelchDesc)
        d.n, d.segments_per_psd, d.window_type, d.max_psd, d.device_id = n, segments_per_psd, window_type, max_psd, device_id
        d.sample_kind, d.enob, d.correct_dc = kind, enob, int(bool(correct_dc))
        self.n, self.k, self.max_psd, self.hop, self.kind = n, segments_per_psd, max_psd, n // 2, kind
        self.bytes_per_sample = capi.BYTES_PER_SAMPLE[kind]
        self._h = C.c_void_p()
        capi.check(self._L.scn_welch_create(C.byref(d), C.byref(self._h)), "scn_welch_create")
        self._npsd = [0] * capi.NUM_SLOTS
        self._keep = [None] * capi.NUM_SLOTS

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.scn_welch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def samples(self, n_psd):
        n = C.c_size_t()
        capi.check(self._L.scn_welch_samples(self._h, n_psd, C.byref(n)), "scn_welch_samples")
        return n.value

    def partition(self, n_psd):
        """(parts, column groups, segments per column group) of a submit of n_psd PSDs -- scn_welch_partition."""
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        capi.check(self._L.scn_welch_partition(self._h, n_psd, C.byref(a), C.byref(b), C.byref(c)), "scn_welch_partition")
        return a.value, b.value, c.value

    def host_buffer(self, slot):
        """pinned staging view (max_psd PSDs worth of samples): complex64 for float samples, bytes for the integer wire formats"""
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        capi.check(self._L.scn_welch_host_buffer(self._h, slot, C.byref(ptr), C.byref(nbytes)), "scn_welch_host_buffer")
        if self.kind != capi.KIND_FLOAT_COMPLEX:
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(nbytes.value,))
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(nbytes.value // 4,)).view(np.complex64)

    def submit(self, slot, n_psd):
        capi.check(self._L.scn_welch_submit(self._h, slot, n_psd), "scn_welch_submit")
        self._npsd[slot] = n_psd

    def submit_device(self, slot, d_samples, n_psd, d_psd_db=None, sync_producer=True):
        if sync_producer:
            import torch

            torch.cuda.current_stream(d_samples.device).synchronize()
        assert d_samples.is_contiguous() and d_samples.numel() * d_samples.element_size() >= self.samples(n_psd) * self.bytes_per_sample
        out = None if d_psd_db is None else C.c_void_p(d_psd_db.data_ptr())
        capi.check(self._L.scn_welch_submit_device(self._h, slot, C.c_void_p(d_samples.data_ptr()), n_psd, out),
                   "scn_welch_submit_device")
        self._npsd[slot] = n_psd
        self._keep[slot] = (d_samples, d_psd_db)

    def collect(self, slot, want_psd=True):
        nb = self._npsd[slot]
        out = np.empty((nb, self.n), np.float32) if want_psd else None
        capi.check(self._L.scn_welch_collect(self._h, slot, None if out is None else out.ctypes.data_as(C.c_void_p)),
                   "scn_welch_collect")
        self._keep[slot] = None
        return out
