"""Python face of the C ABI: a ``Device`` owns one snpgpu context; all arithmetic of the hot path runs in the
HIP kernels behind it.  Nothing here computes a result on the CPU — arrays are only marshalled.

Device-memory plumbing (allocation, streams, collectives) is torch's; host buffers are numpy.
"""
import ctypes as C
import os

import numpy as np

from . import _lib as L
from ._lib import CallerParams, SiteCounts, SnpGpuError  # noqa: F401  (re-exported)

COUNTS_DTYPE = np.dtype([
    ("raw_depth", "<u4"), ("good_depth", "<u4"), ("fwd_good_depth", "<u4"), ("rev_good_depth", "<u4"),
    ("n_symbols", "<u4"), ("ref_base", "u1"), ("cons_base", "u1"), ("filters", "u1"), ("status", "u1"),
    ("sym", "u1", (L.MAX_SYMS,)), ("total", "<u4", (L.MAX_SYMS,)), ("fwd", "<u4", (L.MAX_SYMS,)),
    ("rev", "<u4", (L.MAX_SYMS,))])
assert COUNTS_DTYPE.itemsize == 128
LINE_WIDE, LINE_SYMS = 0xFF, 3
LINE_DTYPE = np.dtype([("raw_depth", "<u4"), ("total", "<u2", (LINE_SYMS,)), ("fwd", "<u2", (LINE_SYMS,)), ("rev", "<u2", (LINE_SYMS,)),
                       ("sym", "u1", (LINE_SYMS,)), ("ref_base", "u1"), ("cons_base", "u1"), ("filters", "u1"), ("status", "u1"),
                       ("n_symbols", "u1"), ("site_flags", "u1"), ("reserved", "u1")])
assert LINE_DTYPE.itemsize == 32
VARSCAN_DTYPE = np.dtype([("line_off", "<u8"), ("sdp", "<u4"), ("dp", "<u4"), ("total", "<u4"), ("rdf", "<u4"), ("rdr", "<u4"),
                          ("ref_qual_sum", "<u4"), ("adf", "<u4"), ("adr", "<u4"), ("alt_qual_sum", "<u4"), ("ref_base", "u1"),
                          ("alt_base", "u1"), ("reserved", "u1", (2,))])
assert VARSCAN_DTYPE.itemsize == 48

_SCAN_CODES = {1: "line has fewer than 2 fields", 2: "position field is not an unsigned decimal integer",
               3: "non-ASCII byte in pileup"}


class PileupFormatError(ValueError):
    """The pileup text is malformed in a way that makes the reference raise (pileup.py:425-426, 224-237).
    ``reference_exception`` is the exception class the reference raises for the same input (ValueError or IndexError;
    None where the reference does not raise and this build refuses the input); the CLI re-raises as that class so that
    the error log names the same exception type."""

    def __init__(self, message, reference_exception=ValueError):
        ValueError.__init__(self, message)
        self.reference_exception = reference_exception


class PileupIOError(IOError):
    """A pileup file could not be opened or read (the reference's open() fails the same way, pileup.py:405/417)."""


# exception class of the reference per scan code (pileup.py:425 unpacking / :426 int()) and per site status
# (pileup.py:224-225 IndexError, :225 ValueError, :237 IndexError); None: the reference accepts the input
_SCAN_EXC = {1: ValueError, 2: ValueError, 3: None}


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_params(min_base_quality=0, min_cons_freq=0.6, min_cons_depth=1, min_cons_strand_depth=0,
                min_cons_strand_bias=0.0):
    return CallerParams(int(min_base_quality), int(min_cons_depth), int(min_cons_strand_depth), 0,
                        float(min_cons_freq), float(min_cons_strand_bias))


class SiteSet(object):
    """The (chrom, pos) set handed to pileup.Reader (pileup.py:396-403), resident on the device."""

    def __init__(self, device, keys, flags):
        """keys: sequence of (chrom: bytes, pos: int); flags: parallel sequence of SITE_* bit masks.
        Duplicated keys are merged (flags OR-ed).  ``index_of`` maps each input key to its device slot
        (-1 for positions that cannot occur in a pileup: negative or >= 2**32)."""
        self.device = device
        contigs = sorted({k[0] for k in keys})
        cid = {c: i for i, c in enumerate(contigs)}
        n = len(keys)
        packed = np.empty(n, dtype=np.uint64)
        ok = np.ones(n, dtype=bool)
        for i, (c, p) in enumerate(keys):
            if 0 <= p < (1 << 32):
                packed[i] = (cid[c] << 32) | p
            else:
                packed[i] = 0
                ok[i] = False
        fl = np.asarray(flags, dtype=np.uint8).reshape(n) if n else np.zeros(0, np.uint8)
        uniq, inv = np.unique(packed[ok], return_inverse=True)
        ufl = np.zeros(len(uniq), dtype=np.uint8)
        np.bitwise_or.at(ufl, inv, fl[ok])
        self.index_of = np.full(n, -1, dtype=np.int64)
        self.index_of[ok] = inv
        self.contigs = contigs
        self.keys = np.ascontiguousarray(uniq, dtype=np.uint64)
        self.flags = np.ascontiguousarray(ufl)
        self._create()

    @classmethod
    def from_arrays(cls, device, contigs, keys, flags, device_contigs=None):
        """contigs: bytewise sorted, unique list of contig names (bytes); keys: uint64 (contig index << 32 | pos), strictly
        increasing; flags: SITE_* per key.  No Python loop per key (``index_of`` is the identity and is not materialised).
        device_contigs: the names as the pileups the DEVICE reads spell them (utf8_names.escape_name of names that are not
        plain ASCII: same order, same equalities); the writers that take this site set keep printing ``contigs``."""
        self = cls.__new__(cls)
        self.device = device
        self.contigs = list(contigs)
        self.device_contigs = list(device_contigs) if device_contigs is not None else None
        self.keys = np.ascontiguousarray(keys, dtype=np.uint64)
        self.flags = np.ascontiguousarray(flags, dtype=np.uint8)
        self.index_of = None
        self._create()
        return self

    @classmethod
    def from_lists(cls, device, lists):
        """lists: [(contig names (str or bytes), contig index per record, position per record, SITE_* flag)] — the array
        triples of utils.read_vcf_site_arrays / read_snp_position_arrays.  Returns (SiteSet over the union, one array of
        slots per list: -1 for a position that cannot occur in a pileup (negative or >= 2**32))."""
        as_bytes = [[n if isinstance(n, bytes) else n.encode("utf-8") for n in names] for names, _, _, _ in lists]
        contigs = sorted({n for names in as_bytes for n in names})
        cid = {c: i for i, c in enumerate(contigs)}
        parts, oks, fl = [], [], []
        for names, (_, cidx, pos, flag) in zip(as_bytes, lists):
            pos = np.asarray(pos, dtype=np.int64)
            ok = (pos >= 0) & (pos < (1 << 32))
            lut = np.asarray([cid[c] for c in names] + [0], dtype=np.uint64)
            key = (lut[np.asarray(cidx, dtype=np.int64)] << np.uint64(32)) | (pos & 0xFFFFFFFF).astype(np.uint64) if len(pos) else np.zeros(0, np.uint64)
            parts.append(key[ok])
            oks.append(ok)
            fl.append(np.full(int(ok.sum()), flag, dtype=np.uint8))
        every = np.concatenate(parts) if parts else np.zeros(0, np.uint64)
        uniq, inv = np.unique(every, return_inverse=True)
        flags = np.zeros(len(uniq), dtype=np.uint8)
        if len(every):
            np.bitwise_or.at(flags, inv, np.concatenate(fl))
        slots, at = [], 0
        for ok in oks:
            k = int(ok.sum())
            sl = np.full(len(ok), -1, dtype=np.int64)
            sl[ok] = inv[at:at + k]
            at += k
            slots.append(sl)
        return cls.from_arrays(device, contigs, uniq, flags), slots

    def _create(self):
        device, contigs = self.device, self.contigs

        def table(names_list):
            names = b"".join(names_list)
            offs = np.zeros(len(names_list) + 1, dtype=np.uint32)
            if names_list:
                offs[1:] = np.cumsum([len(c) for c in names_list])
            return (np.frombuffer(names, dtype=np.uint8) if names else np.zeros(0, np.uint8)), offs
        self._names, self._offs = table(contigs)               # what the host writers print
        dev_names, dev_offs = table(self.device_contigs) if getattr(self, "device_contigs", None) is not None else (self._names, self._offs)
        h = C.c_void_p()
        device._check(device.lib.snpgpu_siteset_create(
            device.ctx, _ptr(dev_names), _ptr(dev_offs), len(contigs), _ptr(self.keys), _ptr(self.flags),
            len(self.keys), C.byref(h)))
        self.handle = h
        device._children.add(self)                         # the library object points at the context: it must go first

    def __len__(self):
        return len(self.keys)

    def key_tuples(self):
        return [(self.contigs[int(k) >> 32], int(k) & 0xFFFFFFFF) for k in self.keys]

    def close(self):
        if self.handle:
            if self.device.ctx:                            # (a context that is gone has taken its site sets with it)
                self.device.lib.snpgpu_siteset_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def spill_reference_field(spill, index):
    """The reference-base field (bytes) of spill record `index` whose ref_len is > 1: its first SPILL_REF bytes are in the
    record's ref[], the rest — for a field longer than that — fills the records that follow it (include/snpgpu.h)."""
    n = int(spill[index]["ref_len"])
    if n <= L.SPILL_REF:
        return bytes(spill[index]["ref"][:n])
    size = SPILL_DTYPE.itemsize
    more = (n - L.SPILL_REF + size - 1) // size
    if index + more >= len(spill):
        raise ValueError("spill record %d: a reference field of %d bytes needs %d more records, %d are there" % (index, n, more, len(spill) - index - 1))
    raw = np.ascontiguousarray(spill[index:index + 1 + more]).view(np.uint8).reshape(-1)
    return bytes(raw[size - L.SPILL_REF:size - L.SPILL_REF + n])


SPILL_DTYPE = np.dtype([("n", "<u4"), ("ref_len", "<u4"), ("depth64", "<i8"), ("sym", "u1", (L.SPILL_SYMS,)),
                        ("total", "<u4", (L.SPILL_SYMS,)), ("fwd", "<u4", (L.SPILL_SYMS,)), ("rev", "<u4", (L.SPILL_SYMS,)),
                        ("ref", "u1", (L.SPILL_REF,))])
assert SPILL_DTYPE.itemsize == C.sizeof(L.SymbolSpill)


def pack_line_records(counts, flags):
    """What k_compact_lines does, in numpy: (LINE_DTYPE records, indices of the wide lines, their COUNTS_DTYPE records) of per-line
    COUNTS_DTYPE records and site flags.  A line is packed when its record is well-formed (status <= ST_OK), has at most three symbols
    and no spill record, every count of the three fits 16 bits, nothing is counted past the symbols it says it has, and its depths are
    the sums of the symbols' counts.  The tests hold the device's packing against this; a host without a device can make the
    records snpgpu_format_line_rows takes."""
    n = len(counts)
    tot, fwd, rev = (counts[k].astype(np.int64) for k in ("total", "fwd", "rev"))
    nsym = counts["n_symbols"].astype(np.int64)
    ok = (counts["status"] <= L.ST_OK) & (nsym <= LINE_SYMS)
    for a in (tot, fwd, rev):
        ok &= (a[:, :LINE_SYMS] < 65536).all(axis=1)
    ok &= (counts["good_depth"] == tot[:, :LINE_SYMS].sum(axis=1)) & (counts["fwd_good_depth"] == fwd[:, :LINE_SYMS].sum(axis=1)) & \
          (counts["rev_good_depth"] == rev[:, :LINE_SYMS].sum(axis=1))
    for k in range(LINE_SYMS):
        ok &= (nsym > k) | (tot[:, k] == 0)
    recs = np.zeros(n, dtype=LINE_DTYPE)
    recs["raw_depth"] = counts["raw_depth"]
    for name in ("total", "fwd", "rev"):
        recs[name] = (counts[name][:, :LINE_SYMS] & 0xFFFF).astype(np.uint16)
    recs["sym"] = counts["sym"][:, :LINE_SYMS]
    for name in ("ref_base", "cons_base", "filters", "status"):
        recs[name] = counts[name]
    recs["n_symbols"] = np.where(ok, nsym & 0xFF, LINE_WIDE).astype(np.uint8)
    recs["site_flags"] = flags
    wide_index = np.nonzero(~ok)[0].astype(np.uint32)
    return recs, wide_index, counts[wide_index].copy()


def expand_line_records(recs, wide_index, wide):
    """(site flags, COUNTS_DTYPE records) of every line from the 32-byte records and the wide lines' full ones — what call_all_lines
    returns, rebuilt on the host (tests, and callers that want the per-line numbers rather than rows)."""
    n = len(recs)
    out = np.zeros(n, dtype=COUNTS_DTYPE)
    for name in ("raw_depth", "ref_base", "cons_base", "filters", "status"):
        out[name] = recs[name]
    out["n_symbols"] = recs["n_symbols"]
    for name, total in (("total", "good_depth"), ("fwd", "fwd_good_depth"), ("rev", "rev_good_depth")):
        out[name][:, :LINE_SYMS] = recs[name]
        out[total] = recs[name].astype(np.uint32).sum(axis=1)
    out["sym"][:, :LINE_SYMS] = recs["sym"]
    if len(wide_index):
        if not (recs["n_symbols"][wide_index] == LINE_WIDE).all() or int((recs["n_symbols"] == LINE_WIDE).sum()) != len(wide_index):
            raise ValueError("the wide list does not match the records marked wide")
        out[wide_index] = wide
    return recs["site_flags"].copy(), out


def symbol_count(counts):
    """Distinct symbols of the records (the low byte of n_symbols; the rest points at a spill record)."""
    return counts["n_symbols"] & 0xFF


class SpillOverflow(RuntimeError):
    """A call asked for more spill records than the context's arena held; repeating the call finds a larger one."""


def _again_on_spill_overflow(method):
    """Repeat a device call whose positions needed more spill records than the arena had (it grows between the attempts)."""
    import functools

    @functools.wraps(method)
    def wrapped(self, *args, **kwargs):
        for _ in range(4):
            try:
                return method(self, *args, **kwargs)
            except SpillOverflow:
                continue
        return method(self, *args, **kwargs)
    return wrapped


class ConsensusResult(object):
    __slots__ = ("bases", "filters", "counts", "status", "n_lines", "n_matched", "depth_sum", "line_offsets", "spill")

    def __init__(self, bases, filters, counts, status, spill=None):
        self.bases, self.filters, self.counts, self.status = bases, filters, counts, status
        self.n_lines, self.n_matched, self.depth_sum = int(status[1]), int(status[2]), int(status[3])
        self.line_offsets = None
        self.spill = spill                                   # SPILL_DTYPE records of the call, or None (no position has > 8 symbols)


class Device(object):
    def __init__(self, index=None):
        self.lib = L.load()
        if index is None:
            index = int(os.environ.get("SNPGPU_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        self.index = index
        h = C.c_void_p()
        rc = self.lib.snpgpu_ctx_create(index, C.byref(h))
        if rc != 0:
            raise SnpGpuError(rc, "no usable gfx950 device %d (the HIP path has no CPU fallback)" % index)
        self.ctx = h
        import weakref
        self._children = weakref.WeakSet()                  # site sets and pileup stores of this context

    def read_symbol_spill(self, counts=None):
        """The spill records of the context's last call that produced per-site records (positions with more than 8 distinct
        symbols, a reference-base field of several bytes, a depth outside 32 bits); with `counts`: None unless one of those
        records points at one.  Raises SpillOverflow when the call asked for more records than the context's arena held: the
        caller repeats the call (the arena grows at the next one)."""
        if counts is not None and not (counts["n_symbols"] >> 8).any():
            return None
        n = C.c_uint32()
        self._check(self.lib.snpgpu_symbol_spill_read(self.ctx, None, 0, C.byref(n)))
        if n.value > int(self.lib.snpgpu_symbol_spill_capacity(self.ctx)):
            raise SpillOverflow(n.value)
        out = np.zeros(max(n.value, 1), dtype=SPILL_DTYPE)
        self._check(self.lib.snpgpu_symbol_spill_read(self.ctx, _ptr(out), n.value, C.byref(n)))
        return out[:n.value]

    # ---- plumbing ------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            msg = self.lib.snpgpu_last_error(self.ctx)
            raise SnpGpuError(rc, msg.decode("utf-8", "replace") if msg else "")

    def close(self):
        if self.ctx:
            for child in list(self._children):              # they hold pointers into the context: closed before it, whoever forgot
                child.close()
            self.lib.snpgpu_ctx_destroy(self.ctx)
            self.ctx = None

    def use_torch_stream(self):
        """Enqueue on torch's current stream so torch.cuda events / collectives order with our kernels."""
        if L.loaded_without_torch:
            raise RuntimeError("libsnpgpu.so was loaded with the system HIP runtime (console-script mode); import torch "
                               "before snp_pipeline_amd to share streams and device pointers with it")
        import torch
        self._check(self.lib.snpgpu_ctx_set_stream(self.ctx, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def sync(self):
        self._check(self.lib.snpgpu_ctx_sync(self.ctx))

    def kernel_timing(self, enable):
        self._check(self.lib.snpgpu_ctx_kernel_timing(self.ctx, 1 if enable else 0))

    def kernel_time_ms(self, kernel):
        """(total ms, launches) of kernel 0=scan 1=call 2=distance 3=site calling (all of a file's launches) since the last query."""
        ms, n = C.c_float(), C.c_uint32()
        self._check(self.lib.snpgpu_ctx_kernel_time_ms(self.ctx, kernel, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def timer_start(self):
        self._check(self.lib.snpgpu_timer_start(self.ctx))

    def timer_stop_ms(self):
        ms = C.c_float()
        self._check(self.lib.snpgpu_timer_stop_ms(self.ctx, C.byref(ms)))
        return ms.value

    # ---- call_consensus ------------------------------------------------------------------------
    def siteset(self, keys, flags):
        return SiteSet(self, keys, flags)

    def siteset_from_lists(self, lists):
        return SiteSet.from_lists(self, lists)

    @staticmethod
    def scan_error(status):
        """(exception, byte offset of its line) for the first line whose chrom / position columns the reader cannot take, or
        (None, None)."""
        w0 = int(status[0])
        if w0 == 0xFFFFFFFFFFFFFFFF:
            return None, None
        code, off = w0 & 0xFF, (w0 >> 8) - 1
        err = PileupFormatError("pileup: %s at byte offset %d" % (_SCAN_CODES.get(code, "malformed line"), off),
                                _SCAN_EXC.get(code, ValueError))
        err.scan_code = code
        return err, off

    @staticmethod
    def raise_scan_status(status):
        err, _ = Device.scan_error(status)
        if err is not None:
            raise err

    def raise_first_error(self, status, res=None, check=True, wanted=None):
        """The reference reads a pileup top to bottom and ends at the FIRST line it cannot take: a line whose first two columns
        the reader cannot split or convert (pileup.py:425-426, any line), or a line at a listed position that Record cannot be
        built from (pileup.py:224-237).  Both kinds are found here by different kernels, so the earlier one in the file is
        picked by byte offset (per-site offsets: ``res.line_offsets``; without them a reader-level error goes first)."""
        scan, s_off = self.scan_error(status)
        site, t_off = self.site_error(res, wanted) if (check and res is not None) else (None, None)
        if scan is not None and (site is None or t_off is None or s_off <= t_off or scan.scan_code == 3):
            raise scan
        if site is not None:
            raise site

    @_again_on_spill_overflow
    def call_consensus(self, siteset, pileup, params, want_counts=False, want_depth_sum=False, check=True):
        """pileup: bytes-like (host).  Returns ConsensusResult over siteset.keys order.  check=False: do not raise for
        the per-site failures the reference raises on (malformed line at a listed position); the scan-level errors
        (malformed chrom / position column anywhere in the file) always raise.  (A buffer that repeats a listed position: the result
        knows the last line of the position only; the commands go through files and raise_file_errors, which looks at every line.)"""
        buf = np.frombuffer(pileup, dtype=np.uint8)
        n = len(siteset)
        bases = np.empty(n, dtype=np.uint8)
        filters = np.empty(n, dtype=np.uint8)
        counts = np.zeros(n, dtype=COUNTS_DTYPE) if want_counts else None
        status = np.zeros(L.SCAN_STATUS_WORDS, dtype=np.uint64)
        rc = self.lib.snpgpu_call_consensus(self.ctx, siteset.handle, _ptr(buf) if len(buf) else None, len(buf),
                                            C.byref(params), _ptr(bases), _ptr(filters), _ptr(counts), _ptr(status),
                                            1 if want_depth_sum else 0)
        scan_failed = rc in (L.E_PILEUP, L.E_UNSUPPORTED) and int(status[0]) != 0xFFFFFFFFFFFFFFFF
        if not scan_failed:
            self._check(rc)
        res = ConsensusResult(bases, filters, counts, status, self.read_symbol_spill(counts) if want_counts and not scan_failed else None)
        if scan_failed or (check and self.site_error(res)[0] is not None):
            res.line_offsets = self.line_offsets(siteset)       # which malformed line comes first in the file
            self.raise_first_error(status, res, check)
        return res

    @staticmethod
    def site_error(res, wanted=None):
        """(exception, byte offset of its line or None) for the first line at a listed position that makes the reference raise
        while building a Record — first in file order when the result carries line offsets — or (None, None).  wanted: a mask
        over the site set's slots, the positions THIS file is asked about (a batch shares one set between files with different
        exclude lists; the reference builds no Record at a position that is on neither of a sample's lists)."""
        if res.counts is not None:
            codes = res.counts["status"]
            is_bad = codes > L.ST_OK
        else:
            codes = res.filters & 0x7F
            is_bad = (res.filters & 0x80) != 0
        if wanted is not None:
            is_bad = is_bad & np.asarray(wanted, dtype=bool)
        bad = np.nonzero(is_bad)[0]
        if not len(bad):
            return None, None
        k, off = int(bad[0]), None
        if res.line_offsets is not None and len(res.line_offsets) >= len(codes):
            lo = np.asarray(res.line_offsets)[bad]
            j = int(np.argmin(lo))
            k, off = int(bad[j]), int(lo[j]) - 1
        what, exc = {L.ST_SHORT_LINE: ("line has fewer than 4 fields", IndexError),
                     L.ST_BAD_DEPTH: ("depth field is not an unsigned decimal integer", ValueError),
                     L.ST_NO_QUALS: ("depth > 0 but no quality field", IndexError),
                     }.get(int(codes[k]), ("malformed line", ValueError))
        return PileupFormatError("pileup line for site #%d: %s" % (k, what), exc), off

    @staticmethod
    def raise_site_status(res):
        """Per-line failures that make the reference raise while building a Record."""
        err, _ = Device.site_error(res)
        if err is not None:
            raise err

    @_again_on_spill_overflow
    def call_consensus_files(self, siteset, paths, params, want_counts=False, want_line_offsets=False,
                             want_depth_sum=False, chunk_bytes=0, n_readers=0, n_staging=0, n_slots=0, exclude=None):
        """Streamed ingestion of pileup FILES (snpgpu_call_consensus_files): reader threads -> pinned staging -> copy
        stream -> scan of the tiles that have landed; no file is resident in host memory.
        Returns (results, rcs, stats): one ConsensusResult per path (``.line_offsets`` set when asked for), the per-file
        return codes (0, E_IO, E_PILEUP, E_UNSUPPORTED) and the StreamStats of the call.  Nothing is raised per file:
        use ``raise_file_status``.  exclude: per path, the site-set slots of that file's own exclude list (arrays; slots < 0
        are ignored) — the file is called with SITE_EXCLUDED on them on top of the set's flags."""
        n_files, n = len(paths), len(siteset)
        excl_off = excl_slots = None
        if exclude is not None:
            lists = [np.asarray(e, dtype=np.int64) for e in exclude]
            lists = [e[e >= 0].astype(np.uint32) for e in lists]
            excl_off = np.zeros(n_files + 1, dtype=np.uint32)
            if lists:
                np.cumsum([len(e) for e in lists], out=excl_off[1:])
            excl_slots = np.ascontiguousarray(np.concatenate(lists) if lists else np.zeros(0, np.uint32), dtype=np.uint32)
            if len(excl_slots) == 0:
                excl_slots = np.zeros(1, dtype=np.uint32)
        enc = [os.fsencode(p) for p in paths]
        arr = (C.c_char_p * max(n_files, 1))(*enc)
        bases = np.empty((n_files, n), dtype=np.uint8)
        filters = np.empty((n_files, n), dtype=np.uint8)
        counts = np.zeros((n_files, n), dtype=COUNTS_DTYPE) if want_counts else None
        line_off = np.zeros((n_files, n), dtype=np.uint64) if want_line_offsets else None
        status = np.zeros((n_files, L.SCAN_STATUS_WORDS), dtype=np.uint64)
        rcs = np.zeros(max(n_files, 1), dtype=np.int32)
        opts = L.StreamOpts(int(chunk_bytes), int(n_staging), int(n_readers), int(n_slots), 1 if want_depth_sum else 0)
        stats = L.StreamStats()
        self._check(self.lib.snpgpu_call_consensus_files(
            self.ctx, siteset.handle, arr, n_files, C.byref(params), _ptr(excl_off), _ptr(excl_slots), _ptr(bases), _ptr(filters),
            _ptr(counts), _ptr(line_off), _ptr(status), _ptr(rcs), C.byref(opts), C.byref(stats)))
        results = []
        spill = self.read_symbol_spill(counts) if want_counts else None      # one spill per call: shared by its files
        for f in range(n_files):
            r = ConsensusResult(bases[f], filters[f], counts[f] if want_counts else None, status[f], spill)
            r.line_offsets = line_off[f] if want_line_offsets else None
            results.append(r)
        return results, rcs[:n_files], stats

    @_again_on_spill_overflow
    def call_all_lines(self, siteset, path, params, capacity=0, check=True, listed_only=False):
        """call_consensus --vcfAllPos: a record for EVERY line of the pileup file, in file order.
        Returns (line_offsets + 1, line site flags, counts records).  Raises like the reference for malformed lines
        (check=False: only for malformed chrom / position columns; the caller looks at the records it uses; listed_only: the
        Record-level failures that count are those of lines at positions of the site set — the reader with a position set builds no
        Record from the others, pileup.py:427-429 — and the first line the reference cannot take, of either kind, decides)."""
        n_lines = C.c_uint64()
        status = np.zeros(L.SCAN_STATUS_WORDS, dtype=np.uint64)
        while True:
            cap = int(capacity)
            off = np.zeros(max(cap, 1), dtype=np.uint64)
            flags = np.zeros(max(cap, 1), dtype=np.uint8)
            counts = np.zeros(max(cap, 1), dtype=COUNTS_DTYPE)
            rc = self.lib.snpgpu_call_all_lines_file(self.ctx, siteset.handle, os.fsencode(path), C.byref(params), cap,
                                                     C.byref(n_lines), _ptr(off), _ptr(flags), _ptr(counts), _ptr(status))
            if rc == L.E_IO:
                raise PileupIOError("cannot open or read the pileup file %s" % path)
            if rc in (L.E_PILEUP, L.E_UNSUPPORTED) and int(status[0]) != 0xFFFFFFFFFFFFFFFF:
                if check and 0 < n_lines.value <= cap:          # the records are there: a Record-level failure earlier in the file goes first
                    keep = np.nonzero(flags[:n_lines.value])[0] if listed_only else slice(0, n_lines.value)
                    res = ConsensusResult(None, None, counts[:n_lines.value][keep], status)
                    res.line_offsets = off[:n_lines.value][keep]
                    self.raise_first_error(status, res, True)
                self.raise_scan_status(status)
            self._check(rc)
            if n_lines.value <= cap:
                break
            capacity = n_lines.value
        n = n_lines.value
        if check:
            self.raise_site_status(ConsensusResult(None, None, counts[:n][flags[:n] != 0] if listed_only else counts[:n], status))
        self.last_spill = self.read_symbol_spill(counts[:n])     # (for the rows of these records: vcf_writer.write_all_positions_vcf)
        return off[:n], flags[:n], counts[:n]

    def call_all_lines_compact(self, siteset, path, params, capacity=0, wide_capacity=4096):
        """call_all_lines with 32-byte records (snpgpu_call_all_lines_compact_file): a fifth of the bytes over the host link.
        Returns (line_offsets + 1, LINE_DTYPE records — site_flags inside —, indices of the wide lines (ascending), their COUNTS_DTYPE
        records); ``expand_line_records`` turns that into what call_all_lines returns.  Raises for a malformed chrom / position
        column only; the Record-level failures are in the wide records' status (the caller looks at the lines it uses)."""
        n_lines, n_wide = C.c_uint64(), C.c_uint32()
        status = np.zeros(L.SCAN_STATUS_WORDS, dtype=np.uint64)
        cap, wcap = int(capacity), int(wide_capacity)
        while True:
            off = np.empty(max(cap, 1), dtype=np.uint64)
            recs = np.empty(max(cap, 1), dtype=LINE_DTYPE)
            widx = np.empty(max(wcap, 1), dtype=np.uint32)
            wide = np.empty(max(wcap, 1), dtype=COUNTS_DTYPE)
            rc = self.lib.snpgpu_call_all_lines_compact_file(self.ctx, siteset.handle, os.fsencode(path), C.byref(params), cap, C.byref(n_lines), _ptr(off),
                                                             _ptr(recs), wcap, C.byref(n_wide), _ptr(widx), _ptr(wide), _ptr(status))
            if rc == L.E_IO:
                raise PileupIOError("cannot open or read the pileup file %s" % path)
            if rc in (L.E_PILEUP, L.E_UNSUPPORTED) and int(status[0]) != 0xFFFFFFFFFFFFFFFF:
                self.raise_scan_status(status)
            self._check(rc)
            if n_lines.value <= cap and n_wide.value <= wcap:
                break
            cap, wcap = max(cap, n_lines.value), max(wcap, n_wide.value)
        n, w = n_lines.value, n_wide.value
        if (wide[:w]["n_symbols"] >> 8).any():
            spill = self.read_symbol_spill()
            if ((wide[:w]["n_symbols"] >> 8) == 0xFFFFFF).any():
                raise SpillOverflow(len(spill))
            self.last_spill = spill
        else:
            self.last_spill = None
        return off[:n], recs[:n], widx[:w], wide[:w]

    def write_all_positions_vcf(self, siteset, pileup_path, params, vcf_path, header_text, filter_names, preserve_ref_case, failed_snp_gt,
                                only_listed=False, check=True):
        """call_consensus --vcfAllPos from file to file (snpgpu_write_all_positions_vcf): the pileup through the device, a row per line
        formatted by the library's host threads straight into vcf_path behind header_text.  Returns (lines, rows written).  Raises what
        the reference raises for the first line it cannot take (nothing is written then)."""
        n_lines, n_rows, bad_line, bad_off = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        bad = np.zeros(1, dtype=COUNTS_DTYPE)
        status = np.zeros(L.SCAN_STATUS_WORDS, dtype=np.uint64)
        fn = (C.c_char_p * 6)(*[n.encode("ascii") for n in filter_names])
        rc = self.lib.snpgpu_write_all_positions_vcf(self.ctx, siteset.handle, os.fsencode(pileup_path), C.byref(params), os.fsencode(vcf_path),
                                                     header_text.encode("utf-8"), fn, 1 if preserve_ref_case else 0, failed_snp_gt.encode("ascii"),
                                                     1 if only_listed else 0, 1 if check else 0, C.byref(n_lines), C.byref(n_rows), C.byref(bad_line),
                                                     C.byref(bad_off), _ptr(bad), _ptr(status))
        if rc == L.E_IO and not os.access(pileup_path, os.R_OK):
            raise PileupIOError("cannot open or read the pileup file %s" % pileup_path)
        if rc == L.E_IO:
            # the output could not be written: end as `open(vcf_path, "w")` of the reference's writer does (vcf_writer.py:92-99) —
            # IOError / PermissionError / FileNotFoundError with the path in it
            with open(vcf_path, "ab"):
                pass
            raise IOError("cannot write %s" % vcf_path)
        has_bad = bad_line.value != 0xFFFFFFFFFFFFFFFF
        if rc in (L.E_PILEUP, L.E_UNSUPPORTED) and int(status[0]) != 0xFFFFFFFFFFFFFFFF:
            if check and has_bad:                                 # a Record-level failure earlier in the file goes first
                res = ConsensusResult(None, None, bad, status)
                res.line_offsets = np.array([bad_off.value], dtype=np.uint64)
                self.raise_first_error(status, res, True)
            self.raise_scan_status(status)
        self._check(rc)
        if has_bad:
            res = ConsensusResult(None, None, bad, status)
            res.line_offsets = np.array([bad_off.value], dtype=np.uint64)
            self.raise_site_status(res)
        return int(n_lines.value), int(n_rows.value)

    def varscan_file(self, path, params, capacity=65536):
        """Phase-1 site calling over a pileup file: numpy records (VARSCAN_DTYPE) of every (line, allele) that passes the
        count tests of `VarScan mpileup2snp`, in file order.  Raises PileupFormatError for a malformed line."""
        n = C.c_uint32()
        status = np.zeros(2, dtype=np.uint64)
        while True:
            cap = int(capacity)
            sites = np.zeros(max(cap, 1), dtype=VARSCAN_DTYPE)
            rc = self.lib.snpgpu_varscan_file(self.ctx, os.fsencode(path), C.byref(params), cap, _ptr(sites), C.byref(n), _ptr(status))
            if rc == L.E_IO:
                raise PileupIOError("cannot open or read the pileup file %s" % path)
            if rc == L.E_PILEUP:
                raise PileupFormatError("Invalid format for pileup at byte %d of %s" % (int(status[0]), path), ValueError)
            self._check(rc)
            if n.value <= cap:
                return sites[:n.value], int(status[1])
            capacity = max(n.value, 2 * cap)

    def varscan_files(self, paths, params, capacity=32768):
        """varscan_file for many pileups in one streamed call (a file's kernels run while the next file is read and copied).
        Returns [(records, lines)] in input order; a file that failed carries an exception object in place of its records
        (PileupIOError / PileupFormatError), so that the caller can treat it as a sample error and go on."""
        n = len(paths)
        if n == 0:
            return []
        if n > 64:                                              # the shared record array is n x capacity: keep it small
            out = []
            for k in range(0, n, 64):
                out.extend(self.varscan_files(paths[k:k + 64], params, capacity))
            return out
        arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
        sites = np.empty((n, max(int(capacity), 1)), dtype=VARSCAN_DTYPE)       # only the rows the library reports are read
        counts = np.zeros(n, dtype=np.uint32)
        status = np.zeros((n, 2), dtype=np.uint64)
        rcs = np.zeros(n, dtype=np.int32)
        self._check(self.lib.snpgpu_varscan_files(self.ctx, arr, n, C.byref(params), int(capacity), _ptr(sites), _ptr(counts), _ptr(status), _ptr(rcs)))
        out = []
        for i, path in enumerate(paths):
            if rcs[i] == L.E_IO:
                out.append((PileupIOError("cannot open or read the pileup file %s" % path), 0))
            elif rcs[i] == L.E_PILEUP:
                out.append((PileupFormatError("Invalid format for pileup at byte %d of %s" % (int(status[i, 0]), path), ValueError), int(status[i, 1])))
            elif counts[i] > capacity:
                out.append(self.varscan_file(path, params, capacity=int(counts[i])))           # more records than the shared array holds
            else:
                out.append((sites[i, :counts[i]].copy(), int(status[i, 1])))
        return out

    def raise_file_status(self, path, rc, res, check=True, wanted=None):
        """Raise for one file of call_consensus_files the way call_consensus does for its single pileup (wanted: see site_error)."""
        if rc == L.E_IO:
            raise PileupIOError("cannot open or read the pileup file %s" % path)
        if rc in (L.E_PILEUP, L.E_UNSUPPORTED):
            self.raise_first_error(res.status, res, check, wanted)
        if rc != 0:
            raise SnpGpuError(int(rc), "pileup %s" % path)
        if check:
            err, _ = self.site_error(res, wanted)
            if err is not None:
                raise err

    def raise_file_errors(self, siteset, path, params, rc, res, wanted=None):
        """raise_file_status and check_repeated_positions as ONE decision, which is what the reference's single pass is: it ends at
        the first line it cannot take.  Where no listed position comes twice the per-site result knows every matching line and
        raise_file_status decides; where one does, it knows only the LAST line of that position — an earlier malformed line of the same
        position is hidden behind a later one, which may lie behind a reader-level error (found by the fuzz campaign of round 6: a
        position three times, two of its lines malformed, a bad position column between them) — so every line at a listed position is
        looked at in file order by the all-lines pass, beside the reader-level error."""
        repeats = res.line_offsets is not None and res.n_matched > int(np.count_nonzero(res.line_offsets))
        if not repeats or rc == L.E_IO:
            self.raise_file_status(path, rc, res, True, wanted)
            return
        self.call_all_lines(siteset, path, params, capacity=res.n_lines, check=True, listed_only=True)
        if rc != 0 and rc not in (L.E_PILEUP, L.E_UNSUPPORTED):
            raise SnpGpuError(int(rc), "pileup %s" % path)

    def check_repeated_positions(self, siteset, path, params, res):
        """A pileup that repeats a listed position: the per-site result knows the LAST line of a position, the reference builds a
        Record from every one of them in file order (call_consensus.py:161-176) and ends at the first it cannot build.  When the
        file has more matching lines than positions with a line, the all-lines pass looks at each (a second read of the file;
        a sorted pileup never takes it).  Lines count as listed by the flags of `siteset`."""
        if res.line_offsets is None or res.n_matched <= int(np.count_nonzero(res.line_offsets)):
            return
        _, flags, counts = self.call_all_lines(siteset, path, params, capacity=res.n_lines, check=False)
        bad = np.nonzero((flags != 0) & (counts["status"] > L.ST_OK))[0]
        if len(bad):
            err, _ = self.site_error(ConsensusResult(None, None, counts[bad[:1]], res.status))
            raise err

    def line_offsets(self, siteset):
        """1 + byte offset of the pileup line used for each site by the last call_consensus on this set (0 = none)."""
        out = np.zeros(len(siteset), dtype=np.uint64)
        self._check(self.lib.snpgpu_siteset_line_offsets(self.ctx, siteset.handle, _ptr(out)))
        return out

    def call_consensus_dev(self, siteset, d_pileup_ptr, nbytes, params, d_bases, d_filters, d_status, d_counts=None,
                           want_depth_sum=False):
        """All arguments are device pointers (ints); asynchronous on the context's stream."""
        self._check(self.lib.snpgpu_call_consensus_dev(
            self.ctx, siteset.handle, C.c_void_p(d_pileup_ptr), nbytes, C.byref(params), C.c_void_p(d_bases),
            C.c_void_p(d_filters), C.c_void_p(d_counts) if d_counts else None, C.c_void_p(d_status),
            1 if want_depth_sum else 0))

    def call_consensus_batch_dev(self, siteset, d_pileups_ptr, offsets, params, d_bases, d_filters, d_status, sizes=None):
        """One scan launch + one call launch for the whole batch.  Sample i is bytes [offsets[i], offsets[i] + sizes[i])
        of the device buffer; without `sizes` the samples are packed and offsets has one entry more than samples."""
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        if sizes is None:
            n, sz = len(offs) - 1, None
        else:
            sz = np.ascontiguousarray(sizes, dtype=np.uint64)
            n = len(sz)
            if len(offs) < n:
                raise ValueError("offsets shorter than sizes")
        self._check(self.lib.snpgpu_call_consensus_batch_dev(
            self.ctx, siteset.handle, C.c_void_p(d_pileups_ptr), _ptr(offs), _ptr(sz) if sz is not None else None, n,
            C.byref(params), C.c_void_p(d_bases), C.c_void_p(d_filters), C.c_void_p(d_status)))

    def call_consensus_many_dev(self, siteset, d_ptrs, sizes, params, d_bases, d_filters, d_status, d_counts=0, d_line_off=0,
                                d_site_flags=0, want_depth_sum=False):
        """Samples anywhere in device memory (the resident pileups of the pipeline): d_ptrs / sizes per sample; the d_*
        arguments are device pointers (ints) of [n][n_sites] outputs; asynchronous."""
        n = len(sizes)
        ptrs = (C.c_void_p * max(n, 1))(*[C.c_void_p(int(p)) for p in d_ptrs])
        sz = np.ascontiguousarray(sizes, dtype=np.uint64)
        opt = lambda v: C.c_void_p(v) if v else None     # noqa: E731
        self._check(self.lib.snpgpu_call_consensus_many_dev(
            self.ctx, siteset.handle, ptrs, _ptr(sz), n, C.byref(params), opt(d_site_flags), C.c_void_p(d_bases), C.c_void_p(d_filters),
            opt(d_counts), opt(d_line_off), C.c_void_p(d_status), 1 if want_depth_sum else 0))

    def region_flow_dev(self, d_base, d_filters, d_line_off, n_samples, n_sites, d_cols, d_col_of, n_cols, d_excl_off, d_excl_slots,
                        d_out_base, d_out_filters, d_err):
        """The preserved flow from the result of the full-list call (snpgpu_region_flow_dev); device pointers; asynchronous."""
        opt = lambda v: C.c_void_p(v) if v else None     # noqa: E731
        self._check(self.lib.snpgpu_region_flow_dev(self.ctx, opt(d_base), opt(d_filters), opt(d_line_off), n_samples, n_sites, opt(d_cols),
                                                    opt(d_col_of), n_cols, opt(d_excl_off), opt(d_excl_slots), opt(d_out_base),
                                                    opt(d_out_filters), C.c_void_p(d_err)))

    def rows_copy_dev(self, d_src, src_stride, d_dst, dst_stride, n_rows, row_bytes, d_src_index=0, d_dst_index=0):
        """dst row dst_index[r] <- src row src_index[r] (snpgpu_rows_copy_dev); device pointers, strides in bytes, index lists of uint32 or 0
        for "row r itself"; asynchronous on the context's stream."""
        opt = lambda v: C.c_void_p(v) if v else None     # noqa: E731
        self._check(self.lib.snpgpu_rows_copy_dev(self.ctx, opt(d_src), int(src_stride), opt(d_src_index), opt(d_dst), int(dst_stride), opt(d_dst_index),
                                                  int(n_rows), int(row_bytes)))

    def varscan_dev(self, d_ptr, nbytes, params, capacity=65536):
        """varscan_file for a pileup that is in device memory."""
        n = C.c_uint32()
        status = np.zeros(2, dtype=np.uint64)
        while True:
            cap = int(capacity)
            sites = np.zeros(max(cap, 1), dtype=VARSCAN_DTYPE)
            rc = self.lib.snpgpu_varscan_dev(self.ctx, C.c_void_p(int(d_ptr)), int(nbytes), C.byref(params), cap, _ptr(sites), C.byref(n), _ptr(status))
            if rc == L.E_PILEUP:
                raise PileupFormatError("Invalid format for pileup at byte %d" % int(status[0]), ValueError)
            self._check(rc)
            if n.value <= cap:
                return sites[:n.value], int(status[1])
            capacity = max(n.value, 2 * cap)

    def varscan_batch_dev(self, d_ptrs, sizes, params, capacity=32768):
        """varscan_dev for many pileups in device memory with ONE scan launch (snpgpu_varscan_batch_dev).  Returns one
        (records, n_lines) pair, or a PileupFormatError, per pileup; a pileup with more records than `capacity` is repeated alone."""
        n = len(d_ptrs)
        if n == 0:
            return []
        ptrs = (C.c_void_p * n)(*[C.c_void_p(int(p)) if int(s) else None for p, s in zip(d_ptrs, sizes)])
        nb = np.asarray([int(s) for s in sizes], dtype=np.uint64)
        sites = np.zeros((n, max(int(capacity), 1)), dtype=VARSCAN_DTYPE)
        counts = np.zeros(n, dtype=np.uint32)
        status = np.zeros((n, 2), dtype=np.uint64)
        rcs = np.zeros(n, dtype=np.int32)
        self._check(self.lib.snpgpu_varscan_batch_dev(self.ctx, ptrs, _ptr(nb), n, C.byref(params), int(capacity), _ptr(sites), _ptr(counts), _ptr(status), _ptr(rcs)))
        out = []
        for i in range(n):
            if rcs[i] == L.E_PILEUP:
                out.append(PileupFormatError("Invalid format for pileup at byte %d" % int(status[i, 0]), ValueError))
            elif rcs[i] != 0:
                out.append(RuntimeError("site calling failed for pileup %d (code %d)" % (i, int(rcs[i]))))
            elif counts[i] > capacity:
                out.append(self.varscan_dev(d_ptrs[i], sizes[i], params, capacity=int(counts[i])))
            else:
                out.append((sites[i, :counts[i]].copy(), int(status[i, 1])))
        return out

    # ---- the exchange steps of the sharded path: RCCL behind the C ABI (csrc/comm.hip) --------------------------------------
    def comm_available(self):
        return bool(self.lib.snpgpu_comm_available())

    def comm_version(self):
        v = C.c_int()
        return v.value if self.lib.snpgpu_comm_version(C.byref(v)) == 0 else None

    def comm_unique_id(self):
        """128 bytes that name a communicator: rank 0 makes them and hands them to the other ranks."""
        buf = C.create_string_buffer(128)
        rc = self.lib.snpgpu_comm_unique_id(buf)
        if rc != 0:
            raise RuntimeError("RCCL is not available (snpgpu_comm_unique_id: %d)" % rc)
        return buf.raw

    def comm_init(self, rank, nranks, unique_id):
        self._check(self.lib.snpgpu_comm_init(self.ctx, int(rank), int(nranks), C.c_char_p(bytes(unique_id))))
        self.comm_rank, self.comm_nranks = int(rank), int(nranks)

    def comm_destroy(self):
        if self.ctx:
            self.lib.snpgpu_comm_destroy(self.ctx)
        self.comm_rank, self.comm_nranks = 0, 0

    def comm_abort(self):
        if self.ctx:
            self.lib.snpgpu_comm_abort(self.ctx)
        self.comm_rank, self.comm_nranks = 0, 0

    def comm_info(self):
        r, n, c = C.c_int(), C.c_int(), C.c_int()
        self._check(self.lib.snpgpu_comm_info(self.ctx, C.byref(r), C.byref(n), C.byref(c)))
        return {"rank": r.value, "nranks": n.value, "rccl_comm_count": c.value, "rccl_version": self.comm_version()}

    def allgather_dev(self, d_send, d_recv, bytes_per_rank):
        self._check(self.lib.snpgpu_allgather(self.ctx, C.c_void_p(int(d_send)), C.c_void_p(int(d_recv)), int(bytes_per_rank)))

    def allgatherv_dev(self, d_send, d_recv, nbytes, offsets):
        b = np.ascontiguousarray(nbytes, dtype=np.uint64)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._check(self.lib.snpgpu_allgatherv(self.ctx, C.c_void_p(int(d_send)) if int(d_send) else None, C.c_void_p(int(d_recv)), _ptr(b), _ptr(o)))

    def alltoallv_dev(self, d_send, send_bytes, d_recv, recv_bytes):
        sb = np.ascontiguousarray(send_bytes, dtype=np.uint64)
        rb = np.ascontiguousarray(recv_bytes, dtype=np.uint64)
        self._check(self.lib.snpgpu_alltoallv(self.ctx, C.c_void_p(int(d_send)) if int(d_send) else None, _ptr(sb),
                                              C.c_void_p(int(d_recv)) if int(d_recv) else None, _ptr(rb)))

    def stream_wait(self, timeout_ms):
        """Wait for the context's stream, at most timeout_ms; raises TimeoutError when it is still busy then."""
        rc = self.lib.snpgpu_stream_wait(self.ctx, int(timeout_ms))
        if rc == L.E_TIMEOUT:
            raise TimeoutError("the device stream was still busy after %d ms" % int(timeout_ms))
        self._check(rc)

    def tiles_gather_dev(self, d_matrix, n_padded, d_rows, d_cols, n_tiles, d_out):
        self._check(self.lib.snpgpu_tiles_gather_dev(self.ctx, C.c_void_p(int(d_matrix)), int(n_padded), C.c_void_p(int(d_rows)), C.c_void_p(int(d_cols)),
                                                     int(n_tiles), C.c_void_p(int(d_out))))

    def tiles_scatter_dev(self, d_tiles, d_rows, d_cols, n_tiles, d_matrix, n_padded):
        self._check(self.lib.snpgpu_tiles_scatter_dev(self.ctx, C.c_void_p(int(d_tiles)), C.c_void_p(int(d_rows)), C.c_void_p(int(d_cols)), int(n_tiles),
                                                      C.c_void_p(int(d_matrix)), int(n_padded)))

    def group_check_dev(self, d_filters, d_counts, d_line_off, d_wanted, d_excl_off, d_excl_slots, n_samples, n_sites, d_out):
        opt = lambda v: C.c_void_p(int(v)) if v else None     # noqa: E731
        self._check(self.lib.snpgpu_group_check_dev(self.ctx, opt(d_filters), opt(d_counts), opt(d_line_off), opt(d_wanted), opt(d_excl_off),
                                                    opt(d_excl_slots), int(n_samples), int(n_sites), C.c_void_p(int(d_out))))

    def pileups(self, budget_bytes=0):
        return Pileups(self, budget_bytes)

    # ---- distance ------------------------------------------------------------------------------
    def packed_row_bytes(self, n_sites):
        return int(self.lib.snpgpu_packed_row_bytes(n_sites))

    def distance(self, symbols):
        """symbols: (n, s) uint8 array of sequence bytes.  Returns (n, n) int32."""
        sym = np.ascontiguousarray(symbols, dtype=np.uint8)
        n, s = sym.shape
        out = np.zeros((n, n), dtype=np.int32)
        self._check(self.lib.snpgpu_distance(self.ctx, _ptr(sym) if sym.size else None, n, s, _ptr(out)))
        return out

    def pack_matrix_dev(self, d_symbols, n_rows, n_sites, row_stride, d_packed):
        self._check(self.lib.snpgpu_pack_matrix_dev(self.ctx, C.c_void_p(d_symbols), n_rows, n_sites, row_stride,
                                                    C.c_void_p(d_packed)))

    def distance_packed_dev(self, d_packed, n_rows, n_sites, d_out, tile_rank=0, tile_nranks=1):
        self._check(self.lib.snpgpu_distance_packed_dev(self.ctx, C.c_void_p(d_packed), n_rows, n_sites, tile_rank,
                                                        tile_nranks, C.c_void_p(d_out)))

    # ---- filter_regions ------------------------------------------------------------------------
    def dense_windows(self, positions, seg_off, max_snps, windows):
        pos = np.ascontiguousarray(positions, dtype=np.int64)
        seg = np.ascontiguousarray(seg_off, dtype=np.uint32)
        ms = np.ascontiguousarray(max_snps, dtype=np.int32)
        ws = np.ascontiguousarray(windows, dtype=np.int32)
        cap = max(1, len(pos) * max(1, len(ms)))
        o_s, o_e, o_g = np.empty(cap, np.int64), np.empty(cap, np.int64), np.empty(cap, np.uint32)
        n = C.c_uint32()
        self._check(self.lib.snpgpu_dense_windows(self.ctx, _ptr(pos), _ptr(seg), len(seg) - 1, _ptr(ms), _ptr(ws),
                                                  len(ms), _ptr(o_s), _ptr(o_e), _ptr(o_g), C.byref(n)))
        return o_s[:n.value], o_e[:n.value], o_g[:n.value]

    def merge_regions(self, group, start, end):
        g = np.ascontiguousarray(group, dtype=np.uint32)
        s = np.ascontiguousarray(start, dtype=np.int64)
        e = np.ascontiguousarray(end, dtype=np.int64)
        m = len(g)
        og, os_, oe = np.empty(max(m, 1), np.uint32), np.empty(max(m, 1), np.int64), np.empty(max(m, 1), np.int64)
        n = C.c_uint32()
        self._check(self.lib.snpgpu_merge_regions(self.ctx, _ptr(g), _ptr(s), _ptr(e), m, _ptr(og), _ptr(os_), _ptr(oe),
                                                  C.byref(n)))
        return og[:n.value], os_[:n.value], oe[:n.value]

    def in_regions(self, pos_group, positions, reg_off, reg_start, reg_end):
        pg = np.ascontiguousarray(pos_group, dtype=np.uint32)
        ps = np.ascontiguousarray(positions, dtype=np.int64)
        ro = np.ascontiguousarray(reg_off, dtype=np.uint32)
        rs = np.ascontiguousarray(reg_start, dtype=np.int64)
        re_ = np.ascontiguousarray(reg_end, dtype=np.int64)
        out = np.zeros(len(ps), dtype=np.uint8)
        self._check(self.lib.snpgpu_in_regions(self.ctx, _ptr(pg), _ptr(ps), len(ps), _ptr(ro), _ptr(rs), _ptr(re_),
                                               len(ro) - 1, _ptr(out)))
        return out.astype(bool)

    # ---- merge_sites ---------------------------------------------------------------------------
    def merge_sites(self, keys, sample_of_key):
        k = np.ascontiguousarray(keys, dtype=np.uint64)
        s = np.ascontiguousarray(sample_of_key, dtype=np.uint32)
        m = len(k)
        uniq, off, car = np.empty(max(m, 1), np.uint64), np.zeros(m + 1, np.uint32), np.empty(max(m, 1), np.uint32)
        nu, nc = C.c_uint32(), C.c_uint32()
        self._check(self.lib.snpgpu_merge_sites(self.ctx, _ptr(k), _ptr(s), m, _ptr(uniq), _ptr(off), _ptr(car),
                                                C.byref(nu), C.byref(nc)))
        return uniq[:nu.value], off[:nu.value + 1], car[:nc.value]

    # ---- device-pointer forms of the small steps (asynchronous; arguments are device pointers as ints) ---------
    def dense_windows_dev(self, d_positions, d_seg_off, n_segs, n_pos, max_snps, windows, d_out_start, d_out_end, d_out_seg, d_out_n):
        ms = np.ascontiguousarray(max_snps, dtype=np.int32)
        ws = np.ascontiguousarray(windows, dtype=np.int32)
        self._check(self.lib.snpgpu_dense_windows_dev(self.ctx, C.c_void_p(d_positions), C.c_void_p(d_seg_off), n_segs, n_pos,
                                                      _ptr(ms), _ptr(ws), len(ms), C.c_void_p(d_out_start), C.c_void_p(d_out_end),
                                                      C.c_void_p(d_out_seg), C.c_void_p(d_out_n)))

    def merge_regions_dev(self, d_group, d_start, d_end, n, d_out_group, d_out_start, d_out_end, d_out_n):
        self._check(self.lib.snpgpu_merge_regions_dev(self.ctx, C.c_void_p(d_group), C.c_void_p(d_start), C.c_void_p(d_end), n,
                                                      C.c_void_p(d_out_group), C.c_void_p(d_out_start), C.c_void_p(d_out_end),
                                                      C.c_void_p(d_out_n)))

    def in_regions_dev(self, d_pos_group, d_positions, n_pos, d_reg_off, d_reg_start, d_reg_end, n_groups, d_out_flag):
        self._check(self.lib.snpgpu_in_regions_dev(self.ctx, C.c_void_p(d_pos_group), C.c_void_p(d_positions), n_pos,
                                                   C.c_void_p(d_reg_off), C.c_void_p(d_reg_start), C.c_void_p(d_reg_end), n_groups,
                                                   C.c_void_p(d_out_flag)))

    def merge_sites_dev(self, d_keys, d_samples, n, d_out_unique, d_out_off, d_out_carrier, d_out_n):
        """Union of site keys straight from device tensors (the C1 all-gather of the sharded pipeline): d_out_n[0] unique
        keys, d_out_n[1] carriers; capacities n, n + 1, n."""
        self._check(self.lib.snpgpu_merge_sites_dev(self.ctx, C.c_void_p(d_keys), C.c_void_p(d_samples), n, C.c_void_p(d_out_unique),
                                                    C.c_void_p(d_out_off), C.c_void_p(d_out_carrier), C.c_void_p(d_out_n)))

    # ---- synthetic pileups ---------------------------------------------------------------------
    def synth_reference_dev(self, seed, genome_len, d_ref):
        self._check(self.lib.snpgpu_synth_reference_dev(self.ctx, seed, genome_len, C.c_void_p(d_ref)))

    def synth_pileup_dev(self, seed, sample, genome_len, d_ref, d_site_alt, d_out, capacity, contig=b"synth_chr1",
                         mean_depth=30.0, p_same=0.15, p_other=0.02, n_clades=10):
        p = L.SynthParams(seed, sample, genome_len, mean_depth, p_same, p_other, n_clades, contig)
        nbytes = C.c_size_t()
        self._check(self.lib.snpgpu_synth_pileup_dev(self.ctx, C.byref(p), C.c_void_p(d_ref),
                                                     C.c_void_p(d_site_alt) if d_site_alt else None,
                                                     C.c_void_p(d_out) if d_out else None, capacity, C.byref(nbytes)))
        return nbytes.value


class Pileups(object):
    """Pileup files kept in device memory between site calling and the consensus scan (snpgpu_pileups): the input side of
    ``hot_path_batch``.  ``ingest`` streams files in (site calling on each while the next arrives) and keeps them while the
    budget lasts; ``get`` returns where file i lives."""

    def __init__(self, device, budget_bytes=0):
        self.device = device
        h = C.c_void_p()
        device._check(device.lib.snpgpu_pileups_create(device.ctx, int(budget_bytes), C.byref(h)))
        self.handle = h
        device._children.add(self)

    def ingest(self, paths, params=None, capacity=16384, done=None):
        """One streamed call over `paths`.  Returns (records array [n][capacity], counts, status [n][2], rcs); with `done`
        (an int32 array of len(paths)) the caller may watch done[f] become 1 from another thread and use row f then."""
        n = len(paths)
        arr = (C.c_char_p * max(n, 1))(*[os.fsencode(p) for p in paths])
        cap = int(capacity) if params is not None else 0
        sites = np.empty((n, max(cap, 1)), dtype=VARSCAN_DTYPE)
        counts = np.zeros(max(n, 1), dtype=np.uint32)
        status = np.zeros((max(n, 1), 2), dtype=np.uint64)
        rcs = np.zeros(max(n, 1), dtype=np.int32)
        self.device._check(self.device.lib.snpgpu_pileups_ingest(
            self.device.ctx, self.handle, arr, n, C.byref(params) if params is not None else None, cap, _ptr(sites), _ptr(counts),
            _ptr(status), _ptr(rcs), _ptr(done) if done is not None else None))
        return sites, counts[:n], status[:n], rcs[:n]

    def ingest_buffers(self, n, capacity):
        """The output arrays of one ingest call, allocated by the caller so that a consumer thread can read row f as soon as
        done[f] is set: (sites, counts, status, rcs, done)."""
        return (np.empty((n, max(int(capacity), 1)), dtype=VARSCAN_DTYPE), np.zeros(max(n, 1), dtype=np.uint32),
                np.zeros((max(n, 1), 2), dtype=np.uint64), np.zeros(max(n, 1), dtype=np.int32), np.zeros(max(n, 1), dtype=np.int32))

    def ingest_into(self, paths, params, capacity, bufs):
        n = len(paths)
        sites, counts, status, rcs, done = bufs
        arr = (C.c_char_p * max(n, 1))(*[os.fsencode(p) for p in paths])
        self.device._check(self.device.lib.snpgpu_pileups_ingest(
            self.device.ctx, self.handle, arr, n, C.byref(params) if params is not None else None, int(capacity) if params is not None else 0,
            _ptr(sites), _ptr(counts), _ptr(status), _ptr(rcs), _ptr(done)))

    def __len__(self):
        return int(self.device.lib.snpgpu_pileups_count(self.handle))

    def get(self, index):
        """(device pointer or 0 when the file is not resident, size in bytes)"""
        p, n = C.c_void_p(), C.c_uint64()
        if self.device.lib.snpgpu_pileups_get(self.handle, int(index), C.byref(p), C.byref(n)) != 0:
            raise IndexError(index)
        return int(p.value or 0), int(n.value)

    def stats(self):
        st = L.PileupsStats()
        self.device.lib.snpgpu_pileups_get_stats(self.handle, C.byref(st))
        return st

    def close(self):
        if self.handle:
            if self.device.ctx:
                self.device.lib.snpgpu_pileups_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_consensus_files(jobs, siteset, filter_names, preserve_ref_case, failed_snp_gt, n_threads=0, spill=None):
    """jobs: list of dicts with the fields of snpgpu_consensus_job (numpy arrays for the pointers; absent = NULL).  Writes
    the consensus FASTA / VCF files of all jobs on host threads (csrc/vcf_rows.hip).  spill: Device.read_symbol_spill() of the
    call the records came from.  Returns [(rc, rows)] per job."""
    lib = L.load()
    n = len(jobs)
    arr = (L.ConsensusJob * max(n, 1))()
    keep = []

    def ptr(a):
        if a is None:
            return None
        keep.append(a)
        return a.ctypes.data

    def text(t):
        if t is None:
            return None
        b = t if isinstance(t, bytes) else os.fsencode(t)
        keep.append(b)
        return b

    for j, job in enumerate(jobs):
        seq = job.get("sequence")
        arr[j].fasta_path = text(job.get("fasta_path"))
        arr[j].fasta_id = text(job.get("fasta_id"))
        arr[j].sequence = ptr(seq)
        arr[j].n_bases = len(seq) if seq is not None else 0
        arr[j].vcf_path = text(job.get("vcf_path"))
        arr[j].vcf_header = text(job.get("vcf_header"))
        arr[j].counts = ptr(job.get("counts"))
        arr[j].line_off = ptr(job.get("line_off"))
        arr[j].row_filters = ptr(job.get("row_filters"))
        arr[j].site_in_flow = ptr(job.get("site_in_flow"))
    fn = (C.c_char_p * 6)(*[x.encode("ascii") for x in filter_names])
    rc = lib.snpgpu_write_consensus_files(arr, n, len(siteset), _ptr(siteset._names), _ptr(siteset._offs), _ptr(siteset.keys), fn,
                                          1 if preserve_ref_case else 0, failed_snp_gt.encode("ascii"),
                                          _ptr(spill) if spill is not None and len(spill) else None, len(spill) if spill is not None else 0, int(n_threads))
    if rc != 0:
        raise SnpGpuError(rc, "snpgpu_write_consensus_files")
    return [(int(arr[j].rc), int(arr[j].n_rows)) for j in range(n)]


_default = None
_slot_lock = None


def device_count():
    return int(L.load().snpgpu_device_count())


def cpu_budget():
    """The host threads this process plans with (snpgpu_cpu_budget: affinity mask, cgroup quota, SNPGPU_MAX_CPU_CORES — the MaxCpuCores
    of run.py:387-400 — divided by the ranks that share the node) as a dict; `budget` is the figure for Python-side pools, `readers`
    and `writers` are what the library's file entry points start by default."""
    b = L.CpuBudget()
    rc = L.load().snpgpu_cpu_budget(C.byref(b))
    if rc != 0:
        raise RuntimeError("snpgpu_cpu_budget failed (%d)" % rc)
    return {name: int(getattr(b, name)) for name, _ in L.CpuBudget._fields_}


def set_local_ranks(n):
    """How many processes of this job share the node (0: back to SNPGPU_LOCAL_RANKS / LOCAL_WORLD_SIZE / 1)."""
    L.load().snpgpu_set_local_ranks(int(n))


def set_max_cpu_cores(n):
    """The MaxCpuCores of this process's node (0: back to SNPGPU_MAX_CPU_CORES / no cap)."""
    L.load().snpgpu_set_max_cpu_cores(int(n))


def host_threads(at_most, share=1):
    """Workers for a Python-side pool: the budget divided by `share`, between 1 and at_most."""
    return max(1, min(int(at_most), cpu_budget()["budget"] // max(1, share)))


def acquire_device_slot(n_devices, max_per_device=None, lock_dir=None, poll_seconds=0.05):
    """Device assignment for the reference's per-sample process array (run.py:709-710 starts up to max_cpu_cores
    ``cfsan_snp_pipeline call_consensus`` processes at once, none of which knows about the others).  Every process
    takes an advisory lock on one of ``n_devices x max_per_device`` slot files: the first free slot in a sweep that
    prefers the least loaded device (slot 0 of every device before slot 1 of any), or — while all are taken — it keeps
    sweeping until a holder goes away.  So the processes spread round-robin over the visible GPUs and at most
    ``max_per_device`` contexts exist per GPU at a time; the lock dies with the process.  Returns (device index, open
    lock file)."""
    import fcntl
    import time
    from . import _paths
    if max_per_device is None:
        max_per_device = int(os.environ.get("SNPGPU_MAX_PROCS_PER_DEVICE", "4"))
    max_per_device = max(1, max_per_device)
    lock_dir = lock_dir or os.environ.get("SNPGPU_LOCK_DIR")
    if lock_dir:
        os.makedirs(lock_dir, exist_ok=True)       # a directory the caller named (several users of one node may share its slots)
    else:
        try:
            lock_dir = _paths.private_dir()        # (refused when /tmp/snpgpu-<uid> is somebody else's)
        except _paths.UnsafeDirectory as err:
            # nowhere to keep slot files that is provably ours: the processes of this user cannot see each other, so each takes a
            # device by its process id (spread, not capped) and says why
            import sys
            sys.stderr.write("snpgpu: no device slots (%s); set SNPGPU_LOCK_DIR to a directory of your own\n" % err)
            return os.getpid() % n_devices, None
    start = os.getpid() % n_devices
    order = [((start + i) % n_devices, j) for j in range(max_per_device) for i in range(n_devices)]
    while True:
        for dev, j in order:
            f = open(os.path.join(lock_dir, "dev%d.slot%d" % (dev, j)), "a+")
            try:
                fcntl.flock(f, fcntl.LOCK_EX | fcntl.LOCK_NB)
                return dev, f
            except (IOError, OSError):
                f.close()
        time.sleep(poll_seconds)                   # everything is busy: look again shortly


def default_device():
    """The process-wide device.  SNPGPU_DEVICE / LOCAL_RANK pin it; otherwise a console-script process takes a device
    slot (see acquire_device_slot), anything else uses device 0."""
    global _default, _slot_lock
    if _default is None:
        pinned = os.environ.get("SNPGPU_DEVICE", os.environ.get("LOCAL_RANK"))
        if pinned is None and L.TORCH_FREE_OK:
            n = device_count()
            if n > 0:
                index, _slot_lock = acquire_device_slot(n)
                _default = Device(index)
                return _default
        _default = Device()
    return _default
