"""CPU restatement of the pileup -> consensus path.  TEST INFRASTRUCTURE ONLY.

This module is the checker for the HIP kernels.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package (``snp_pipeline_amd``) never does.

It restates, on ``bytes`` and with explicit automata instead of regexes, what
the reference computes in

* ``snppipeline/pileup.py:209-274``   Record parsing, quality filter, histograms
* ``snppipeline/pileup.py:276-325``   caret / indel / dollar stripping
* ``snppipeline/pileup.py:408-429``   Reader: per line split + (chrom,pos) lookup
* ``snppipeline/pileup.py:492-590``   ConsensusCaller filters
* ``snppipeline/call_consensus.py:147-188``  per-sample driver (Region filter,
  '-' mapping, last duplicate line wins, snplist-order output)

Parity is PINNED: ``oracle/gen_golden.py`` imports the real reference in the
build container and writes ``tests/golden/pileup_vectors.json.gz`` (doctest
vectors + fuzzed lines + whole-file consensus runs); ``tests/test_oracle.py``
checks this module against those vectors.
"""

from dataclasses import dataclass, field

# bytes that ``str.split()`` treats as separators, restricted to ASCII
# (pileup.py:206/424 split on any whitespace).
WS = frozenset(b"\t\n\x0b\x0c\r\x1c\x1d\x1e\x1f ")

# failed-filter bit order == the order pileup.py:564-584 appends them, then
# call_consensus.py:165-168 appends Region.
F_RAWDPTH, F_VARFREQ, F_DEPTH, F_STRDPTH, F_STRBIAS, F_REGION = 1, 2, 4, 8, 16, 32

INDEL_CAP = 1 << 30   # any count >= string length behaves the same (slice clips)


def iter_lines(data):
    """Lines as CPython's text-mode file iterator yields them (universal
    newlines: ``\\n``, ``\\r\\n`` and a lone ``\\r`` all end a line), without the
    terminator."""
    n = len(data)
    if b"\r" not in data:                     # common case: plain '\n' files
        start = 0
        while start < n:
            end = data.find(b"\n", start)
            if end < 0:
                end = n
            yield start, data[start:end]
            start = end + 1
        return
    start = 0
    i = 0
    while i < n:
        c = data[i]
        if c == 0x0A:
            yield start, data[start:i]
            i += 1
            start = i
        elif c == 0x0D:
            yield start, data[start:i]
            i += 1
            if i < n and data[i] == 0x0A:
                i += 1
            start = i
        else:
            i += 1
    if start < n:
        yield start, data[start:n]


def split_fields(line):
    """``line.rstrip().split()`` on ASCII bytes."""
    out = []
    cur = bytearray()
    for c in line:
        if c in WS:
            if cur:
                out.append(bytes(cur))
                cur = bytearray()
        else:
            cur.append(c)
    if cur:
        out.append(bytes(cur))
    return out


def strip_bases(raw):
    """pileup.py:276-325 restated as two left-to-right automata.

    Pass 1 drops every ``^`` together with the byte after it (a trailing lone
    ``^`` stays).  Pass 2 works on that result: every ``[+-]`` directly
    followed by a digit opens a marker; the marker and its digit run always
    vanish, and its count is added to a running *debt* that swallows the
    following non-marker bytes one by one.  Debts of neighbouring markers add
    up, which is exactly what the reference's back-to-front slicing does when
    one marker's tail runs into the next marker.  ``$`` bytes that
    survive are dropped last.
    """
    s1 = bytearray()
    i, n = 0, len(raw)
    while i < n:
        if raw[i] == 0x5E and i + 1 < n:       # '^' + any next byte
            i += 2
        else:
            s1.append(raw[i])
            i += 1
    out = bytearray()
    debt = 0
    i, n = 0, len(s1)
    while i < n:
        c = s1[i]
        if c in (0x2B, 0x2D) and i + 1 < n and 0x30 <= s1[i + 1] <= 0x39:
            j = i + 1
            cnt = 0
            while j < n and 0x30 <= s1[j] <= 0x39:
                cnt = min(cnt * 10 + (s1[j] - 0x30), INDEL_CAP)
                j += 1
            debt = min(debt + cnt, INDEL_CAP)
            i = j
        elif debt > 0:
            debt -= 1
            i += 1
        else:
            if c != 0x24:                      # '$'
                out.append(c)
            i += 1
    return bytes(out)


def _upper(c):
    return c - 32 if 0x61 <= c <= 0x7A else c


def _lower(c):
    return c + 32 if 0x41 <= c <= 0x5A else c


@dataclass
class Record:
    chrom: bytes
    position: int
    reference_base: bytes
    raw_depth: int
    good_depth: int = 0
    forward_good_depth: int = 0
    reverse_good_depth: int = 0
    base_good_depth: dict = field(default_factory=dict)          # upper byte -> n
    forward_base_good_depth: dict = field(default_factory=dict)
    reverse_base_good_depth: dict = field(default_factory=dict)
    most_common_good_bases: list = None                          # ranked bytes or None


def parse_record(fields, min_base_quality):
    """pileup.py:209-274 on an already split line (list of bytes)."""
    rec = Record(chrom=fields[0], position=int(fields[1]),
                 reference_base=fields[2], raw_depth=int(fields[3]))
    if rec.raw_depth == 0 or len(fields) < 5:
        return rec
    bases = strip_bases(fields[4])
    quals = fields[5]                      # IndexError when absent, as pileup.py:237
    # bases_str.replace('.', ref.upper()).replace(',', ref.lower()), pileup.py:255-258, for a field of any length: a ','
    # that the first replace brought in (the field itself holds one) is hit by the second replace too
    ref_lo = bytes(_lower(c) for c in rec.reference_base)
    for_dot = b"".join(ref_lo if c == 0x2C else bytes([_upper(c)]) for c in rec.reference_base)
    total, fwd, rev = {}, {}, {}
    good = nf = nr = 0
    for b0, q in zip(bases, quals):        # zip truncates at the shorter one
        if q - 33 < min_base_quality:
            continue
        good += 1                          # good_depth counts reads (pileup.py:250), whatever they are spelled out to
        for b in (for_dot if b0 == 0x2E else ref_lo if b0 == 0x2C else (b0,)):
            u = _upper(b)
            total[u] = total.get(u, 0) + 1
            if b <= 0x5A:
                nf += 1
                fwd[b] = fwd.get(b, 0) + 1
            elif b >= 0x61:
                nr += 1
                rev[u] = rev.get(u, 0) + 1
    rec.good_depth, rec.forward_good_depth, rec.reverse_good_depth = good, nf, nr
    rec.base_good_depth, rec.forward_base_good_depth, rec.reverse_base_good_depth = total, fwd, rev
    if good >= 1:
        rec.most_common_good_bases = [k for k, _ in sorted(total.items(), key=lambda kv: (-kv[1], kv[0]))]
    return rec


@dataclass
class CallerParams:
    min_base_quality: int = 0
    min_cons_freq: float = 0.6
    min_cons_depth: int = 1
    min_cons_strand_depth: int = 0
    min_cons_strand_bias: float = 0.0


def call_record(rec, p):
    """pileup.py:550-590.  Returns (base byte, failed-filter bitmask)."""
    if rec.most_common_good_bases is None:
        return 0x2D, F_RAWDPTH
    cons = rec.most_common_good_bases[0]
    n = rec.base_good_depth.get(cons, 0)
    nf = rec.forward_base_good_depth.get(cons, 0)
    nr = rec.reverse_base_good_depth.get(cons, 0)
    mask = 0
    if n < rec.good_depth * p.min_cons_freq:
        mask |= F_VARFREQ
    if n < p.min_cons_depth:
        mask |= F_DEPTH
    if nf < p.min_cons_strand_depth or nr < p.min_cons_strand_depth:
        mask |= F_STRDPTH
    bias = n * p.min_cons_strand_bias
    if nf < bias or nr < bias:
        mask |= F_STRBIAS
    if len(rec.reference_base) == 1:       # (a one-character base never equals a longer field, pileup.py:586-587)
        ref = rec.reference_base[0]
        if cons == _upper(ref):
            cons = ref
    return cons, mask


def filter_names(p):
    """pileup.py:467-471 — names in bit order, Region last."""
    return ["RawDpth", "VarFreq%d" % int(100 * p.min_cons_freq), "Depth%d" % p.min_cons_depth,
            "StrDpth%d" % p.min_cons_strand_depth, "StrBias%d" % int(100 * p.min_cons_strand_bias), "Region"]


def scan_sites(data, wanted, min_base_quality):
    """pileup.py:423-429: yield a Record for every line whose (chrom,pos) is in
    ``wanted`` (a set of (bytes,int)); every line must split into >= 2 fields
    with an integer second field, as in the reference."""
    # bytes.split() knows the ASCII separators except FS/GS/RS/US; fall back to the explicit splitter for those
    exotic = any(c in data for c in (b"\x1c", b"\x1d", b"\x1e", b"\x1f"))
    for _, line in iter_lines(data):
        f = split_fields(line) if exotic else line.split()
        chrom, pos = f[:2]                 # ValueError on short lines, like the reference
        if (chrom, int(pos)) in wanted:
            yield parse_record(f, min_base_quality)


def call_consensus_sites(data, snp_list, excluded, p):
    """call_consensus.py:147-188.  ``snp_list``: list of (chrom bytes, pos) in
    snplist order; ``excluded``: set of the same.  Returns the consensus bytes
    in snplist order and {key: (record, base, mask)} for every parsed line (the
    last line for a key wins)."""
    wanted = set(snp_list) | set(excluded)
    snps = set(snp_list)
    called = {}
    detail = {}
    for rec in scan_sites(data, wanted, p.min_base_quality):
        key = (rec.chrom, rec.position)
        base, mask = call_record(rec, p)
        if key in excluded:
            mask |= F_REGION
        if key in snps:
            called[key] = 0x2D if (mask or base == 0x2A) else base
        detail[key] = (rec, base, mask)
    return bytes(called.get(k, 0x2D) for k in snp_list), detail


def depth_sum(data):
    """collect_metrics.py:325-340: the sum of int(tokens[3]) over the lines that have one; a line with fewer fields or a
    4th field that is not an integer is skipped.  (avePileupDepth is this sum / reference length, "%.2f".)"""
    total = 0
    for _, ln in iter_lines(data):
        tokens = split_fields(ln)
        try:
            total += int(tokens[3].decode())
        except (ValueError, IndexError):
            pass
    return total
