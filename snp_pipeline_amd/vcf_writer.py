"""consensus.vcf writer (the reference's vcf_writer.SingleSampleWriter, snppipeline/vcf_writer.py:92-134, 295-435).

Every number in a row (SDP, RD, AD[], RDF, RDR, ADF[], ADR[], ALT order, FT) is a by-product of the per-site
caller kernel (``snpgpu_site_counts``); this file only lays the text out.  Header layout follows what PyVCF3's
Writer emits for the reference's template: plain ``##key=value`` lines, INFO, FORMAT, FILTER, ``#CHROM``
(pinned by the lambda consensus*.vcf fixtures, which the reference's own test compares modulo fileDate/source).
"""
from __future__ import print_function

import datetime

import numpy as np

from . import _lib as L
from . import utils

FORMAT_IDS = "GT:SDP:RD:AD:RDF:RDR:ADF:ADR:FT"
_FORMAT_LINES = [
    '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
    '##FORMAT=<ID=SDP,Number=1,Type=Integer,Description="Raw read depth">',
    '##FORMAT=<ID=RD,Number=1,Type=Integer,Description="Depth of reference-supporting bases">',
    '##FORMAT=<ID=AD,Number=A,Type=Integer,Description="Depth of variant-supporting bases (comma-separated depth per alt allele)">',
    '##FORMAT=<ID=RDF,Number=1,Type=Integer,Description="Depth of reference-supporting bases on forward strand">',
    '##FORMAT=<ID=RDR,Number=1,Type=Integer,Description="Depth of reference-supporting bases on reverse strand">',
    '##FORMAT=<ID=ADF,Number=A,Type=Integer,Description="Depth of variant-supporting bases on forward strand (comma-separated depth per alt allele)">',
    '##FORMAT=<ID=ADR,Number=A,Type=Integer,Description="Depth of variant-supporting bases on reverse strand (comma-separated depth per alt allele)">',
    '##FORMAT=<ID=FT,Number=1,Type=String,Description="Genotype filters using the same codes as the FILTER data element">',
]


def filter_descriptions(min_cons_freq, min_cons_depth, min_cons_strand_depth, min_cons_strand_bias):
    """Names in failed-filter bit order (pileup.py:467-490) + Region (call_consensus.py:156)."""
    return [
        ("RawDpth", "No read depth"),
        ("VarFreq" + str(int(100 * min_cons_freq)), "Variant base frequency below %.2f" % min_cons_freq),
        ("Depth" + str(min_cons_depth), "Less than %i supporting reads" % min_cons_depth),
        ("StrDpth" + str(min_cons_strand_depth), "Less than %i variant-supporing reads on at least one strand" % min_cons_strand_depth),
        ("StrBias" + str(int(100 * min_cons_strand_bias)), "Fraction of variant supporting reads below %.2f on one strand" % min_cons_strand_bias),
        ("Region", "Position is in dense region of snps or near the end of the contig."),
    ]


def header_lines(sample_id, filters, reference, now=None):
    now = now or datetime.datetime.now()
    out = ["##fileformat=VCFv4.2", now.strftime("##fileDate=%Y%m%d"), "##source=CFSAN SNP-Pipeline %s" % utils.__version__,
           "##reference=%s" % reference,
           '##INFO=<ID=NS,Number=1,Type=Integer,Description="Number of samples with data">']
    out.extend(_FORMAT_LINES)
    out.append('##FILTER=<ID=PASS,Description="All filters passed">')
    out.extend('##FILTER=<ID=%s,Description="%s">' % (n, d) for n, d in filters)
    out.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s" % sample_id)
    return out


def row_from_counts(chrom, pos, c, filter_names, preserve_ref_case, failed_snp_gt, spill=None):
    """One VCF data line from a snpgpu_site_counts record (vcf_writer.py:295-379); spill: the call's spill records, for a
    position with more than 8 distinct symbols."""
    ref = chr(int(c["ref_base"]))
    upper_ref = ref.upper()
    if not preserve_ref_case:
        ref = upper_ref
    mask = int(c["filters"])
    failed = [filter_names[i] for i in range(6) if mask >> i & 1]
    n_sym, code = int(c["n_symbols"]) & 0xFF, int(c["n_symbols"]) >> 8
    ranked = [(chr(int(c["sym"][r])), int(c["total"][r]), int(c["fwd"][r]), int(c["rev"][r])) for r in range(min(n_sym, L.MAX_SYMS))]
    raw_depth = int(c["raw_depth"])
    if n_sym > L.MAX_SYMS or code:
        extra = max(n_sym - L.MAX_SYMS, 0)
        if spill is None or code == 0 or code - 1 >= len(spill) or int(spill[code - 1]["n"]) != extra:
            raise ValueError("%s:%d has %d distinct symbols (or a long reference field) and no spill record; the device record keeps %d"
                             % (chrom, pos, n_sym, L.MAX_SYMS))
        more = spill[code - 1]
        if int(more["depth64"]) != 0:                   # a depth column outside 0 .. 2^32 - 1 ("-3", "5000000000": int() takes them)
            raw_depth = int(more["depth64"])
        ranked += [(chr(int(more["sym"][r])), int(more["total"][r]), int(more["fwd"][r]), int(more["rev"][r])) for r in range(extra)]
        if int(more["ref_len"]) > 1:                    # a reference field of several bytes: the string itself, equal to no symbol
            from .device import spill_reference_field
            ref = spill_reference_field(spill, code - 1).decode("latin-1")
            upper_ref = ref.upper()
            if not preserve_ref_case:
                ref = upper_ref
    syms = [t[0] for t in ranked]
    total = {t[0]: t[1] for t in ranked}
    fwd = {t[0]: t[2] for t in ranked}
    rev = {t[0]: t[3] for t in ranked}
    if int(c["good_depth"]) == 0:                       # most_common_good_bases is None
        alt, gt, ad, adf, adr = [], ".", "0", "0", "0"
    else:
        alt = [s for s in syms if s != upper_ref]
        if not alt:
            gt, ad, adf, adr = "0", "0", "0", "0"
        else:
            gt = "0" if syms[0] == upper_ref else "1"
            ad = ",".join(str(total[s]) for s in alt)
            adf = ",".join(str(fwd.get(s, 0)) for s in alt)
            adr = ",".join(str(rev.get(s, 0)) for s in alt)
        if failed:
            gt = "." if failed_snp_gt == "." else ("0" if failed_snp_gt == "0" else "1")
    ft = ";".join(failed) if failed else "PASS"
    data = ":".join([gt, str(raw_depth), str(total.get(upper_ref, 0)), ad, str(fwd.get(upper_ref, 0)),
                     str(rev.get(upper_ref, 0)), adf, adr, ft])
    return "\t".join([chrom, str(pos), ".", ref, ",".join(alt) if alt else ".", ".", ft, "NS=1", FORMAT_IDS, data])


def format_rows(counts, order, contig_names, contig_name_off, site_keys, filter_names, preserve_ref_case, failed_snp_gt, spill=None):
    """The data lines of ``order`` as one bytes object, formatted by the library (snpgpu_format_vcf_rows: the same layout as
    row_from_counts, which stays as the readable statement of it and is tested against it row by row)."""
    import ctypes as C
    lib = L.load()
    counts = np.ascontiguousarray(counts)
    order = np.ascontiguousarray(order, dtype=np.uint32)
    keys = np.ascontiguousarray(site_keys, dtype=np.uint64)
    names = np.ascontiguousarray(contig_names, dtype=np.uint8)
    offs = np.ascontiguousarray(contig_name_off, dtype=np.uint32)
    fn = (C.c_char_p * 6)(*[n.encode("ascii") for n in filter_names])
    bad = C.c_int32(-1)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)      # noqa: E731
    gt = failed_snp_gt.encode("ascii")
    if spill is not None and len(spill):
        spill = np.ascontiguousarray(spill)
        sp, n_sp = ptr(spill), len(spill)
    else:
        sp, n_sp = None, 0
    need = lib.snpgpu_format_vcf_rows(ptr(counts), ptr(order), len(order), ptr(names), ptr(offs), ptr(keys), fn,
                                      1 if preserve_ref_case else 0, gt, sp, n_sp, None, 0, C.byref(bad))
    if bad.value >= 0:
        slot = int(order[bad.value])
        raise ValueError("site #%d has %d distinct symbols and no spill record; the device record keeps %d"
                         % (slot, int(counts[slot]["n_symbols"]) & 0xFF, L.MAX_SYMS))
    buf = C.create_string_buffer(max(int(need), 1))
    lib.snpgpu_format_vcf_rows(ptr(counts), ptr(order), len(order), ptr(names), ptr(offs), ptr(keys), fn,
                               1 if preserve_ref_case else 0, gt, sp, n_sp, buf, int(need), C.byref(bad))
    return buf.raw[:int(need)]


def write_consensus_vcf(path, sample_id, args, siteset, result, line_offsets, parsed=None):
    """Rows for every parsed position (snplist and exclude positions that have a pileup line), in pileup order.  parsed:
    bool per site — the positions THIS sample parses, when the site set also serves other samples' exclude lists."""
    filters = filter_descriptions(args.minConsFreq, args.minConsDpth, args.minConsStrdDpth, args.minConsStrdBias)
    names = [n for n, _ in filters]
    ok = result.counts["status"] == L.ST_OK
    have = np.nonzero(ok if parsed is None else (ok & parsed))[0]
    order = have[np.argsort(line_offsets[have], kind="stable")]
    rows = format_rows(result.counts, order, siteset._names, siteset._offs, siteset.keys, names, args.vcfPreserveRefCase,
                       args.vcfFailedSnpGt, spill=getattr(result, "spill", None))
    with open(path, "wb") as f:
        f.write(("\n".join(header_lines(sample_id, filters, args.vcfRefName)) + "\n").encode("ascii"))
        f.write(rows)


def write_all_positions_vcf_from_pileup(dev, siteset, path, sample_id, args, pileup_path, params, only_listed=False, check=True):
    """--vcfAllPos (call_consensus.py:148-151, vcf_writer.py:381-435): one row per pileup LINE, in file order — or, only_listed, per
    line at a listed position (a pileup that repeats positions, call_consensus.py:178-180) — from file to file inside the library
    (snpgpu_write_all_positions_vcf: 32-byte records back over the host link, rows formatted by its host threads; a 5 Mbp sample's
    5 M rows in tens of milliseconds where the row-by-row form below takes a minute).  Raises what the reference raises for the
    first line it cannot take; the file is not written then.  Returns (lines, rows)."""
    filters = filter_descriptions(args.minConsFreq, args.minConsDpth, args.minConsStrdDpth, args.minConsStrdBias)
    header = "\n".join(header_lines(sample_id, filters, args.vcfRefName)) + "\n"
    return dev.write_all_positions_vcf(siteset, pileup_path, params, path, header, [n for n, _ in filters], args.vcfPreserveRefCase,
                                       args.vcfFailedSnpGt, only_listed=only_listed, check=check)


def format_line_rows(pileup, line_offsets, recs, wide_index, wide, filter_names, preserve_ref_case, failed_snp_gt, only_listed=False, spill=None):
    """consensus.vcf rows (bytes) of the lines whose 32-byte records are `recs` (device.LINE_DTYPE; the wide ones in `wide`), CHROM and POS
    from the pileup text itself: snpgpu_format_line_rows, the library's host formatter behind the file-to-file writer."""
    import ctypes as C
    lib = L.load()
    text = np.frombuffer(pileup, dtype=np.uint8)
    off = np.ascontiguousarray(line_offsets, dtype=np.uint64)
    recs = np.ascontiguousarray(recs)
    widx = np.ascontiguousarray(wide_index, dtype=np.uint32)
    wide = np.ascontiguousarray(wide)
    fn = (C.c_char_p * 6)(*[n.encode("ascii") for n in filter_names])
    ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None and len(a) else None      # noqa: E731
    sp = np.ascontiguousarray(spill) if spill is not None and len(spill) else None
    n_rows, bad = C.c_uint64(), C.c_int64(-1)
    args = (ptr(text), len(text), ptr(off), ptr(recs), len(recs), ptr(widx), ptr(wide), len(widx), fn, 1 if preserve_ref_case else 0,
            failed_snp_gt.encode("ascii"), ptr(sp), len(sp) if sp is not None else 0, 1 if only_listed else 0)
    need = lib.snpgpu_format_line_rows(*args, None, 0, C.byref(n_rows), C.byref(bad))
    if bad.value >= 0:
        raise ValueError("line #%d: its record cannot be written (more symbols than a record keeps and no spill record, or an offset outside the text)" % bad.value)
    buf = C.create_string_buffer(max(int(need), 1))
    lib.snpgpu_format_line_rows(*args, buf, int(need), C.byref(n_rows), C.byref(bad))
    return buf.raw[:int(need)], int(n_rows.value)


def write_all_positions_vcf(path, sample_id, args, pileup_path, line_offsets, counts, spill=None):
    """The same file row by row in Python, from the per-line records of ``Device.call_all_lines``; CHROM and POS are the first two
    fields of the line itself.  The readable statement of the layout: the tests hold the library's rows against it."""
    import mmap
    filters = filter_descriptions(args.minConsFreq, args.minConsDpth, args.minConsStrdDpth, args.minConsStrdBias)
    names = [n for n, _ in filters]
    with open(path, "w") as f:
        f.write("\n".join(header_lines(sample_id, filters, args.vcfRefName)) + "\n")
        if len(line_offsets) == 0:
            return
        with open(pileup_path, "rb") as pf:
            mm = mmap.mmap(pf.fileno(), 0, access=mmap.ACCESS_READ)
            try:
                for i in range(len(line_offsets)):
                    start = int(line_offsets[i]) - 1
                    fields = mm[start:start + 256].split(None, 2)
                    if len(fields) < 3:                          # a very long contig name: take the whole line
                        end = mm.find(b"\n", start)
                        fields = mm[start:end if end >= 0 else len(mm)].split(None, 2)
                    f.write(row_from_counts(fields[0].decode("ascii"), int(fields[1]), counts[i], names,
                                            args.vcfPreserveRefCase, args.vcfFailedSnpGt, spill=spill) + "\n")
            finally:
                mm.close()
