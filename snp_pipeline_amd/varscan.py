"""Phase-1 site calling (`VarScan mpileup2snp --output-vcf 1`) on the device: var.flt.vcf from reads.all.pileup.

Replaces the VarScan v2.3.9 jar that snppipeline/call_sites.py:89-108 shells out to (third-party, not in the reference
tree).  The device pass (csrc/varscan.hip through ``Device.varscan_file``) does the per-line counting and the
min-coverage / min-reads2 / min-avg-qual / min-var-freq tests and returns one record per passing (line, allele); this
module finishes the few records that come back: Fisher's exact test against a 0.1 % error model (``PVAL``, ``GQ``,
``--p-value``), the strand filter, the homozygous threshold, and VarScan's VCF 4.1 text.  The text and the arithmetic are
pinned by the 69 019 data lines of the reference's bundled var.flt.vcf fixtures (tests/test_host_cpu.py feeds every line's
own counts back in); how unusual read-base strings are counted is not (see oracle/varscan_oracle.py and DESIGN.md).
"""
from __future__ import print_function

import math
import mmap
import shlex
from decimal import ROUND_HALF_EVEN, Decimal

from . import _lib as L

FORMAT_KEYS = "GT:GQ:SDP:DP:RD:AD:FREQ:PVAL:RBQ:ABQ:RDF:RDR:ADF:ADR"

_HEADER_LINES = [
    "##fileformat=VCFv4.1",
    "##source=VarScan2",
    '##INFO=<ID=ADP,Number=1,Type=Integer,Description="Average per-sample depth of bases with Phred score >= {q}">',
    '##INFO=<ID=WT,Number=1,Type=Integer,Description="Number of samples called reference (wild-type)">',
    '##INFO=<ID=HET,Number=1,Type=Integer,Description="Number of samples called heterozygous-variant">',
    '##INFO=<ID=HOM,Number=1,Type=Integer,Description="Number of samples called homozygous-variant">',
    '##INFO=<ID=NC,Number=1,Type=Integer,Description="Number of samples not called">',
    '##FILTER=<ID=str10,Description="Less than 10% or more than 90% of variant supporting reads on one strand">',
    '##FILTER=<ID=indelError,Description="Likely artifact due to indel reads at this position">',
    '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
    '##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Genotype Quality">',
    '##FORMAT=<ID=SDP,Number=1,Type=Integer,Description="Raw Read Depth as reported by SAMtools">',
    '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Quality Read Depth of bases with Phred score >= {q}">',
    '##FORMAT=<ID=RD,Number=1,Type=Integer,Description="Depth of reference-supporting bases (reads1)">',
    '##FORMAT=<ID=AD,Number=1,Type=Integer,Description="Depth of variant-supporting bases (reads2)">',
    '##FORMAT=<ID=FREQ,Number=1,Type=String,Description="Variant allele frequency">',
    '##FORMAT=<ID=PVAL,Number=1,Type=String,Description="P-value from Fisher\'s Exact Test">',
    '##FORMAT=<ID=RBQ,Number=1,Type=Integer,Description="Average quality of reference-supporting bases (qual1)">',
    '##FORMAT=<ID=ABQ,Number=1,Type=Integer,Description="Average quality of variant-supporting bases (qual2)">',
    '##FORMAT=<ID=RDF,Number=1,Type=Integer,Description="Depth of reference-supporting bases on forward strand (reads1plus)">',
    '##FORMAT=<ID=RDR,Number=1,Type=Integer,Description="Depth of reference-supporting bases on reverse strand (reads1minus)">',
    '##FORMAT=<ID=ADF,Number=1,Type=Integer,Description="Depth of variant-supporting bases on forward strand (reads2plus)">',
    '##FORMAT=<ID=ADR,Number=1,Type=Integer,Description="Depth of variant-supporting bases on reverse strand (reads2minus)">',
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSample1",
]


class Options(object):
    """mpileup2snp's options.  VarScan's own defaults, overridden by ``--name value`` pairs of the ExtraParams string
    (snppipeline.conf:199 passes ``--min-avg-qual 15 --min-var-freq 0.90 --min-reads2 5``)."""

    def __init__(self, extra_params=""):
        self.min_coverage = 8
        self.min_reads2 = 2
        self.min_avg_qual = 15
        self.min_var_freq = 0.20
        self.min_freq_for_hom = 0.75
        self.p_value = 0.99
        self.strand_filter = 1
        words = shlex.split(extra_params or "")
        kinds = {"--min-coverage": ("min_coverage", int), "--min-reads2": ("min_reads2", int), "--min-avg-qual": ("min_avg_qual", int),
                 "--min-var-freq": ("min_var_freq", float), "--min-freq-for-hom": ("min_freq_for_hom", float),
                 "--p-value": ("p_value", float), "--strand-filter": ("strand_filter", int)}
        i = 0
        while i < len(words):
            w = words[i]
            if w in kinds and i + 1 < len(words):
                name, conv = kinds[w]
                setattr(self, name, conv(words[i + 1]))
                i += 2
            elif w.startswith("--") and i + 1 < len(words) and not words[i + 1].startswith("--"):
                i += 2                                     # an option this step has no use for (--output-vcf 1, --variants 1)
            else:
                i += 1
        if self.min_coverage < 0 or self.min_reads2 < 0 or self.min_avg_qual < 0:
            raise ValueError("negative mpileup2snp threshold")

    def device_params(self):
        return L.VarscanParams(self.min_coverage, self.min_reads2, self.min_avg_qual, 0, self.min_var_freq)


def header_text(min_avg_qual=15):
    return "".join(line.replace("{q}", str(min_avg_qual)) + "\n" for line in _HEADER_LINES)


# ---- Fisher's exact test on 2 x 2 tables, terms from log-factorials --------------------------------------------------
class _Hypergeometric(object):
    def __init__(self):
        self.logfact = [0.0]

    def _upto(self, n):
        t = self.logfact
        while len(t) <= n:
            t.append(t[-1] + math.log(len(t)))

    def term(self, a, b, c, d):
        n = a + b + c + d
        self._upto(n)
        t = self.logfact
        return math.exp(t[a + b] + t[c + d] + t[a + c] + t[b + d] - (t[a] + t[b] + t[c] + t[d] + t[n]))

    def right_tail(self, a, b, c, d):
        total = self.term(a, b, c, d)
        steps = c if c < b else b
        for k in range(1, steps + 1):
            total += self.term(a + k, b - k, c - k, d + k)
        return total

    def two_tails(self, a, b, c, d):
        here = self.term(a, b, c, d)
        total = here
        for k in range(1, min(a, d) + 1):
            t = self.term(a - k, b + k, c + k, d - k)
            if t <= here:
                total += t
        for k in range(1, min(b, c) + 1):
            t = self.term(a + k, b - k, c - k, d + k)
            if t <= here:
                total += t
        return total


_HG = _Hypergeometric()
_PVALUES = {}


def variant_p_value(reads1, reads2):
    """VarScan.getSignificance: (reads1, reads2) against the split a 0.001 error rate predicts at that coverage."""
    key = (reads1, reads2)
    p = _PVALUES.get(key)
    if p is None:
        cover = reads1 + reads2
        expected2 = int(cover * 0.001)
        p = _PVALUES[key] = _HG.right_tail(cover - expected2, expected2, reads1, reads2)
    return p


def _sci(p):
    """Java DecimalFormat("0.####E0")."""
    if p == 0.0:
        return "0E0"
    d = Decimal(p)
    exp10 = d.adjusted()
    mant = d.scaleb(-exp10).quantize(Decimal("0.0001"), rounding=ROUND_HALF_EVEN)
    if mant >= 10:
        exp10 += 1
        mant = d.scaleb(-exp10).quantize(Decimal("0.0001"), rounding=ROUND_HALF_EVEN)
    text = format(mant, "f").rstrip("0").rstrip(".")
    return "%sE%d" % (text, exp10)


_PERCENT_TEXT = {}


def _percent(part, whole):
    """Java DecimalFormat("###.##") of part / whole * 100, with the % sign."""
    key = (part, whole)
    text = _PERCENT_TEXT.get(key)
    if text is None:
        value = Decimal((float(part) / float(whole)) * 100.0).quantize(Decimal("0.01"), rounding=ROUND_HALF_EVEN)
        text = _PERCENT_TEXT[key] = format(value, "f").rstrip("0").rstrip(".") + "%"
    return text


def strand_filter_fails(rdf, rdr, adf, adr):
    """str10: >90 % of the variant reads on one strand, a reference count of 2+ that is not itself that lopsided, and a
    two-tailed Fisher p < 0.01 between the two strand splits."""
    var_plus = float(adf) / float(adf + adr)
    if 0.10 <= var_plus <= 0.90 or rdf + rdr < 2:
        return False
    ref_plus = float(rdf) / float(rdf + rdr)
    return 0.10 <= ref_plus <= 0.90 and _HG.two_tails(rdf, rdr, adf, adr) < 0.01


_P_TEXT = {}


def data_line(chrom, pos, ref, alt, sdp, dp, total, rdf, rdr, rbq, adf, adr, abq, p, homozygous, filter_text="PASS"):
    pt = _P_TEXT.get(p)
    if pt is None:
        pt = _P_TEXT[p] = (255 if p <= 0.0 else min(255, int(-10.0 * math.log10(p))), _sci(p))
    rd, ad = rdf + rdr, adf + adr
    sample = "%s:%d:%d:%d:%d:%d:%s:%s:%d:%d:%d:%d:%d:%d" % ("1/1" if homozygous else "0/1", pt[0], sdp, dp, rd, ad, _percent(ad, total), pt[1],
                                                            rbq, abq, rdf, rdr, adf, adr)
    info = "ADP=%d;WT=0;HET=%d;HOM=%d;NC=0" % (dp, 0 if homozygous else 1, 1 if homozygous else 0)
    return "%s\t%s\t.\t%s\t%s\t.\t%s\t%s\t%s\t%s\n" % (chrom, pos, ref, alt, filter_text, info, FORMAT_KEYS, sample)


def rows_from_records(records, pileup_bytes, opts):
    """records: Device.varscan_file's array (file order); pileup_bytes: the file (an mmap).  Yields the data lines."""
    cols = [records[k].tolist() for k in ("line_off", "sdp", "dp", "total", "rdf", "rdr", "ref_qual_sum", "adf", "adr", "alt_qual_sum",
                                          "ref_base", "alt_base")]
    rows = list(zip(*cols))
    i, n = 0, len(rows)
    strand = bool(opts.strand_filter)
    while i < n:
        off = rows[i][0]
        best = None
        best_ad = -1
        while i < n and rows[i][0] == off:                      # the alleles of one line: most reads wins, first on ties
            r = rows[i]
            ad = r[7] + r[8]
            p = variant_p_value(r[4] + r[5], ad)
            if p <= opts.p_value and ad > best_ad:
                best, best_ad, best_p = r, ad, p
            i += 1
        if best is None:
            continue
        _, sdp, dp, total, rdf, rdr, rq, adf, adr, aq, ref, alt = best
        t1 = pileup_bytes.find(b"\t", off)
        t2 = pileup_bytes.find(b"\t", t1 + 1)
        rd = rdf + rdr
        fails = strand and strand_filter_fails(rdf, rdr, adf, adr)
        yield data_line(pileup_bytes[off:t1].decode("latin-1"), pileup_bytes[t1 + 1:t2].decode("latin-1"), chr(ref), chr(alt), sdp, dp, total, rdf, rdr,
                        rq // rd if rd else 0, adf, adr, aq // best_ad, best_p, float(best_ad) / float(total) >= opts.min_freq_for_hom,
                        "str10" if fails else "PASS")


def mpileup2snp(device, pileup_path, vcf_path, opts):
    """reads.all.pileup -> var.flt.vcf.  Returns (lines in the pileup, sites written)."""
    records, n_lines = device.varscan_file(pileup_path, opts.device_params())
    n_rows = 0
    with open(vcf_path, "w", encoding="latin-1", newline="\n") as out:           # contig names pass through byte for byte
        out.write(header_text(opts.min_avg_qual))
        if len(records):
            with open(pileup_path, "rb") as f:
                view = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
                try:
                    for line in rows_from_records(records, view, opts):
                        out.write(line)
                        n_rows += 1
                finally:
                    view.close()
    return n_lines, n_rows
