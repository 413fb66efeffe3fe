"""CPU restatement of phase-1 site calling: ``VarScan mpileup2snp --output-vcf 1`` on a one-sample pileup.
TEST INFRASTRUCTURE ONLY (same rules as pileup_oracle.py: tests/, smoke() and bench.py's cpu_baseline may import it).

What it replaces: the reference does not implement this step, it shells out to a third-party jar —
``snppipeline/call_sites.py:89-108`` runs ``java -jar VarScan.jar mpileup2snp reads.all.pileup --output-vcf 1
$VarscanMpileup2snp_ExtraParams`` (defaults ``--min-avg-qual 15 --min-var-freq 0.90 --min-reads2 5``,
``data/configuration/snppipeline.conf:199``) and keeps stdout as ``var.flt.vcf``.  The dependency is VarScan **v2.3.9**
(``docs/faq.rst:26, :103``; ``Dockerfile``/``environment.yml``); its source is NOT under /root/reference and there is no JVM in
this image, so this file restates VarScan's published algorithm (net.sf.varscan: ``VarScan.getReadCounts``,
``VarScan.callPosition``, ``VarScan.getSignificance`` + ``FishersExact``, ``CallMpileup``'s VCF branch).

PARITY — what is pinned and what is not:

* PINNED by the reference's golden data: the 58 bundled ``var.flt.vcf`` files (lambda 4, agona 6, listeria 48 samples; 69 019
  data lines, inside ``tests/golden/fixtures/*/expected.tar.xz``).  Every line must come back byte-identical from
  ``vcf_row`` when it is fed the line's own counts (``tests/test_oracle.py``): that fixes the header text, the column
  layout, ``PVAL`` (Fisher's exact test against a 0.001 error model, ``0.####E0``), ``GQ`` (``int(-10 log10 p)``, cap 255),
  ``GT``/``HET``/``HOM`` (homozygous at >= 75 %), ``ADP``, ``FREQ``'s ``#.##%`` text, and that ``FREQ``'s denominator is NOT ``DP``
  and NOT ``RD + AD`` (it counts indel-supporting reads, which have no quality and are not in DP; 836 distinct (RD, AD)
  pairs, 0 misses).  The lines also bound the selection rules from inside: min DP 8, min AD >= 5, min ABQ 15, min FREQ
  90 %, strand filter never fails a site whose reference count is < 2 (1 016 such lines with all variant reads on one
  strand are present), and where the reference count is >= 2 the strand test's p-value is >= 0.01 on every line.
* UNPINNED: how VarScan turns the read-base string into those counts on unusual input (an indel at the very end of the
  string, indel lengths of 4+ digits, reference skips ``<`` ``>``), and which lines it *drops* — the reference ships no pileup
  (``.MISSING_LARGE_BLOBS``) and no VarScan, so no (pileup line -> row) pair exists to check against.  Those rules follow
  VarScan's published source as restated below; DESIGN.md carries the same caveat.  "parity unpinned" for read counting.
"""

import math
import re
from decimal import ROUND_HALF_EVEN, Decimal

VCF_HEADER = (
    "##fileformat=VCFv4.1\n"
    "##source=VarScan2\n"
    '##INFO=<ID=ADP,Number=1,Type=Integer,Description="Average per-sample depth of bases with Phred score >= %(q)d">\n'
    '##INFO=<ID=WT,Number=1,Type=Integer,Description="Number of samples called reference (wild-type)">\n'
    '##INFO=<ID=HET,Number=1,Type=Integer,Description="Number of samples called heterozygous-variant">\n'
    '##INFO=<ID=HOM,Number=1,Type=Integer,Description="Number of samples called homozygous-variant">\n'
    '##INFO=<ID=NC,Number=1,Type=Integer,Description="Number of samples not called">\n'
    '##FILTER=<ID=str10,Description="Less than 10%% or more than 90%% of variant supporting reads on one strand">\n'
    '##FILTER=<ID=indelError,Description="Likely artifact due to indel reads at this position">\n'
    '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n'
    '##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Genotype Quality">\n'
    '##FORMAT=<ID=SDP,Number=1,Type=Integer,Description="Raw Read Depth as reported by SAMtools">\n'
    '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Quality Read Depth of bases with Phred score >= %(q)d">\n'
    '##FORMAT=<ID=RD,Number=1,Type=Integer,Description="Depth of reference-supporting bases (reads1)">\n'
    '##FORMAT=<ID=AD,Number=1,Type=Integer,Description="Depth of variant-supporting bases (reads2)">\n'
    '##FORMAT=<ID=FREQ,Number=1,Type=String,Description="Variant allele frequency">\n'
    '##FORMAT=<ID=PVAL,Number=1,Type=String,Description="P-value from Fisher\'s Exact Test">\n'
    '##FORMAT=<ID=RBQ,Number=1,Type=Integer,Description="Average quality of reference-supporting bases (qual1)">\n'
    '##FORMAT=<ID=ABQ,Number=1,Type=Integer,Description="Average quality of variant-supporting bases (qual2)">\n'
    '##FORMAT=<ID=RDF,Number=1,Type=Integer,Description="Depth of reference-supporting bases on forward strand (reads1plus)">\n'
    '##FORMAT=<ID=RDR,Number=1,Type=Integer,Description="Depth of reference-supporting bases on reverse strand (reads1minus)">\n'
    '##FORMAT=<ID=ADF,Number=1,Type=Integer,Description="Depth of variant-supporting bases on forward strand (reads2plus)">\n'
    '##FORMAT=<ID=ADR,Number=1,Type=Integer,Description="Depth of variant-supporting bases on reverse strand (reads2minus)">\n'
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSample1\n"
)


class Params(object):
    """mpileup2snp options (VarScan defaults, then the pipeline's ExtraParams on top)."""

    def __init__(self, min_coverage=8, min_reads2=2, min_avg_qual=15, min_var_freq=0.2, min_freq_for_hom=0.75, p_value=0.99,
                 strand_filter=1):
        self.min_coverage, self.min_reads2, self.min_avg_qual = min_coverage, min_reads2, min_avg_qual
        self.min_var_freq, self.min_freq_for_hom, self.p_value, self.strand_filter = min_var_freq, min_freq_for_hom, p_value, strand_filter


PIPELINE_DEFAULTS = dict(min_avg_qual=15, min_var_freq=0.90, min_reads2=5)      # snppipeline.conf:199


# ---- FishersExact: hypergeometric terms from a table of log-factorials ------------------------------------------------
_LOGFACT = [0.0]


def _lf(n):
    while len(_LOGFACT) <= n:
        _LOGFACT.append(_LOGFACT[-1] + math.log(len(_LOGFACT)))
    return _LOGFACT[n]


def _term(a, b, c, d):
    return math.exp(_lf(a + b) + _lf(c + d) + _lf(a + c) + _lf(b + d) - (_lf(a) + _lf(b) + _lf(c) + _lf(d) + _lf(a + b + c + d)))


def right_tailed_p(a, b, c, d):
    p = _term(a, b, c, d)
    for _ in range(min(c, b)):
        a, b, c, d = a + 1, b - 1, c - 1, d + 1
        p += _term(a, b, c, d)
    return p


def two_tailed_p(a, b, c, d):
    base = _term(a, b, c, d)
    p = base
    x = (a, b, c, d)
    for _ in range(min(a, d)):
        x = (x[0] - 1, x[1] + 1, x[2] + 1, x[3] - 1)
        t = _term(*x)
        if t <= base:
            p += t
    x = (a, b, c, d)
    for _ in range(min(b, c)):
        x = (x[0] + 1, x[1] - 1, x[2] - 1, x[3] + 1)
        t = _term(*x)
        if t <= base:
            p += t
    return p


def significance(reads1, reads2):
    """VarScan.getSignificance(obsReads1, obsReads2): the observed pair against what a 0.1 % error rate would give."""
    coverage = reads1 + reads2
    exp2 = int(coverage * 0.001)
    return right_tailed_p(coverage - exp2, exp2, reads1, reads2)


def java_sci(p):
    """DecimalFormat("0.####E0")."""
    if p == 0:
        return "0E0"
    e = int(math.floor(math.log10(p)))
    m = Decimal(p).scaleb(-e)
    if m >= 10:
        e += 1
        m = Decimal(p).scaleb(-e)
    elif m < 1:
        e -= 1
        m = Decimal(p).scaleb(-e)
    q = m.quantize(Decimal("0.0001"), rounding=ROUND_HALF_EVEN)
    if q >= 10:
        e += 1
        q = Decimal(p).scaleb(-e).quantize(Decimal("0.0001"), rounding=ROUND_HALF_EVEN)
    s = str(q).rstrip("0").rstrip(".")
    return "%sE%d" % (s, e)


def java_percent(reads2, total):
    """DecimalFormat("###.##") of the double reads2 / total * 100, then '%'."""
    v = (float(reads2) / float(total)) * 100.0
    s = str(Decimal(v).quantize(Decimal("0.01"), rounding=ROUND_HALF_EVEN))
    return s.rstrip("0").rstrip(".") + "%"


# ---- VarScan.getReadCounts ---------------------------------------------------------------------------------------------
class Counts(object):
    def __init__(self):
        self.ref = [0, 0, 0]                    # forward, reverse, quality sum   ('.' / ',' at quality >= min)
        self.alt = {}                           # 'A','C','G','T' -> [forward, reverse, quality sum]
        self.indel = 0                          # reads carrying +n / -n (no quality of their own: always counted)

    def total(self):
        return self.ref[0] + self.ref[1] + sum(v[0] + v[1] for v in self.alt.values()) + self.indel


def quality_depth(quals, min_qual):
    """VarScan.qualityDepth: quality characters at or above the threshold — every read, whatever its base."""
    return sum(1 for q in quals if q - 33 >= min_qual)


def read_counts(bases, quals, min_qual):
    """Walk the read-base string with a quality cursor j (bases: bytes, quals: bytes)."""
    c = Counts()
    i, j, n = 0, 0, len(bases)
    while i < n:
        ch = bases[i]
        q = quals[j] - 33 if j < len(quals) else 0
        if ch == 0x2E or ch == 0x2C:                                     # . ,
            if q >= min_qual:
                c.ref[0 if ch == 0x2E else 1] += 1
                c.ref[2] += q
            j += 1
        elif ch in b"ACGTacgt":
            if q >= min_qual:
                key = chr(ch).upper()
                v = c.alt.setdefault(key, [0, 0, 0])
                v[0 if ch < 0x61 else 1] += 1
                v[2] += q
            j += 1
        elif ch == 0x2B or ch == 0x2D:                                   # + - : digits, then that many bases; no quality
            k = i + 1
            size = 0
            while k < n and 0x30 <= bases[k] <= 0x39:
                size = size * 10 + bases[k] - 0x30
                k += 1
            if k > i + 1:
                c.indel += 1
                i = k + size - 1
        elif ch == 0x4E or ch == 0x6E or ch == 0x2A:                     # N n * : not counted, but they own a quality
            j += 1
        elif ch == 0x5E:                                                 # ^ : the next byte is a mapping quality
            i += 1
        # '$' and anything else: skipped, no quality consumed
        i += 1
    return c


# ---- VarScan.callPosition (SNP part) + CallMpileup's strand filter and VCF line --------------------------------------
def call_line(ref, depth, bases, quals, prm):
    """One pileup line -> None or a dict with everything the VCF line shows."""
    if depth < prm.min_coverage:
        return None
    dp = quality_depth(quals, prm.min_avg_qual)
    if dp < prm.min_coverage:
        return None
    c = read_counts(bases, quals, prm.min_avg_qual)
    ref = ref.upper()
    total = c.total()
    reads1 = c.ref[0] + c.ref[1]
    best = None
    for allele in sorted(c.alt):
        if allele == ref:
            continue
        f, r, qs = c.alt[allele]
        reads2 = f + r
        if reads2 == 0:
            continue
        avg2 = qs // reads2
        freq = float(reads2) / float(total)
        if reads2 >= prm.min_reads2 and avg2 >= prm.min_avg_qual and freq >= prm.min_var_freq:
            p = significance(reads1, reads2)
            if p <= prm.p_value and (best is None or reads2 > best["AD"]):
                best = dict(ALT=allele, AD=reads2, ADF=f, ADR=r, ABQ=avg2, p=p)
    if best is None:
        return None
    best.update(REF=ref, SDP=depth, DP=dp, RD=reads1, RDF=c.ref[0], RDR=c.ref[1], RBQ=(c.ref[2] // reads1 if reads1 else 0), total=total)
    best["hom"] = float(best["AD"]) / float(total) >= prm.min_freq_for_hom
    best["FILTER"] = "PASS"
    if prm.strand_filter:
        var_plus = float(best["ADF"]) / float(best["AD"])
        if (var_plus < 0.10 or var_plus > 0.90) and reads1 > 1:
            ref_plus = float(c.ref[0]) / float(reads1)
            if two_tailed_p(c.ref[0], c.ref[1], best["ADF"], best["ADR"]) < 0.01 and 0.10 <= ref_plus <= 0.90:
                best["FILTER"] = "str10"
    return best


def vcf_row(chrom, pos, r):
    """The data line for one called site (r: the dict of call_line, or one rebuilt from a fixture line)."""
    p = r["p"]
    gq = 255 if p <= 0 else min(255, int(-10.0 * math.log10(p)))
    hom = r["hom"]
    sample = ":".join([("1/1" if hom else "0/1"), str(gq), str(r["SDP"]), str(r["DP"]), str(r["RD"]), str(r["AD"]),
                       java_percent(r["AD"], r["total"]), java_sci(p), str(r["RBQ"]), str(r["ABQ"]), str(r["RDF"]), str(r["RDR"]),
                       str(r["ADF"]), str(r["ADR"])])
    info = "ADP=%d;WT=0;HET=%d;HOM=%d;NC=0" % (r["DP"], 0 if hom else 1, 1 if hom else 0)
    return "\t".join([chrom, pos, ".", r["REF"], r["ALT"], ".", r["FILTER"], info, "GT:GQ:SDP:DP:RD:AD:FREQ:PVAL:RBQ:ABQ:RDF:RDR:ADF:ADR", sample]) + "\n"


def mpileup2snp(data, prm):
    """Whole pileup (bytes) -> the text VarScan prints.  Lines are split on TAB (String.split("\\t")); a line needs six
    non-empty leading columns."""
    out = [VCF_HEADER % {"q": prm.min_avg_qual}]
    for line in re.split(b"\r\n|\r|\n", data):                 # BufferedReader.readLine(): a line ends at LF, CR or CR LF
        if not line:
            continue
        f = line.split(b"\t")
        while f and f[-1] == b"":
            f.pop()                                            # Java drops trailing empty strings
        if len(f) < 6 or not (f[0] and f[1] and f[2] and f[3]):
            raise ValueError("Invalid format for pileup: %r" % line[:60])
        if len(f[2]) != 1 or not f[3].isdigit() or len(f[3]) > 9:
            raise ValueError("refused (this build and its oracle): reference column longer than one byte, or a depth that is not a plain integer: %r" % line[:60])
        r = call_line(f[2].decode("latin-1"), int(f[3]), f[4], f[5], prm)
        if r is not None:
            out.append(vcf_row(f[0].decode("latin-1"), f[1].decode("latin-1"), r))
    return "".join(out)


def row_from_fixture_line(line):
    """Invert one bundled var.flt.vcf data line into the dict vcf_row takes (FREQ's denominator is searched: the smallest
    total >= RD + AD whose percentage text matches)."""
    f = line.rstrip("\n").split("\t")
    v = dict(zip(f[8].split(":"), f[9].split(":")))
    r = dict((k, int(v[k])) for k in ("SDP", "DP", "RD", "AD", "RBQ", "ABQ", "RDF", "RDR", "ADF", "ADR"))
    r.update(REF=f[3], ALT=f[4], FILTER=f[6], hom=v["GT"] == "1/1", p=significance(r["RD"], r["AD"]))
    total = r["RD"] + r["AD"]
    while java_percent(r["AD"], total) != v["FREQ"]:
        total += 1
        if total > 4 * (r["RD"] + r["AD"]) + 64:
            raise ValueError("no denominator reproduces FREQ in %r" % line)
    r["total"] = total
    return f[0], f[1], r
