"""Phase-1 site calling (`VarScan mpileup2snp --output-vcf 1`) on the device: var.flt.vcf from reads.all.pileup.

Replaces the VarScan v2.3.9 jar that snppipeline/call_sites.py:89-108 shells out to (third-party, not in the reference
tree).  The device pass (csrc/varscan.hip through ``Device.varscan_file``) does the per-line counting and the
min-coverage / min-reads2 / min-avg-qual / min-var-freq tests and returns one record per passing (line, allele); this
module parses the options, and hands the few records that come back to the library's host code (csrc/varscan_rows.hip):
Fisher's exact test against a 0.1 % error model (``PVAL``, ``GQ``, ``--p-value``), the strand filter, the homozygous
threshold, and VarScan's VCF 4.1 text.  The text and the arithmetic are pinned by the 69 019 data lines of the reference's
bundled var.flt.vcf fixtures (tests/test_host_cpu.py feeds every line's own counts back in); how unusual read-base strings
are counted is not (see oracle/varscan_oracle.py and DESIGN.md).
"""
from __future__ import print_function

import mmap
import shlex

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


def format_rows(records, pileup_bytes, opts):
    """records: Device.varscan_file's array (file order); pileup_bytes: the file (bytes, or an mmap).  Returns (the data lines as
    bytes, their number) — Fisher's exact test, --p-value, the strand filter, GT and the VCF text are the library's host code
    (csrc/varscan_rows.hip)."""
    import ctypes as C
    import numpy as np
    lib = L.load()
    fin = L.VarscanFinish(opts.p_value, opts.min_freq_for_hom, 1 if opts.strand_filter else 0, 0)
    recs = np.ascontiguousarray(records)
    view = np.frombuffer(pileup_bytes, dtype=np.uint8)
    n_rows = C.c_uint32()
    cap = 256 * len(recs) + 4096
    try:
        while True:
            out = C.create_string_buffer(cap)
            need = lib.snpgpu_varscan_format_rows(recs.ctypes.data, len(recs), view.ctypes.data if len(view) else None, len(view), C.byref(fin),
                                                  out, cap, C.byref(n_rows))
            if need <= cap:
                return out.raw[:need], n_rows.value
            cap = need
    finally:
        del view                                               # an mmap cannot be closed while a view of it exists


def _write_vcf(vcf_path, pileup_path, records, opts):
    n_rows = 0
    with open(vcf_path, "wb") as out:                           # contig names pass through byte for byte
        out.write(header_text(opts.min_avg_qual).encode("ascii"))
        if len(records):
            with open(pileup_path, "rb") as f:
                view = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
                try:
                    text, n_rows = format_rows(records, view, opts)
                    out.write(text)
                finally:
                    view.close()
    return n_rows


def mpileup2snp_files(device, pileup_paths, vcf_paths, opts, group=6):
    """Many samples: streamed device calls over groups of `group` files in a helper thread (the library releases the GIL), while
    this thread turns the records of the group before into VCF files.  Returns [(lines, sites written) or the exception of
    that sample] in input order.  A sample whose VCF cannot be written (unwritable directory, full disk) carries that exception
    and the others go on; whatever happens, the helper thread has been stopped and joined when this function returns."""
    import queue
    import threading
    n = len(pileup_paths)
    done = queue.Queue(maxsize=2)
    stop = threading.Event()

    def put(item):
        while not stop.is_set():
            try:
                done.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def produce():
        try:
            for g0 in range(0, n, group):
                if stop.is_set() or not put((g0, device.varscan_files(pileup_paths[g0:g0 + group], opts.device_params()))):
                    return
        except BaseException as err:                            # noqa: B902 — handed to the consumer
            put((None, err))
        put((None, None))

    producer = threading.Thread(target=produce)
    producer.start()
    results = [None] * n
    failure = None
    try:
        while True:
            g0, batch = done.get()
            if g0 is None:
                if batch is None:
                    break
                failure = batch
                continue
            for k, (records, n_lines) in enumerate(batch):
                i = g0 + k
                if isinstance(records, Exception):
                    results[i] = records
                    continue
                try:
                    results[i] = (n_lines, _write_vcf(vcf_paths[i], pileup_paths[i], records, opts))
                except Exception as err:                        # noqa: B902 — this sample's error; the stream goes on
                    results[i] = err
    finally:
        stop.set()                                              # on any exit: no producer left blocked on the queue or inside the device
        while producer.is_alive():
            try:
                done.get(timeout=0.05)
            except queue.Empty:
                pass
        producer.join()
    if failure is not None:
        raise failure
    return results


def mpileup2snp(device, pileup_path, vcf_path, opts):
    """reads.all.pileup -> var.flt.vcf.  Returns (lines in the pileup, sites written)."""
    records, n_lines = device.varscan_file(pileup_path, opts.device_params())
    return n_lines, _write_vcf(vcf_path, pileup_path, records, opts)
