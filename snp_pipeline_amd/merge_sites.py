"""merge_sites subcommand: union of the SNP positions of all samples -> snplist.txt.

Host mirror of snppipeline/merge_sites.py:12-133.  The union itself (sort + unique + per-site carrier lists) runs on
the device (``Device.merge_sites``, csrc/regions.hip); this file parses the VCFs' CHROM/POS columns, applies the
``--maxsnps`` sample exclusion and writes the two text outputs.
"""
from __future__ import print_function

import os

import numpy as np

from . import utils


def merge_site_lists(dev, per_sample_sites):
    """per_sample_sites: list (sorted-dir order) of iterables of (chrom, pos).  Returns (keys, carriers) with keys
    sorted as the reference sorts (chrom str, pos int) tuples and carriers = list of sample indices per key."""
    contigs = sorted({c for sites in per_sample_sites for c, _ in sites})
    cid = {c: i for i, c in enumerate(contigs)}
    n = sum(len(s) for s in per_sample_sites)
    keys = np.empty(n, dtype=np.uint64)
    samp = np.empty(n, dtype=np.uint32)
    i = 0
    for si, sites in enumerate(per_sample_sites):
        for c, p in sites:
            if not 0 <= p < (1 << 32):
                raise ValueError("VCF position %r out of range" % (p,))
            keys[i] = (cid[c] << 32) | p
            samp[i] = si
            i += 1
    uniq, off, car = dev.merge_sites(keys, samp)
    out_keys = [(contigs[int(k) >> 32], int(k) & 0xFFFFFFFF) for k in uniq]
    carriers = [car[off[j]:off[j + 1]] for j in range(len(uniq))]
    return out_keys, carriers


def merge_site_arrays(dev, per_sample):
    """The same union from arrays: per_sample = list (sorted-dir order) of (contig names, contig index per record, position
    per record) as utils.read_vcf_site_arrays returns them.  Returns (sorted contig names, unique keys (contig << 32 | pos),
    carrier offsets, carrier sample indices) — numpy all the way, for snpgpu_write_snplist."""
    contigs = sorted({c for names, _, _ in per_sample for c in names})
    cid = {c: i for i, c in enumerate(contigs)}
    keys, samp = [], []
    for si, (names, cidx, pos) in enumerate(per_sample):
        if len(pos) == 0:
            continue
        if pos.min() < 0 or pos.max() >= (1 << 32):
            raise ValueError("VCF position out of range")
        lut = np.asarray([cid[c] for c in names], dtype=np.uint64)
        keys.append((lut[cidx] << np.uint64(32)) | pos.astype(np.uint64))
        samp.append(np.full(len(pos), si, dtype=np.uint32))
    if not keys:
        return contigs, np.zeros(0, np.uint64), np.zeros(1, np.uint32), np.zeros(0, np.uint32)
    uniq, off, car = dev.merge_sites(np.concatenate(keys), np.concatenate(samp))
    return contigs, uniq, off, car


def write_snplist(path, contigs, uniq, off, car, sample_names):
    """utils.write_list_of_snps through the library's host formatter (csrc/vcf_in.hip)."""
    from . import _lib as L

    def blob(strings):
        raw = [x.encode("utf-8") for x in strings]
        o = np.zeros(len(raw) + 1, dtype=np.uint64)
        if raw:
            np.cumsum([len(b) for b in raw], out=o[1:])
        return b"".join(raw), o

    cb, co = blob(contigs)
    sb, so = blob(sample_names)
    uniq = np.ascontiguousarray(uniq, dtype=np.uint64)
    off = np.ascontiguousarray(off, dtype=np.uint32)
    car = np.ascontiguousarray(car, dtype=np.uint32)
    rc = L.load().snpgpu_write_snplist(os.fsencode(path), cb, co.ctypes.data, uniq.ctypes.data, len(uniq), off.ctypes.data, car.ctypes.data,
                                       sb, so.ctypes.data)
    if rc == L.E_IO:
        raise IOError("cannot write %s" % path)
    if rc != 0:
        raise RuntimeError("snpgpu_write_snplist failed (%d)" % rc)


def merge_sites(args):
    """Entry point of ``cfsan_snp_pipeline merge_sites`` (cfsan_snp_pipeline.py:329-340)."""
    utils.print_log_header()
    utils.print_arguments(args)

    sample_directories_list_path = args.sampleDirsFile
    if utils.verify_non_empty_input_files("File of sample directories", [sample_directories_list_path]) > 0:
        utils.global_error(None)
    with open(sample_directories_list_path, "r") as f:
        unsorted_dirs = [line.rstrip() for line in f]
    unsorted_dirs = [d for d in unsorted_dirs if d]
    sorted_dirs = sorted(unsorted_dirs)

    snp_list_file_path = args.snpListFile
    vcf_file_name = args.vcfFileName
    list_of_vcf_files = [os.path.join(d, vcf_file_name) for d in sorted_dirs]
    bad = utils.verify_non_empty_input_files("VCF file", list_of_vcf_files)
    if bad == len(list_of_vcf_files):
        utils.global_error("Error: all %d VCF files were missing or empty." % bad)
    elif bad > 0:
        utils.sample_error("Error: %d VCF files were missing or empty." % bad, continue_possible=True)

    if not (args.forceFlag or utils.target_needs_rebuild(list_of_vcf_files, snp_list_file_path)):
        utils.verbose_print("SNP list %s has already been freshly built.  Use the -f option to force a rebuild." % snp_list_file_path)
        return

    names, site_arrays, excluded_dirs = [], [], set()
    for sample_dir, vcf_file_path in zip(sorted_dirs, list_of_vcf_files):
        if not os.path.isfile(vcf_file_path) or os.path.getsize(vcf_file_path) == 0:
            continue
        utils.verbose_print("Processing VCF file %s" % vcf_file_path)
        sample_name = os.path.basename(os.path.dirname(vcf_file_path))
        contig_names, cidx, pos = utils.read_vcf_site_arrays(vcf_file_path)
        if args.maxSnps >= 0:
            n_snps = len(np.unique((cidx.astype(np.int64) << 40) ^ pos)) if len(pos) else 0      # size of the reference's snp_set
            if n_snps > args.maxSnps:
                utils.verbose_print("Excluding sample %s having %d snps." % (sample_name, n_snps))
                excluded_dirs.add(sample_dir)
                continue
        names.append(sample_name)
        site_arrays.append((contig_names, cidx, pos))

    from .device import default_device
    contigs, uniq, off, car = merge_site_arrays(default_device(), site_arrays)
    utils.verbose_print('Found %d snp positions across %d sample vcf files.' % (len(uniq), len(list_of_vcf_files)))
    write_snplist(snp_list_file_path, contigs, uniq, off, car, names)

    with open(args.filteredSampleDirsFile, "w") as f:
        for sample_dir in unsorted_dirs:                      # original order (merge_sites.py:127-131)
            if sample_dir not in excluded_dirs:
                f.write("%s\n" % sample_dir)
