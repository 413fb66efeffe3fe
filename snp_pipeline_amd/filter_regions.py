"""filter_regions subcommand: drop SNPs at contig edges and in abnormally dense windows.

Host mirror of snppipeline/filter_regions.py (filter_regions :74, filter_regions_across_samples :205,
filter_regions_per_sample :300, collect_dense_regions :386, write_outgroup_... :431,
write_preserved_and_removed_vcf_files :460).  File handling, freshness and the error protocol live here; the
arithmetic — dense-window test, interval merge, position-in-region classification — runs in the HIP kernels
behind ``Device.dense_windows / merge_regions / in_regions`` (csrc/regions.hip).
"""
from __future__ import print_function

import itertools
import os
import shutil
import sys

import numpy as np

from . import utils

UNKNOWN_CONTIG_LENGTH = sys.maxsize          # filter_regions.py:417 fallback for contigs missing from the FASTA


def _edge_intervals(length, edge_length):
    if length <= 2 * edge_length:
        return [(0, length)]
    return [(0, edge_length), (length - edge_length, length)]


def sites_to_arrays(sites):
    """[(contig, pos), ...] -> (contig names in order of first appearance, contig index per record, position per record): the
    form utils.read_vcf_site_arrays returns."""
    names, index = [], {}
    for c, _ in sites:
        if c not in index:
            index[c] = len(names)
            names.append(c)
    return (names, np.fromiter((index[c] for c, _ in sites), dtype=np.uint32, count=len(sites)),
            np.fromiter((p for _, p in sites), dtype=np.int64, count=len(sites)))


def compute_bad_regions(dev, samples, contig_lengths, edge_length, max_snps_list, window_list, per_sample=False):
    """samples: list (one per non-outgroup sample) of [(contig, pos), ...] in file order, or of the array triples of
    sites_to_arrays / utils.read_vcf_site_arrays.

    Returns, for mode all, {contig: [(start, end), ...]} merged; for per_sample=True a list of such dicts.
    Every interval comes from the device: candidates from the dense-window kernel, the union from the merge
    kernel.  The host only lays out segments (one per sample and contig, numpy) and the contig-edge intervals (parameters,
    not data)."""
    samples = [smp if isinstance(smp, tuple) and len(smp) == 3 and isinstance(smp[1], np.ndarray) else sites_to_arrays(smp) for smp in samples]
    # group ids: mode all -> contig; mode each -> (sample, contig)
    group_of = {}
    groups = []

    def gid(sample_idx, contig):
        key = (sample_idx if per_sample else -1, contig)
        if key not in group_of:
            group_of[key] = len(groups)
            groups.append(key)
        return group_of[key]

    seg_positions, seg_group = [], []
    for si, (names, cidx, pos) in enumerate(samples):
        if len(pos) == 0:
            continue
        # one segment per contig, contigs in order of first appearance, positions in file order (a stable sort by contig index)
        order = np.argsort(cidx, kind="stable")
        sorted_c = cidx[order]
        cuts = np.flatnonzero(np.diff(sorted_c)) + 1
        starts = np.concatenate(([0], cuts))
        ends = np.concatenate((cuts, [len(order)]))
        ordered_pos = pos[order]
        for a, b in zip(starts.tolist(), ends.tolist()):
            seg_positions.append(ordered_pos[a:b])
            seg_group.append(gid(si, names[int(sorted_c[a])]))
    g_list, s_list, e_list = [], [], []
    for (si, contig), g in sorted(group_of.items(), key=lambda kv: kv[1]):
        for a, b in _edge_intervals(contig_lengths.get(contig, UNKNOWN_CONTIG_LENGTH), edge_length):
            g_list.append(g), s_list.append(a), e_list.append(b)
    if seg_positions:
        seg_off = np.zeros(len(seg_positions) + 1, dtype=np.uint32)
        seg_off[1:] = np.cumsum([len(p) for p in seg_positions])
        flat = np.concatenate(seg_positions).astype(np.int64, copy=False)
        cs, ce, cseg = dev.dense_windows(flat, seg_off, max_snps_list, window_list)
        seg_group_arr = np.asarray(seg_group, dtype=np.uint32)
        g_all = np.concatenate([np.asarray(g_list, dtype=np.uint32), seg_group_arr[cseg]])
        s_all = np.concatenate([np.asarray(s_list, dtype=np.int64), cs])
        e_all = np.concatenate([np.asarray(e_list, dtype=np.int64), ce])
    else:
        g_all, s_all, e_all = (np.asarray(g_list, dtype=np.uint32), np.asarray(s_list, dtype=np.int64),
                               np.asarray(e_list, dtype=np.int64))
    mg, ms, me = dev.merge_regions(g_all, s_all, e_all)
    if per_sample:
        out = [dict() for _ in samples]
        for g, a, b in zip(mg.tolist(), ms.tolist(), me.tolist()):
            si, contig = groups[g]
            out[si].setdefault(contig, []).append((a, b))
        return out
    out = {}
    for g, a, b in zip(mg.tolist(), ms.tolist(), me.tolist()):
        out.setdefault(groups[g][1], []).append((a, b))
    return out


def classify_records(dev, sites, regions):
    """sites: [(contig, pos)] or an array triple; regions: {contig: merged [(start, end)]}.  Returns a bool array, True = removed."""
    names, cidx, pos = sites if isinstance(sites, tuple) and len(sites) == 3 and isinstance(sites[1], np.ndarray) else sites_to_arrays(sites)
    if len(pos) == 0:
        return np.zeros(0, dtype=bool)
    contigs = sorted(regions)
    cid = {c: i for i, c in enumerate(contigs)}
    reg_off = np.zeros(len(contigs) + 1, dtype=np.uint32)
    rs, re_ = [], []
    for i, c in enumerate(contigs):
        rs.extend(a for a, _ in regions[c])
        re_.extend(b for _, b in regions[c])
        reg_off[i + 1] = len(rs)
    lut = np.asarray([cid[c] for c in names], dtype=np.uint32)                  # KeyError like bad_regions_dict[contig]
    return dev.in_regions(lut[cidx], pos.astype(np.int64, copy=False), reg_off, rs, re_)


def removed_flags(dev, contigs, contig_lengths, rec_sample, rec_cid, rec_pos, n_samples, edge_length, max_snps_list, window_list, per_sample=False):
    """The whole region step for ALL samples at once, on arrays (10 000 samples x 1 500 records at BASELINE configs[4]: no Python
    loop per sample, three device calls in total).  Records of the samples that take part (outgroup samples left out), grouped
    by sample: rec_sample (index 0 .. n_samples-1, non-decreasing), rec_cid (index into `contigs`), rec_pos.  Returns the
    removed flag of every record: it lies in a bad region of its contig — contig edges (filter_regions.py:416-422), dense windows
    (find_dense_regions :17-71) — of any sample (mode all) or of its own sample (mode each)."""
    n = len(rec_pos)
    if n == 0:
        return np.zeros(0, dtype=bool)
    n_c = max(1, len(contigs))
    rec_sample = np.asarray(rec_sample, dtype=np.int64)
    rec_cid64 = np.asarray(rec_cid, dtype=np.int64)
    rec_pos = np.asarray(rec_pos, dtype=np.int64)
    # one segment per (sample, contig): records in file order inside (the device sorts the positions of a segment itself)
    key = rec_sample * n_c + rec_cid64
    order = None
    if (key[1:] < key[:-1]).any():                            # a VCF whose contigs are interleaved: group them, file order kept
        order = np.argsort(key, kind="stable")
        key = key[order]
    cuts = np.flatnonzero(key[1:] != key[:-1]) + 1
    seg_start = np.concatenate(([0], cuts))
    seg_off = np.concatenate((seg_start, [n])).astype(np.uint32)
    seg_key = key[seg_start]
    seg_cid = seg_key % n_c
    seg_group = (seg_key if per_sample else seg_cid).astype(np.int64)        # mode each: (sample, contig); mode all: contig
    flat = rec_pos if order is None else rec_pos[order]
    cs, ce, cseg = dev.dense_windows(flat, seg_off, max_snps_list, window_list)
    # contig-edge intervals, once per group that has records
    groups, first = np.unique(seg_group, return_index=True)
    lengths = np.asarray([contig_lengths.get(c, UNKNOWN_CONTIG_LENGTH) for c in contigs] + [UNKNOWN_CONTIG_LENGTH], dtype=np.int64)
    g_len = lengths[seg_cid[first]]
    whole = g_len <= 2 * edge_length
    eg = np.concatenate((groups, groups[~whole]))
    es = np.concatenate((np.zeros(len(groups), np.int64), (g_len - edge_length)[~whole]))
    ee = np.concatenate((np.where(whole, g_len, edge_length), g_len[~whole]))
    # dense group ids for the device (uint32): rank among the groups present
    rank_of_seg = np.searchsorted(groups, seg_group)
    g_all = np.concatenate((np.searchsorted(groups, eg), rank_of_seg[cseg])).astype(np.uint32)
    mg, ms, me = dev.merge_regions(g_all, np.concatenate((es, cs)), np.concatenate((ee, ce)))
    reg_off = np.zeros(len(groups) + 1, dtype=np.uint32)
    np.cumsum(np.bincount(mg, minlength=len(groups)), out=reg_off[1:])
    rec_group = np.repeat(rank_of_seg, np.diff(seg_off.astype(np.int64))).astype(np.uint32)
    flags = dev.in_regions(rec_group, flat, reg_off, ms, me)
    if order is not None:
        out = np.empty(n, dtype=bool)
        out[order] = flags
        return out
    return flags


_STRUCTURED = ("##INFO=", "##FORMAT=", "##FILTER=", "##ALT=", "##contig=")


def reorder_header(header_lines):
    """Header as PyVCF3's Writer re-emits a Reader template: plain ``##key=value`` lines, then INFO, FORMAT,
    FILTER, ALT, contig, then the ``#CHROM`` line (pinned by the lambda var.flt_preserved.vcf fixtures)."""
    plain = [h for h in header_lines if h.startswith("##") and not h.startswith(_STRUCTURED)]
    out = list(plain)
    for prefix in ("##INFO=", "##FORMAT=", "##FILTER=", "##ALT=", "##contig="):
        out.extend(h for h in header_lines if h.startswith(prefix))
    out.extend(h for h in header_lines if h.startswith("#") and not h.startswith("##"))
    return out


class DataLines(object):
    """The data lines of a plain VCF as ONE byte array plus the line lengths (1 500 lines per file, 10 000 files: no Python
    object per line).  ``select(keep)`` is the text of the lines `keep` marks, in file order."""

    def __init__(self, data, lengths):
        self.data, self.lengths = data, lengths

    def __len__(self):
        return len(self.lengths)

    def select(self, keep):
        return self.data[np.repeat(np.asarray(keep, dtype=bool), self.lengths)].tobytes()


def _write_vcf(path, header, data_lines, keep=None):
    """header lines + the data lines `keep` marks (all of them when None)."""
    if isinstance(data_lines, DataLines):
        with open(path, "wb") as f:
            f.write("".join(header).encode("ascii"))
            f.write(data_lines.data.tobytes() if keep is None else data_lines.select(keep))
        return
    with open(path, "w") as f:
        f.writelines(header)
        f.writelines(data_lines if keep is None else itertools.compress(data_lines, np.asarray(keep, dtype=bool).tolist()))


def write_outgroup_preserved_and_removed_vcf_files(vcf_file_path, header):
    preserved = vcf_file_path[:-4] + "_preserved.vcf"
    removed = vcf_file_path[:-4] + "_removed.vcf"
    try:
        _write_vcf(removed, reorder_header(header), [])
    except (IOError, OSError):
        if os.path.exists(removed):
            os.remove(removed)
        utils.sample_error("Error: Cannot create the file for removed SNPs: %s." % removed, continue_possible=True)
        return
    shutil.copyfile(vcf_file_path, preserved)


def write_preserved_and_removed_vcf_files(vcf_file_path, header, data_lines, removed_flags):
    preserved = vcf_file_path[:-4] + "_preserved.vcf"
    removed = vcf_file_path[:-4] + "_removed.vcf"
    hdr = reorder_header(header)
    try:
        _write_vcf(preserved, hdr, data_lines, ~np.asarray(removed_flags, dtype=bool))
    except (IOError, OSError):
        if os.path.exists(preserved):
            os.remove(preserved)
        utils.sample_error("Error: Cannot create the file for preserved SNPs: %s." % preserved, continue_possible=True)
        return
    try:
        _write_vcf(removed, hdr, data_lines, np.asarray(removed_flags, dtype=bool))
    except (IOError, OSError):
        if os.path.exists(removed):
            os.remove(removed)
        utils.sample_error("Error: Cannot create the file for removed SNPs: %s." % removed, continue_possible=True)


def _read_vcf(vcf_path):
    """(header lines, data lines, sites as arrays): the columns come from the library's reader (utils.read_vcf_site_arrays), the
    lines from one readlines(); a file outside the reader's plain case is read by utils.read_vcf_sites, line by line."""
    names, cidx, pos = utils.read_vcf_site_arrays(vcf_path)     # (a file without a "#CHROM" line: utils.read_vcf_sites says what becomes of it)
    # the plain case on bytes: ASCII, "\n" line ends, the header first, one record per remaining line
    with open(vcf_path, "rb") as f:
        raw = f.read()
    if raw.isascii() and b"\r" not in raw and len(raw):
        arr = np.frombuffer(raw, dtype=np.uint8)
        starts = np.concatenate(([0], np.flatnonzero(arr == 10) + 1))
        if starts[-1] == len(arr):
            starts = starts[:-1]
        is_header = arr[starts] == 35                          # '#'
        n_header = int(np.count_nonzero(is_header))
        if is_header[:n_header].all() and len(starts) - n_header == len(pos):
            first = int(starts[n_header]) if n_header < len(starts) else len(arr)
            lengths = np.diff(np.concatenate((starts[n_header:], [len(arr)])))
            return raw[:first].decode("ascii").splitlines(True), DataLines(arr[first:], lengths), (names, cidx, pos)
    with open(vcf_path, "r") as f:
        lines = f.readlines()
    header = [ln for ln in lines if ln.startswith("#")]
    data_lines = [ln for ln in lines if not ln.startswith("#") and not ln.isspace()]
    if len(data_lines) != len(pos):                             # (cannot happen for the files the plain reader accepts)
        header, data_lines, sites = utils.read_vcf_sites(vcf_path)
        return header, data_lines, sites_to_arrays(sites)
    return header, data_lines, (names, cidx, pos)


def filter_regions(args):
    """Entry point of ``cfsan_snp_pipeline filter_regions`` (cfsan_snp_pipeline.py:309-324)."""
    utils.print_log_header()
    utils.print_arguments(args)

    sample_directories_list_path = args.sampleDirsFile
    ref_fasta_path = args.refFastaFile
    force_flag = args.forceFlag
    vcf_file_name = args.vcfFileName
    edge_length = args.edgeLength
    window_size_list = args.windowSizeList
    max_num_snps_list = args.maxSnpsList
    out_group_list_path = args.outGroupFile
    filter_across_samples = args.mode == "all"

    if utils.verify_non_empty_input_files("File of sample directories", [sample_directories_list_path]) > 0:
        utils.global_error(None)
    with open(sample_directories_list_path, "r") as f:
        dirs = [line.rstrip() for line in f]
    sorted_dirs = sorted(d for d in dirs if d)
    list_of_vcf_files = [os.path.join(d, vcf_file_name) for d in sorted_dirs]
    bad = utils.verify_non_empty_input_files("VCF file", list_of_vcf_files)
    if bad == len(list_of_vcf_files):
        utils.global_error("Error: all %d VCF files were missing or empty." % bad)
    elif bad > 0:
        utils.sample_error("Error: %d VCF files were missing or empty." % bad, continue_possible=True)
    if utils.verify_non_empty_input_files("Reference file", [ref_fasta_path]) > 0:
        utils.global_error(None)

    outgroup = []
    if out_group_list_path is not None:
        if utils.verify_non_empty_input_files("File of outgroup samples", [out_group_list_path]) > 0:
            utils.global_error(None)
        try:
            with open(out_group_list_path, "r") as f:
                outgroup = sorted(line.rstrip() for line in f)
        except (IOError, OSError):
            utils.global_error("Error: Cannot open the file containing the list of outgroup samples!")
    try:
        contig_length_dict = utils.read_fasta_lengths(ref_fasta_path)
    except (IOError, OSError, UnicodeDecodeError):
        utils.global_error("Error: cannot open the reference fastq file, or fail to read the contigs in the reference fastq file.")

    # ---- which samples need a rebuild (filter_regions.py:246-256 / :339-350) -------------------------------
    common_inputs = [ref_fasta_path] + ([out_group_list_path] if out_group_list_path else [])
    if filter_across_samples:
        common_inputs = common_inputs + list_of_vcf_files
    # (make-style: a target is stale when it is missing, empty, or older than its newest source.  The newest source is found
    # once — the reference stats every VCF for every sample, 10^8 calls at 10 000 samples.)
    def newest(paths):
        return max([os.stat(p_).st_mtime for p_ in paths if os.path.isfile(p_)] or [float("-inf")])

    def stale(target, newest_source):
        return (not os.path.isfile(target) or os.path.getsize(target) == 0 or newest_source > os.stat(target).st_mtime)

    common_newest = newest(common_inputs)
    need_rebuild = {}
    for vcf_path in list_of_vcf_files:
        src = common_newest if filter_across_samples else max(common_newest, newest([vcf_path]))
        need_rebuild[vcf_path] = force_flag or stale(vcf_path[:-4] + "_preserved.vcf", src) or stale(vcf_path[:-4] + "_removed.vcf", src)
    if not any(need_rebuild.values()):
        utils.verbose_print("All preserved and removed vcf files are already freshly built.  Use the -f option to force a rebuild.")
        return

    # ---- read the VCFs (mode all reads every sample; mode each only those being rebuilt) ---------------------
    from .device import default_device
    dev = default_device()
    parsed = []          # (vcf_path, header, data_lines, sites)
    for vcf_path in list_of_vcf_files:
        if not filter_across_samples and not need_rebuild[vcf_path]:
            continue
        try:
            header, data_lines, sites = _read_vcf(vcf_path)
        except (IOError, OSError):
            utils.sample_error("Error: Cannot open the input vcf file: %s." % vcf_path, continue_possible=True)
            continue
        sample_id = utils.sample_id_from_file(vcf_path)
        utils.verbose_print("Processing sample %s" % sample_id)
        if sample_id in outgroup:
            if filter_across_samples or need_rebuild[vcf_path]:
                write_outgroup_preserved_and_removed_vcf_files(vcf_path, header)
            continue
        parsed.append((vcf_path, header, data_lines, sites))

    # every record of every sample that takes part, as arrays over one table of contig names
    contigs = sorted({c for _, _, _, (names, _, _) in parsed for c in names})
    cid = {c: i for i, c in enumerate(contigs)}
    cidx_all, pos_all, counts = [], [], []
    for _, _, _, (names, cidx, pos) in parsed:
        lut = np.asarray([cid[c] for c in names] + [0], dtype=np.int64)
        cidx_all.append(lut[cidx.astype(np.int64)])
        pos_all.append(np.asarray(pos, dtype=np.int64))
        counts.append(len(pos))
    n_rec = int(sum(counts))
    rec_sample = np.repeat(np.arange(len(parsed), dtype=np.int64), counts)
    removed = removed_flags(dev, contigs, contig_length_dict, rec_sample, np.concatenate(cidx_all) if n_rec else np.zeros(0, np.int64),
                            np.concatenate(pos_all) if n_rec else np.zeros(0, np.int64), len(parsed), edge_length, max_num_snps_list,
                            window_size_list, per_sample=not filter_across_samples)
    at = 0
    for (vcf_path, header, data_lines, _), k in zip(parsed, counts):
        if need_rebuild[vcf_path]:
            write_preserved_and_removed_vcf_files(vcf_path, header, data_lines, removed[at:at + k])
        at += k
