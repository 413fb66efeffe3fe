"""CPU restatement of the region filter, site merge and distance arithmetic.
TEST INFRASTRUCTURE ONLY — see ``oracle/pileup_oracle.py`` for the rules.

Restates (reference file:line):

* ``snppipeline/filter_regions.py:17-71``    find_dense_regions
* ``snppipeline/filter_regions.py:386-428``  collect_dense_regions (edges + rules)
* ``snppipeline/filter_regions.py:205-297, 300-383``  mode all / each region sets
* ``snppipeline/utils.py:1168-1282``         merge_regions
* ``snppipeline/utils.py:1285-1318``         in_region
* ``snppipeline/merge_sites.py:91-117`` + ``utils.py:1056-1070``  site union + snplist lines
* ``snppipeline/utils.py:1135-1165``         calculate_sequence_distance
* ``snppipeline/distance.py:76-115``         FASTA parse, id sort, both TSV layouts

Pinned by ``tests/golden/steps_vectors.json.gz`` (outputs of the real reference
functions, written by ``oracle/gen_golden.py``) and by the reference's bundled
ExpectedResults trees copied under ``tests/golden/fixtures``.
"""

import sys

UNKNOWN_CONTIG_LENGTH = sys.maxsize      # filter_regions.py:417


def dense_windows(max_snps, window, positions):
    """Candidate intervals before merging: sorted ``positions``; i is dense iff
    ``p[i+M]`` exists and ``p[i] + W - 1 >= p[i+M]``."""
    out = []
    for i in range(len(positions) - max_snps):
        a, b = positions[i], positions[i + max_snps]
        if a + window - 1 >= b:
            out.append((a, b))
    return out


def merge_regions(regions):
    """Sort, drop contained intervals, join overlapping or adjacent ones."""
    merged = []
    for start, end in sorted(regions):
        if merged:
            ls, le = merged[-1]
            if start >= ls and end <= le:
                continue
            if start <= le + 1 and end > le:
                merged[-1] = (ls, end)
                continue
        merged.append((start, end))
    return merged


def find_dense_regions(max_snps, window, positions):
    return merge_regions(dense_windows(max_snps, window, positions))


def in_region(pos, regions):
    return any(a <= pos <= b for a, b in regions)


def edge_regions(contig_length, edge_length):
    if contig_length <= 2 * edge_length:
        return [(0, contig_length)]
    return [(0, edge_length), (contig_length - edge_length, contig_length)]


def collect_dense_regions(sample_sites, bad, contig_lengths, edge_length, max_snps_list, window_list):
    """``sample_sites``: {contig: [pos,...]} of ONE sample (file order).  Adds
    that sample's edge + dense intervals to ``bad`` in place."""
    for contig, plist in sample_sites.items():
        if contig not in bad:
            bad[contig] = edge_regions(contig_lengths.get(contig, UNKNOWN_CONTIG_LENGTH), edge_length)
        srt = sorted(plist)
        for m, w in zip(max_snps_list, window_list):
            bad[contig].extend(find_dense_regions(m, w, srt))


def bad_regions(samples, contig_lengths, edge_length, max_snps_list, window_list, mode="all", outgroup=()):
    """``samples``: ordered list of (sample_id, [(contig,pos),...]).  Returns
    mode all: one {contig: merged regions}; mode each: {sample_id: {...}}.
    Outgroup samples contribute nothing (and are never filtered)."""
    def by_contig(sites):
        d = {}
        for c, p in sites:
            d.setdefault(c, []).append(p)
        return d

    if mode == "all":
        bad = {}
        for sid, sites in samples:
            if sid in outgroup:
                continue
            collect_dense_regions(by_contig(sites), bad, contig_lengths, edge_length, max_snps_list, window_list)
        return {c: merge_regions(r) for c, r in bad.items()}
    out = {}
    for sid, sites in samples:
        if sid in outgroup:
            continue
        bad = {}
        collect_dense_regions(by_contig(sites), bad, contig_lengths, edge_length, max_snps_list, window_list)
        out[sid] = {c: merge_regions(r) for c, r in bad.items()}
    return out


def merge_sites(samples, max_snps=-1):
    """``samples``: list of (sample_dir, sample_name, [(contig,pos),...]) in
    SORTED-dir order.  Returns (sorted [(key, [names...])], excluded dirs)."""
    sites = {}
    excluded = set()
    for sdir, name, recs in samples:
        uniq = set(recs)
        if max_snps >= 0 and len(uniq) > max_snps:
            excluded.add(sdir)
            continue
        for k in uniq:
            sites.setdefault(k, []).append(name)
    return [(k, sites[k]) for k in sorted(sites)], excluded


def snplist_text(merged):
    return "".join("%s\t%d\t%d\t%s\n" % (k[0], k[1], len(names), "\t".join(names)) for k, names in merged)


_ACGT = frozenset("ACGT")


def sequence_distance(a, b):
    a, b = a.upper(), b.upper()
    n = 0
    for i in range(len(a)):
        x, y = a[i], b[i]
        if x in _ACGT and y in _ACGT and x != y:
            n += 1
    return n


def parse_snpma(text):
    seqs = {}
    cur = None
    for line in text.split("\n"):
        if line.startswith(">"):
            cur = line.lstrip(">")
            seqs[cur] = ""
        elif cur is not None or line:
            if cur is None:                     # distance.py:84: curr_sample is read before any header has bound it
                raise UnboundLocalError("local variable 'curr_sample' referenced before assignment")
            seqs[cur] += line
    return seqs


def distance_tables(seqs):
    """Returns (ids sorted, {(i,j): d}) like distance.py:90-98."""
    ids = sorted(seqs)
    d = {}
    for i, a in enumerate(ids):
        for b in ids[i + 1:]:
            d[(a, b)] = d[(b, a)] = sequence_distance(seqs[a], seqs[b])
    return ids, d


def pairwise_text(ids, d):
    rows = ["Seq1\tSeq2\tDistance\n"]
    for a in ids:
        for b in ids:
            rows.append("%s\t%s\t%i\n" % (a, b, d.get((a, b), 0)))
    return "".join(rows)


def matrix_text(ids, d):
    rows = ["\t%s\n" % "\t".join(ids)]
    for a in ids:
        rows.append("%s\t%s\n" % (a, "\t".join(str(d.get((a, b), 0)) for b in ids)))
    return "".join(rows)
