"""Helpers shared by the -m gpu parity tests: run the HIP path through the C ABI and the oracle on the same bytes."""
import numpy as np

from oracle import pileup_oracle as po
from snp_pipeline_amd import _lib as L
from snp_pipeline_amd import device as dev


def get_device():
    return dev.default_device()


def gpu_consensus(d, data, snp_list, excluded, p, want_counts=True):
    """Returns (consensus bytes in snp_list order, ConsensusResult, SiteSet)."""
    keys = list(snp_list) + [k for k in excluded if k not in set(snp_list)]
    snps, excl = set(snp_list), set(excluded)
    flags = [(L.SITE_IN_SNPLIST if k in snps else 0) | (L.SITE_EXCLUDED if k in excl else 0) for k in keys]
    ss = d.siteset(keys, flags)
    prm = dev.make_params(p.min_base_quality, p.min_cons_freq, p.min_cons_depth, p.min_cons_strand_depth,
                          p.min_cons_strand_bias)
    res = d.call_consensus(ss, data, prm, want_counts=want_counts)
    idx = ss.index_of[:len(snp_list)]
    cons = bytes(int(res.bases[i]) if i >= 0 else 0x2D for i in idx)
    return cons, res, ss


def check_against_oracle(d, data, snp_list, excluded, p):
    want, detail = po.call_consensus_sites(data, snp_list, set(excluded), p)
    got, res, ss = gpu_consensus(d, data, snp_list, excluded, p)
    assert got == want
    # every parsed position: counts, ranking, caller output
    for slot, key in enumerate(ss.key_tuples()):
        c = res.counts[slot]
        if key not in detail:
            assert c["status"] == L.ST_NO_LINE and res.bases[slot] == 0x2D and res.filters[slot] == 0
            continue
        rec, base, mask = detail[key]
        assert c["status"] == L.ST_OK
        raw = int(c["raw_depth"])
        if not 0 <= rec.raw_depth < (1 << 32):                      # "-3", "5000000000": the value is in the position's spill record
            raw = int(res.spill[(int(c["n_symbols"]) >> 8) - 1]["depth64"])
        assert (raw, c["good_depth"], c["fwd_good_depth"], c["rev_good_depth"]) == \
            (rec.raw_depth, rec.good_depth, rec.forward_good_depth, rec.reverse_good_depth), key
        assert c["cons_base"] == base and c["filters"] == mask, key
        ranked = rec.most_common_good_bases or []
        assert c["n_symbols"] & 0xFF == len(ranked)
        for r, sym in enumerate(ranked[:L.MAX_SYMS]):
            assert c["sym"][r] == sym
            assert c["total"][r] == rec.base_good_depth[sym]
            assert c["fwd"][r] == rec.forward_base_good_depth.get(sym, 0)
            assert c["rev"][r] == rec.reverse_base_good_depth.get(sym, 0)
        if len(ranked) > L.MAX_SYMS:                                # ranks 8, 9, ...: the position's spill record
            more = res.spill[(int(c["n_symbols"]) >> 8) - 1]
            assert more["n"] == len(ranked) - L.MAX_SYMS
            for r, sym in enumerate(ranked[L.MAX_SYMS:]):
                assert (more["sym"][r], more["total"][r], more["fwd"][r], more["rev"][r]) == \
                    (sym, rec.base_good_depth[sym], rec.forward_base_good_depth.get(sym, 0), rec.reverse_base_good_depth.get(sym, 0)), (key, sym)
        else:
            assert (c["n_symbols"] >> 8 == 0) == (0 <= rec.raw_depth < (1 << 32))      # (a spill record only for a wide depth)
    # the throughput path (no per-site counts: one lane per site, leftovers by the wave-per-site kernel) must agree
    got2, res2, _ = gpu_consensus(d, data, snp_list, excluded, p, want_counts=False)
    assert got2 == want
    assert bytes(res2.bases) == bytes(res.bases) and bytes(res2.filters) == bytes(res.filters)
    return res
