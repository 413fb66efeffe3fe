#!/usr/bin/env python3
"""Development helper (no GPU needed): a timed differential campaign of the library's HOST code against the plain Python line loops
it stands in for, on mutated files: the VCF CHROM/POS reader (csrc/vcf_in.hip vs utils.read_vcf_sites), the snplist reader (vs
utils.read_snp_position_list, the reference's own loop: utils.py:1073-1088), the multi-FASTA loader (csrc/fasta_in.hip vs
snp_matrix.read_matrix = distance.py:76-84) and the byte-level filter_regions reader (filter_regions._read_vcf vs read_vcf_sites).
Either side may raise: then both must, with the same class.  Usage: python tools/fuzz_host.py [seconds] [first seed]."""
import os
import random
import sys
import tempfile
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mutate(rng, base, alphabet, n_ops):
    buf = bytearray(base)
    for _ in range(n_ops):
        if not buf:
            buf = bytearray(b"\n")
        at = rng.randrange(len(buf))
        op = rng.random()
        ch = rng.choice(alphabet)
        if op < 0.35:
            buf[at] = ch
        elif op < 0.6:
            buf.insert(at, ch)
        elif op < 0.75:
            del buf[at]
        elif op < 0.85:
            end = buf.find(b"\n", at)
            del buf[at:end if end >= 0 else len(buf)]
        else:
            lines = bytes(buf).split(b"\n")
            i, j = rng.randrange(len(lines)), rng.randrange(len(lines))
            if op < 0.93:
                lines.insert(j, lines[i])
            else:
                lines[i], lines[j] = lines[j], lines[i]
            buf = bytearray(b"\n".join(lines))
    return bytes(buf)


def outcome(fn):
    try:
        return ("ok", fn())
    except Exception as e:                                       # noqa: B902 — the class is the datum
        return ("raised", type(e).__name__)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time()) % 1000000
    import numpy as np
    from snp_pipeline_amd import filter_regions as fr
    from snp_pipeline_amd import snp_matrix, utils
    tmp = tempfile.mkdtemp(prefix="fuzz_host_")
    path = os.path.join(tmp, "f")
    counts = {"vcf": 0, "snplist": 0, "fasta": 0, "vcf_rows": 0, "tsv": 0, "raised": 0}
    from oracle import steps_oracle as so
    from snp_pipeline_amd import distance as dmod
    from snp_pipeline_amd import merge_sites as msites
    from oracle import fuzz
    from oracle import pileup_oracle as po
    from oracle import vcf_oracle as vo
    from snp_pipeline_amd import _lib as L
    from snp_pipeline_amd import vcf_writer
    from snp_pipeline_amd.device import COUNTS_DTYPE, SPILL_DTYPE
    t_end = time.time() + seconds
    seed = seed0
    while time.time() < t_end:
        seed += 1
        rng = random.Random(seed)
        kind = ("vcf", "snplist", "fasta", "vcf_rows", "tsv")[seed % 5]
        data = b""
        contigs = [rng.choice(["c", "ctg|%d" % rng.randint(1, 9), "NODE_%d_cov_1.5" % rng.randint(1, 99), "x" * rng.randint(1, 30)]) for _ in range(rng.randint(1, 3))]
        eol = rng.choice([b"\n", b"\n", b"\r\n"])
        n = rng.choice([0, 1, 5, 60, 400])
        try:
            if kind == "vcf":
                head = b"##fileformat=VCFv4.1" + eol + b"##INFO=<ID=ADP,Number=1>" + eol + b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS" + eol
                body = b"".join(("%s\t%d\t.\tA\tC\t.\tPASS\tADP=9\tGT\t1/1" % (rng.choice(contigs), rng.randint(1, 5000000))).encode() + eol for _ in range(n))
                data = mutate(rng, head + body, b"\t\t\n\r #0123456789-+_.cx|", rng.choice([0, 1, 2, 4]))
                with open(path, "wb") as f:
                    f.write(data)
                want = outcome(lambda: utils.read_vcf_sites(path)[2])
                got = outcome(lambda: (lambda r: [(r[0][int(c)], int(p)) for c, p in zip(r[1], r[2])])(utils.read_vcf_site_arrays(path)))
                assert got == want, ("vcf arrays", got if got[0] == "raised" else len(got[1]), want if want[0] == "raised" else len(want[1]))
                got2 = outcome(lambda: (lambda r: (len(r[1]), [(r[2][0][int(c)], int(p)) for c, p in zip(r[2][1], r[2][2])]))(fr._read_vcf(path)))
                if want[0] == "ok":
                    assert got2 == ("ok", (len(want[1]), want[1])), ("filter_regions reader", got2[0])
                else:
                    assert got2 == want, ("filter_regions reader", got2, want)
            elif kind == "snplist":
                body = b"".join(("%s\t%d\t%d\t%s" % (rng.choice(contigs), rng.randint(1, 5000000), 2, "s1\ts2")).encode() + eol for _ in range(max(n, 1)))
                data = mutate(rng, body, b"\t\t\n\r 0123456789-+_.cx|", rng.choice([0, 1, 2, 4]))
                with open(path, "wb") as f:
                    f.write(data)
                want = outcome(lambda: utils.read_snp_position_list(path))
                got = outcome(lambda: (lambda r: [(r[0][int(c)], int(p)) for c, p in zip(r[1], r[2])])(utils.read_snp_position_arrays(path)))
                assert got == want, ("snplist arrays", got if got[0] == "raised" else len(got[1]), want if want[0] == "raised" else len(want[1]))
            elif kind == "tsv":
                # the two distance TSV files (csrc/tsv_out.hip vs distance.py:100-114) and the snplist text (vs utils.py:1056-1070)
                m = rng.choice([0, 1, 2, 7, 40])
                ids = sorted(set("".join(rng.choice("abcXYZ019_-|. ") for _ in range(rng.randint(1, 12))) for _ in range(m)))
                mat = np.asarray([[rng.choice([0, 1, 9, 10, 99, 12345, 2 ** 31 - 1]) for _ in ids] for _ in ids], dtype=np.int32).reshape(len(ids), len(ids))
                table = {(a, b): int(mat[i, j]) for i, a in enumerate(ids) for j, b in enumerate(ids)}
                if ids:
                    dmod.write_pairwise(path, ids, mat)
                    assert open(path).read() == so.pairwise_text(ids, table), "pairwise TSV"
                    dmod.write_matrix(path, ids, mat)
                    assert open(path).read() == so.matrix_text(ids, table), "matrix TSV"
                samples = ["s%d" % k for k in range(rng.randint(1, 5))]
                sites = sorted(set((rng.randrange(len(contigs)), rng.randint(0, 4000000000)) for _ in range(n)))
                uniq = np.asarray([(c << 32) | p_ for c, p_ in sites], dtype=np.uint64)
                carriers = [sorted(rng.sample(range(len(samples)), rng.randint(1, len(samples)))) for _ in sites]
                off = np.cumsum([0] + [len(c) for c in carriers]).astype(np.uint32)
                car = np.asarray([x for c in carriers for x in c], dtype=np.uint32)
                msites.write_snplist(path, contigs, uniq, off, car, samples)
                got_text = open(path).read()
                utils.write_list_of_snps(path, [(contigs[c], p_) for c, p_ in sites], [[samples[x] for x in c] for c in carriers])
                assert got_text == open(path).read(), "snplist text"
                want = ("ok", None)
            elif kind == "vcf_rows":
                # consensus.vcf rows: records as the device fills them, built here from the oracle's Records of fuzzed lines ->
                # the library's formatter and the Python row function against the restatement of vcf_writer.py:295-379
                q, c_, D, d_, b_ = rng.choice([(0, 0.6, 1, 0, 0.0), (15, 0.9, 5, 2, 0.1), (30, 0.75, 2, 1, 0.25)])
                prm = po.CallerParams(q, c_, D, d_, b_)
                names = [nm for nm, _ in vcf_writer.filter_descriptions(c_, D, d_, b_)]
                chrom = rng.choice(contigs)
                records, rows = [], []
                while len(records) < 60:
                    ln = fuzz.fuzz_line(rng, chrom=chrom)
                    try:
                        rec = po.parse_record(po.split_fields(ln.encode()), q)
                    except (IndexError, ValueError):
                        continue
                    if len(rec.reference_base) != 1 or not (0 <= rec.raw_depth < 2 ** 32):
                        continue
                    base, mask = po.call_record(rec, prm)
                    if rng.random() < 0.1:
                        mask |= po.F_REGION
                    records.append((rec, mask))
                recs = np.zeros(len(records), dtype=COUNTS_DTYPE)
                spill = np.zeros(len(records), dtype=SPILL_DTYPE)
                n_spill = 0
                keys = np.zeros(len(records), dtype=np.uint64)
                gt, keep_case = rng.choice([".", "0", "1"]), rng.random() < 0.5
                for i, (rec, mask) in enumerate(records):
                    cc = recs[i]
                    ranked = rec.most_common_good_bases or []
                    cc["ref_base"], cc["raw_depth"], cc["good_depth"], cc["filters"] = rec.reference_base[0], rec.raw_depth, rec.good_depth, mask
                    cc["n_symbols"] = len(ranked)
                    for r, sym in enumerate(ranked):
                        tgt, k = (cc, r) if r < L.MAX_SYMS else (spill[n_spill], r - L.MAX_SYMS)
                        tgt["sym"][k], tgt["total"][k] = sym, rec.base_good_depth[sym]
                        tgt["fwd"][k], tgt["rev"][k] = rec.forward_base_good_depth.get(sym, 0), rec.reverse_base_good_depth.get(sym, 0)
                    if len(ranked) > L.MAX_SYMS:
                        spill[n_spill]["n"] = len(ranked) - L.MAX_SYMS
                        n_spill += 1
                        cc["n_symbols"] = len(ranked) | (n_spill << 8)
                    keys[i] = rec.position
                    failed = [names[b] for b in range(6) if mask >> b & 1]
                    rows.append(vo.vcf_row(rec, failed or None, gt, keep_case))
                want_text = "".join(r + "\n" for r in rows)
                py = "".join(vcf_writer.row_from_counts(chrom, int(keys[j]), recs[j], names, keep_case, gt, spill=spill[:n_spill]) + "\n" for j in range(len(records)))
                assert py == want_text, "row_from_counts differs from the restatement"
                cname = np.frombuffer(chrom.encode(), dtype=np.uint8)
                lib = vcf_writer.format_rows(recs, np.arange(len(records), dtype=np.uint32), cname, np.array([0, len(cname)], dtype=np.uint32), keys, names,
                                             keep_case, gt, spill=spill[:n_spill] if n_spill else None)
                assert lib == want_text.encode("latin-1"), "the library's rows differ from the restatement"
                want = ("ok", None)
            else:
                recs = []
                for k in range(rng.randint(0, 6)):
                    seq = "".join(rng.choice("ACGTacgt-N") for _ in range(rng.choice([0, 1, 59, 60, 61, 200])))
                    width = rng.choice([60, 7, 10 ** 6])
                    recs.append(b">" + rng.choice(contigs).encode() + (b" extra words" if rng.random() < 0.2 else b"") + eol
                                + b"".join(seq[i:i + width].encode() + eol for i in range(0, len(seq), width)))
                data = mutate(rng, b"".join(recs) or b"\n", b"\n\r >ACGTacgt-N \t", rng.choice([0, 1, 2, 4]))
                with open(path, "wb") as f:
                    f.write(data)
                want = outcome(lambda: snp_matrix.read_matrix(path))

                def load():
                    ids, mat, lens = snp_matrix.load_matrix(path)
                    out = {}
                    for r, i in enumerate(ids):                  # the last record of a repeated id, as the dict keeps it
                        out[i] = bytes(mat[r, :int(lens[r])]).decode("utf-8")
                    return out
                got = outcome(load)
                assert got == want, ("fasta loader", got if got[0] == "raised" else sorted(got[1])[:3], want if want[0] == "raised" else sorted(want[1])[:3])
            counts[kind] += 1
            counts["raised"] += want[0] == "raised"
        except Exception:                                        # noqa: B902
            keep = os.path.join(ROOT, "gpurun_out", "fuzz_host")
            os.makedirs(keep, exist_ok=True)
            with open(os.path.join(keep, "%s_%d.bin" % (kind, seed)), "wb") as f:
                f.write(data)
            print("DISAGREEMENT kind=%s seed=%d\n%s" % (kind, seed, traceback.format_exc()[-1200:]))
            print("agreed before that:", counts)
            sys.exit(1)
    print("fuzz host: %.0f s, seeds %d..%d, all agreed: %r" % (seconds, seed0 + 1, seed, counts))
    assert np is not None


if __name__ == "__main__":
    main()
