#!/usr/bin/env python3
"""Development helper: a timed differential campaign of the device paths against the oracle on FRESH seeds (the -m gpu tests use
fixed ones).  Usage: python tools/fuzz_campaign.py [seconds] [first seed]   -> one line per kind with the number of cases that
agreed; the first disagreement is written to gpurun_out/fuzz/ (input bytes + what differed) and ends the run with exit code 1.

Kinds: fuzz.fuzz_line files (adversarial read-base columns, odd separators) with three caller parameter sets; synthetic pileups of
random shape (depth 2-300, 1-12 contigs with names of 1-50 bytes, CR LF / mixed / VT FF line ends, repeated positions) through the
file-level scan + call; site calling (fuzz.varscan_pileup / varscan_adversarial) with two option sets against the VarScan
restatement."""
import os
import random
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time()) % 1000000
    import tempfile
    from oracle import fuzz
    from oracle import pileup_oracle as po
    from oracle import varscan_oracle as vo
    from snp_pipeline_amd import varscan
    from tests.gpu_util import check_against_oracle, get_device, gpu_consensus
    d = get_device()
    out_dir = os.path.join(ROOT, "gpurun_out", "fuzz")
    tmp = tempfile.mkdtemp(prefix="fuzz_")
    counts = {"lines": 0, "shapes": 0, "sites": 0, "mutants": 0, "mutants_raising": 0, "mutants_refused": 0, "site_mutants": 0, "site_mutants_raising": 0, "distance": 0, "regions": 0, "allpos": 0, "allpos_raising": 0}
    from oracle import vcf_oracle as vco
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    import numpy as np
    from oracle import steps_oracle as so
    from snp_pipeline_amd import _lib as L
    from snp_pipeline_amd import device as devmod

    def mutate(rng, base, alphabet, n_ops):
        buf = bytearray(base)
        for _ in range(n_ops):
            at = rng.randrange(len(buf))
            if len(buf) > 9000 and rng.random() < 0.35:           # near an edge of the scan's 4 KiB tiles (lines, CRs, TABs that straddle one)
                at = min(len(buf) - 1, max(0, rng.randrange(1, len(buf) // 4096 + 1) * 4096 + rng.randint(-24, 24)))
            op = rng.random()
            ch = rng.choice(alphabet)
            if op < 0.35:
                buf[at] = ch
            elif op < 0.6:
                buf.insert(at, ch)
            elif op < 0.75:
                del buf[at]
            elif op < 0.83:                                      # a field boundary: remove what is left of the line
                end = buf.find(b"\n", at)
                del buf[at:end if end >= 0 else len(buf)]
            else:                                                # whole lines: one copied to another place, or two swapped, or one dropped
                lines = bytes(buf).split(b"\n")
                i, j = rng.randrange(len(lines)), rng.randrange(len(lines))
                if op < 0.91:
                    lines.insert(j, lines[i])
                elif op < 0.96:
                    lines[i], lines[j] = lines[j], lines[i]
                else:
                    del lines[i]
                buf = bytearray(b"\n".join(lines))
            if not buf:
                buf = bytearray(b"\n")
        return bytes(buf)

    def consensus_through_files(data, keys, excluded, p):
        """The per-sample command's path: streamed file, status checks in the reference's order, consensus in snplist order."""
        path = os.path.join(tmp, "m.pileup")
        with open(path, "wb") as f:
            f.write(data)
        allk = list(keys) + [k for k in excluded if k not in set(keys)]
        flags = [(L.SITE_IN_SNPLIST if k in set(keys) else 0) | (L.SITE_EXCLUDED if k in set(excluded) else 0) for k in allk]
        ss = d.siteset(allk, flags)
        try:
            prm = devmod.make_params(p.min_base_quality, p.min_cons_freq, p.min_cons_depth, p.min_cons_strand_depth, p.min_cons_strand_bias)
            results, rcs, _ = d.call_consensus_files(ss, [path], prm, want_counts=True, want_line_offsets=True)
            d.raise_file_errors(ss, path, prm, int(rcs[0]), results[0])
            idx = ss.index_of[:len(keys)]
            return bytes(int(results[0].bases[i]) if i >= 0 else 0x2D for i in idx)
        finally:
            ss.close()
    from snp_pipeline_amd.device import PileupFormatError
    param_sets = (po.CallerParams(), po.CallerParams(15, 0.9, 5, 2, 0.1), po.CallerParams(30, 0.75, 2, 1, 0.25), po.CallerParams(0, 0.6, 3, 0, 0.0))
    vs_cases = (("--min-avg-qual 15 --min-var-freq 0.90 --min-reads2 5", dict(vo.PIPELINE_DEFAULTS)),
                ("--min-avg-qual 0 --min-var-freq 0.05 --min-reads2 1 --min-coverage 1 --strand-filter 0",
                 dict(min_avg_qual=0, min_var_freq=0.05, min_reads2=1, min_coverage=1, strand_filter=0)))

    def fail(kind, seed, data, err):
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "%s_%d.pileup" % (kind, seed)), "wb") as f:
            f.write(data)
        with open(os.path.join(out_dir, "%s_%d.txt" % (kind, seed)), "w") as f:
            f.write(err)
        print("DISAGREEMENT kind=%s seed=%d (%d bytes): %s" % (kind, seed, len(data), err.splitlines()[-1] if err else ""))
        print("agreed before that:", counts)
        sys.exit(1)

    t_end = time.time() + seconds
    seed = seed0
    while time.time() < t_end:
        seed += 1
        rng = random.Random(seed)
        kind = ("lines", "shapes", "sites", "mutants", "site_mutants", "mutants", "distance", "regions", "allpos")[seed % 9]
        data = b""
        try:
            if kind == "lines":
                lines, keys, pos = [], [], 0
                chrom = rng.choice(["chrF", "c", "a_rather_long_contig_name|with|bars.1", "x" * rng.randint(1, 44)])
                n = rng.choice([40, 400, 1500])
                while len(lines) < n:
                    ln = fuzz.fuzz_line(rng, chrom=chrom)
                    f = po.split_fields(ln.encode())
                    try:
                        po.parse_record(f, 0)
                    except (IndexError, ValueError):
                        continue
                    pos += rng.choice([1, 1, 1, 2, 7, 1000])
                    f[1] = str(pos).encode()
                    lines.append(b"\t".join(f))
                    keys.append((f[0], pos))
                data = b"\n".join(lines) + (b"\n" if rng.random() < 0.8 else b"")
                step = rng.choice([1, 2, 9])
                check_against_oracle(d, data, keys[::step] if step > 1 else keys, keys[::13], rng.choice(param_sets))
            elif kind == "shapes":
                contigs = tuple(rng.choice(["c", "ctg%d" % rng.randint(1, 99), "NODE_%d_length_%d_cov_1.5" % (rng.randint(1, 999), rng.randint(100, 99999)),
                                            "n" * rng.randint(17, 50)]) + ("_%d" % k) for k in range(rng.choice([1, 1, 2, 3, 3, 6, 12])))      # (6, 12: several contig changes per scan tile)
                depth = rng.choice([2, 8, 15, 30, 30, 100, 300])
                glen = rng.choice([g for g in (40, 300, 2000, 9000, 30000) if g * depth * len(contigs) <= 600000 and (g > 40 or len(contigs) > 3)])   # (40: a contig of less than a tile)
                data, _, sites = fuzz.synth_pileup(seed, genome_len=glen, contigs=contigs, mean_depth=depth, n_sites=rng.choice([5, 60, 250]))
                variant = rng.choice([None, None, "crlf", "mixed", "vt_ff", "repeats", "shuffle", "interleave", "blanks"])
                if variant in ("shuffle", "interleave", "blanks"):
                    body = data.split(b"\n")[:-1]
                    if variant == "shuffle":                         # no order at all: the bitmap window and the contig hint never settle
                        rng.shuffle(body)
                    elif variant == "interleave":                    # the contigs' lines dealt round robin: the hint changes on every line
                        per = {}
                        for ln in body:
                            per.setdefault(ln.split(b"\t")[0], []).append(ln)
                        body = [ln for group in zip(*[v[:min(map(len, per.values()))] for v in per.values()]) for ln in group]
                    else:                                            # every separator a run of blanks: every line takes the exact parser
                        body = [b"  ".join(ln.split(b"\t")) for ln in body]
                    data = b"\n".join(body) + b"\n"
                elif variant:
                    data = fuzz.with_line_ends(data, variant, seed)
                keys = sorted(sites)
                p = rng.choice(param_sets)
                want, _ = po.call_consensus_sites(data, keys, set(keys[::11]), p)
                for want_counts in (True, False):
                    got, _, _ = gpu_consensus(d, data, keys, keys[::11], p, want_counts=want_counts)
                    assert got == want, "consensus differs (want_counts=%s, line ends %s)" % (want_counts, variant)
            elif kind == "mutants":
                # a well-formed pileup with a few ASCII bytes changed, inserted or removed: the same consensus, or the same exception
                # class as the reference's text-mode reader / Record raises first (pileup.py:224-237, 425-426)
                base, _, sites = fuzz.synth_pileup(seed, genome_len=rng.choice([200, 1500, 1500]), mean_depth=rng.choice([6, 30, 30]), n_sites=rng.choice([20, 150, 600]),
                                                   contigs=(rng.choice(["c1", "contig_with_a_longer_name_%d" % seed]),))
                data = mutate(rng, base, b"\t\t\n\r \x0b\x0c0123456789-+_*ACGTacgt.,^$<>!I~xX", rng.choice([1, 1, 2, 4]))
                keys = sorted(sites)
                p = rng.choice(param_sets)
                try:
                    want = po.call_consensus_sites(data, keys, set(keys[::11]), p)[0]
                except (ValueError, IndexError) as e:
                    want = type(e)
                try:
                    got = consensus_through_files(data, keys, keys[::11], p)
                except PileupFormatError as e:
                    got = e.reference_exception
                    if got is None:                                  # an input this build refuses by name: counted, not compared
                        counts["mutants_refused"] += 1
                        continue
                assert got == want, "device %r, oracle %r" % (got, want)
                counts["mutants_raising"] += isinstance(want, type)
            elif kind == "allpos":
                # the per-sample command with a consensus.vcf (rows for the listed positions, or --vcfAllPos: for EVERY line), on
                # fuzzed lines with a few mutations: the same rows as the restatement of the writer, or the same exception class
                lines, keys, pos = [], [], 0
                chrom = rng.choice(["chrF", "c", "a_rather_long_contig_name|with|bars.1"])
                while len(lines) < rng.choice([30, 300]):
                    ln = fuzz.fuzz_line(rng, chrom=chrom)
                    f = po.split_fields(ln.encode())
                    try:
                        rec = po.parse_record(f, 0)
                    except (IndexError, ValueError):
                        continue
                    if len(rec.reference_base) != 1 or rec.reference_base[0] >= 0x80:
                        continue
                    pos += rng.choice([1, 1, 2, 50])
                    f[1] = str(pos).encode()
                    if rng.random() < 0.03:                          # a reference field of several bytes, up to a few spill records long
                        f[2] = "".join(rng.choice("ACGTNacgtn.,*") for _ in range(rng.choice([2, 3, 64, 65, 300, 1704, 1705, 3500]))).encode()
                    lines.append(b"\t".join(f))
                    keys.append((f[0], pos))
                data = b"\n".join(lines) + b"\n"
                if rng.random() < 0.4:
                    data = mutate(rng, data, b"\t\t\n\r 0123456789-+*ACGTacgt.,^$<>!I~", rng.choice([1, 2]))
                snps = keys[::rng.choice([1, 2, 7])]
                excl = keys[::9]
                all_pos = rng.random() < 0.6
                q, c_, D, d_, b_ = rng.choice([(0, 0.6, 1, 0, 0.0), (15, 0.9, 5, 2, 0.1), (30, 0.75, 2, 1, 0.25)])
                p = po.CallerParams(q, c_, D, d_, b_)
                gt, keep_case = rng.choice([".", "0", "1"]), rng.random() < 0.5
                sdir = os.path.join(tmp, "sampleA")
                os.makedirs(sdir, exist_ok=True)
                with open(os.path.join(sdir, "reads.all.pileup"), "wb") as f:
                    f.write(data)
                with open(os.path.join(tmp, "snplist.txt"), "w") as f:
                    f.write("".join("%s\t%d\t1\ts\n" % (c.decode(), pp) for c, pp in snps))
                with open(os.path.join(sdir, "excl.vcf"), "w") as f:
                    f.write("##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n")
                    f.write("".join("%s\t%d\t.\tA\tC\t.\tPASS\t.\tGT\t1/1\n" % (c.decode(), pp) for c, pp in excl))
                line = ("call_consensus -v 0 -f -l %s/snplist.txt -o %s/consensus.fasta -e %s/excl.vcf -q %d -c %s -D %d -d %d -b %s --vcfFileName all.vcf "
                        "--vcfFailedSnpGt %s%s%s %s/reads.all.pileup" % (tmp, sdir, sdir, q, c_, D, d_, b_, gt, " --vcfPreserveRefCase" if keep_case else "",
                                                                        " --vcfAllPos" if all_pos else "", sdir))
                wanted = set(snps) | set(excl)
                names = po.filter_names(p)
                try:
                    rows = []
                    for _, ln in po.iter_lines(data):
                        f = po.split_fields(ln)
                        if all_pos:                                  # pileup.py:418-421: a Record from every line (IndexError for a short one)
                            rec = po.parse_record(f, q)
                            key = (rec.chrom, rec.position)
                        else:                                        # pileup.py:423-429: two fields unpacked (ValueError), Records at listed positions
                            chrom_, pos_ = f[:2]
                            key = (chrom_, int(pos_.decode()))
                            if key not in wanted:
                                continue
                            rec = po.parse_record(f, q)
                        base, mask = po.call_record(rec, p)
                        if key in set(excl):
                            mask |= 32
                        rows.append(vco.vcf_row(rec, [names[i] for i in range(6) if mask >> i & 1] or None, gt, preserve_ref_case=keep_case))
                    want = rows
                except (ValueError, IndexError) as e:
                    want = type(e)
                os.environ.pop("errorOutputFile", None)
                try:
                    cli.run_command_from_line(line)
                    got = [x for x in open(os.path.join(sdir, "all.vcf"), encoding="latin-1").read().split("\n") if x and not x.startswith("#")]
                except (ValueError, IndexError) as e:
                    got = type(e) if not isinstance(e, PileupFormatError) else "refused"
                if got == "refused":
                    counts["mutants_refused"] += 1
                    continue
                if isinstance(want, list) and len(set(k for k in keys)) == len(keys) and not all_pos:
                    pass
                assert got == want, "consensus.vcf: command %r, restatement %r" % (got if isinstance(got, type) else len(got), want if isinstance(want, type) else len(want))
                counts["allpos_raising"] += isinstance(want, type)
            elif kind == "distance":
                # K5 on shapes around its tile edges (128 x 128 pairs, 64-site words), any byte as a symbol
                n = rng.choice([1, 2, 3, 63, 64, 65, 127, 128, 129, 200, 257])
                s = rng.choice([0, 1, 15, 16, 17, 63, 64, 65, 255, 256, 1000, 4097])
                nrng = np.random.default_rng(seed)
                alphabet = np.frombuffer(rng.choice([b"ACGT-", b"ACGTacgtNn-*RY", bytes(range(33, 127))]), dtype=np.uint8)      # (ASCII: str.upper() of other text can change its length)
                sym = alphabet[nrng.integers(0, len(alphabet), size=(n, s))]
                data = sym.tobytes()
                got = d.distance(sym)
                up = np.where((sym >= 97) & (sym <= 122), sym - 32, sym)
                ok = np.isin(up, np.frombuffer(b"ACGT", dtype=np.uint8))
                want = np.zeros((n, n), dtype=np.int64)
                for i in range(n):                                       # (a numpy statement of utils.py:1135-1165, anchored below)
                    want[i] = ((up[i][None, :] != up) & ok[i][None, :] & ok).sum(axis=1)
                assert np.array_equal(got, want), "distance matrix differs at %r" % (np.argwhere(got != want)[:3].tolist(),)
                for _ in range(min(3, n)):
                    i, j = rng.randrange(n), rng.randrange(n)
                    assert so.sequence_distance(sym[i].tobytes().decode("latin-1"), sym[j].tobytes().decode("latin-1")) == int(got[i, j])
            elif kind == "regions":
                # K3 / K4 through the subcommands' device entry points against the restatement: dense windows + merged regions
                # + classification on random sorted positions with random rules
                n_pos = rng.choice([0, 1, 2, 10, 200, 3000])
                span = rng.choice([50, 1000, 100000, 5000000])
                pos = sorted(rng.randint(1, span) for _ in range(n_pos))
                rules = rng.choice([([3, 2, 1], [1000, 125, 15]), ([1], [1]), ([2], [500]), ([6, 1], [2000, 60])])
                data = repr((pos, rules)).encode()
                regs = []
                for m, w in zip(*rules):
                    regs.extend(so.find_dense_regions(m, w, pos))
                extra_regs = [(a, a + rng.randint(0, 40)) for a in (rng.randint(0, span) for _ in range(rng.randint(0, 5)))]   # (edge regions and the like)
                want_regions = so.merge_regions(regs + extra_regs)
                starts, ends, _ = d.dense_windows(np.asarray(pos, dtype=np.int64), np.asarray([0, len(pos)], dtype=np.uint32), rules[0], rules[1])
                all_s = np.concatenate([starts, np.asarray([a for a, _ in extra_regs], dtype=np.int64)])
                all_e = np.concatenate([ends, np.asarray([b for _, b in extra_regs], dtype=np.int64)])
                _, ms, me = d.merge_regions(np.zeros(len(all_s), dtype=np.uint32), all_s, all_e)
                # (adjacent intervals: the reference joins them only when the later one reaches further — utils.py:1168-1282; the
                # device list may keep them apart, which classifies every position the same way: compare the covered positions)
                covered = lambda regions: sorted(set(p for a, b in regions for p in (a, b))) and [(a, b) for a, b in so.merge_regions([(a, b) for a, b in regions])]   # noqa: E731
                probe = sorted(set(pos + [p + dlt for a, b in want_regions for p in (a, b) for dlt in (-1, 0, 1)] + [rng.randint(0, span + 50) for _ in range(50)]))
                probe = [p for p in probe if p >= 0]
                got_in = d.in_regions(np.zeros(len(probe), dtype=np.uint32), np.asarray(probe, dtype=np.int64), np.asarray([0, len(ms)], dtype=np.uint32), ms, me)
                want_in = [so.in_region(p, want_regions) for p in probe]
                assert list(got_in) == want_in, "classification by merged dense regions differs"
                assert covered is not None
            elif kind == "site_mutants":
                base = fuzz.varscan_pileup(seed, rng.choice([60, 600]), eol=rng.choice([b"\n", b"\n", b"\r\n"]))
                data = mutate(rng, base, b"\t\t\n\r 0123456789-+*ACGTNacgtn.,^$!I5~", rng.choice([1, 1, 2, 3]))
                path, out = os.path.join(tmp, "p.pileup"), os.path.join(tmp, "p.vcf")
                with open(path, "wb") as f:
                    f.write(data)
                extra, kw = vs_cases[seed % 2]
                try:
                    want = vo.mpileup2snp(data, vo.Params(**kw))
                except ValueError:
                    want = ValueError
                try:
                    varscan.mpileup2snp(d, path, out, varscan.Options(extra))
                    got = open(out, encoding="latin-1").read()
                except PileupFormatError:
                    got = ValueError
                assert got == want, "site calling: device %r, restatement %r" % (str(got)[-200:], str(want)[-200:])
                counts["site_mutants_raising"] += want is ValueError
            else:
                data = fuzz.varscan_adversarial(seed, rng.choice([300, 2000])) if seed % 2 else \
                    fuzz.varscan_pileup(seed, rng.choice([200, 3000, 12000]), eol=rng.choice([b"\n", b"\n", b"\r\n"]),
                                        depths=rng.choice([(0, 0, 3, 7, 8, 9, 12, 20, 30, 30, 45, 80), (150, 200, 120, 0), (8, 8, 9, 30), (1500, 30, 0)]))
                path, out = os.path.join(tmp, "p.pileup"), os.path.join(tmp, "p.vcf")
                with open(path, "wb") as f:
                    f.write(data)
                for extra, kw in vs_cases:
                    varscan.mpileup2snp(d, path, out, varscan.Options(extra))
                    assert open(out, encoding="latin-1").read() == vo.mpileup2snp(data, vo.Params(**kw)), "var.flt.vcf differs for %r" % extra
            counts[kind] += 1
        except SystemExit:
            raise
        except Exception:                                        # noqa: B902 — any disagreement or refusal is the finding
            fail(kind, seed, data, traceback.format_exc())
    print("fuzz campaign: %.0f s, seeds %d..%d, all agreed: %r" % (seconds, seed0 + 1, seed, counts))


if __name__ == "__main__":
    main()
