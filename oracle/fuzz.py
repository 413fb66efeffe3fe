"""Seeded generators of pileup text for parity tests.  TEST INFRASTRUCTURE ONLY.

``fuzz_line`` produces single lines that exercise the grammar corners listed in
SURVEY.md A.1 (caret runs, indel tails that collide, length mismatches, exotic
bytes, lower-case reference, depth 0 ...).  ``synth_pileup`` produces a small
well-formed genome-wide pileup in the shape SURVEY.md 8(d) describes, on the
host, for sizes the Python oracle finishes in seconds.
"""

import random

_QUAL_LO, _QUAL_HI = 33, 74


def _bases_token(rng, ref_like=True):
    r = rng.random()
    if r < 0.45:
        return rng.choice(".,")
    if r < 0.75:
        return rng.choice("ACGTNacgtn")
    if r < 0.80:
        return "*"
    if r < 0.86:
        return "^" + chr(rng.randint(33, 126)) + rng.choice(".,ACGTacgt")
    if r < 0.91:
        return rng.choice(".,ACGT") + "$"
    if r < 0.97:
        k = rng.randint(1, 12)
        return rng.choice(".,ACGT") + rng.choice("+-") + str(k) + "".join(rng.choice("ACGTNacgtn") for _ in range(k))
    return rng.choice("#<>*")


def _adversarial(rng):
    """Strings that are not valid samtools output but that the reference still
    has a defined answer for."""
    pool = [
        "^^.A", "^+2AC.", "^$.$", ".^", "^", "^^", "^^^", ".+3A-1C.", ".+2A-1CG.", "+1+1AA.", ".-2+1A.",
        "..+9AC", ".+1A2C", ".+A.", "+1$A", ".+0A", "-0", "+", "-", "+-1A.", ".+12ACGTACGTACGTA,",
        ".+1^A,C", ".^+1A", "$$$", ".$+1A,", "+2^AC..", "+1^", ".,+3AC", "+007ACGTACG.", "1.2,", ".+2A^B-1C,,,",
        "..-1^", "+1+", "+1-", "^-1A.", ".+1", ".-", "+99999999999999999999A.", "..+1A+1", "[]_`\\", "{|}~",
        "Rr.,", "*#<>", "....^~a", ",,,,$", "a+1gc-2tta",
    ]
    s = rng.choice(pool)
    if rng.random() < 0.5:
        s = "".join(_bases_token(rng) for _ in range(rng.randint(0, 6))) + s
    if rng.random() < 0.5:
        s = s + "".join(_bases_token(rng) for _ in range(rng.randint(0, 6)))
    return s


def fuzz_line(rng, chrom="chrF"):
    """One pileup line (str, no newline) plus nothing else; may be malformed."""
    pos = rng.randint(1, 5_000_000)
    ref = rng.choice("ACGTNacgtnRY*")
    mode = rng.random()
    if mode < 0.04:
        return "%s\t%d\t%s\t0\t*\t*" % (chrom, pos, ref)
    if mode < 0.06:
        return "%s\t%d\t%s\t0" % (chrom, pos, ref)
    if mode < 0.08:
        return "%s\t%d\t%s\t%d" % (chrom, pos, ref, rng.randint(1, 9))      # 4 fields, depth>0 -> empty record
    if mode < 0.30:
        bases = _adversarial(rng)
    else:
        depth_t = rng.choice([1, 2, 3, 5, 8, 13, 30, 30, 30, 64, 65, 130, 300])
        bases = "".join(_bases_token(rng) for _ in range(depth_t))
    # quality string: usually one char per surviving base, sometimes off by a few
    approx = sum(1 for c in bases if c in ".,ACGTNacgtn*#<>")
    qlen = max(0, approx + rng.choice([0, 0, 0, 0, 0, 0, -2, -1, 1, 3]))
    if rng.random() < 0.03:
        qlen = 0
    quals = "".join(chr(rng.randint(_QUAL_LO, _QUAL_HI)) for _ in range(qlen))
    depth = rng.choice([approx, approx, approx, max(1, approx + 1), 1])
    sep = "\t" if rng.random() < 0.93 else rng.choice([" ", "\t\t", " \t", "\x0b", "\x1f"])
    fields = [chrom, str(pos), ref, str(max(depth, 1)), bases]
    if qlen > 0 or rng.random() < 0.5:
        fields.append(quals)
    line = sep.join(f for f in fields)
    if rng.random() < 0.03:
        line = line + rng.choice([" ", "\t", "  "])
    return line


def synth_pileup(seed, genome_len=4000, contigs=("synth_chr1",), mean_depth=30, n_sites=60,
                 carrier_frac=0.3, skip_frac=0.01, site_margin=1):
    """Small well-formed pileup.  Returns (text bytes, reference {contig: str},
    site list [(contig bytes, pos)], sorted)."""
    rng = random.Random(seed)
    lines = []
    refs = {}
    sites = []
    for contig in contigs:
        ref = "".join(rng.choice("ACGT") for _ in range(genome_len))
        refs[contig] = ref
        lo, hi = site_margin, genome_len - site_margin + 1
        chosen = sorted(rng.sample(range(lo, hi), min(n_sites, hi - lo)))
        alts = {p: rng.choice([b for b in "ACGT" if b != ref[p - 1]]) for p in chosen}
        carried = {p for p in chosen if rng.random() < carrier_frac}
        sites.extend((contig.encode(), p) for p in chosen)
        for pos in range(1, genome_len + 1):
            if rng.random() < skip_frac:
                continue                                   # uncovered position: no line
            depth = max(0, int(rng.gauss(mean_depth, mean_depth ** 0.5)))
            if depth == 0:
                continue
            r = ref[pos - 1]
            toks = []
            quals = []
            for _ in range(depth):
                fwd = rng.random() < 0.5
                if pos in carried and rng.random() < 0.97:
                    b = alts[pos]
                    t = b if fwd else b.lower()
                elif rng.random() < 0.005:
                    b = rng.choice([x for x in "ACGT" if x != r])
                    t = b if fwd else b.lower()
                elif rng.random() < 0.0005:
                    t = "*"
                else:
                    t = "." if fwd else ","
                if rng.random() < 1 / 150:
                    t = "^" + chr(33 + rng.randint(0, 42)) + t
                if rng.random() < 1e-3:
                    k = rng.randint(1, 3)
                    seq = "".join(rng.choice("ACGT") for _ in range(k))
                    t += rng.choice("+-") + str(k) + (seq if fwd else seq.lower())
                if rng.random() < 1 / 150:
                    t += "$"
                toks.append(t)
                quals.append(chr(33 + min(41, max(2, int(round(rng.gauss(35, 5)))))))
            lines.append("%s\t%d\t%s\t%d\t%s\t%s\n" % (contig, pos, r, depth, "".join(toks), "".join(quals)))
    return "".join(lines).encode(), refs, sorted(sites)


def odd_depth_lines(seed, eol=b"\n", n=4000):
    """A pileup whose lines exercise every shape the depth column can take: depths of 1 to 6 digits, two- and three-field
    lines, a multi-byte reference field, doubled separators, a depth that is not a number, a line that ends after the depth."""
    rng = random.Random(seed)
    lines = []
    for pos in range(1, n + 1):
        r = rng.random()
        depth = rng.choice([0, 1, 7, 30, 250, 999, 1000, 9999, 10000, 123456])
        if r < 0.80:
            lines.append(b"c9\t%d\tA\t%d\t%s\t%s" % (pos, depth, b"." * min(depth, 40), b"I" * min(depth, 40)))
        elif r < 0.84:
            lines.append(b"c9\t%d" % pos)
        elif r < 0.88:
            lines.append(b"c9\t%d\tA" % pos)
        elif r < 0.92:
            lines.append(b"c9\t%d\tACG\t%d\t...\tIII" % (pos, depth))
        elif r < 0.95:
            lines.append(b"c9\t%d\tA\t\t%d\t.\tI" % (pos, depth))
        elif r < 0.98:
            lines.append(b"c9 %d A %dx . I" % (pos, depth))
        else:
            lines.append(b"c9\t%d\tA\t%d" % (pos, depth))
    return eol.join(lines) + eol


def varscan_pileup(seed, n_lines=3000, contigs=("ctgA", "ctg_B|2"), eol=b"\n", depths=(0, 0, 3, 7, 8, 9, 12, 20, 30, 30, 45, 80)):
    """A TAB-separated one-sample pileup for the phase-1 site caller: mostly reference-matching columns of depth 0..80,
    every ~12th line a variant column (one dominant alternate allele at 60..100 %, sometimes a second one, strands from
    balanced to one-sided), with indels, read starts / ends, N, '*', low qualities, depth-0 lines ("*\\t*"), qualities
    shorter or longer than the bases, a seventh column now and then."""
    rng = random.Random(seed)
    out = []
    pos = 0
    for k in range(n_lines):
        chrom = contigs[0] if k < n_lines // 2 else contigs[-1]
        pos += rng.choice((1, 1, 1, 1, 2, 17))
        ref = rng.choice("ACGTacgtN")
        depth = rng.choice(depths)
        if depth == 0:
            out.append(("%s\t%d\t%s\t0\t*\t*" % (chrom, pos, ref)).encode())
            continue
        variant = rng.random() < 0.085
        alt = rng.choice([b for b in "ACGT" if b != ref.upper()])
        alt2 = rng.choice([b for b in "ACGT" if b not in (ref.upper(), alt)])
        frac = rng.choice((0.55, 0.8, 0.88, 0.9, 0.93, 1.0, 1.0)) if variant else 0.02
        fwd_p = rng.choice((0.0, 0.05, 0.5, 0.5, 0.5, 0.95, 1.0))
        var_fwd_p = rng.choice((fwd_p, fwd_p, 0.0, 0.04, 0.97, 1.0))
        toks, quals = [], []
        for _ in range(depth):
            r = rng.random()
            fwd = rng.random() < (var_fwd_p if r < frac else fwd_p)
            if r < frac:
                b = alt if fwd else alt.lower()
            elif r < frac + 0.03:
                b = alt2 if fwd else alt2.lower()
            elif r < frac + 0.05:
                b = rng.choice("Nn*")
            else:
                b = "." if fwd else ","
            t = b
            r = rng.random()
            if r < 0.04:
                t = "^" + chr(rng.randint(33, 126)) + t
            elif r < 0.08:
                t = t + "$"
            elif r < 0.12 and b != "*":
                n = rng.choice((1, 1, 2, 3, 11))
                t = t + rng.choice("+-") + str(n) + "".join(rng.choice("ACGTN" if fwd else "acgtn") for _ in range(n))
            toks.append(t)
            quals.append(chr(33 + (rng.randint(0, 14) if rng.random() < 0.12 else rng.randint(15, 41))))
        q = "".join(quals)
        r = rng.random()
        if r < 0.01:
            q = q[:-1]
        elif r < 0.02:
            q = q + "I"
        line = "%s\t%d\t%s\t%d\t%s\t%s" % (chrom, pos, ref, depth, "".join(toks), q)
        if rng.random() < 0.02:
            line += "\t" + "]" * depth
        out.append(line.encode())
    return eol.join(out) + (eol if rng.random() < 0.5 else b"")


def cohort_pileups(seed, n_samples=6, genome_len=5000, contigs=("ctg1", "ctg2"), mean_depth=26, n_scattered=36):
    """A small outbreak: one reference, shared SNP sites (scattered ones, a dense cluster, two near a contig end), samples in
    two clades that carry a site with probability 0.85 / 0.1, reads with sequencing errors, read starts / ends, a few indels,
    '*', low qualities.  Returns (reference {contig: str}, [pileup bytes per sample])."""
    rng = random.Random(seed)
    refs = {c: "".join(rng.choice("ACGT") for _ in range(genome_len)) for c in contigs}
    plan = {}                                                  # (contig, pos) -> (alt, clade)
    for c in contigs:
        pos = set(rng.sample(range(150, genome_len - 150), n_scattered))
        start = rng.randrange(1000, genome_len - 1000)
        pos.update(start + k for k in (0, 9, 23, 40, 41, 77))  # dense cluster
        pos.update((40, genome_len - 30))                      # edge sites
        for p in sorted(pos):
            plan[(c, p)] = (rng.choice([b for b in "ACGT" if b != refs[c][p - 1]]), rng.randrange(2))
    piles = []
    for s in range(n_samples):
        srng = random.Random(seed * 1000 + s)
        carried = {k for k, (_, clade) in plan.items() if srng.random() < (0.85 if clade == s % 2 else 0.1)}
        lines = []
        for c in contigs:
            ref = refs[c]
            for pos in range(1, genome_len + 1):
                if srng.random() < 0.004:
                    continue
                depth = max(0, int(srng.gauss(mean_depth, mean_depth ** 0.5)))
                r = ref[pos - 1]
                if depth == 0:
                    lines.append("%s\t%d\t%s\t0\t*\t*\n" % (c, pos, r))
                    continue
                toks, quals = [], []
                for _ in range(depth):
                    fwd = srng.random() < 0.5
                    if (c, pos) in carried and srng.random() < 0.97:
                        b = plan[(c, pos)][0]
                        t = b if fwd else b.lower()
                    elif srng.random() < 0.004:
                        b = srng.choice([x for x in "ACGT" if x != r])
                        t = b if fwd else b.lower()
                    elif srng.random() < 0.002:
                        t = srng.choice("*Nn")
                    else:
                        t = "." if fwd else ","
                    if srng.random() < 1 / 120:
                        t = "^" + chr(33 + srng.randint(0, 42)) + t
                    if t != "*" and srng.random() < 2e-3:
                        k = srng.randint(1, 3)
                        seq = "".join(srng.choice("ACGT") for _ in range(k))
                        t += srng.choice("+-") + str(k) + (seq if fwd else seq.lower())
                    if srng.random() < 1 / 120:
                        t += "$"
                    toks.append(t)
                    quals.append(chr(33 + min(41, max(2, int(round(srng.gauss(33, 7)))))))
                lines.append("%s\t%d\t%s\t%d\t%s\t%s\n" % (c, pos, r, depth, "".join(toks), "".join(quals)))
        piles.append("".join(lines).encode())
    return refs, piles


def varscan_adversarial(seed, n_lines=2000):
    """Lines whose read-base column is arbitrary printable text (carets at the end, signs followed by huge or missing
    numbers, indel tails that run off the column, digits everywhere, bytes >= 0x80) and whose quality column has any length
    and any non-TAB byte: the walk must agree with its restatement on every one of them."""
    rng = random.Random(seed)
    alphabet = ".,.,.,ACGTacgtNn*^$+-0123456789<>#!~XYZxyz"
    out = []
    for k in range(n_lines):
        depth = rng.choice((8, 9, 15, 30, 60, 200, 999999999))
        n = rng.randint(8, 70)
        bases = bytearray()
        for _ in range(n):
            r = rng.random()
            if r < 0.8:
                bases.append(ord(rng.choice(alphabet)))
            elif r < 0.9:
                bases += rng.choice((b"+", b"-")) + str(rng.choice((0, 1, 2, 5, 12, 300, 10 ** 15))).encode() + b"ACgtN"[:rng.randint(0, 5)]
            elif r < 0.95:
                bases.append(rng.randint(0x80, 0xFF))
            else:
                bases.append(rng.choice((0x20, 0x0B, 0x0C, 0x7F, 0x01)))
        if rng.random() < 0.1:
            bases += rng.choice((b"^", b"+", b"-", b"+1", b"^^", b"+3A"))
        nq = rng.choice((n, n, n, n // 2, n + 9, 1))
        quals = bytes(rng.choice((rng.randint(33, 126), rng.randint(33, 60), rng.randint(0x80, 0xFF), 0x20)) for _ in range(nq))
        quals = quals.replace(b"\t", b"!").replace(b"\n", b"!").replace(b"\r", b"!")
        ref = rng.choice("ACGTNacgtn*.")
        out.append(b"c%d\t%d\t%s\t%d\t%s\t%s" % (k % 3, k + 1, ref.encode(), depth, bytes(bases), quals))
    return b"\n".join(out) + b"\n"


def with_line_ends(data, variant, seed=0):
    """The same pileup with other line ends: "crlf" (every line), "mixed" (LF / CR LF / lone CR at random), "vt_ff" (a '\\v' or
    '\\f' — whitespace to str.split(), no line end to the text-mode reader — before some line ends), or with "repeats" (positions that
    come a second time, later in the file).  Deterministic in seed."""
    rng = random.Random(1000 + seed)
    lines = data.split(b"\n")[:-1]
    if variant == "crlf":
        return b"\r\n".join(lines) + b"\r\n"
    if variant == "mixed":
        return b"".join(ln + rng.choice((b"\n", b"\n", b"\r\n", b"\r")) for ln in lines)
    if variant == "vt_ff":
        return b"".join(ln + rng.choice((b"", b"", b"\x0b", b"\x0c", b" \x0b")) + b"\n" for ln in lines)
    if variant == "repeats":
        # one line in twelve comes again further down (out of position order) with other read bases: the last line of a
        # position is the one that counts (call_consensus.py:171-176)
        out = list(lines)
        for ln in lines:
            f = ln.split(b"\t")
            if len(f) >= 6 and rng.random() < 1 / 12.0:
                swap = bytes.maketrans(b"ACGTacgt.,", b"CATGcatgAa")
                f[4] = f[4].translate(swap)
                out.insert(rng.randrange(len(out) // 2, len(out) + 1), b"\t".join(f))
        return b"\n".join(out) + b"\n"
    raise ValueError(variant)


BAD_LINE_SCENARIOS = ("one_field", "blank", "bad_position", "listed_five_fields", "listed_three_fields", "listed_bad_depth",
                      "unlisted_five_fields", "unlisted_bad_depth", "listed_negative_position_text")


def with_bad_line(data, scenario, listed):
    """The pileup with ONE line changed or inserted, about the middle of the file.  listed: set of (chrom bytes, pos) that the
    caller will parse.  What the reference does with it (pileup.py:423-429, 209-237): a line with fewer than two fields or a
    position that int() refuses raises ValueError wherever it is; the rest of a line is only looked at for listed positions."""
    lines = data.split(b"\n")[:-1]
    mid = len(lines) // 2
    def find(want_listed):
        for k in list(range(mid, len(lines))) + list(range(mid)):
            f = lines[k].split(b"\t")
            if len(f) >= 6 and f[3] != b"0" and ((f[0], int(f[1])) in listed) == want_listed:
                return k, f
        raise ValueError("no such line")
    if scenario == "one_field":
        lines.insert(mid, b"justonefield")
    elif scenario == "blank":
        lines.insert(mid, b"")
    elif scenario == "bad_position":
        k, f = find(False)
        f[1] = f[1] + b"x"
        lines[k] = b"\t".join(f)
    elif scenario == "listed_negative_position_text":
        k, f = find(False)
        f[1] = b"-" + f[1]                                      # an integer for int(): in no site set, no error
        lines[k] = b"\t".join(f)
    elif scenario in ("listed_five_fields", "unlisted_five_fields"):
        k, f = find(scenario.startswith("listed"))
        lines[k] = b"\t".join(f[:5])
    elif scenario == "listed_three_fields":
        k, f = find(True)
        lines[k] = b"\t".join(f[:3])
    elif scenario in ("listed_bad_depth", "unlisted_bad_depth"):
        k, f = find(scenario.startswith("listed"))
        f[3] = b"x7"
        lines[k] = b"\t".join(f)
    else:
        raise ValueError(scenario)
    return b"\n".join(lines) + b"\n"


def vcf_cohort(seed, n_samples=6):
    """SNP records of a small cohort for the region filter: ({contig: length}, {sample name: [(contig, pos), ...] in file order}).
    Shared scattered sites, one dense cluster that only some samples carry, sites near both contig ends, a contig the reference
    FASTA does not list last in name order, and one sample with a single record."""
    rng = random.Random(seed)
    lengths = {"ctgB": 6000, "ctgA": 3500, "c": 900}
    shared = [("ctgB", p) for p in sorted(rng.sample(range(700, 5300), 14))] + [("ctgA", p) for p in sorted(rng.sample(range(700, 2800), 8))]
    cluster = [("ctgB", 2000 + k) for k in (0, 7, 19, 40, 41, 90, 118)]
    ends = [("ctgB", 30), ("ctgB", 5990), ("ctgA", 480), ("ctgA", 3021), ("c", 450)]
    cohort = {}
    for s in range(n_samples):
        recs = [r for r in shared if rng.random() < 0.7]
        if s % 2 == 0:
            recs += [r for r in cluster if rng.random() < 0.9]
        recs += [r for r in ends if rng.random() < 0.5]
        recs += [("ctgB", rng.randrange(600, 5400)) for _ in range(rng.randint(0, 3))]
        recs = sorted(set(recs), key=lambda r: (r[0], r[1]))
        if s == n_samples - 1:
            recs = recs[:1]
        cohort["smp%02d" % s] = recs
    return lengths, cohort


def vcf_text(records):
    head = "##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSample1\n"
    return head + "".join("%s\t%d\t.\tA\tG\t.\tPASS\tADP=20\tGT\t1/1\n" % (c, p) for c, p in records)


def untidy_snpmas():
    """SNP matrix files the distance step must read as the reference's line loop does (distance.py:76-84): [(name, text)]."""
    rng = random.Random(77)
    letters = "ACGTacgt-NnRY*"

    def seq(n):
        return "".join(rng.choice(letters) for _ in range(n))

    def fasta(recs, width=60, eol="\n"):
        return "".join(">" + name + eol + "".join(s[i:i + width] + eol for i in range(0, len(s), width)) for name, s in recs)

    out = []
    out.append(("plain", fasta([("s%02d" % (7 - k), seq(300)) for k in range(7)])))
    out.append(("repeated_id", fasta([("b", seq(120)), ("a", seq(120)), ("b", seq(120)), ("c", seq(120))])))
    out.append(("growing_lengths", fasta([("zeta", seq(500)), ("alpha", seq(300)), ("mid", seq(400)), ("beta", seq(310)), ("omega", seq(500))])))
    out.append(("shorter_later", fasta([("a", seq(50)), ("b", seq(30))])))
    out.append(("crlf_and_unwrapped", fasta([("x", seq(200)), ("y", seq(200))], width=10 ** 6, eol="\r\n")))
    out.append(("header_with_blanks", fasta([("sample one extra words", seq(70)), (">double", seq(70)), ("plain", seq(70))])))
    out.append(("an_empty_record", fasta([("a", ""), ("b", seq(40)), ("c", seq(40))])))
    out.append(("text_before_the_first_header", "ACGT\n" + fasta([("a", seq(20)), ("b", seq(20))])))
    out.append(("one_sample", fasta([("only", seq(33))])))
    return out


def pileup_at_positions(seed, contig_positions, refs=None, neighbours=2, mean_depth=24):
    """A well-formed pileup with lines only at the given positions (and a few neighbours): what a sample looks like whose
    var.flt.vcf exists already and whose reads are not at hand.  contig_positions: [(contig str, sorted positions)] in file
    order; refs: {contig: sequence} for the reference column (N when absent).  At about a third of the listed positions every
    read shows one variant letter; elsewhere the reads match."""
    rng = random.Random(seed)
    out = []
    for contig, positions in contig_positions:
        seq = (refs or {}).get(contig)
        want = set()
        for p in positions:
            for q in range(p - neighbours, p + neighbours + 1):
                if q >= 1 and (seq is None or q <= len(seq)):
                    want.add(q)
        listed = set(positions)
        name = contig.encode()
        for p in sorted(want):
            r = seq[p - 1].upper() if seq is not None else "N"
            depth = max(1, int(rng.gauss(mean_depth, mean_depth ** 0.5)))
            alt = rng.choice([b for b in "ACGT" if b != r]) if (p in listed and rng.random() < 0.35) else None
            toks = []
            for _ in range(depth):
                fwd = rng.random() < 0.5
                t = (alt if fwd else alt.lower()) if alt else ("." if fwd else ",")
                toks.append(t)
            quals = "".join(chr(33 + rng.randint(20, 40)) for _ in range(depth))
            out.append(b"%s\t%d\t%s\t%d\t%s\t%s\n" % (name, p, r.encode(), depth, "".join(toks).encode(), quals.encode()))
    return b"".join(out)
