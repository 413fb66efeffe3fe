"""CPU restatement of the consensus.vcf row layout.  TEST INFRASTRUCTURE ONLY.

Restates snppipeline/vcf_writer.py:295-379 (_make_vcf_record_from_pileup) + the text PyVCF3's Writer emits for it.
Pinned by the reference's own doctest answers (vcf_writer.py:400-429), repeated in tests/test_oracle.py, and by the
row syntax of the bundled lambda consensus*.vcf fixtures.
"""

FORMAT_IDS = "GT:SDP:RD:AD:RDF:RDR:ADF:ADR:FT"


def vcf_row(rec, failed, failed_snp_gt=".", preserve_ref_case=False):
    """rec: pileup_oracle.Record; failed: list of filter names or None."""
    ref = rec.reference_base.decode()
    upper_ref = ref.upper()
    if not preserve_ref_case:
        ref = upper_ref
    ur = ord(upper_ref) if len(upper_ref) == 1 else None      # a field of several characters equals no symbol
    if rec.most_common_good_bases is None:
        alt, gt, ad, adf, adr = [], ".", "0", "0", "0"
    else:
        alt = [b for b in rec.most_common_good_bases if b != ur]
        if not alt:
            gt, ad, adf, adr = "0", "0", "0", "0"
        else:
            gt = "0" if rec.most_common_good_bases[0] == ur else "1"
            ad = ",".join(str(rec.base_good_depth.get(b, 0)) for b in alt)
            adf = ",".join(str(rec.forward_base_good_depth.get(b, 0)) for b in alt)
            adr = ",".join(str(rec.reverse_base_good_depth.get(b, 0)) for b in alt)
        if failed:
            gt = "." if failed_snp_gt == "." else ("0" if failed_snp_gt == "0" else "1")
    ft = ";".join(failed) if failed else "PASS"
    data = ":".join([gt, str(rec.raw_depth), str(rec.base_good_depth.get(ur, 0)), ad,
                     str(rec.forward_base_good_depth.get(ur, 0)), str(rec.reverse_base_good_depth.get(ur, 0)), adf, adr, ft])
    return "\t".join([rec.chrom.decode(), str(rec.position), ".", ref, ",".join(chr(b) for b in alt) if alt else ".", ".",
                      ft, "NS=1", FORMAT_IDS, data])
