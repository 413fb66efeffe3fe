// consensus.vcf data lines from per-site records: host-side text formatting (no device code).
//
// Replaces the per-record work of vcf_writer.SingleSampleWriter.write_from_pileup / _make_vcf_record_from_pileup
// (snppipeline/vcf_writer.py:295-435) + the text PyVCF3's Writer emits for it.  Formatting 50 k rows takes Python 0.33 s per
// sample process — more than everything else the call_consensus subcommand does; here it is a few milliseconds.  The layout
// is pinned by the eight lambda consensus*.vcf fixtures (every row, byte for byte) and by the reference's doctest answers.
#include <string.h>

#include "internal.h"

namespace {

struct Out {
    char *p;
    size_t cap, n;          // n counts every byte, also those that did not fit
    void put(char c) { if (n < cap) p[n] = c; ++n; }
    void puts_(const char *s) { while (*s) put(*s++); }
    void putn(const char *s, size_t len) { for (size_t i = 0; i < len; ++i) put(s[i]); }
    void putu(uint64_t v) {
        char tmp[24];
        int k = 0;
        do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (k) put(tmp[--k]);
    }
};

}  // namespace

extern "C" size_t snpgpu_format_vcf_rows(const snpgpu_site_counts *counts, const uint32_t *order, uint32_t n_rows,
                                         const uint8_t *contig_names, const uint32_t *contig_name_off, const uint64_t *site_keys,
                                         const char *const *filter_names, int preserve_ref_case, char failed_snp_gt,
                                         char *out, size_t capacity, int32_t *out_bad_row) {
    Out o{out, out ? capacity : 0, 0};
    if (out_bad_row) *out_bad_row = -1;
    for (uint32_t r = 0; r < n_rows; ++r) {
        const uint32_t idx = order ? order[r] : r;
        const snpgpu_site_counts &c = counts[idx];
        if (c.n_symbols > SNPGPU_MAX_SYMS) {                    // the record keeps 8 symbols: the caller raises
            if (out_bad_row && *out_bad_row < 0) *out_bad_row = (int32_t)r;
            continue;
        }
        char ref = (char)c.ref_base, upper_ref = ref;
        if (upper_ref >= 'a' && upper_ref <= 'z') upper_ref = (char)(upper_ref - 32);
        if (!preserve_ref_case) ref = upper_ref;
        // failed filters in bit order
        char ft[256];
        size_t ftn = 0;
        for (int b = 0; b < 6; ++b)
            if (c.filters >> b & 1) {
                const size_t len = strlen(filter_names[b]);
                if (ftn + len + 2 < sizeof ft) {
                    if (ftn) ft[ftn++] = ';';
                    memcpy(ft + ftn, filter_names[b], len);
                    ftn += len;
                }
            }
        const bool failed = ftn != 0;
        if (!failed) { memcpy(ft, "PASS", 4); ftn = 4; }
        // ALT = ranked symbols other than the (upper-case) reference
        int alt[SNPGPU_MAX_SYMS], n_alt = 0, ref_at = -1;
        for (uint32_t k = 0; k < c.n_symbols; ++k) {
            if ((char)c.sym[k] == upper_ref) ref_at = (int)k; else alt[n_alt++] = (int)k;
        }
        char gt;
        const bool none = c.good_depth == 0;                    // most_common_good_bases is None
        if (none) { gt = '.'; n_alt = 0; }
        else {
            gt = n_alt == 0 ? '0' : ((char)c.sym[0] == upper_ref ? '0' : '1');
            if (failed) gt = failed_snp_gt == '.' ? '.' : (failed_snp_gt == '0' ? '0' : '1');
        }
        const uint32_t cid = (uint32_t)(site_keys[idx] >> 32);
        o.putn((const char *)contig_names + contig_name_off[cid], contig_name_off[cid + 1] - contig_name_off[cid]);
        o.put('\t'); o.putu(site_keys[idx] & 0xFFFFFFFFull);
        o.puts_("\t.\t"); o.put(ref); o.put('\t');
        if (n_alt == 0) o.put('.');
        else for (int k = 0; k < n_alt; ++k) { if (k) o.put(','); o.put((char)c.sym[alt[k]]); }
        o.puts_("\t.\t"); o.putn(ft, ftn);
        o.puts_("\tNS=1\tGT:SDP:RD:AD:RDF:RDR:ADF:ADR:FT\t");
        o.put(gt); o.put(':'); o.putu(c.raw_depth); o.put(':');
        o.putu(ref_at >= 0 && !none ? c.total[ref_at] : 0); o.put(':');
        if (n_alt == 0) o.put('0'); else for (int k = 0; k < n_alt; ++k) { if (k) o.put(','); o.putu(c.total[alt[k]]); }
        o.put(':'); o.putu(ref_at >= 0 && !none ? c.fwd[ref_at] : 0);
        o.put(':'); o.putu(ref_at >= 0 && !none ? c.rev[ref_at] : 0); o.put(':');
        if (n_alt == 0) o.put('0'); else for (int k = 0; k < n_alt; ++k) { if (k) o.put(','); o.putu(c.fwd[alt[k]]); }
        o.put(':');
        if (n_alt == 0) o.put('0'); else for (int k = 0; k < n_alt; ++k) { if (k) o.put(','); o.putu(c.rev[alt[k]]); }
        o.put(':'); o.putn(ft, ftn); o.put('\n');
    }
    return o.n;
}
