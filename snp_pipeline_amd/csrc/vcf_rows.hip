// consensus.vcf data lines from per-site records: host-side text formatting (no device code).
//
// Replaces the per-record work of vcf_writer.SingleSampleWriter.write_from_pileup / _make_vcf_record_from_pileup
// (snppipeline/vcf_writer.py:295-435) + the text PyVCF3's Writer emits for it.  Formatting 50 k rows takes Python 0.33 s per
// sample process — more than everything else the call_consensus subcommand does; here it is a few milliseconds.  The layout
// is pinned by the eight lambda consensus*.vcf fixtures (every row, byte for byte) and by the reference's doctest answers.
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

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


// One data line.  `mask`: the failed-filter bits of the row (the record's own, or — for the preserved flow of the pipeline —
// the record's plus Region).  Returns false for a record with more symbols than it keeps.
// head / head_len (nullable): "CHROM\tPOS" as text, for rows whose position is the pileup line's own (--vcfAllPos) instead of a site key.
bool put_row(Out &o, const snpgpu_site_counts &c, uint32_t mask, uint64_t key, const uint8_t *contig_names, const uint32_t *contig_name_off,
             const char *const *filter_names, int preserve_ref_case, char failed_snp_gt, const snpgpu_symbol_spill *spill, uint32_t n_spill,
             const char *head = nullptr, size_t head_len = 0) {
    // the ranked symbols: eight in the record, the rest (rare) in the position's spill record
    const uint32_t n_symbols = c.n_symbols & 0xFFu, spill_code = c.n_symbols >> 8;
    const snpgpu_symbol_spill *more = nullptr;
    if (n_symbols > SNPGPU_MAX_SYMS || spill_code != 0) {
        if (!spill || spill_code == 0 || spill_code - 1 >= n_spill) return false;
        more = &spill[spill_code - 1];
        if (more->n != (n_symbols > SNPGPU_MAX_SYMS ? n_symbols - SNPGPU_MAX_SYMS : 0u) || more->n > SNPGPU_SPILL_SYMS) return false;
        // (a field of more than SNPGPU_SPILL_REF bytes goes on in the records behind its own: they have to be there)
        if (more->ref_len > SNPGPU_SPILL_REF &&
            (more->ref_len - SNPGPU_SPILL_REF + sizeof(snpgpu_symbol_spill) - 1) / sizeof(snpgpu_symbol_spill) > (size_t)(n_spill - spill_code)) return false;
    }
    // a reference field of several bytes: REF shows the string, and no single symbol equals it (vcf_writer.py:295-331)
    const bool long_ref = more && more->ref_len > 1;
    auto sym_of = [&](int k) -> char { return (char)(k < SNPGPU_MAX_SYMS ? c.sym[k] : more->sym[k - SNPGPU_MAX_SYMS]); };
    auto total_of = [&](int k) -> uint32_t { return k < SNPGPU_MAX_SYMS ? c.total[k] : more->total[k - SNPGPU_MAX_SYMS]; };
    auto fwd_of = [&](int k) -> uint32_t { return k < SNPGPU_MAX_SYMS ? c.fwd[k] : more->fwd[k - SNPGPU_MAX_SYMS]; };
    auto rev_of = [&](int k) -> uint32_t { return k < SNPGPU_MAX_SYMS ? c.rev[k] : more->rev[k - SNPGPU_MAX_SYMS]; };
    char ref = (char)c.ref_base, upper_ref = ref;
    if (upper_ref >= 'a' && upper_ref <= 'z') upper_ref = (char)(upper_ref - 32);
    if (!preserve_ref_case) ref = upper_ref;
    // failed filters in bit order
    char ft[256];
    size_t ftn = 0;
    for (int b = 0; b < 6; ++b)
        if (mask >> b & 1) {
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
    int alt[SNPGPU_MAX_SYMS + SNPGPU_SPILL_SYMS], n_alt = 0, ref_at = -1;
    for (uint32_t k = 0; k < n_symbols; ++k) {
        if (!long_ref && sym_of((int)k) == upper_ref) ref_at = (int)k; else alt[n_alt++] = (int)k;
    }
    char gt;
    const bool none = c.good_depth == 0;                    // most_common_good_bases is None
    if (none) { gt = '.'; n_alt = 0; }
    else {
        gt = n_alt == 0 ? '0' : (!long_ref && sym_of(0) == upper_ref ? '0' : '1');
        if (failed) gt = failed_snp_gt == '.' ? '.' : (failed_snp_gt == '0' ? '0' : '1');
    }
    if (head) o.putn(head, head_len);
    else {
        const uint32_t cid = (uint32_t)(key >> 32);
        o.putn((const char *)contig_names + contig_name_off[cid], contig_name_off[cid + 1] - contig_name_off[cid]);
        o.put('\t'); o.putu(key & 0xFFFFFFFFull);
    }
    o.puts_("\t.\t");
    if (long_ref) {
        const uint8_t *field = (const uint8_t *)(more + 1) - SNPGPU_SPILL_REF;        // = more->ref, and on into the next records
        for (uint32_t i = 0; i < more->ref_len; ++i) {
            char ch = (char)field[i];
            if (!preserve_ref_case && ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);
            o.put(ch);
        }
    } else o.put(ref);
    o.put('\t');
    if (n_alt == 0) o.put('.');
    else for (int k = 0; k < n_alt; ++k) { if (k) o.put(','); o.put(sym_of(alt[k])); }
    o.puts_("\t.\t"); o.putn(ft, ftn);
    o.puts_("\tNS=1\tGT:SDP:RD:AD:RDF:RDR:ADF:ADR:FT\t");
    o.put(gt); o.put(':');
    if (more && more->depth64 != 0) { if (more->depth64 < 0) o.put('-'); o.putu(more->depth64 < 0 ? 0ull - (uint64_t)more->depth64 : (uint64_t)more->depth64); }
    else o.putu(c.raw_depth);
    o.put(':');
    o.putu(ref_at >= 0 && !none ? total_of(ref_at) : 0); o.put(':');
    if (n_alt == 0) o.put('0'); else for (int k = 0; k < n_alt; ++k) { if (k) o.put(','); o.putu(total_of(alt[k])); }
    o.put(':'); o.putu(ref_at >= 0 && !none ? fwd_of(ref_at) : 0);
    o.put(':'); o.putu(ref_at >= 0 && !none ? rev_of(ref_at) : 0); o.put(':');
    if (n_alt == 0) o.put('0'); else for (int k = 0; k < n_alt; ++k) { if (k) o.put(','); o.putu(fwd_of(alt[k])); }
    o.put(':');
    if (n_alt == 0) o.put('0'); else for (int k = 0; k < n_alt; ++k) { if (k) o.put(','); o.putu(rev_of(alt[k])); }
    o.put(':'); o.putn(ft, ftn); o.put('\n');
    return true;
}

// ---- rows of --vcfAllPos: one per pileup line, CHROM and POS from the line's own text ---------------------------------------------
__host__ inline bool host_is_ws(uint8_t c) { return (uint8_t)(c - 9u) <= 4u || (uint8_t)(c - 28u) <= 4u; }     // str.split() on ASCII (pileup.py:424)

inline char *put_u32(char *p, uint32_t v) {
    char tmp[10];
    int k = 0;
    do { tmp[k++] = (char)('0' + v % 10u); v /= 10u; } while (v);
    while (k) *p++ = tmp[--k];
    return p;
}

// "CHROM\tPOS" of the line that starts at text[at]: the first two whitespace-separated fields (fields = line.split(), pileup.py:424),
// POS the way str(int(field)) prints it (sign, leading zeros and underscores gone: pileup.py:426, vcf_writer.py:406).  The scan has
// already refused lines without two fields or with a position int() does not take.  Returns the bytes written to `out` (cap bytes), or
// 0 when they do not fit.
size_t line_head(const uint8_t *text, uint64_t n, uint64_t at, char *out, size_t cap) {
    uint64_t p = at;
    while (p < n && host_is_ws(text[p]) && text[p] != '\n' && text[p] != '\r') ++p;
    const uint64_t c0 = p;
    while (p < n && !host_is_ws(text[p])) ++p;
    const uint64_t c1 = p;
    while (p < n && host_is_ws(text[p]) && text[p] != '\n' && text[p] != '\r') ++p;
    const uint64_t p0 = p;
    while (p < n && !host_is_ws(text[p])) ++p;
    const uint64_t p1 = p;
    if (c1 - c0 + (p1 - p0) + 3 > cap) return 0;
    size_t k = 0;
    memcpy(out, text + c0, c1 - c0); k += c1 - c0;
    out[k++] = '\t';
    uint64_t q = p0;
    bool neg = false;
    if (q < p1 && (text[q] == '+' || text[q] == '-')) { neg = text[q] == '-'; ++q; }
    while (q < p1 && (text[q] == '0' || text[q] == '_')) ++q;            // leading zeros (and the underscores between them)
    if (q == p1) { out[k++] = '0'; return k; }                            // int("-0") prints 0
    if (neg) out[k++] = '-';
    for (; q < p1; ++q) if (text[q] != '_') out[k++] = (char)text[q];
    return k;
}

// The row of a packed record (at most three symbols: vcf_writer.py:295-379 with every list up to three long), written with plain
// pointer stores: `p` has room for SNPGPU_PACKED_ROW_MAX bytes beside the head.
constexpr size_t PACKED_ROW_TAIL_MAX = 192;
char *put_packed_row(char *p, const snpgpu_line_record &r, const char *ft, size_t ftn, bool failed, int preserve_ref_case, char failed_snp_gt) {
    char ref = (char)r.ref_base, upper_ref = ref;
    if (upper_ref >= 'a' && upper_ref <= 'z') upper_ref = (char)(upper_ref - 32);
    if (!preserve_ref_case) ref = upper_ref;
    const uint32_t nsym = r.n_symbols;
    const bool none = (uint32_t)r.total[0] + r.total[1] + r.total[2] == 0;
    int ref_at = -1, alt[SNPGPU_LINE_SYMS], n_alt = 0;
    for (uint32_t k = 0; k < nsym; ++k) { if ((char)r.sym[k] == upper_ref) ref_at = (int)k; else alt[n_alt++] = (int)k; }
    char gt;
    if (none) { gt = '.'; n_alt = 0; }
    else {
        gt = n_alt == 0 ? '0' : ((char)r.sym[0] == upper_ref ? '0' : '1');
        if (failed) gt = failed_snp_gt == '.' ? '.' : (failed_snp_gt == '0' ? '0' : '1');
    }
    memcpy(p, "\t.\t", 3); p += 3;
    *p++ = ref; *p++ = '\t';
    if (n_alt == 0) *p++ = '.';
    else for (int k = 0; k < n_alt; ++k) { if (k) *p++ = ','; *p++ = (char)r.sym[alt[k]]; }
    memcpy(p, "\t.\t", 3); p += 3;
    memcpy(p, ft, ftn); p += ftn;
    static const char mid[] = "\tNS=1\tGT:SDP:RD:AD:RDF:RDR:ADF:ADR:FT\t";
    memcpy(p, mid, sizeof mid - 1); p += sizeof mid - 1;
    *p++ = gt; *p++ = ':';
    p = put_u32(p, r.raw_depth); *p++ = ':';
    const bool have_ref = ref_at >= 0 && !none;
    p = put_u32(p, have_ref ? r.total[ref_at] : 0u); *p++ = ':';
    if (n_alt == 0) *p++ = '0'; else for (int k = 0; k < n_alt; ++k) { if (k) *p++ = ','; p = put_u32(p, r.total[alt[k]]); }
    *p++ = ':'; p = put_u32(p, have_ref ? r.fwd[ref_at] : 0u);
    *p++ = ':'; p = put_u32(p, have_ref ? r.rev[ref_at] : 0u); *p++ = ':';
    if (n_alt == 0) *p++ = '0'; else for (int k = 0; k < n_alt; ++k) { if (k) *p++ = ','; p = put_u32(p, r.fwd[alt[k]]); }
    *p++ = ':';
    if (n_alt == 0) *p++ = '0'; else for (int k = 0; k < n_alt; ++k) { if (k) *p++ = ','; p = put_u32(p, r.rev[alt[k]]); }
    *p++ = ':'; memcpy(p, ft, ftn); p += ftn;
    *p++ = '\n';
    return p;
}

bool write_all(const char *path, const char *a, size_t na, const char *b, size_t nb) {
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    if (fd < 0) return false;
    bool ok = true;
    for (int part = 0; part < 2 && ok; ++part) {
        const char *p = part ? b : a;
        size_t left = part ? nb : na;
        while (left) {
            const ssize_t w = write(fd, p, left);
            if (w < 0) { if (errno == EINTR) continue; ok = false; break; }
            p += w;
            left -= (size_t)w;
        }
    }
    if (close(fd) != 0) ok = false;
    return ok;
}

// consensus.fasta + consensus.vcf of one sample and flow
void write_consensus_job(snpgpu_consensus_job &job, uint32_t n_sites, const uint8_t *contig_names, const uint32_t *contig_name_off, const uint64_t *site_keys,
                         const char *const *filter_names, int preserve_ref_case, char failed_snp_gt, const snpgpu_symbol_spill *spill, uint32_t n_spill,
                         std::vector<char> &text, std::vector<uint32_t> &order) {
    job.rc = SNPGPU_OK;
    job.n_rows = 0;
    if (job.fasta_path) {
        // Bio.SeqIO's FASTA layout (call_consensus.py:189-192): ">id", then the sequence in lines of 60
        const size_t idn = strlen(job.fasta_id);
        text.resize(idn + 2 + job.n_bases + job.n_bases / 60 + 2);
        char *p = text.data();
        *p++ = '>';
        memcpy(p, job.fasta_id, idn); p += idn;
        *p++ = '\n';
        for (uint64_t i = 0; i < job.n_bases; i += 60) {
            const uint64_t n = job.n_bases - i < 60 ? job.n_bases - i : 60;
            memcpy(p, job.sequence + i, n); p += n;
            *p++ = '\n';
        }
        if (!write_all(job.fasta_path, text.data(), (size_t)(p - text.data()), nullptr, 0)) job.rc = SNPGPU_E_IO;
    }
    if (job.vcf_path) {
        // a row for every parsed position that has a pileup line, in pileup order (call_consensus.py:161-180)
        order.clear();
        bool sorted = true;
        uint64_t prev = 0;
        for (uint32_t i = 0; i < n_sites; ++i) {
            if (job.counts[i].status != SNPGPU_ST_OK) continue;
            const uint32_t mask = job.row_filters ? job.row_filters[i] : job.counts[i].filters;
            if (job.site_in_flow && !job.site_in_flow[i] && !(mask & SNPGPU_F_REGION)) continue;
            order.push_back(i);
            if (job.line_off[i] < prev) sorted = false;
            prev = job.line_off[i];
        }
        if (!sorted) std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return job.line_off[x] < job.line_off[y]; });
        text.resize(order.size() * 96 + 4096);
        for (;;) {
            Out o{text.data(), text.size(), 0};
            bool bad = false;
            for (uint32_t i : order) {
                const uint32_t mask = job.row_filters ? job.row_filters[i] : job.counts[i].filters;
                if (!put_row(o, job.counts[i], mask & 0x3Fu, site_keys[i], contig_names, contig_name_off, filter_names, preserve_ref_case, failed_snp_gt, spill, n_spill)) { bad = true; break; }
            }
            if (bad) { job.rc = SNPGPU_E_UNSUPPORTED; break; }
            if (o.n > text.size()) { text.resize(o.n + 4096); continue; }
            if (!write_all(job.vcf_path, job.vcf_header, strlen(job.vcf_header), text.data(), o.n)) job.rc = SNPGPU_E_IO;
            job.n_rows = (uint32_t)order.size();
            break;
        }
    }
}

}  // namespace

// Rows of the lines [lo, hi) of a pileup whose bytes are text[0, nbytes): appended to `out`.  recs[i - first] / line_off[i - first] belong to line i;
// wide lines (recs[i].n_symbols == SNPGPU_LINE_WIDE) take their full record from wide[] — wide_index is ascending, so a range walks
// it from its first entry on.  only_listed: rows for the lines whose position is in the site set only.  Returns false for a record
// the writer refuses (more symbols than it keeps and no spill record): *bad_line says which.
bool snpgpu_line_rows_into(const uint8_t *text, uint64_t nbytes, const uint64_t *line_off, const snpgpu_line_record *recs, uint64_t first, uint64_t lo, uint64_t hi,
                             const uint32_t *wide_index, const snpgpu_site_counts *wide, uint32_t n_wide, const char *const *filter_names,
                             int preserve_ref_case, char failed_snp_gt, const snpgpu_symbol_spill *spill, uint32_t n_spill, int only_listed,
                             std::vector<char> &out, uint64_t *n_rows, uint64_t *bad_line) {
    // the FT text of every filter mask (64 of them), once
    char ft_text[64][256];
    size_t ft_len[64];
    for (uint32_t mask = 0; mask < 64; ++mask) {
        size_t ftn = 0;
        for (int b = 0; b < 6; ++b)
            if (mask >> b & 1) {
                const size_t len = strlen(filter_names[b]);
                if (ftn + len + 2 < sizeof ft_text[0]) {
                    if (ftn) ft_text[mask][ftn++] = ';';
                    memcpy(ft_text[mask] + ftn, filter_names[b], len);
                    ftn += len;
                }
            }
        if (!ftn) { memcpy(ft_text[mask], "PASS", 4); ftn = 4; }
        ft_len[mask] = ftn;
    }
    uint32_t w = (uint32_t)(std::lower_bound(wide_index, wide_index + n_wide, (uint32_t)lo) - wide_index);
    size_t used = out.size();
    out.resize(used + (size_t)(hi - lo) * 128 + 4096);          // a row of a 30x line is ~95 bytes: one allocation for nearly every range
    uint64_t rows = 0;
    char head[512];
    std::vector<char> long_head;
    for (uint64_t i = lo; i < hi; ++i) {
        const snpgpu_line_record &r = recs[i - first];
        const bool is_wide = r.n_symbols == SNPGPU_LINE_WIDE;
        const uint32_t wi = is_wide ? w++ : 0u;
        if (only_listed && !r.site_flags) continue;
        const uint64_t at = line_off[i - first] - 1;
        const char *hp = head;
        size_t hn = line_head(text, nbytes, at, head, sizeof head);
        if (!hn) {                                               // a contig name of hundreds of bytes
            uint64_t e = at;
            while (e < nbytes && text[e] != '\n' && text[e] != '\r') ++e;
            long_head.resize((size_t)(e - at) + 8);
            hn = line_head(text, nbytes, at, long_head.data(), long_head.size());
            hp = long_head.data();
        }
        if (!is_wide) {
            const uint32_t mask = r.filters & 0x3Fu;
            const size_t need = hn + PACKED_ROW_TAIL_MAX + 2 * ft_len[mask];
            if (out.size() < used + need) out.resize((used + need) * 2);
            char *p = out.data() + used;
            memcpy(p, hp, hn);
            p = put_packed_row(p + hn, r, ft_text[mask], ft_len[mask], mask != 0, preserve_ref_case, failed_snp_gt);
            used = (size_t)(p - out.data());
        } else {
            if (wi >= n_wide || wide_index[wi] != (uint32_t)i) { *bad_line = i; return false; }
            const snpgpu_site_counts &c = wide[wi];
            for (;;) {
                Out o{out.data() + used, out.size() - used, 0};
                if (!put_row(o, c, c.filters & 0x3Fu, 0, nullptr, nullptr, filter_names, preserve_ref_case, failed_snp_gt, spill, n_spill, hp, hn)) { *bad_line = i; return false; }
                if (o.n > o.cap) { out.resize((used + o.n) * 2 + 4096); continue; }
                used += o.n;
                break;
            }
        }
        ++rows;
    }
    out.resize(used);
    *n_rows = rows;
    return true;
}

// The same as a plain function of host arrays (a host that took the records of snpgpu_call_all_lines_compact_file and wants rows; the
// CPU tests and the sanitizer runs of this text-parsing code): rows of lines [0, n_lines) into out[0, capacity); returns the bytes the
// rows take (call with capacity 0 to size the buffer).
extern "C" size_t snpgpu_format_line_rows(const uint8_t *pileup, uint64_t nbytes, const uint64_t *line_off, const snpgpu_line_record *records,
                                          uint64_t n_lines, const uint32_t *wide_index, const snpgpu_site_counts *wide, uint32_t n_wide,
                                          const char *const *filter_names, int preserve_ref_case, char failed_snp_gt,
                                          const snpgpu_symbol_spill *spill, uint32_t n_spill, int only_listed, char *out, size_t capacity,
                                          uint64_t *out_n_rows, int64_t *out_bad_line) {
    if (out_n_rows) *out_n_rows = 0;
    if (out_bad_line) *out_bad_line = -1;
    if (n_lines && (!pileup || !line_off || !records || !filter_names)) return 0;
    for (uint64_t i = 0; i < n_lines; ++i)
        if (line_off[i] == 0 || line_off[i] > nbytes) { if (out_bad_line) *out_bad_line = (int64_t)i; return 0; }     // (offsets are 1 + byte offset)
    std::vector<char> text;
    uint64_t rows = 0, bad = 0;
    static const uint32_t no_index = 0;
    static const snpgpu_site_counts no_wide = {};
    if (!snpgpu_line_rows_into(pileup, nbytes, line_off, records, 0, 0, n_lines, n_wide ? wide_index : &no_index, n_wide ? wide : &no_wide, n_wide, filter_names,
                               preserve_ref_case, failed_snp_gt, spill, n_spill, only_listed, text, &rows, &bad)) {
        if (out_bad_line) *out_bad_line = (int64_t)bad;
        return 0;
    }
    if (out_n_rows) *out_n_rows = rows;
    if (out && capacity) memcpy(out, text.data(), text.size() < capacity ? text.size() : capacity);
    return text.size();
}

extern "C" size_t snpgpu_format_vcf_rows(const snpgpu_site_counts *counts, const uint32_t *order, uint32_t n_rows,
                                         const uint8_t *contig_names, const uint32_t *contig_name_off, const uint64_t *site_keys,
                                         const char *const *filter_names, int preserve_ref_case, char failed_snp_gt,
                                         const snpgpu_symbol_spill *spill, uint32_t n_spill,
                                         char *out, size_t capacity, int32_t *out_bad_row) {
    Out o{out, out ? capacity : 0, 0};
    if (out_bad_row) *out_bad_row = -1;
    for (uint32_t r = 0; r < n_rows; ++r) {
        const uint32_t idx = order ? order[r] : r;
        const snpgpu_site_counts &c = counts[idx];
        if (!put_row(o, c, c.filters & 0x3Fu, site_keys[idx], contig_names, contig_name_off, filter_names, preserve_ref_case, failed_snp_gt, spill, n_spill)) {
            if (out_bad_row && *out_bad_row < 0) *out_bad_row = (int32_t)r;   // more than 8 symbols and no spill record: the caller raises
        }
    }
    return o.n;
}

// The output files of call_consensus for many (sample, flow) pairs at once, on `n_threads` host threads (0: one per job up to
// the hardware's): the text of 125 samples x 2 flows x 50 000 rows is a gigabyte — formatted and written by one thread it would
// take longer than the pileups take to cross the host link.
extern "C" int snpgpu_write_consensus_files(snpgpu_consensus_job *jobs, uint32_t n_jobs, uint32_t n_sites, const uint8_t *contig_names,
                                            const uint32_t *contig_name_off, const uint64_t *site_keys, const char *const *filter_names,
                                            int preserve_ref_case, char failed_snp_gt, const snpgpu_symbol_spill *spill, uint32_t n_spill,
                                            uint32_t n_threads) {
    if (n_jobs && !jobs) return SNPGPU_E_ARG;
    if (!n_threads) {
        n_threads = snpgpu_cpu_threads(64);                       // one job per thread, within this process's CPU budget
    }
    if (n_threads > n_jobs) n_threads = n_jobs;
    if (n_threads < 1) n_threads = 1;
    std::atomic<uint32_t> next{0};
    auto work = [&]() {
        std::vector<char> text;
        std::vector<uint32_t> order;
        for (;;) {
            const uint32_t j = next.fetch_add(1);
            if (j >= n_jobs) return;
            write_consensus_job(jobs[j], n_sites, contig_names, contig_name_off, site_keys, filter_names, preserve_ref_case, failed_snp_gt, spill, n_spill, text, order);
        }
    };
    std::vector<std::thread> pool;
    try {
        for (uint32_t t = 1; t < n_threads; ++t) pool.emplace_back(work);
    } catch (...) {}                                            // fewer threads than asked for: the others do the work
    work();
    for (auto &t : pool) t.join();
    return SNPGPU_OK;
}
