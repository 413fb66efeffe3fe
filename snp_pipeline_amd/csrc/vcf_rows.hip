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
bool put_row(Out &o, const snpgpu_site_counts &c, uint32_t mask, uint64_t key, const uint8_t *contig_names, const uint32_t *contig_name_off,
             const char *const *filter_names, int preserve_ref_case, char failed_snp_gt, const snpgpu_symbol_spill *spill, uint32_t n_spill) {
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
    const uint32_t cid = (uint32_t)(key >> 32);
    o.putn((const char *)contig_names + contig_name_off[cid], contig_name_off[cid + 1] - contig_name_off[cid]);
    o.put('\t'); o.putu(key & 0xFFFFFFFFull);
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
