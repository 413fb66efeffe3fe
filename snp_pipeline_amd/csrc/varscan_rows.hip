// var.flt.vcf data lines from the site-calling records: host-side arithmetic and text (no device code).
//
// The second half of `VarScan mpileup2snp --output-vcf 1` for the (line, allele) records of csrc/varscan.hip:
// VarScan.getSignificance + FishersExact (PVAL, GQ, --p-value), the choice among the alleles of one line, the strand
// filter, the homozygous threshold, and the VCF 4.1 line as VarScan v2.3.9 prints it (snppipeline/call_sites.py:99-105
// keeps that output as var.flt.vcf).  Pinned by the 69 019 data lines of the reference's bundled var.flt.vcf files: each is
// rebuilt from its own counts (tests/test_host_cpu.py).  libm's log / exp / log10 and printf's exact decimal rounding give
// the digits Java's Math and DecimalFormat (HALF_EVEN) give on all of them.
#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "internal.h"

namespace {

struct Fisher {
    std::vector<double> lf{0.0};
    void upto(uint64_t n) {
        while (lf.size() <= n) lf.push_back(lf.back() + log((double)lf.size()));
    }
    double term(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
        const uint64_t n = a + b + c + d;
        upto(n);
        return exp(lf[a + b] + lf[c + d] + lf[a + c] + lf[b + d] - (lf[a] + lf[b] + lf[c] + lf[d] + lf[n]));
    }
    double right_tail(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
        double total = term(a, b, c, d);
        const uint64_t steps = c < b ? c : b;
        for (uint64_t k = 1; k <= steps; ++k) total += term(a + k, b - k, c - k, d + k);
        return total;
    }
    double two_tails(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
        const double here = term(a, b, c, d);
        double total = here;
        for (uint64_t k = 1; k <= (a < d ? a : d); ++k) { const double t = term(a - k, b + k, c + k, d - k); if (t <= here) total += t; }
        for (uint64_t k = 1; k <= (b < c ? b : c); ++k) { const double t = term(a + k, b - k, c - k, d + k); if (t <= here) total += t; }
        return total;
    }
    // VarScan.getSignificance(reads1, reads2): against the split a 0.001 error rate predicts at that coverage
    double variant_p(uint32_t reads1, uint32_t reads2) {
        const uint64_t cover = (uint64_t)reads1 + reads2;
        const uint64_t expected2 = (uint64_t)((double)cover * 0.001);
        return right_tail(cover - expected2, expected2, reads1, reads2);
    }
};

struct Out {
    char *p;
    size_t cap, n;
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

void strip_zeros(char *s) {                                     // "98.10" -> "98.1", "100.00" -> "100"
    char *dot = strchr(s, '.');
    if (!dot) return;
    char *e = s + strlen(s);
    while (e > dot + 1 && e[-1] == '0') --e;
    if (e == dot + 1) e = dot;
    *e = 0;
}

void put_sci(Out &o, double p) {                                // DecimalFormat("0.####E0")
    if (p == 0.0) { o.puts_("0E0"); return; }
    char buf[64];
    snprintf(buf, sizeof buf, "%.4e", p);
    char *e = strchr(buf, 'e');
    const int ex = atoi(e + 1);
    *e = 0;
    strip_zeros(buf);
    o.puts_(buf);
    o.put('E');
    if (ex < 0) { o.put('-'); o.putu((uint64_t)(-ex)); } else o.putu((uint64_t)ex);
}

void put_percent(Out &o, uint32_t part, uint32_t whole) {       // DecimalFormat("###.##") + '%'
    char buf[64];
    snprintf(buf, sizeof buf, "%.2f", ((double)part / (double)whole) * 100.0);
    strip_zeros(buf);
    o.puts_(buf);
    o.put('%');
}

}  // namespace

extern "C" size_t snpgpu_varscan_format_rows(const snpgpu_varscan_site *sites, uint32_t n_sites, const uint8_t *pileup, uint64_t pileup_bytes,
                                             const snpgpu_varscan_finish *fin, char *out, size_t capacity, uint32_t *out_rows) {
    Out o{out, out ? capacity : 0, 0};
    Fisher fisher;
    uint32_t rows = 0;
    for (uint32_t i = 0; i < n_sites;) {
        // the alleles of one line: most variant reads wins among those within --p-value, the first on ties
        const uint64_t off = sites[i].line_off;
        const snpgpu_varscan_site *best = nullptr;
        uint32_t best_ad = 0;
        double best_p = 1.0;
        for (; i < n_sites && sites[i].line_off == off; ++i) {
            const snpgpu_varscan_site &s = sites[i];
            const uint32_t ad = s.adf + s.adr;
            const double p = fisher.variant_p(s.rdf + s.rdr, ad);
            if (p <= fin->p_value && (!best || ad > best_ad)) { best = &s; best_ad = ad; best_p = p; }
        }
        if (!best || off >= pileup_bytes) continue;
        const snpgpu_varscan_site &s = *best;
        const uint32_t rd = s.rdf + s.rdr, ad = best_ad;
        // chrom and position as they stand in the pileup line
        uint64_t t1 = off, t2;
        while (t1 < pileup_bytes && pileup[t1] != '\t') ++t1;
        t2 = t1 + 1;
        while (t2 < pileup_bytes && pileup[t2] != '\t') ++t2;
        if (t2 >= pileup_bytes) continue;
        bool str10 = false;
        if (fin->strand_filter) {
            const double var_plus = (double)s.adf / (double)ad;
            if ((var_plus < 0.10 || var_plus > 0.90) && rd >= 2) {
                const double ref_plus = (double)s.rdf / (double)rd;
                str10 = ref_plus >= 0.10 && ref_plus <= 0.90 && fisher.two_tails(s.rdf, s.rdr, s.adf, s.adr) < 0.01;
            }
        }
        const bool hom = (double)ad / (double)s.total >= fin->min_freq_for_hom;
        const uint32_t gq = best_p <= 0.0 ? 255u : (uint32_t)fmin(255.0, fmax(0.0, trunc(-10.0 * log10(best_p))));
        o.putn((const char *)pileup + off, (size_t)(t1 - off)); o.put('\t');
        o.putn((const char *)pileup + t1 + 1, (size_t)(t2 - t1 - 1));
        o.puts_("\t.\t"); o.put((char)s.ref_base); o.put('\t'); o.put((char)s.alt_base);
        o.puts_("\t.\t"); o.puts_(str10 ? "str10" : "PASS");
        o.puts_("\tADP="); o.putu(s.dp); o.puts_(";WT=0;HET="); o.put(hom ? '0' : '1'); o.puts_(";HOM="); o.put(hom ? '1' : '0');
        o.puts_(";NC=0\tGT:GQ:SDP:DP:RD:AD:FREQ:PVAL:RBQ:ABQ:RDF:RDR:ADF:ADR\t");
        o.puts_(hom ? "1/1:" : "0/1:"); o.putu(gq); o.put(':'); o.putu(s.sdp); o.put(':'); o.putu(s.dp); o.put(':'); o.putu(rd); o.put(':');
        o.putu(ad); o.put(':'); put_percent(o, ad, s.total); o.put(':'); put_sci(o, best_p); o.put(':');
        o.putu(rd ? s.ref_qual_sum / rd : 0); o.put(':'); o.putu(s.alt_qual_sum / ad); o.put(':');
        o.putu(s.rdf); o.put(':'); o.putu(s.rdr); o.put(':'); o.putu(s.adf); o.put(':'); o.putu(s.adr); o.put('\n');
        ++rows;
    }
    if (out_rows) *out_rows = rows;
    return o.n;
}
