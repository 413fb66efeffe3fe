// CHROM / POS of every record of a VCF, and the snplist text: host-side parsing and formatting (no device code).
//
// snpgpu_vcf_sites replaces what utils.convert_vcf_file_to_snp_set (utils.py:1113-1132) and filter_regions.py:408-410 get
// from PyVCF3's Reader — the first two columns of every data line — for 10 000 files of ~1 500 records at configs[4]
// (12 s of Python line handling per pass).  It takes the plain case only: TAB-separated columns, POS of 1..10 plain
// digits below 2^32, text-mode line ends; anything else (and data before the header: PyVCF takes such a line AS the header) is reported
// as SNPGPU_E_UNSUPPORTED and the caller reads that file with its Python reader, which knows the corner cases.
// snpgpu_write_snplist replaces utils.write_list_of_snps (utils.py:1056-1070): "%s\t%d\t%d\t%s\n" per site.
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "internal.h"

// is_vcf: '#' lines are the header (and must come first), blank lines are skipped; a snplist has neither — every line is data
static int read_sites(bool is_vcf, const char *path, uint64_t capacity, uint32_t *out_pos, uint32_t *out_contig, uint64_t *out_n_records,
                      char *out_names, uint64_t names_capacity, uint64_t *out_name_off, uint32_t names_max, uint32_t *out_n_names) {
    if (!path || !out_n_records || !out_n_names || (capacity && (!out_pos || !out_contig)) || (names_max && (!out_names || !out_name_off)))
        return SNPGPU_E_ARG;
    *out_n_records = 0;
    *out_n_names = 0;
    int fd = open(path, O_RDONLY | O_CLOEXEC);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { if (fd >= 0) close(fd); return SNPGPU_E_IO; }
    const size_t n = (size_t)st.st_size;
    std::vector<char> buf(n);
    for (size_t got = 0; got < n;) {
        const ssize_t r = read(fd, buf.data() + got, n - got);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) { close(fd); return SNPGPU_E_IO; }
        got += (size_t)r;
    }
    close(fd);
    const char *p = buf.data();
    std::unordered_map<std::string, uint32_t> ids;
    std::vector<std::string> names;
    const char *last_name = nullptr;
    size_t last_len = 0;
    uint32_t last_id = 0;
    uint64_t n_rec = 0;
    bool header_seen = false;
    size_t pos = 0;
    while (pos < n) {
        // one text-mode line [b, e)
        const size_t b = pos;
        const char *nl = (const char *)memchr(p + pos, '\n', n - pos);
        size_t e = nl ? (size_t)(nl - p) : n;
        const char *cr = (const char *)memchr(p + pos, '\r', e - pos);
        if (cr) { e = (size_t)(cr - p); pos = e + 1; if (pos < n && p[pos] == '\n') ++pos; }
        else pos = nl ? e + 1 : n;
        if (is_vcf && e > b && p[b] == '#') { header_seen = header_seen || !(e - b >= 2 && p[b + 1] == '#'); continue; }   // "##" lines are meta
        bool blank = true;
        for (size_t k = b; k < e && blank; ++k) blank = p[k] == ' ' || (p[k] >= 9 && p[k] <= 13) || (p[k] >= 28 && p[k] <= 31);
        if (blank) { if (is_vcf) continue; return SNPGPU_E_UNSUPPORTED; }   // (utils.read_snp_position_list raises for a blank line)
        if (is_vcf && !header_seen) return SNPGPU_E_UNSUPPORTED; // PyVCF takes this line as the column header: left to the Python reader
        const char *t1 = (const char *)memchr(p + b, '\t', e - b);
        if (!t1 || t1 == p + b) return SNPGPU_E_UNSUPPORTED;
        if (!is_vcf)                                            // str.split() cuts at ANY white space: a name with a blank in it is not plain
            for (const char *q = p + b; q < t1; ++q)
                if (*q == ' ' || (*q >= 9 && *q <= 13) || (*q >= 28 && *q <= 31)) return SNPGPU_E_UNSUPPORTED;
        const char *d0 = t1 + 1;
        const char *t2 = (const char *)memchr(d0, '\t', (size_t)(p + e - d0));
        const char *d1 = t2 ? t2 : p + e;
        if (d1 == d0 || d1 - d0 > 10) return SNPGPU_E_UNSUPPORTED;
        uint64_t v = 0;
        for (const char *q = d0; q < d1; ++q) {
            if (*q < '0' || *q > '9') return SNPGPU_E_UNSUPPORTED;
            v = v * 10 + (uint64_t)(*q - '0');
        }
        if (v >> 32) return SNPGPU_E_UNSUPPORTED;
        const size_t len = (size_t)(t1 - (p + b));
        uint32_t id;
        if (last_name && len == last_len && memcmp(last_name, p + b, len) == 0) id = last_id;
        else {
            std::string key(p + b, len);
            auto it = ids.find(key);
            if (it == ids.end()) { id = (uint32_t)names.size(); ids.emplace(key, id); names.push_back(key); }
            else id = it->second;
            last_name = p + b; last_len = len; last_id = id;
        }
        if (n_rec < capacity) { out_pos[n_rec] = (uint32_t)v; out_contig[n_rec] = id; }
        ++n_rec;
    }
    *out_n_records = n_rec;
    *out_n_names = (uint32_t)names.size();
    uint64_t at = 0;
    if (names.size() <= names_max) {
        bool fits = true;
        for (size_t i = 0; i < names.size() && fits; ++i) fits = (at += names[i].size()) <= names_capacity;
        if (fits) {
            at = 0;
            out_name_off[0] = 0;
            for (size_t i = 0; i < names.size(); ++i) {
                memcpy(out_names + at, names[i].data(), names[i].size());
                at += names[i].size();
                out_name_off[i + 1] = at;
            }
            return SNPGPU_OK;
        }
    }
    return names.empty() ? SNPGPU_OK : SNPGPU_E_NOMEM;           // the caller comes back with more room for the names
}

extern "C" int snpgpu_vcf_sites(const char *path, uint64_t capacity, uint32_t *out_pos, uint32_t *out_contig, uint64_t *out_n_records,
                                char *out_names, uint64_t names_capacity, uint64_t *out_name_off, uint32_t names_max, uint32_t *out_n_names) {
    return read_sites(true, path, capacity, out_pos, out_contig, out_n_records, out_names, names_capacity, out_name_off, names_max, out_n_names);
}

// The first two columns of snplist.txt (utils.read_snp_position_list, utils.py:1073-1088): the same plain case, every line a record.
extern "C" int snpgpu_snplist_sites(const char *path, uint64_t capacity, uint32_t *out_pos, uint32_t *out_contig, uint64_t *out_n_records,
                                    char *out_names, uint64_t names_capacity, uint64_t *out_name_off, uint32_t names_max, uint32_t *out_n_names) {
    return read_sites(false, path, capacity, out_pos, out_contig, out_n_records, out_names, names_capacity, out_name_off, names_max, out_n_names);
}

extern "C" int snpgpu_write_snplist(const char *path, const char *contig_names, const uint64_t *contig_off, const uint64_t *keys, uint64_t n_sites,
                                    const uint32_t *carrier_off, const uint32_t *carriers, const char *sample_names, const uint64_t *sample_off) {
    if (!path || (n_sites && (!contig_names || !contig_off || !keys || !carrier_off || !carriers || !sample_names || !sample_off))) return SNPGPU_E_ARG;
    int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    if (fd < 0) return SNPGPU_E_IO;
    std::vector<char> buf;
    buf.reserve((size_t)8 << 20);
    bool failed = false;
    auto flush = [&]() {
        size_t done = 0;
        while (!failed && done < buf.size()) {
            const ssize_t w = write(fd, buf.data() + done, buf.size() - done);
            if (w < 0) { if (errno == EINTR) continue; failed = true; break; }
            done += (size_t)w;
        }
        buf.clear();
    };
    auto put_u = [&](uint64_t v) {
        char tmp[24];
        int k = 0;
        do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (k) buf.push_back(tmp[--k]);
    };
    for (uint64_t i = 0; i < n_sites && !failed; ++i) {
        const uint32_t c = (uint32_t)(keys[i] >> 32);
        buf.insert(buf.end(), contig_names + contig_off[c], contig_names + contig_off[c + 1]);
        buf.push_back('\t'); put_u(keys[i] & 0xFFFFFFFFull);
        buf.push_back('\t'); put_u(carrier_off[i + 1] - carrier_off[i]);
        buf.push_back('\t');
        for (uint32_t k = carrier_off[i]; k < carrier_off[i + 1]; ++k) {
            if (k > carrier_off[i]) buf.push_back('\t');
            const uint32_t sidx = carriers[k];
            buf.insert(buf.end(), sample_names + sample_off[sidx], sample_names + sample_off[sidx + 1]);
        }
        buf.push_back('\n');
        if (buf.size() > ((size_t)6 << 20)) flush();
    }
    flush();
    if (close(fd) != 0) failed = true;
    return failed ? SNPGPU_E_IO : SNPGPU_OK;
}
