// snpma.fasta -> ids + a samples x sites byte matrix: host-side parsing (no device code).
//
// Replaces the read loop of snppipeline/distance.py:76-84 (text-mode lines, a line that starts with '>' opens a record
// named by the rest of the line without its leading '>'s, every other line is appended to the current record).  At
// BASELINE configs[4] the file is 2 GB in 3.3e7 lines: 23 s of Python string work in front of a 38 ms kernel; here two
// passes over an mmap with memchr.  Line ends as Python's universal newlines: "\n", "\r\n" and a lone "\r".
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "internal.h"

namespace {

struct Mapped {
    const uint8_t *p = nullptr;
    size_t n = 0;
    int fd = -1;
    bool ok = false;
    explicit Mapped(const char *path) {
        fd = open(path, O_RDONLY | O_CLOEXEC);
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return;
        n = (size_t)st.st_size;
        if (n) {
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) return;
            p = (const uint8_t *)m;
            (void)madvise(m, n, MADV_SEQUENTIAL);
        }
        ok = true;
    }
    ~Mapped() {
        if (p) munmap((void *)p, n);
        if (fd >= 0) close(fd);
    }
};

// the next line [b, e) and the start of the one after it
inline bool next_line(const uint8_t *p, size_t n, size_t &pos, size_t &b, size_t &e) {
    if (pos >= n) return false;
    b = pos;
    const uint8_t *nl = (const uint8_t *)memchr(p + pos, '\n', n - pos);
    size_t stop = nl ? (size_t)(nl - p) : n;
    const uint8_t *cr = (const uint8_t *)memchr(p + pos, '\r', stop - pos);      // (rare: only looked for inside the line)
    if (cr) {
        e = (size_t)(cr - p);
        pos = e + 1;
        if (pos < n && p[pos] == '\n') ++pos;
        return true;
    }
    e = stop;
    pos = nl ? stop + 1 : n;
    return true;
}

template <typename OnHeader, typename OnData>
void walk(const Mapped &m, OnHeader on_header, OnData on_data) {
    size_t pos = 0, b, e;
    while (next_line(m.p, m.n, pos, b, e)) {
        if (e > b && m.p[b] == '>') {
            size_t s = b;
            while (s < e && m.p[s] == '>') ++s;                  // lstrip('>')
            on_header(s, e);
        } else {
            on_data(b, e);
        }
    }
}

}  // namespace

extern "C" int snpgpu_fasta_scan(const char *path, uint64_t *out_n_records, uint64_t *out_max_len, uint64_t *out_names_bytes) {
    if (!path || !out_n_records || !out_max_len || !out_names_bytes) return SNPGPU_E_ARG;
    Mapped m(path);
    if (!m.ok) return SNPGPU_E_IO;
    uint64_t n_rec = 0, cur = 0, longest = 0, names = 0;
    bool orphan = false;
    walk(m,
         [&](size_t s, size_t e) { if (cur > longest) longest = cur; cur = 0; ++n_rec; names += e - s; },
         [&](size_t b, size_t e) { if (!n_rec) orphan = true; cur += e - b; });
    if (cur > longest) longest = cur;
    *out_n_records = n_rec;
    *out_max_len = longest;
    *out_names_bytes = names;
    return orphan ? SNPGPU_E_UNSUPPORTED : SNPGPU_OK;              // sequence text before the first header
}

extern "C" int snpgpu_fasta_load(const char *path, uint64_t n_records, uint64_t row_stride, uint8_t pad, uint8_t *out_matrix, uint64_t *out_len,
                                 char *out_names, uint64_t *out_name_off) {
    if (!path || (n_records && (!out_len || !out_name_off || !out_names)) || (n_records && row_stride && !out_matrix)) return SNPGPU_E_ARG;
    Mapped m(path);
    if (!m.ok) return SNPGPU_E_IO;
    uint64_t rec = 0, name_at = 0;
    bool overflow = false;
    if (out_name_off) out_name_off[0] = 0;
    walk(m,
         [&](size_t s, size_t e) {
             if (rec >= n_records) { overflow = true; ++rec; return; }
             if (e > s) memcpy(out_names + name_at, m.p + s, e - s);
             name_at += e - s;
             out_name_off[rec + 1] = name_at;
             out_len[rec] = 0;
             ++rec;
         },
         [&](size_t b, size_t e) {
             if (!rec || rec > n_records) { if (!rec && e > b) overflow = true; return; }
             uint64_t &len = out_len[rec - 1];
             if (len + (e - b) > row_stride) { overflow = true; return; }
             if (e > b) memcpy(out_matrix + (rec - 1) * row_stride + len, m.p + b, e - b);
             len += e - b;
         });
    if (overflow || rec != n_records) return SNPGPU_E_ARG;         // the file changed between the two passes
    for (uint64_t r = 0; r < n_records; ++r)
        if (out_len[r] < row_stride) memset(out_matrix + r * row_stride + out_len[r], pad, row_stride - out_len[r]);
    return SNPGPU_OK;
}
