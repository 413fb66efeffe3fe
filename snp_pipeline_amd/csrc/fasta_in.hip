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

#include <thread>
#include <vector>

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

// the next line [b, e) and the start of the one after it: the nearest "\n" or "\r" ends it (looked for in growing windows, so
// that neither a file without "\n" nor one without "\r" is scanned to its end for every line)
inline bool next_line(const uint8_t *p, size_t n, size_t &pos, size_t &b, size_t &e) {
    if (pos >= n) return false;
    b = pos;
    size_t scan = pos, win = 256;
    for (;;) {
        const size_t len = n - scan < win ? n - scan : win;
        const uint8_t *nl = (const uint8_t *)memchr(p + scan, '\n', len);
        const uint8_t *cr = (const uint8_t *)memchr(p + scan, '\r', nl ? (size_t)(nl - (p + scan)) : len);
        if (cr) {
            e = (size_t)(cr - p);
            pos = e + 1;
            if (pos < n && p[pos] == '\n') ++pos;
            return true;
        }
        if (nl) { e = (size_t)(nl - p); pos = e + 1; return true; }
        scan += len;
        if (scan >= n) { e = n; pos = n; return true; }
        if (win < ((size_t)1 << 20)) win *= 4;
    }
}

// walk the lines that START in [lo, hi) (hi a range bound, not a line bound: the last line may run past it)
template <typename OnHeader, typename OnData>
void walk_range(const Mapped &m, size_t lo, size_t hi, OnHeader on_header, OnData on_data) {
    size_t pos = lo, b, e;
    if (lo > 0 && lo < m.n) {
        // `lo` starts a line when the byte before it ends one: "\n", or a "\r" that `lo` does not complete to "\r\n"; else the
        // line that covers lo belongs to the range before: skip to the start of the next one
        const uint8_t prev = m.p[lo - 1];
        if (prev == '\r' && m.p[lo] == '\n') pos = lo + 1;
        else if (prev != '\n' && prev != '\r') (void)next_line(m.p, m.n, pos, b, e);
    }
    while (pos < hi && next_line(m.p, m.n, pos, b, e)) {
        if (e > b && m.p[b] == '>') {
            size_t s = b;
            while (s < e && m.p[s] == '>') ++s;                  // lstrip('>')
            on_header(s, e);
        } else {
            on_data(b, e);
        }
    }
}

template <typename OnHeader, typename OnData>
void walk(const Mapped &m, OnHeader on_header, OnData on_data) { walk_range(m, 0, m.n, on_header, on_data); }

// The file in byte ranges, one per thread (2 GB of text in 3.3e7 lines at BASELINE configs[4]: one thread needs seconds).
struct RangeInfo {
    size_t lo = 0, hi = 0;
    uint64_t lead = 0;                                          // data bytes before the range's first header (they belong to a record of an earlier range)
    bool lead_any = false;                                      // ... any data LINE there (an empty one counts for the orphan test)
    std::vector<uint64_t> rec_len;                              // data bytes, within this range, of each record that starts in it
    uint64_t names = 0;
};

unsigned fasta_threads(size_t n) {
    unsigned t = snpgpu_writer_threads(0);
    const size_t per = (size_t)16 << 20;                        // not worth a thread below 16 MB each
    if (n / per < t) t = (unsigned)(n / per);
    return t < 1 ? 1 : t;
}

std::vector<RangeInfo> survey(const Mapped &m) {
    const unsigned T = fasta_threads(m.n);
    std::vector<RangeInfo> r(T);
    for (unsigned t = 0; t < T; ++t) { r[t].lo = m.n / T * t; r[t].hi = t + 1 == T ? m.n : m.n / T * (t + 1); }
    auto work = [&](unsigned t) {
        RangeInfo &ri = r[t];
        walk_range(m, ri.lo, ri.hi,
                   [&](size_t s, size_t e) { ri.rec_len.push_back(0); ri.names += e - s; },
                   [&](size_t b, size_t e) { if (ri.rec_len.empty()) { ri.lead += e - b; ri.lead_any = true; } else ri.rec_len.back() += e - b; });
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    return r;
}

}  // namespace

extern "C" int snpgpu_fasta_scan(const char *path, uint64_t *out_n_records, uint64_t *out_max_len, uint64_t *out_names_bytes) {
    if (!path || !out_n_records || !out_max_len || !out_names_bytes) return SNPGPU_E_ARG;
    Mapped m(path);
    if (!m.ok) return SNPGPU_E_IO;
    const std::vector<RangeInfo> r = survey(m);
    uint64_t n_rec = 0, cur = 0, longest = 0, names = 0;
    bool orphan = false;
    for (const RangeInfo &ri : r) {
        if (n_rec == 0 && ri.lead_any) orphan = true;           // sequence text before the first header
        cur += ri.lead;
        for (uint64_t len : ri.rec_len) { if (cur > longest) longest = cur; cur = len; ++n_rec; }
        names += ri.names;
    }
    if (cur > longest) longest = cur;
    *out_n_records = n_rec;
    *out_max_len = n_rec ? longest : 0;
    *out_names_bytes = names;
    return orphan ? SNPGPU_E_UNSUPPORTED : SNPGPU_OK;
}

extern "C" int snpgpu_fasta_load(const char *path, uint64_t n_records, uint64_t row_stride, uint8_t pad, uint8_t *out_matrix, uint64_t *out_len,
                                 char *out_names, uint64_t *out_name_off) {
    if (!path || (n_records && (!out_len || !out_name_off || !out_names)) || (n_records && row_stride && !out_matrix)) return SNPGPU_E_ARG;
    Mapped m(path);
    if (!m.ok) return SNPGPU_E_IO;
    const std::vector<RangeInfo> r = survey(m);
    // where every range starts: its first record, the bytes its leading lines add to the record before, its first name byte
    const size_t T = r.size();
    std::vector<uint64_t> rec0(T + 1, 0), carry(T, 0), name0(T + 1, 0);
    uint64_t cur = 0;
    bool overflow = false;
    for (size_t t = 0; t < T; ++t) {
        carry[t] = cur;
        cur += r[t].lead;
        if (rec0[t] == 0 && r[t].lead_any) overflow = true;     // text before the first header: snpgpu_fasta_scan has refused the file
        for (uint64_t len : r[t].rec_len) {
            if (cur > row_stride) overflow = true;
            cur = len;
        }
        rec0[t + 1] = rec0[t] + r[t].rec_len.size();
        name0[t + 1] = name0[t] + r[t].names;
    }
    if (cur > row_stride) overflow = true;
    if (overflow || rec0[T] != n_records) return SNPGPU_E_ARG;  // the file changed between the two calls
    if (out_name_off) out_name_off[0] = 0;
    std::vector<uint8_t> bad(T, 0);
    auto work = [&](size_t t) {
        uint64_t rec = rec0[t], name_at = name0[t], len = carry[t];   // (rec: records opened so far; the current one is rec - 1)
        walk_range(m, r[t].lo, r[t].hi,
                   [&](size_t s, size_t e) {
                       if (e > s) memcpy(out_names + name_at, m.p + s, e - s);
                       name_at += e - s;
                       out_name_off[rec + 1] = name_at;
                       len = 0;
                       ++rec;
                   },
                   [&](size_t b, size_t e) {
                       if (rec == 0) return;
                       if (len + (e - b) > row_stride) { bad[t] = 1; return; }
                       if (e > b) memcpy(out_matrix + (rec - 1) * row_stride + len, m.p + b, e - b);
                       len += e - b;
                   });
        // the length of the range's last record is complete only when no later range adds leading lines to it: the lengths are
        // set after the join, from the survey
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    for (size_t t = 0; t < T; ++t) if (bad[t]) return SNPGPU_E_ARG;
    {   // record lengths from the survey (a record may collect bytes from several ranges)
        uint64_t rec = 0;
        cur = 0;
        for (size_t t = 0; t < T; ++t) {
            if (rec) out_len[rec - 1] += r[t].lead;
            for (uint64_t len : r[t].rec_len) { out_len[rec] = len; ++rec; }
        }
    }
    auto pad_rows = [&](size_t t) {
        for (uint64_t row = n_records * t / T; row < n_records * (t + 1) / T; ++row)
            if (out_len[row] < row_stride) memset(out_matrix + row * row_stride + out_len[row], pad, row_stride - out_len[row]);
    };
    th.clear();
    for (size_t t = 1; t < T; ++t) th.emplace_back(pad_rows, t);
    pad_rows(0);
    for (auto &x : th) x.join();
    return SNPGPU_OK;
}
