// snp_distance_pairwise.tsv / snp_distance_matrix.tsv from the N x N matrix: host-side text formatting (no device code).
//
// Replaces the two print loops of snppipeline/distance.py:100-115.  At BASELINE configs[4] (10 000 samples) the pairwise
// file has 10^8 lines: Python's "%s\t%s\t%i\n" loop takes ~35 s for what the distance kernel computed in 38 ms; here the
// rows are formatted into a 4 MiB buffer and written with plain write(2) calls.
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <unistd.h>

#include <thread>
#include <vector>

#include "internal.h"

namespace {

struct FileOut {
    int fd = -1;
    std::vector<char> buf;
    size_t n = 0;
    bool failed = false;
    explicit FileOut(const char *path) : buf((size_t)4 << 20) {
        fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
        failed = fd < 0;
    }
    ~FileOut() { if (fd >= 0) close(fd); }
    void flush() {
        size_t done = 0;
        while (!failed && done < n) {
            const ssize_t w = write(fd, buf.data() + done, n - done);
            if (w < 0) { if (errno == EINTR) continue; failed = true; break; }
            done += (size_t)w;
        }
        n = 0;
    }
    // room for `len` more bytes (len <= half the buffer)
    char *reserve(size_t len) { if (n + len > buf.size()) flush(); return buf.data() + n; }
    void put(const char *s, size_t len) {
        while (len) {                                           // names longer than the buffer are cut into pieces
            const size_t part = len < buf.size() / 2 ? len : buf.size() / 2;
            memcpy(reserve(part), s, part);
            n += part; s += part; len -= part;
        }
    }
    void put(char c) { *reserve(1) = c; ++n; }
    void put_i32(int32_t v) {                                   // "%i" / str(int)
        char *p = reserve(12);
        char tmp[12];
        int k = 0;
        uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
        do { tmp[k++] = (char)('0' + u % 10); u /= 10; } while (u);
        if (v < 0) tmp[k++] = '-';
        for (int i = 0; i < k; ++i) p[i] = tmp[k - 1 - i];
        n += (size_t)k;
    }
    int finish() {
        flush();
        if (fd >= 0 && close(fd) != 0) failed = true;
        fd = -1;
        return failed ? SNPGPU_E_IO : SNPGPU_OK;
    }
};
// bytes of str(v)
inline uint32_t digits_i32(int32_t v) {
    uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v, k = v < 0 ? 2u : 1u;
    while (u >= 10) { u /= 10; ++k; }
    return k;
}

// Rows [r0, r1) of one of the two layouts into `o` (a FileOut-like sink with put / put_i32).
template <typename Sink>
void format_rows(Sink &o, int layout, const char *ids, const uint64_t *id_off, uint32_t n, const int32_t *matrix, uint64_t row_stride, uint32_t r0, uint32_t r1) {
    auto name = [&](uint32_t i) { o.put(ids + id_off[i], (size_t)(id_off[i + 1] - id_off[i])); };
    for (uint32_t i = r0; i < r1 && !o.failed; ++i) {
        const int32_t *row = matrix + (size_t)i * row_stride;
        if (layout == SNPGPU_TSV_PAIRWISE) {
            for (uint32_t j = 0; j < n; ++j) { name(i); o.put('\t'); name(j); o.put('\t'); o.put_i32(row[j]); o.put('\n'); }
        } else {
            name(i);
            for (uint32_t j = 0; j < n; ++j) { o.put('\t'); o.put_i32(row[j]); }
            o.put('\n');
        }
    }
}

// A sink that writes its buffer at a running offset of an open file (several threads, each its own range of the file).
struct RangeOut {
    int fd;
    uint64_t at;
    std::vector<char> buf;
    size_t n = 0;
    bool failed = false;
    RangeOut(int fd_, uint64_t at_) : fd(fd_), at(at_), buf((size_t)4 << 20) {}
    void flush() {
        size_t done = 0;
        while (!failed && done < n) {
            const ssize_t w = pwrite(fd, buf.data() + done, n - done, (off_t)(at + done));
            if (w < 0) { if (errno == EINTR) continue; failed = true; break; }
            done += (size_t)w;
        }
        at += n;
        n = 0;
    }
    char *reserve(size_t len) { if (n + len > buf.size()) flush(); return buf.data() + n; }
    void put(const char *s, size_t len) {
        while (len) {
            const size_t part = len < buf.size() / 2 ? len : buf.size() / 2;
            memcpy(reserve(part), s, part);
            n += part; s += part; len -= part;
        }
    }
    void put(char c) { *reserve(1) = c; ++n; }
    void put_i32(int32_t v) {
        char *p = reserve(12);
        char tmp[12];
        int k = 0;
        uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
        do { tmp[k++] = (char)('0' + u % 10); u /= 10; } while (u);
        if (v < 0) tmp[k++] = '-';
        for (int i = 0; i < k; ++i) p[i] = tmp[k - 1 - i];
        n += (size_t)k;
    }
};

}  // namespace

// At 10 000 samples the pairwise file is 3 GB in 10^8 lines: formatted by one thread that alone takes longer than everything
// else the distance subcommand does.  So: the byte length of every row's text first (threads over row blocks), a prefix sum
// gives every block its place in the file, and the blocks are formatted and written with pwrite() side by side.
extern "C" int snpgpu_write_distance_tsv(const char *path, int layout, const char *ids, const uint64_t *id_off, uint32_t n,
                                         const int32_t *matrix, uint64_t row_stride) {
    if (!path || (layout != SNPGPU_TSV_PAIRWISE && layout != SNPGPU_TSV_MATRIX) || (n && (!ids || !id_off || !matrix)) || row_stride < n)
        return SNPGPU_E_ARG;
    unsigned T = snpgpu_writer_threads(0);
    if ((uint64_t)n * n < ((uint64_t)1 << 22)) T = 1;            // small matrices: one thread, one stream of write() calls
    if (T > n) T = n ? n : 1;
    if (T <= 1) {
        FileOut o(path);
        if (o.failed) return SNPGPU_E_IO;
        if (layout == SNPGPU_TSV_PAIRWISE) o.put("Seq1\tSeq2\tDistance\n", 19);           // distance.py:100-105
        else {                                                  // distance.py:107-114
            for (uint32_t j = 0; j < n; ++j) { o.put('\t'); o.put(ids + id_off[j], (size_t)(id_off[j + 1] - id_off[j])); }
            if (n == 0) o.put('\t');                            // '\t%s\n' % '\t'.join([])
            o.put('\n');
        }
        format_rows(o, layout, ids, id_off, n, matrix, row_stride, 0, n);
        return o.finish();
    }
    // ---- sizes ----
    const uint64_t names_total = id_off[n] - id_off[0];
    std::vector<uint64_t> block_bytes(T, 0);
    auto block = [&](unsigned t, uint32_t &r0, uint32_t &r1) { r0 = (uint32_t)((uint64_t)n * t / T); r1 = (uint32_t)((uint64_t)n * (t + 1) / T); };
    auto size_rows = [&](unsigned t) {
        uint32_t r0, r1;
        block(t, r0, r1);
        uint64_t total = 0;
        for (uint32_t i = r0; i < r1; ++i) {
            const int32_t *row = matrix + (size_t)i * row_stride;
            uint64_t d = 0;
            for (uint32_t j = 0; j < n; ++j) d += digits_i32(row[j]);
            const uint64_t li = id_off[i + 1] - id_off[i];
            total += layout == SNPGPU_TSV_PAIRWISE ? (uint64_t)n * (li + 3) + names_total + d      // n lines: id_i \t id_j \t d \n
                                                   : li + (uint64_t)n + d + 1;                      // id_i (\t d) x n \n
        }
        block_bytes[t] = total;
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; ++t) th.emplace_back(size_rows, t);
        size_rows(0);
        for (auto &x : th) x.join();
    }
    const uint64_t header = layout == SNPGPU_TSV_PAIRWISE ? 19 : names_total + n + 1;
    std::vector<uint64_t> at(T + 1, header);
    for (unsigned t = 0; t < T; ++t) at[t + 1] = at[t] + block_bytes[t];
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    if (fd < 0) return SNPGPU_E_IO;
    bool failed = ftruncate(fd, (off_t)at[T]) != 0;
    {
        RangeOut h(fd, 0);
        if (layout == SNPGPU_TSV_PAIRWISE) h.put("Seq1\tSeq2\tDistance\n", 19);
        else {
            for (uint32_t j = 0; j < n; ++j) { h.put('\t'); h.put(ids + id_off[j], (size_t)(id_off[j + 1] - id_off[j])); }
            h.put('\n');
        }
        h.flush();
        failed = failed || h.failed || h.at != header;
    }
    std::vector<uint8_t> bad(T, 0);
    auto write_rows = [&](unsigned t) {
        uint32_t r0, r1;
        block(t, r0, r1);
        RangeOut o(fd, at[t]);
        format_rows(o, layout, ids, id_off, n, matrix, row_stride, r0, r1);
        o.flush();
        bad[t] = o.failed || o.at != at[t + 1];
    };
    if (!failed) {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; ++t) th.emplace_back(write_rows, t);
        write_rows(0);
        for (auto &x : th) x.join();
        for (unsigned t = 0; t < T; ++t) failed = failed || bad[t];
    }
    if (close(fd) != 0) failed = true;
    return failed ? SNPGPU_E_IO : SNPGPU_OK;
}
