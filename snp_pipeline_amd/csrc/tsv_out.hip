// snp_distance_pairwise.tsv / snp_distance_matrix.tsv from the N x N matrix: host-side text formatting (no device code).
//
// Replaces the two print loops of snppipeline/distance.py:100-115.  At BASELINE configs[4] (10 000 samples) the pairwise
// file has 10^8 lines: Python's "%s\t%s\t%i\n" loop takes ~35 s for what the distance kernel computed in 38 ms; here the
// rows are formatted into a 4 MiB buffer and written with plain write(2) calls.
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <unistd.h>

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

}  // namespace

extern "C" int snpgpu_write_distance_tsv(const char *path, int layout, const char *ids, const uint64_t *id_off, uint32_t n,
                                         const int32_t *matrix, uint64_t row_stride) {
    if (!path || (layout != SNPGPU_TSV_PAIRWISE && layout != SNPGPU_TSV_MATRIX) || (n && (!ids || !id_off || !matrix)) || row_stride < n)
        return SNPGPU_E_ARG;
    FileOut o(path);
    if (o.failed) return SNPGPU_E_IO;
    auto name = [&](uint32_t i) { o.put(ids + id_off[i], (size_t)(id_off[i + 1] - id_off[i])); };
    if (layout == SNPGPU_TSV_PAIRWISE) {                        // distance.py:100-105
        o.put("Seq1\tSeq2\tDistance\n", 19);
        for (uint32_t i = 0; i < n && !o.failed; ++i) {
            const int32_t *row = matrix + (size_t)i * row_stride;
            for (uint32_t j = 0; j < n; ++j) {
                name(i); o.put('\t'); name(j); o.put('\t'); o.put_i32(row[j]); o.put('\n');
            }
        }
    } else {                                                    // distance.py:107-114
        for (uint32_t j = 0; j < n; ++j) { o.put('\t'); name(j); }
        if (n == 0) o.put('\t');                                // '\t%s\n' % '\t'.join([])
        o.put('\n');
        for (uint32_t i = 0; i < n && !o.failed; ++i) {
            const int32_t *row = matrix + (size_t)i * row_stride;
            name(i);
            for (uint32_t j = 0; j < n; ++j) { o.put('\t'); o.put_i32(row[j]); }
            o.put('\n');
        }
    }
    return o.finish();
}
