/* snpgpu.h — C ABI of the MI355X (gfx950) hot path of the CFSAN SNP Pipeline.
 *
 * The reference (CFSAN-Biostatistics/snp-pipeline v2.2.1) is pure Python and has
 * no FFI; its operator boundary is the `cfsan_snp_pipeline <subcommand>` CLI
 * (snppipeline/cfsan_snp_pipeline.py:309-457).  The Python host layer in
 * snp_pipeline_amd/ mirrors that CLI and calls the entry points below through
 * ctypes.  Each entry point names the reference code it replaces.
 *
 * Conventions
 *  - every function returns 0 on success, a negative SNPGPU_E_* code on failure;
 *    a message is available from snpgpu_last_error(ctx).  Nothing throws.
 *  - `*_dev` functions take DEVICE pointers, enqueue work on the context's
 *    stream and return without synchronising; the others take HOST pointers and
 *    are synchronous.  The caller owns every buffer it passes.
 *  - a context is bound to one (process, device) and is not thread-safe.
 *  - there is no CPU fallback: without a gfx950 device snpgpu_ctx_create fails.
 */
#ifndef SNPGPU_H
#define SNPGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNPGPU_ABI_VERSION 7

/* ---- error codes ------------------------------------------------------- */
#define SNPGPU_OK            0
#define SNPGPU_E_HIP        -1   /* HIP runtime error (message has the hipError string) */
#define SNPGPU_E_ARG        -2   /* invalid argument */
#define SNPGPU_E_NOMEM      -3
#define SNPGPU_E_PILEUP     -4   /* malformed pileup text; see snpgpu_scan_status */
#define SNPGPU_E_UNSUPPORTED -5  /* input the reference accepts but this build refuses (reported loudly) */
#define SNPGPU_E_IO         -6   /* a file could not be opened, read or written */
#define SNPGPU_E_TIMEOUT    -7   /* snpgpu_stream_wait: the stream was still busy when the time was up */

/* ---- failed-filter bits, in the order pileup.py:564-584 appends them and
 *      call_consensus.py:165-168 appends "Region" ------------------------- */
#define SNPGPU_F_RAWDPTH  0x01
#define SNPGPU_F_VARFREQ  0x02
#define SNPGPU_F_DEPTH    0x04
#define SNPGPU_F_STRDPTH  0x08
#define SNPGPU_F_STRBIAS  0x10
#define SNPGPU_F_REGION   0x20

/* ---- site flags --------------------------------------------------------- */
#define SNPGPU_SITE_IN_SNPLIST 0x01   /* position is in snplist.txt (call_consensus.py:146) */
#define SNPGPU_SITE_EXCLUDED   0x02   /* position is in the -e exclude VCF (call_consensus.py:121) */

/* ---- per-site status ---------------------------------------------------- */
#define SNPGPU_ST_NO_LINE      0   /* no pileup line for this position: '-' (call_consensus.py:187) */
#define SNPGPU_ST_OK           1
#define SNPGPU_ST_SHORT_LINE   2   /* < 4 fields: IndexError in pileup.py:224-225 */
#define SNPGPU_ST_BAD_DEPTH    3   /* depth field is not [0-9]+: ValueError in pileup.py:225 */
#define SNPGPU_ST_NO_QUALS     4   /* depth > 0 and exactly 5 fields: IndexError in pileup.py:237 */
#define SNPGPU_ST_MULTI_REF    5   /* (ABI <= 4: reference-base field longer than SNPGPU_SPILL_REF bytes; no longer produced) */

typedef struct snpgpu_ctx snpgpu_ctx;
typedef struct snpgpu_siteset snpgpu_siteset;
typedef struct snpgpu_pileups snpgpu_pileups;

/* ConsensusCaller parameters, pileup.py:433-465 (+ Reader's min_base_quality, pileup.py:384). */
typedef struct snpgpu_caller_params {
    int32_t min_base_quality;       /* -q, default 0   */
    int32_t min_cons_depth;         /* -D, default 1   */
    int32_t min_cons_strand_depth;  /* -d, default 0   */
    int32_t reserved;
    double  min_cons_freq;          /* -c, default 0.6 */
    double  min_cons_strand_bias;   /* -b, default 0.0 */
} snpgpu_caller_params;

/* Everything pileup.Record exposes for one position (pileup.py:58-94), with the
 * histograms as a ranked list (count descending, byte ascending, pileup.py:263-266).
 * 128 bytes per site.  Enough to write consensus.vcf rows (vcf_writer.py:295-379). */
#define SNPGPU_MAX_SYMS 8
typedef struct snpgpu_site_counts {
    uint32_t raw_depth;             /* Record.raw_depth */
    uint32_t good_depth;            /* Record.good_depth */
    uint32_t fwd_good_depth;        /* Record.forward_good_depth */
    uint32_t rev_good_depth;        /* Record.reverse_good_depth */
    uint32_t n_symbols;             /* bits 0-7: distinct upper-cased symbols with good depth; when that is more than 8, or the
                                     * reference-base field has more than one byte: bits 8-31 = 1 + index of the position's
                                     * snpgpu_symbol_spill record (0xFFFFFF: there was no room) */
    uint8_t  ref_base;              /* Record.reference_base (its first byte), case preserved */
    uint8_t  cons_base;             /* ConsensusCaller.call_consensus()[0] */
    uint8_t  filters;               /* SNPGPU_F_* mask, incl. REGION */
    uint8_t  status;                /* SNPGPU_ST_* */
    uint8_t  sym[SNPGPU_MAX_SYMS];  /* most_common_good_bases[0..8) */
    uint32_t total[SNPGPU_MAX_SYMS];/* base_good_depth[sym] */
    uint32_t fwd[SNPGPU_MAX_SYMS];  /* forward_base_good_depth[sym] */
    uint32_t rev[SNPGPU_MAX_SYMS];  /* reverse_base_good_depth[sym] */
} snpgpu_site_counts;

/* Ranks 8, 9, ... of a position with more than SNPGPU_MAX_SYMS distinct symbols (pileup.py:259-266 ranks any number of them and
 * vcf_writer.py:317-331 lists every one as an ALT allele): the record itself keeps the first eight, the rest goes to one of
 * the records the context holds (SNPGPU_SPILL_CAP to begin with; more after a call that ran out: snpgpu_symbol_spill_read).  The same record carries a reference-base field of more than one byte
 * (pileup.py:223 takes any string; every '.' / ',' then stands for all of its characters, pileup.py:255-258, and the VCF
 * REF column shows the string): ref_len > 1 and ref[] hold it, n may be 0.  And a depth column outside 0 .. 2^32 - 1 (depth64).  A call that produces per-site records starts with an empty spill;
 * snpgpu_symbol_spill_read copies out what the context's calls have put there since (synchronises the context's stream). */
#define SNPGPU_SPILL_SYMS 120
#define SNPGPU_SPILL_REF  64
#define SNPGPU_SPILL_CAP  1024
typedef struct snpgpu_symbol_spill {
    uint32_t n;                         /* entries used */
    uint32_t ref_len;                   /* 0 or 1: the record's ref_base is the whole field; else the field's length: its first
                                         * SNPGPU_SPILL_REF bytes in ref[], the rest — raw, no header — in the
                                         * ceil((ref_len - SNPGPU_SPILL_REF) / sizeof(record)) records that follow this one (ref[] is
                                         * the last member, so the field is one run of bytes starting at ref) */
    int64_t  depth64;                   /* Record.raw_depth when the record's 32 unsigned bits cannot hold it — int() takes "-3"
                                         * and "5000000000" (pileup.py:225) and consensus.vcf prints what it got; else 0 */
    uint8_t  sym[SNPGPU_SPILL_SYMS];    /* most_common_good_bases[8 + k] */
    uint32_t total[SNPGPU_SPILL_SYMS];
    uint32_t fwd[SNPGPU_SPILL_SYMS];
    uint32_t rev[SNPGPU_SPILL_SYMS];
    uint8_t  ref[SNPGPU_SPILL_REF];     /* Record.reference_base when it is longer than one byte, case preserved */
} snpgpu_symbol_spill;

/* Result words of one pileup scan (device-written, 4 x u64). */
#define SNPGPU_SCAN_STATUS_WORDS 4
/*  [0] first malformed line: ((byte offset + 1) << 8) | code, or UINT64_MAX when clean
 *        code 1: line has < 2 fields      (ValueError at pileup.py:425)
 *        code 2: position is not [0-9]+   (ValueError at pileup.py:426)
 *        code 3: byte >= 0x80 in the file (non-ASCII pileup: unsupported)
 *  [1] number of lines seen, [2] number of lines that matched a site, [3] sum of the depth column over
 *      well-formed lines (collect_metrics.py:325-340 by-product; 0 unless requested) */

/* ---- context ------------------------------------------------------------ */
int  snpgpu_abi_version(void);
int  snpgpu_device_count(void);   /* visible gfx950 devices (0 when there is no usable HIP runtime); creates no context */
int  snpgpu_ctx_create(int device, snpgpu_ctx **out);
void snpgpu_ctx_destroy(snpgpu_ctx *ctx);
const char *snpgpu_last_error(const snpgpu_ctx *ctx);
/* Run on the caller's hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL is HIP's default stream.
 * snpgpu_ctx_reset_stream goes back to the context's own non-blocking stream. */
int  snpgpu_ctx_set_stream(snpgpu_ctx *ctx, void *hip_stream);
int  snpgpu_ctx_reset_stream(snpgpu_ctx *ctx);
int  snpgpu_ctx_sync(snpgpu_ctx *ctx);

/* ---- the CPU budget of the host side (ABI 7) ------------------------------------------------------------------------------
 * Replaces run.py:387-400 (MaxCpuCores: min(psutil.cpu_count(), MaxCpuCores) caps the reference's per-sample process fan-out).
 * Here one process per GPU does its shard's host work on threads (file readers, text writers), so the cap is divided among the
 * processes that share the node:
 *   usable_cpus = min(CPUs in the affinity mask, the cgroup's CPU quota (v2 cpu.max / v1 cfs quota), max_cpu_cores)
 *   budget      = max(1, usable_cpus / local_ranks)        -> readers, writers (what the entry points start by default)
 * max_cpu_cores: snpgpu_set_max_cpu_cores, else the environment variable SNPGPU_MAX_CPU_CORES, else no cap.
 * local_ranks:   snpgpu_set_local_ranks, else SNPGPU_LOCAL_RANKS, else LOCAL_WORLD_SIZE (torch.distributed.run), else 1.
 * An explicit thread count in a call's options (snpgpu_stream_opts.n_readers, n_threads arguments) still wins. */
typedef struct snpgpu_cpu_budget_info {
    uint32_t affinity_cpus;              /* sched_getaffinity */
    uint32_t quota_cpus;                 /* cgroup quota / period, rounded down, at least 1; 0 = no quota */
    uint32_t max_cpu_cores;              /* the cap in force; 0 = none */
    uint32_t usable_cpus;
    uint32_t local_ranks;
    uint32_t budget;                     /* CPUs this process plans with */
    uint32_t readers;                    /* default reader threads of the file entry points */
    uint32_t writers;                    /* default formatting threads of the text writers / the FASTA survey */
} snpgpu_cpu_budget_info;
int  snpgpu_cpu_budget(snpgpu_cpu_budget_info *out);
void snpgpu_set_max_cpu_cores(uint32_t cores);   /* 0 = back to the environment / no cap */
void snpgpu_set_local_ranks(uint32_t ranks);     /* 0 = back to the environment / 1 */
/* Kernel-only elapsed time helpers (HIP events recorded on the context's stream). */
int  snpgpu_timer_start(snpgpu_ctx *ctx);
int  snpgpu_timer_stop_ms(snpgpu_ctx *ctx, float *out_ms);   /* synchronises on the stop event */

/* Per-kernel timing for bench.py: when enabled, HIP events are recorded on the stream around every launch of
 * the scan (0), per-site caller (1) and distance (2) kernels and around everything phase-1 site calling launches for a
 * file (3); snpgpu_ctx_kernel_time_ms synchronises, returns the
 * summed elapsed time and the number of launches of that kernel since the last query, and clears them. */
int  snpgpu_ctx_kernel_timing(snpgpu_ctx *ctx, int enable);
int  snpgpu_ctx_kernel_time_ms(snpgpu_ctx *ctx, int kernel, float *total_ms, uint32_t *launches);

/* ---- site set: the (chrom,pos) set handed to pileup.Reader (pileup.py:396-403) ----
 * contig_names: concatenated names, contig_name_off[n_contigs+1]; names must be sorted bytewise and unique.
 * site_keys: (contig_index << 32) | position, strictly increasing; site_flags: SNPGPU_SITE_* per key.
 * All HOST pointers; the set is copied to the device (bitmap + rank directory). */
int  snpgpu_siteset_create(snpgpu_ctx *ctx, const uint8_t *contig_names, const uint32_t *contig_name_off,
                           uint32_t n_contigs, const uint64_t *site_keys, const uint8_t *site_flags,
                           uint32_t n_sites, snpgpu_siteset **out);
void snpgpu_siteset_destroy(snpgpu_siteset *ss);
uint32_t snpgpu_siteset_size(const snpgpu_siteset *ss);

/* ---- call_consensus: pileup.Reader + Record + ConsensusCaller + the mapping in
 *      call_consensus.py:161-188, for ONE sample ------------------------------------------------
 * d_pileup: raw ASCII of reads.all.pileup in device memory (any alignment).
 * d_out_base[n_sites]:    the byte call_consensus.py writes to the FASTA for that key ('-' when a filter failed,
 *                         the base is '*', or the position has no line).
 * d_out_filters[n_sites]: SNPGPU_F_* mask (0 when the position has no line).
 * d_out_counts[n_sites]:  nullable.
 * d_status[4]:            SNPGPU_SCAN_STATUS_WORDS u64.
 * want_depth_sum: also accumulate status[3] (costs one integer parse per line). */
int  snpgpu_call_consensus_dev(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const void *d_pileup, size_t nbytes,
                               const snpgpu_caller_params *params, uint8_t *d_out_base, uint8_t *d_out_filters,
                               snpgpu_site_counts *d_out_counts, uint64_t *d_status, int want_depth_sum);
/* Many samples resident in one device buffer: ONE scan launch and ONE call launch serve up to 256 samples (the
 * waves of the scan are dealt to the samples in proportion to their sizes), which is what the throughput figures
 * are quoted on.  Sample i is bytes [h_offsets[i], h_offsets[i] + h_sizes[i]) of d_pileups; h_sizes == NULL means
 * the samples are packed back to back, h_sizes[i] = h_offsets[i+1] - h_offsets[i] (h_offsets then has n_samples + 1
 * entries).  Outputs are [n_samples][n_sites] row-major; d_status is [n_samples][4]. */
int  snpgpu_call_consensus_batch_dev(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const void *d_pileups,
                                     const uint64_t *h_offsets, const uint64_t *h_sizes, uint32_t n_samples,
                                     const snpgpu_caller_params *params, uint8_t *d_out_base,
                                     uint8_t *d_out_filters, uint64_t *d_status);
/* Samples anywhere in device memory (d_pileups: HOST array of n device pointers, h_sizes their lengths), with everything the
 * streamed form returns: d_out_counts (nullable) and d_out_line_off (nullable) are [n][n_sites] like the other outputs,
 * d_site_flags (nullable) [n][n_sites] replaces the set's flags per sample (a sample's own exclude list).  Asynchronous. */
int  snpgpu_call_consensus_many_dev(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const void *const *d_pileups,
                                    const uint64_t *h_sizes, uint32_t n_samples, const snpgpu_caller_params *params,
                                    const uint8_t *d_site_flags, uint8_t *d_out_base, uint8_t *d_out_filters,
                                    snpgpu_site_counts *d_out_counts, uint64_t *d_out_line_off, uint64_t *d_status,
                                    int want_depth_sum);
/* Host-buffer form for one pileup (an mmap, bytes read elsewhere): streamed through the same pipeline as
 * snpgpu_call_consensus_files below, synchronous.
 * Returns SNPGPU_E_PILEUP when the scan found a malformed line (status words still filled). */
int  snpgpu_call_consensus(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const uint8_t *pileup, size_t nbytes,
                           const snpgpu_caller_params *params, uint8_t *out_base, uint8_t *out_filters,
                           snpgpu_site_counts *out_counts, uint64_t *out_status, int want_depth_sum);
/* ---- streamed ingestion: pileup FILES -> consensus bytes --------------------------------------------------
 * Replaces the reference's input side of call_consensus: the text-mode line iterator over the open pileup
 * (pileup.py:408-429, driven from call_consensus.py:161) and the one-process-per-sample job array around it
 * (run.py:704-718).  Reader threads pread() the files in chunks of whole 4 KiB scan tiles into pinned staging
 * buffers, a copy stream moves each chunk into the file's device buffer, and the scan runs over the tiles that
 * have landed while the rest of the file is still on its way; the exact parser's queue and the call step run
 * when the file is complete.  Files alternate between device buffers, so the next file streams in during the
 * tail of the previous one.  No file is ever resident in host memory as a whole.
 *
 * Outputs are [n_files][n_sites] row-major HOST arrays: out_base / out_filters as snpgpu_call_consensus_dev,
 * out_counts (nullable) the per-site records, out_line_off (nullable) 1 + byte offset of the line used for
 * each site (0 = none), out_status [n_files][4] the scan status words, out_rc (nullable) per file: 0,
 * SNPGPU_E_IO (could not open / read: its outputs are void), SNPGPU_E_PILEUP or SNPGPU_E_UNSUPPORTED (malformed
 * pileup: see its status words).  The return value is 0 unless the pipeline itself failed.
 *
 * excl_off / excl_slots (both nullable): per-file exclude lists (call_consensus.py:117-123: every sample of step 7.2 has its
 * own var.flt_removed.vcf) as CSR over site-set slots — file f is called with the set's flags plus SNPGPU_SITE_EXCLUDED on
 * slots excl_slots[excl_off[f] .. excl_off[f+1]), so one site set (the union of the snplist and every file's list) and one
 * stream of files serve all samples. */
typedef struct snpgpu_stream_opts {      /* 0 = default everywhere */
    uint32_t chunk_bytes;                /* bytes per host->device copy (default 8 MiB; rounded up to 4 KiB) */
    uint32_t n_staging;                  /* pinned staging buffers (default: readers + 4) */
    uint32_t n_readers;                  /* reader threads (default: snpgpu_cpu_budget().readers — 8 on a big host, fewer on a small one or beside other ranks) */
    uint32_t n_slots;                    /* device file buffers = files in flight (default 2) */
    uint32_t want_depth_sum;             /* also accumulate status[3] (collect_metrics.py:325-340 by-product) */
    uint32_t reserved[3];
} snpgpu_stream_opts;
typedef struct snpgpu_stream_stats {
    uint64_t bytes;                      /* pileup bytes that went through */
    uint64_t n_chunks;
    double   seconds;                    /* wall time of the call */
    double   seconds_waiting_for_readers;/* ... of which the issuing thread waited for a chunk to be read */
    double   seconds_waiting_for_device; /* ... and for the device to finish a file */
    uint32_t n_readers, n_staging, chunk_bytes, reserved;
    double   reader_seconds_reading;     /* summed over the reader threads: inside pread / memcpy */
    double   reader_seconds_waiting;     /* ... waiting for a staging buffer to be copied out */
    double   seconds_enqueueing;         /* issuing thread: inside HIP enqueue calls (copies, events, kernels) */
} snpgpu_stream_stats;
int  snpgpu_call_consensus_files(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const char *const *paths, uint32_t n_files,
                                 const snpgpu_caller_params *params, const uint32_t *excl_off, const uint32_t *excl_slots,
                                 uint8_t *out_base, uint8_t *out_filters,
                                 snpgpu_site_counts *out_counts, uint64_t *out_line_off, uint64_t *out_status,
                                 int32_t *out_rc, const snpgpu_stream_opts *opts, snpgpu_stream_stats *stats);

/* call_consensus --vcfAllPos (call_consensus.py:148-151, pileup.py:418-421): a Record for EVERY line of the pileup,
 * whether its position is listed or not.  Synchronous; host outputs in file order: out_line_off[i] = 1 + byte offset of
 * line i, out_line_flags[i] = SNPGPU_SITE_* of its position (0 when it is not in the site set), out_counts[i] its
 * record (filters include REGION for excluded positions).  *out_n_lines is always set; when it exceeds `capacity`
 * nothing else is written and the caller comes back with larger arrays (capacity 0 just counts the lines).
 * out_status[4]: [0] first line whose chrom / position columns are malformed (else UINT64_MAX), [1] lines. */
int  snpgpu_call_all_lines_file(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const char *path,
                                const snpgpu_caller_params *params, uint64_t capacity, uint64_t *out_n_lines,
                                uint64_t *out_line_off, uint8_t *out_line_flags, snpgpu_site_counts *out_counts,
                                uint64_t *out_status);

/* The same pass with 32-byte records (ABI 7).  Nearly every line of a pileup has at most three distinct symbols and depths far
 * below 65 536: such a line's record is packed into a snpgpu_line_record — everything consensus.vcf prints for it
 * (vcf_writer.py:295-379) — and only the others ("wide": more symbols, a count above 65 535, a spill record, a Record-level error)
 * come back as the full 128-byte record, with their line index.  5 M lines: 200 MB over the host link instead of 685 MB.
 *   good_depth = total[0] + total[1] + total[2], forward_good_depth = sum of fwd[], reverse_good_depth = sum of rev[]
 * (each kept read base counts for exactly one symbol; a line is only packed when these sums hold).
 * out_records[i] is line i of the file (n_symbols == SNPGPU_LINE_WIDE: look the line up in out_wide_index, ascending, and take
 * out_wide[k]).  *out_n_wide is always set; when it exceeds wide_capacity (or *out_n_lines exceeds capacity) nothing else is
 * written and the caller comes back with larger arrays.  Other arguments and the status words as snpgpu_call_all_lines_file. */
#define SNPGPU_LINE_WIDE 0xFF
#define SNPGPU_LINE_SYMS 3
typedef struct snpgpu_line_record {
    uint32_t raw_depth;                     /* Record.raw_depth */
    uint16_t total[SNPGPU_LINE_SYMS];       /* base_good_depth of the ranked symbols */
    uint16_t fwd[SNPGPU_LINE_SYMS];
    uint16_t rev[SNPGPU_LINE_SYMS];
    uint8_t  sym[SNPGPU_LINE_SYMS];         /* most_common_good_bases[0..3) */
    uint8_t  ref_base, cons_base, filters, status;      /* as in snpgpu_site_counts */
    uint8_t  n_symbols;                     /* 0 .. 3, or SNPGPU_LINE_WIDE */
    uint8_t  site_flags;                    /* SNPGPU_SITE_* of the line's position (0: not in the site set) */
    uint8_t  reserved;
} snpgpu_line_record;
int  snpgpu_call_all_lines_compact_file(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const char *path,
                                        const snpgpu_caller_params *params, uint64_t capacity, uint64_t *out_n_lines,
                                        uint64_t *out_line_off, snpgpu_line_record *out_records,
                                        uint32_t wide_capacity, uint32_t *out_n_wide, uint32_t *out_wide_index,
                                        snpgpu_site_counts *out_wide, uint64_t *out_status);

/* Rows of consensus.vcf from such records, host code (no device): the lines [0, n_lines) of the pileup text `pileup[0, nbytes)`, CHROM and
 * POS of a row from the line's own first two fields (POS as int() prints it), the numbers from records[i] — or, where that says
 * SNPGPU_LINE_WIDE, from the entry of wide[] whose wide_index equals i (ascending) — laid out as vcf_writer.py:295-379 does.
 * only_listed: rows for lines with site_flags != 0 only.  Returns the bytes the rows take and writes at most `capacity` of them to out
 * (capacity 0: just the size); *out_n_rows rows; *out_bad_line: -1, or the line whose record the writer refuses (more symbols than it
 * keeps and no spill record; an offset outside the text) — nothing is returned then. */
size_t snpgpu_format_line_rows(const uint8_t *pileup, uint64_t nbytes, const uint64_t *line_off, const snpgpu_line_record *records,
                               uint64_t n_lines, const uint32_t *wide_index, const snpgpu_site_counts *wide, uint32_t n_wide,
                               const char *const *filter_names, int preserve_ref_case, char failed_snp_gt,
                               const snpgpu_symbol_spill *spill, uint32_t n_spill, int only_listed, char *out, size_t capacity,
                               uint64_t *out_n_rows, int64_t *out_bad_line);

/* call_consensus --vcfAllPos from file to file: every row of consensus.vcf (call_consensus.py:148-151, vcf_writer.py:381-435: one
 * row per pileup LINE, in file order; CHROM and POS are the line's own first two fields, POS as int() prints it) formatted by the
 * library's host threads from the 32-byte records, straight into vcf_path behind `header` (the text of the header lines).
 * only_listed != 0: rows for the lines whose position is in the site set only (a pileup that repeats positions: the reference
 * writes a row for every matching line, call_consensus.py:178-180).  The file is written only when no line stops the reference:
 *   returns SNPGPU_E_PILEUP / SNPGPU_E_UNSUPPORTED with out_status as snpgpu_call_all_lines_file for a malformed chrom / position
 *   column; returns SNPGPU_OK with *out_first_bad_line != UINT64_MAX (and that line's record and offset + 1) when `check` is set
 *   and a Record cannot be built from some line (status > SNPGPU_ST_OK) — the caller raises what the reference raises.
 * *out_n_rows: rows written.  filter_names, preserve_ref_case, failed_snp_gt as snpgpu_format_vcf_rows. */
int  snpgpu_write_all_positions_vcf(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const char *pileup_path,
                                    const snpgpu_caller_params *params, const char *vcf_path, const char *header,
                                    const char *const *filter_names, int preserve_ref_case, char failed_snp_gt,
                                    int only_listed, int check, uint64_t *out_n_lines, uint64_t *out_n_rows,
                                    uint64_t *out_first_bad_line, uint64_t *out_first_bad_off, snpgpu_site_counts *out_first_bad,
                                    uint64_t *out_status);

/* ---- phase-1 site calling: the counting + selection half of `VarScan mpileup2snp` -------------------------------
 * Replaces what call_sites.py:89-108 gets from the VarScan v2.3.9 jar (a third-party dependency that is not in the
 * reference tree): VarScan.qualityDepth, VarScan.getReadCounts and the min-coverage / min-reads2 / min-avg-qual /
 * min-var-freq tests of VarScan.callPosition, for every line of a one-sample pileup.  Each (line, allele) that passes
 * leaves one record; Fisher's exact test, --p-value, the strand filter and the VCF text are host work on those
 * records (snp_pipeline_amd/varscan.py). */
typedef struct snpgpu_varscan_params {
    uint32_t min_coverage;          /* --min-coverage, default 8: raw depth AND qualities >= min_avg_qual */
    uint32_t min_reads2;            /* --min-reads2, default 2 (the pipeline passes 5) */
    uint32_t min_avg_qual;          /* --min-avg-qual, default 15 */
    uint32_t reserved;
    double   min_var_freq;          /* --min-var-freq, default 0.20 (the pipeline passes 0.90) */
} snpgpu_varscan_params;
typedef struct snpgpu_varscan_site {
    uint64_t line_off;              /* byte offset of the pileup line in the file (chrom / position text are read there) */
    uint32_t sdp;                   /* depth column */
    uint32_t dp;                    /* qualities >= min_avg_qual */
    uint32_t total;                 /* reads over all alleles at that quality + indel-carrying reads: FREQ's denominator */
    uint32_t rdf, rdr, ref_qual_sum;/* '.' / ',' at that quality, and the sum of their qualities */
    uint32_t adf, adr, alt_qual_sum;/* the same for alt_base */
    uint8_t  ref_base, alt_base;    /* upper case */
    uint8_t  reserved[2];
} snpgpu_varscan_site;
/* path -> records in file order (ties: allele A < C < G < T).  Streams the file to the device, indexes its lines, runs
 * the kernel, copies the records back; synchronous.  *out_n_sites is the number of records found; when it exceeds
 * `capacity` only `capacity` arbitrary ones were written and the caller comes back with a larger array.
 * out_status[2]: [0] byte offset of the first malformed line (fewer than six non-empty TAB-separated columns, a depth
 * that is not a plain integer, a reference column longer than one byte) else UINT64_MAX — the call then returns
 * SNPGPU_E_PILEUP; [1] lines in the file.  SNPGPU_E_IO when the file cannot be read. */
int  snpgpu_varscan_file(snpgpu_ctx *ctx, const char *path, const snpgpu_varscan_params *params, uint32_t capacity,
                         snpgpu_varscan_site *out_sites, uint32_t *out_n_sites, uint64_t *out_status);

/* Many files in one call: the reader threads run ahead across file boundaries, the files alternate between two device
 * slots, and a file's kernels run while the next file is read and copied (the shape of snpgpu_call_consensus_files).
 * out_sites is [n_files][capacity], out_n_sites [n_files], out_status [n_files][2], out_rc [n_files] (SNPGPU_OK, SNPGPU_E_IO or
 * SNPGPU_E_PILEUP per file; the call itself fails only for argument, memory or HIP errors).  A file with more records
 * than `capacity` reports its count in out_n_sites and leaves out_sites undefined for it: repeat it with
 * snpgpu_varscan_file. */
int  snpgpu_varscan_files(snpgpu_ctx *ctx, const char *const *paths, uint32_t n_files, const snpgpu_varscan_params *params,
                          uint32_t capacity, snpgpu_varscan_site *out_sites, uint32_t *out_n_sites, uint64_t *out_status,
                          int32_t *out_rc);

/* The same over a pileup that already is in device memory. */
int  snpgpu_varscan_dev(snpgpu_ctx *ctx, const void *d_pileup, uint64_t nbytes, const snpgpu_varscan_params *params,
                        uint32_t capacity, snpgpu_varscan_site *out_sites, uint32_t *out_n_sites, uint64_t *out_status);

/* The same over MANY pileups that are in device memory (the resident files of a snpgpu_pileups store, synthetic ones): one
 * scan launch over all of them — every wave works inside one file, the waves are dealt to the files by size — so the ramp and
 * the tail of a launch are paid once per call, not once per 0.1 ms file.  Arguments as snpgpu_varscan_files: out_sites is
 * [n_files][capacity], out_n_sites [n_files], out_status [n_files][2], out_rc [n_files] (SNPGPU_OK or SNPGPU_E_PILEUP per
 * file); a file with more records than `capacity` reports its count and leaves its out_sites undefined (repeat it with
 * snpgpu_varscan_dev).  At most 65 535 files per call.  What the reference does per sample with one JVM each
 * (call_sites.py:89-108, run.py:662 runs them as a job array). */
int  snpgpu_varscan_batch_dev(snpgpu_ctx *ctx, const void *const *d_pileups, const uint64_t *nbytes, uint32_t n_files,
                              const snpgpu_varscan_params *params, uint32_t capacity, snpgpu_varscan_site *out_sites,
                              uint32_t *out_n_sites, uint64_t *out_status, int32_t *out_rc);

/* ---- resident pileups: the input side of the one-job pipeline (`cfsan_snp_pipeline hot_path_batch`) ----------------------
 * The reference runs steps 4-11 as separate process arrays over a shared file system (run.py:662-784): call_sites
 * (call_sites.py:89-108) and call_consensus twice (run.py:704-710, :712-718) each read a sample's reads.all.pileup again.
 * A snpgpu_pileups store keeps the files in device memory instead: snpgpu_pileups_ingest streams them in exactly as
 * snpgpu_varscan_files does (same reader threads, staging ring, copy streams; site calling on a file while the next one
 * arrives) but leaves every file where it landed, while `budget_bytes` of device memory last (0: what is free at creation
 * less 24 GiB); files past the budget are site-called through the two streaming slots and stay non-resident.  The consensus
 * step then reads the resident files with snpgpu_call_consensus_many_dev: each pileup crosses the host link once.
 * params == NULL: no site calling (the files are only made resident).  out_done (nullable) [n_files]: set to 1 (release
 * store) when the outputs of file f are final, so that another host thread can turn them into var.flt.vcf while the call is
 * still running.  Other arguments as snpgpu_varscan_files. */
typedef struct snpgpu_pileups_stats {
    uint64_t h2d_bytes;             /* bytes copied host -> device on behalf of the store */
    uint64_t file_bytes;            /* sizes of the ingested files */
    uint64_t resident_bytes;        /* device bytes taken (files + padding) */
    uint64_t budget_bytes;
    uint32_t n_files, n_resident;
    double   seconds;                       /* wall time of the ingest calls, of which the issuing thread spent ... */
    double   seconds_allocating;            /* ... allocating device memory, */
    double   seconds_waiting_for_readers;   /* waiting for a piece of a file to be read, */
    double   seconds_waiting_for_device;    /* and waiting for the device (line counts, results) */
    double   reader_seconds_reading;        /* summed over the reader threads: inside pread */
    double   reader_seconds_waiting;        /* ... waiting for a staging buffer to be copied out */
    double   seconds_preparing;             /* part of `seconds` before the first read starts: files opened and placed, staging memory */
    uint32_t n_readers;                     /* reader threads of the last ingest call (ABI 7: from snpgpu_cpu_budget) */
    uint32_t reserved;
} snpgpu_pileups_stats;
int  snpgpu_pileups_create(snpgpu_ctx *ctx, uint64_t budget_bytes, snpgpu_pileups **out);
void snpgpu_pileups_destroy(snpgpu_pileups *store);
int  snpgpu_pileups_ingest(snpgpu_ctx *ctx, snpgpu_pileups *store, const char *const *paths, uint32_t n_files,
                           const snpgpu_varscan_params *params, uint32_t capacity, snpgpu_varscan_site *out_sites,
                           uint32_t *out_n_sites, uint64_t *out_status, int32_t *out_rc, int32_t *out_done);
uint32_t snpgpu_pileups_count(const snpgpu_pileups *store);
/* file `index` in ingestion order: its device pointer (NULL when it is not resident) and size */
int  snpgpu_pileups_get(const snpgpu_pileups *store, uint32_t index, void **d_ptr, uint64_t *nbytes);
int  snpgpu_pileups_get_stats(const snpgpu_pileups *store, snpgpu_pileups_stats *out);

/* The host half of the same step (no device work, no context): the records of snpgpu_varscan_file, in its order, ->
 * var.flt.vcf data lines.  Per line: the allele with the most variant reads among those whose Fisher p (reads against a
 * 0.1 % error model, VarScan.getSignificance) is <= p_value; strand filter (FILTER str10); GT 1/1 at or above
 * min_freq_for_hom, else 0/1; chrom and position text are taken from `pileup` (the file's bytes, e.g. an mmap) at the
 * record's line offset.  Writes at most `capacity` bytes to `out` (may be NULL), returns the bytes the lines take and
 * leaves their number in *out_rows. */
typedef struct snpgpu_varscan_finish {
    double  p_value;                /* --p-value, default 0.99 */
    double  min_freq_for_hom;       /* --min-freq-for-hom, default 0.75 */
    int32_t strand_filter;          /* --strand-filter, default 1 */
    int32_t reserved;
} snpgpu_varscan_finish;
size_t snpgpu_varscan_format_rows(const snpgpu_varscan_site *sites, uint32_t n_sites, const uint8_t *pileup,
                                  uint64_t pileup_bytes, const snpgpu_varscan_finish *fin, char *out, size_t capacity,
                                  uint32_t *out_rows);

/* consensus.vcf data lines from per-site records — host-side text formatting, no device work, no context
 * (vcf_writer.py:295-435: _make_vcf_record_from_pileup + the text PyVCF3's Writer emits for it).  Row r is the record
 * counts[order[r]] (order == NULL: r) of site site_keys[order[r]] = (contig index << 32) | position, contig names as
 * in snpgpu_siteset_create; filter_names: the six names in SNPGPU_F_* bit order; failed_snp_gt: '.', '0' or '1'.
 * spill / n_spill (nullable / 0): the context's spill records for positions with more than SNPGPU_MAX_SYMS symbols.
 * Writes at most `capacity` bytes to `out` (may be NULL) and returns the number of bytes the rows take; a record with
 * more than SNPGPU_MAX_SYMS symbols whose spill record is not in `spill` is skipped and its row index left in *out_bad_row
 * (else -1). */
size_t snpgpu_format_vcf_rows(const snpgpu_site_counts *counts, const uint32_t *order, uint32_t n_rows,
                              const uint8_t *contig_names, const uint32_t *contig_name_off, const uint64_t *site_keys,
                              const char *const *filter_names, int preserve_ref_case, char failed_snp_gt,
                              const snpgpu_symbol_spill *spill, uint32_t n_spill,
                              char *out, size_t capacity, int32_t *out_bad_row);
/* *out_n = the records the context's calls have asked for since the last call that started an empty spill; up to `capacity` of
 * them are copied to `out`.  *out_n > snpgpu_symbol_spill_capacity(ctx): the arena ran out (the records past it say "no room",
 * nothing is copied) — repeat the call: the context allocates an arena of the size this one asked for at its next call. */
int  snpgpu_symbol_spill_read(snpgpu_ctx *ctx, snpgpu_symbol_spill *out, uint32_t capacity, uint32_t *out_n);
uint32_t snpgpu_symbol_spill_capacity(const snpgpu_ctx *ctx);

/* The output files of call_consensus (call_consensus.py:178-192: consensus.fasta through Bio.SeqIO, consensus.vcf through
 * vcf_writer.SingleSampleWriter) for many (sample, flow) pairs at once, on host threads — host code, no device work, no
 * context.  Per job: fasta_path (nullable) gets ">fasta_id" and `sequence` in lines of 60; vcf_path (nullable) gets vcf_header
 * followed by one row per site whose record has status SNPGPU_ST_OK, in the order of line_off (pileup order) — with
 * site_in_flow (nullable, [n_sites]) only the sites it marks and those whose row mask carries SNPGPU_F_REGION (the parse set
 * of call_consensus.py:147-151 is the snplist plus the sample's exclude list); row_filters (nullable, [n_sites]) replaces the
 * records' own failed-filter masks.  Out: rc (0, SNPGPU_E_IO, or SNPGPU_E_UNSUPPORTED for a record with more than
 * SNPGPU_MAX_SYMS symbols that has no record in spill[n_spill]) and n_rows.  Contig names / site_keys / filter_names as snpgpu_format_vcf_rows; n_threads 0 = as
 * many as there are jobs, up to 64. */
typedef struct snpgpu_consensus_job {
    const char *fasta_path;
    const char *fasta_id;
    const uint8_t *sequence;
    uint64_t n_bases;
    const char *vcf_path;
    const char *vcf_header;
    const snpgpu_site_counts *counts;
    const uint64_t *line_off;
    const uint8_t *row_filters;
    const uint8_t *site_in_flow;
    int32_t rc;
    uint32_t n_rows;
} snpgpu_consensus_job;
int  snpgpu_write_consensus_files(snpgpu_consensus_job *jobs, uint32_t n_jobs, uint32_t n_sites, const uint8_t *contig_names,
                                  const uint32_t *contig_name_off, const uint64_t *site_keys, const char *const *filter_names,
                                  int preserve_ref_case, char failed_snp_gt, const snpgpu_symbol_spill *spill, uint32_t n_spill,
                                  uint32_t n_threads);

/* Both consensus flows from one call (run.py:704-718 calls every sample twice: at the positions of snplist.txt, and at those
 * of snplist_preserved.txt with the sample's var.flt_removed.vcf as exclude file).  From the result of the call over the FULL
 * list — d_base / d_filters / d_line_off, [n_samples][n_sites] — derives the preserved flow: d_out_base [n_samples][n_cols] =
 * the columns d_cols[n_cols] (slots of the preserved list, in its order), d_out_filters [n_samples][n_sites] = the filters,
 * and for every slot of a sample's exclude list (CSR d_excl_off[n_samples + 1] / d_excl_slots) whose position has a pileup
 * line: SNPGPU_F_REGION in the filters and '-' as its base when it is a column (d_col_of[n_sites]: column of a slot or -1) —
 * call_consensus.py:165-176.  d_err: one zeroed word, bit 0 = an exclude slot >= n_sites.  Asynchronous. */
int  snpgpu_region_flow_dev(snpgpu_ctx *ctx, const uint8_t *d_base, const uint8_t *d_filters, const uint64_t *d_line_off,
                            uint32_t n_samples, uint32_t n_sites, const uint32_t *d_cols, const int32_t *d_col_of, uint32_t n_cols,
                            const uint32_t *d_excl_off, const uint32_t *d_excl_slots, uint8_t *d_out_base, uint8_t *d_out_filters,
                            uint32_t *d_err);

/* Whole rows from one device matrix to another by index lists (ABI 7): dst row dst_index[r] <- src row src_index[r], r < n_rows; a null
 * list stands for r itself; strides and row_bytes in bytes; destination rows distinct; asynchronous on the context's stream.  The
 * one-job pipeline's gathers between its steps — the rows of a group's resident samples to their places (run.py:704-718 runs the
 * samples in any order), the packed rows into sorted-id order for the distance step (distance.py:76-84) — without leaving the
 * library's kernels. */
int  snpgpu_rows_copy_dev(snpgpu_ctx *ctx, const void *d_src, uint64_t src_stride, const uint32_t *d_src_index,
                          void *d_dst, uint64_t dst_stride, const uint32_t *d_dst_index, uint32_t n_rows, uint64_t row_bytes);

/* After a call_consensus on `ss`: for every site, 1 + the byte offset of the pileup line that was used (0 = no
 * line).  consensus.vcf rows are written in pileup order (call_consensus.py:161-180), which this recovers.
 * out_line_off[n_sites] is a HOST pointer; synchronous. */
int  snpgpu_siteset_line_offsets(snpgpu_ctx *ctx, const snpgpu_siteset *ss, uint64_t *out_line_off);

/* ---- snp_matrix / distance: utils.calculate_sequence_distance (utils.py:1135-1165) over all pairs
 *      (distance.py:93-98) ----------------------------------------------------------------------
 * The samples x sites matrix is packed 4 bits per site, planar per 32-site word:
 *   packed[row][word] = uint32[4] { valid (upper(c) in ACGT), code bit1, code bit0, lower-case flag }
 * with A=0 C=1 G=2 T=3.  A row holds ceil(n_sites / 32) words rounded up to a multiple of 4 (64 bytes); the padding
 * words are all zero (no valid site).  snpgpu_packed_row_bytes() is the row pitch. */
size_t snpgpu_packed_row_bytes(uint32_t n_sites);
int  snpgpu_pack_matrix_dev(snpgpu_ctx *ctx, const uint8_t *d_symbols, uint32_t n_rows, uint32_t n_sites,
                            size_t row_stride, void *d_packed);
/* d_out is a full n x n int32 matrix.  Only the 128x128 tiles t of the upper triangle with
 * t % tile_nranks == tile_rank are computed; each is written together with its mirror image.
 * Entries of other tiles are left untouched. */
int  snpgpu_distance_packed_dev(snpgpu_ctx *ctx, const void *d_packed, uint32_t n_rows, uint32_t n_sites,
                                uint32_t tile_rank, uint32_t tile_nranks, int32_t *d_out);
/* Host form: symbols is n_rows x n_sites bytes (row-major, the sequences of snpma.fasta); out is n x n int32. */
int  snpgpu_distance(snpgpu_ctx *ctx, const uint8_t *symbols, uint32_t n_rows, uint32_t n_sites, int32_t *out);

/* The first two columns of every record of a VCF (host code): what utils.convert_vcf_file_to_snp_set (utils.py:1113-1132)
 * and filter_regions.py:408-410 read through PyVCF3.  out_pos / out_contig [capacity] in file order (contig = index into
 * the file's names in order of first appearance, returned back to back in out_names with out_name_off[n_names + 1]);
 * *out_n_records may exceed `capacity` (then only the count is valid: come back with more room), SNPGPU_E_NOMEM: more
 * room for the names.  Plain files only — TAB-separated, POS of plain digits below 2^32, header before data: anything else
 * is SNPGPU_E_UNSUPPORTED and belongs to the caller's own reader. */
int  snpgpu_vcf_sites(const char *path, uint64_t capacity, uint32_t *out_pos, uint32_t *out_contig, uint64_t *out_n_records,
                      char *out_names, uint64_t names_capacity, uint64_t *out_name_off, uint32_t names_max,
                      uint32_t *out_n_names);
/* The first two columns of snplist.txt, what utils.read_snp_position_list (utils.py:1073-1088) returns: as snpgpu_vcf_sites,
 * but every line is a record (no header, no blank lines); a line outside the plain case — columns separated by anything but
 * TAB, a position that is not 1-10 plain digits — is SNPGPU_E_UNSUPPORTED and belongs to the caller's own reader. */
int  snpgpu_snplist_sites(const char *path, uint64_t capacity, uint32_t *out_pos, uint32_t *out_contig, uint64_t *out_n_records,
                          char *out_names, uint64_t names_capacity, uint64_t *out_name_off, uint32_t names_max,
                          uint32_t *out_n_names);
/* snplist.txt (utils.write_list_of_snps, utils.py:1056-1070): one "chrom\tpos\tcount\tname..." line per site; keys =
 * (contig index << 32) | pos in output order, carriers of site i = carriers[carrier_off[i], carrier_off[i+1]) as sample
 * indices; contig / sample names back to back with their offset arrays. */
int  snpgpu_write_snplist(const char *path, const char *contig_names, const uint64_t *contig_off, const uint64_t *keys,
                          uint64_t n_sites, const uint32_t *carrier_off, const uint32_t *carriers, const char *sample_names,
                          const uint64_t *sample_off);

/* snpma.fasta into a byte matrix (host code, no device work): replaces the read loop of distance.py:76-84 — text-mode lines
 * ("\n", "\r\n", lone "\r"), a line that starts with '>' opens a record named by the rest of the line without its leading
 * '>'s, every other line is appended to the current record.  Two passes: snpgpu_fasta_scan counts the records, the longest
 * sequence and the bytes of all names (SNPGPU_E_UNSUPPORTED: sequence text before the first header, which the reference
 * cannot handle either); snpgpu_fasta_load fills out_matrix (record r at r * row_stride, padded with `pad` up to row_stride
 * >= the longest sequence), out_len[n_records], and the names back to back with out_name_off[n_records + 1].  Records keep
 * the file's order and its duplicates (the reference's dict keeps the last of equal names: the caller's business). */
int  snpgpu_fasta_scan(const char *path, uint64_t *out_n_records, uint64_t *out_max_len, uint64_t *out_names_bytes);
int  snpgpu_fasta_load(const char *path, uint64_t n_records, uint64_t row_stride, uint8_t pad, uint8_t *out_matrix,
                       uint64_t *out_len, char *out_names, uint64_t *out_name_off);

/* The two text layouts of the distance step, written straight to `path` (host code, no device work): replaces the print
 * loops of distance.py:100-105 (SNPGPU_TSV_PAIRWISE: "Seq1\tSeq2\tDistance" header, one line per ordered pair, the
 * diagonal included) and distance.py:107-114 (SNPGPU_TSV_MATRIX: header "\t" + ids, one row per id).  ids: the names
 * back to back, name i = ids[id_off[i], id_off[i+1]); matrix: HOST int32, row i at matrix + i * row_stride.
 * SNPGPU_E_IO when the file cannot be created or written. */
#define SNPGPU_TSV_PAIRWISE 0
#define SNPGPU_TSV_MATRIX   1
int  snpgpu_write_distance_tsv(const char *path, int layout, const char *ids, const uint64_t *id_off, uint32_t n,
                               const int32_t *matrix, uint64_t row_stride);

/* ---- the exchange steps of the sharded path: RCCL over xGMI, one process per GPU ------------------------
 * The reference fans out one process per sample (run.py:704-718, `run_array` with `max_processes`) over a shared file system
 * and has no exchange step; sharded over GPUs the hot path has three: C1, the all-gather of the per-rank SNP site keys
 * (variable length: snpgpu_allgatherv); C2, the all-gather of the per-rank rows of the packed consensus matrix
 * (snpgpu_allgather, or snpgpu_allgatherv when the last rank holds fewer rows); the row-band exchange of the 128 x 128
 * distance tiles (snpgpu_tiles_gather_dev -> snpgpu_alltoallv -> snpgpu_tiles_scatter_dev).  librccl.so is loaded on first
 * use (snpgpu_comm_available() == 0: not found; single-GPU work is not affected).  Rank 0 makes a 128-byte id and hands it
 * to the other ranks by any means the host program has; every rank then calls snpgpu_comm_init on its own context (its own
 * device).  The collectives are ENQUEUED on the context's stream, like kernels: they return at once, results are valid when
 * the stream has passed them (snpgpu_ctx_sync, or snpgpu_stream_wait with a time limit: a rank that never arrives shows as
 * SNPGPU_E_TIMEOUT instead of a process that hangs).  All ranks must make the same calls in the same order. */
#define SNPGPU_COMM_ID_BYTES 128
int  snpgpu_comm_available(void);
int  snpgpu_comm_version(int *out_version);                    /* ncclGetVersion of the library that was loaded */
int  snpgpu_comm_unique_id(void *out_id);                      /* SNPGPU_COMM_ID_BYTES bytes; rank 0 */
int  snpgpu_comm_init(snpgpu_ctx *ctx, int rank, int nranks, const void *unique_id);
void snpgpu_comm_destroy(snpgpu_ctx *ctx);                     /* (also done by snpgpu_ctx_destroy) */
void snpgpu_comm_abort(snpgpu_ctx *ctx);                       /* without waiting for the stream: cancels what never completed (ncclCommAbort) */
/* this rank, the number of ranks, and what ncclCommCount says (0 without a communicator) */
int  snpgpu_comm_info(const snpgpu_ctx *ctx, int *out_rank, int *out_nranks, int *out_count_from_rccl);
/* block r of d_recv (bytes_per_rank bytes) comes from rank r's d_send */
int  snpgpu_allgather(snpgpu_ctx *ctx, const void *d_send, void *d_recv, size_t bytes_per_rank);
/* rank r contributes bytes[r] bytes, which land at d_recv + offsets[r] on every rank; bytes / offsets: HOST arrays of nranks
 * entries, the same on every rank (sizes are exchanged first, e.g. with snpgpu_allgather of one word) */
int  snpgpu_allgatherv(snpgpu_ctx *ctx, const void *d_send, void *d_recv, const uint64_t *bytes, const uint64_t *offsets);
/* send_bytes[p] bytes of d_send (blocks packed in rank order) go to rank p; recv_bytes[p] bytes from rank p arrive in d_recv
 * (blocks packed in rank order); HOST arrays */
int  snpgpu_alltoallv(snpgpu_ctx *ctx, const void *d_send, const uint64_t *send_bytes, void *d_recv, const uint64_t *recv_bytes);
int  snpgpu_stream_wait(snpgpu_ctx *ctx, uint32_t timeout_ms);
/* 128 x 128 tiles of an n_padded x n_padded int32 matrix (n_padded a multiple of 128) to and from a packed list: tile t is the
 * block at tile row d_tile_rows[t], tile column d_tile_cols[t]; d_out / d_tiles hold 128 * 128 values per tile */
int  snpgpu_tiles_gather_dev(snpgpu_ctx *ctx, const int32_t *d_matrix, uint32_t n_padded, const uint32_t *d_tile_rows,
                             const uint32_t *d_tile_cols, uint32_t n_tiles, int32_t *d_out);
int  snpgpu_tiles_scatter_dev(snpgpu_ctx *ctx, const int32_t *d_tiles, const uint32_t *d_tile_rows, const uint32_t *d_tile_cols,
                              uint32_t n_tiles, int32_t *d_matrix, uint32_t n_padded);
/* What the one-job pipeline asks about a group of samples after scan + call, per sample s -> d_out[s][3] (int64): [0] 1 when a
 * position the sample is asked about is malformed (d_wanted[n_sites]: 1 = every sample is asked about it; plus the sample's
 * own stretch d_excl_slots[d_excl_off[s] .. d_excl_off[s + 1]), both nullable), judged by the records' status bytes when
 * d_counts is given, else by bit 7 of the filter bytes; [1] positions of the set that have a pileup line (d_line_off != 0);
 * [2] positions with a record in the context's spill (d_counts only).  Inputs [n_samples][n_sites]. */
int  snpgpu_group_check_dev(snpgpu_ctx *ctx, const uint8_t *d_filters, const snpgpu_site_counts *d_counts, const uint64_t *d_line_off,
                            const uint8_t *d_wanted, const uint32_t *d_excl_off, const uint32_t *d_excl_slots,
                            uint32_t n_samples, uint32_t n_sites, int64_t *d_out);

/* ---- filter_regions: find_dense_regions (filter_regions.py:17-71) + utils.merge_regions
 *      (utils.py:1267-1282) + utils.in_region (utils.py:1314-1318); merge_sites (merge_sites.py:91-117) ---------
 * Hand-written sort / scan kernels (csrc/prims.h); each step comes as a host-pointer form (synchronous, one
 * synchronisation at its end) and a `_dev` form (device pointers, asynchronous on the context's stream, counts left in
 * device memory: d_out_n[0] = number of outputs, d_out_n[1] = error bits — 1: position outside [0, 2^40),
 * 2: interval with start > end).  Output capacities are stated with each function.
 *
 * dense windows: positions are grouped in segments (one per (sample, contig)); seg_off[n_segs+1]; the positions of a
 * segment may come in any order (they are sorted on the device, filter_regions.py:425).  For every rule r the candidate
 * window (p[i], p[i+max_snps[r]]) is emitted when p[i] + window[r] - 1 >= p[i+max_snps[r]].  out_start/out_end receive the
 * candidates (capacity n_pos * n_rules), out_seg their segment; order: by segment, position, rule.  At most 64 rules. */
int  snpgpu_dense_windows(snpgpu_ctx *ctx, const int64_t *positions, const uint32_t *seg_off, uint32_t n_segs,
                          const int32_t *max_snps, const int32_t *window, uint32_t n_rules,
                          int64_t *out_start, int64_t *out_end, uint32_t *out_seg, uint32_t *out_n);
int  snpgpu_dense_windows_dev(snpgpu_ctx *ctx, const int64_t *d_positions, const uint32_t *d_seg_off, uint32_t n_segs,
                              uint32_t n_pos, const int32_t *max_snps /* host */, const int32_t *window /* host */,
                              uint32_t n_rules, int64_t *d_out_start, int64_t *d_out_end, uint32_t *d_out_seg,
                              uint32_t *d_out_n /* 2 words */);
/* Merge intervals per group (intervals in any order, start <= end): sort by (group,start,end), running max of end, join
 * when start <= last_end + 1.  Output: merged regions ascending by (group, start); capacity n. */
int  snpgpu_merge_regions(snpgpu_ctx *ctx, const uint32_t *group, const int64_t *start, const int64_t *end,
                          uint32_t n, uint32_t *out_group, int64_t *out_start, int64_t *out_end, uint32_t *out_n);
int  snpgpu_merge_regions_dev(snpgpu_ctx *ctx, const uint32_t *d_group, const int64_t *d_start, const int64_t *d_end,
                              uint32_t n, uint32_t *d_out_group, int64_t *d_out_start, int64_t *d_out_end,
                              uint32_t *d_out_n /* 2 words */);
/* in_region for many positions: regions per group must be merged (disjoint, sorted); reg_off[n_groups+1].
 * out_flag[i] = 1 when positions[i] lies in a region of pos_group[i] (inclusive ends). */
int  snpgpu_in_regions(snpgpu_ctx *ctx, const uint32_t *pos_group, const int64_t *positions, uint32_t n_pos,
                       const uint32_t *reg_off, const int64_t *reg_start, const int64_t *reg_end,
                       uint32_t n_groups, uint8_t *out_flag);
int  snpgpu_in_regions_dev(snpgpu_ctx *ctx, const uint32_t *d_pos_group, const int64_t *d_positions, uint32_t n_pos,
                           const uint32_t *d_reg_off, const int64_t *d_reg_start, const int64_t *d_reg_end,
                           uint32_t n_groups, uint8_t *d_out_flag);

/* merge_sites: union of (CHROM,POS) over samples (merge_sites.py:91-117).
 * keys[i] = (contig_index << 32) | pos, sample_of_key[i] = sample index in sorted-dir order; records in any order.
 * Outputs: unique keys ascending, CSR offsets (n_unique+1) and the carrier sample indices in ascending
 * order per key, duplicates of (key, sample) collapsed (the reference builds a set per sample).
 * Capacities: out_unique[n], out_off[n+1], out_carrier[n].  _dev: d_out_n[0] = unique keys, d_out_n[1] = carriers. */
int  snpgpu_merge_sites(snpgpu_ctx *ctx, const uint64_t *keys, const uint32_t *sample_of_key, size_t n,
                        uint64_t *out_unique, uint32_t *out_off, uint32_t *out_carrier,
                        uint32_t *out_n_unique, uint32_t *out_n_carrier);
int  snpgpu_merge_sites_dev(snpgpu_ctx *ctx, const uint64_t *d_keys, const uint32_t *d_sample_of_key, uint32_t n,
                            uint64_t *d_out_unique, uint32_t *d_out_off, uint32_t *d_out_carrier,
                            uint32_t *d_out_n /* 2 words */);

/* ---- synthetic pileups for bench/tests, generated on the device --------------------
 * Fills d_out with the text of one sample's pileup over a single contig and returns its length in *out_nbytes
 * (synchronous; d_out capacity in bytes; d_out == NULL only queries the length).  d_site_alt[genome_len+1]: 0 or the ALT base at each 1-based position. */
typedef struct snpgpu_synth_params {
    uint64_t seed;
    uint32_t sample;
    uint32_t genome_len;
    float    mean_depth;
    float    carrier_p_same_clade;
    float    carrier_p_other_clade;
    uint32_t n_clades;
    char     contig[32];
} snpgpu_synth_params;
int  snpgpu_synth_reference_dev(snpgpu_ctx *ctx, uint64_t seed, uint32_t genome_len, uint8_t *d_ref /* genome_len+1 */);
int  snpgpu_synth_pileup_dev(snpgpu_ctx *ctx, const snpgpu_synth_params *p, const uint8_t *d_ref,
                             const uint8_t *d_site_alt, uint8_t *d_out, size_t capacity, size_t *out_nbytes);

#ifdef __cplusplus
}
#endif
#endif /* SNPGPU_H */
