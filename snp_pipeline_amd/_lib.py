"""ctypes binding of libsnpgpu.so (include/snpgpu.h).  There is no CPU fallback: if the library is missing
or no gfx950 device is visible, the calls raise."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsnpgpu.so")

MAX_SYMS = 8
SCAN_STATUS_WORDS = 4

F_RAWDPTH, F_VARFREQ, F_DEPTH, F_STRDPTH, F_STRBIAS, F_REGION = 1, 2, 4, 8, 16, 32
SITE_IN_SNPLIST, SITE_EXCLUDED = 1, 2
ST_NO_LINE, ST_OK, ST_SHORT_LINE, ST_BAD_DEPTH, ST_NO_QUALS, ST_MULTI_REF = 0, 1, 2, 3, 4, 5
E_HIP, E_ARG, E_NOMEM, E_PILEUP, E_UNSUPPORTED, E_IO, E_TIMEOUT = -1, -2, -3, -4, -5, -6, -7


class CallerParams(C.Structure):
    _fields_ = [("min_base_quality", C.c_int32), ("min_cons_depth", C.c_int32),
                ("min_cons_strand_depth", C.c_int32), ("reserved", C.c_int32),
                ("min_cons_freq", C.c_double), ("min_cons_strand_bias", C.c_double)]


class SiteCounts(C.Structure):
    _fields_ = [("raw_depth", C.c_uint32), ("good_depth", C.c_uint32), ("fwd_good_depth", C.c_uint32),
                ("rev_good_depth", C.c_uint32), ("n_symbols", C.c_uint32), ("ref_base", C.c_uint8),
                ("cons_base", C.c_uint8), ("filters", C.c_uint8), ("status", C.c_uint8),
                ("sym", C.c_uint8 * MAX_SYMS), ("total", C.c_uint32 * MAX_SYMS),
                ("fwd", C.c_uint32 * MAX_SYMS), ("rev", C.c_uint32 * MAX_SYMS)]


SPILL_SYMS, SPILL_REF, SPILL_CAP = 120, 64, 1024


class SymbolSpill(C.Structure):
    _fields_ = [("n", C.c_uint32), ("ref_len", C.c_uint32), ("depth64", C.c_int64), ("sym", C.c_uint8 * SPILL_SYMS),
                ("total", C.c_uint32 * SPILL_SYMS), ("fwd", C.c_uint32 * SPILL_SYMS), ("rev", C.c_uint32 * SPILL_SYMS),
                ("ref", C.c_uint8 * SPILL_REF)]


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("sample", C.c_uint32), ("genome_len", C.c_uint32),
                ("mean_depth", C.c_float), ("carrier_p_same_clade", C.c_float),
                ("carrier_p_other_clade", C.c_float), ("n_clades", C.c_uint32), ("contig", C.c_char * 32)]


class StreamOpts(C.Structure):
    _fields_ = [("chunk_bytes", C.c_uint32), ("n_staging", C.c_uint32), ("n_readers", C.c_uint32),
                ("n_slots", C.c_uint32), ("want_depth_sum", C.c_uint32), ("reserved", C.c_uint32 * 3)]


class StreamStats(C.Structure):
    _fields_ = [("bytes", C.c_uint64), ("n_chunks", C.c_uint64), ("seconds", C.c_double),
                ("seconds_waiting_for_readers", C.c_double), ("seconds_waiting_for_device", C.c_double),
                ("n_readers", C.c_uint32), ("n_staging", C.c_uint32), ("chunk_bytes", C.c_uint32),
                ("reserved", C.c_uint32), ("reader_seconds_reading", C.c_double),
                ("reader_seconds_waiting", C.c_double), ("seconds_enqueueing", C.c_double)]


class VarscanParams(C.Structure):
    _fields_ = [("min_coverage", C.c_uint32), ("min_reads2", C.c_uint32), ("min_avg_qual", C.c_uint32),
                ("reserved", C.c_uint32), ("min_var_freq", C.c_double)]


class VarscanFinish(C.Structure):
    _fields_ = [("p_value", C.c_double), ("min_freq_for_hom", C.c_double), ("strand_filter", C.c_int32), ("reserved", C.c_int32)]


class VarscanSite(C.Structure):
    _fields_ = [("line_off", C.c_uint64), ("sdp", C.c_uint32), ("dp", C.c_uint32), ("total", C.c_uint32),
                ("rdf", C.c_uint32), ("rdr", C.c_uint32), ("ref_qual_sum", C.c_uint32),
                ("adf", C.c_uint32), ("adr", C.c_uint32), ("alt_qual_sum", C.c_uint32),
                ("ref_base", C.c_uint8), ("alt_base", C.c_uint8), ("reserved", C.c_uint8 * 2)]


class PileupsStats(C.Structure):
    _fields_ = [("h2d_bytes", C.c_uint64), ("file_bytes", C.c_uint64), ("resident_bytes", C.c_uint64), ("budget_bytes", C.c_uint64),
                ("n_files", C.c_uint32), ("n_resident", C.c_uint32), ("seconds", C.c_double), ("seconds_allocating", C.c_double),
                ("seconds_waiting_for_readers", C.c_double), ("seconds_waiting_for_device", C.c_double),
                ("reader_seconds_reading", C.c_double), ("reader_seconds_waiting", C.c_double), ("seconds_preparing", C.c_double),
                ("n_readers", C.c_uint32), ("reserved", C.c_uint32)]


class CpuBudget(C.Structure):
    _fields_ = [("affinity_cpus", C.c_uint32), ("quota_cpus", C.c_uint32), ("max_cpu_cores", C.c_uint32), ("usable_cpus", C.c_uint32),
                ("local_ranks", C.c_uint32), ("budget", C.c_uint32), ("readers", C.c_uint32), ("writers", C.c_uint32)]


class ConsensusJob(C.Structure):
    _fields_ = [("fasta_path", C.c_char_p), ("fasta_id", C.c_char_p), ("sequence", C.c_void_p), ("n_bases", C.c_uint64),
                ("vcf_path", C.c_char_p), ("vcf_header", C.c_char_p), ("counts", C.c_void_p), ("line_off", C.c_void_p),
                ("row_filters", C.c_void_p), ("site_in_flow", C.c_void_p), ("rc", C.c_int32), ("n_rows", C.c_uint32)]


assert C.sizeof(SiteCounts) == 128 and C.sizeof(CallerParams) == 32 and C.sizeof(VarscanSite) == 48 and C.sizeof(VarscanParams) == 24

# name -> (restype, argtypes); every exported symbol of include/snpgpu.h
_P = C.c_void_p
SIGNATURES = {
    "snpgpu_abi_version": (C.c_int, []),
    "snpgpu_device_count": (C.c_int, []),
    "snpgpu_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "snpgpu_ctx_destroy": (None, [_P]),
    "snpgpu_last_error": (C.c_char_p, [_P]),
    "snpgpu_ctx_set_stream": (C.c_int, [_P, _P]),
    "snpgpu_ctx_reset_stream": (C.c_int, [_P]),
    "snpgpu_ctx_sync": (C.c_int, [_P]),
    "snpgpu_cpu_budget": (C.c_int, [C.POINTER(CpuBudget)]),
    "snpgpu_set_max_cpu_cores": (None, [C.c_uint32]),
    "snpgpu_set_local_ranks": (None, [C.c_uint32]),
    "snpgpu_timer_start": (C.c_int, [_P]),
    "snpgpu_timer_stop_ms": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "snpgpu_ctx_kernel_timing": (C.c_int, [_P, C.c_int]),
    "snpgpu_ctx_kernel_time_ms": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "snpgpu_siteset_create": (C.c_int, [_P, _P, _P, C.c_uint32, _P, _P, C.c_uint32, C.POINTER(_P)]),
    "snpgpu_siteset_destroy": (None, [_P]),
    "snpgpu_siteset_size": (C.c_uint32, [_P]),
    "snpgpu_call_consensus_dev": (C.c_int, [_P, _P, _P, C.c_size_t, C.POINTER(CallerParams), _P, _P, _P, _P, C.c_int]),
    "snpgpu_call_consensus_batch_dev": (C.c_int, [_P, _P, _P, _P, _P, C.c_uint32, C.POINTER(CallerParams), _P, _P, _P]),
    "snpgpu_call_consensus": (C.c_int, [_P, _P, _P, C.c_size_t, C.POINTER(CallerParams), _P, _P, _P, _P, C.c_int]),
    "snpgpu_call_consensus_files": (C.c_int, [_P, _P, C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(CallerParams), _P, _P, _P, _P,
                                              _P, _P, _P, _P, C.POINTER(StreamOpts), C.POINTER(StreamStats)]),
    "snpgpu_call_consensus_many_dev": (C.c_int, [_P, _P, _P, _P, C.c_uint32, C.POINTER(CallerParams), _P, _P, _P, _P, _P, _P, C.c_int]),
    "snpgpu_region_flow_dev": (C.c_int, [_P, _P, _P, _P, C.c_uint32, C.c_uint32, _P, _P, C.c_uint32, _P, _P, _P, _P, _P]),
    "snpgpu_rows_copy_dev": (C.c_int, [_P, _P, C.c_uint64, _P, _P, C.c_uint64, _P, C.c_uint32, C.c_uint64]),
    "snpgpu_write_consensus_files": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, _P, _P, C.POINTER(C.c_char_p), C.c_int, C.c_char, _P, C.c_uint32,
                                               C.c_uint32]),
    "snpgpu_varscan_dev": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint32, _P, C.POINTER(C.c_uint32), _P]),
    "snpgpu_varscan_batch_dev": (C.c_int, [_P, _P, _P, C.c_uint32, _P, C.c_uint32, _P, _P, _P, _P]),
    "snpgpu_pileups_create": (C.c_int, [_P, C.c_uint64, C.POINTER(_P)]),
    "snpgpu_pileups_destroy": (None, [_P]),
    "snpgpu_pileups_ingest": (C.c_int, [_P, _P, C.POINTER(C.c_char_p), C.c_uint32, _P, C.c_uint32, _P, _P, _P, _P, _P]),
    "snpgpu_pileups_count": (C.c_uint32, [_P]),
    "snpgpu_pileups_get": (C.c_int, [_P, C.c_uint32, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "snpgpu_pileups_get_stats": (C.c_int, [_P, C.POINTER(PileupsStats)]),
    "snpgpu_call_all_lines_file": (C.c_int, [_P, _P, C.c_char_p, C.POINTER(CallerParams), C.c_uint64, C.POINTER(C.c_uint64),
                                             _P, _P, _P, _P]),
    "snpgpu_call_all_lines_compact_file": (C.c_int, [_P, _P, C.c_char_p, C.POINTER(CallerParams), C.c_uint64, C.POINTER(C.c_uint64), _P, _P,
                                                     C.c_uint32, C.POINTER(C.c_uint32), _P, _P, _P]),
    "snpgpu_format_line_rows": (C.c_size_t, [_P, C.c_uint64, _P, _P, C.c_uint64, _P, _P, C.c_uint32, C.POINTER(C.c_char_p), C.c_int, C.c_char, _P, C.c_uint32,
                                             C.c_int, _P, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_int64)]),
    "snpgpu_write_all_positions_vcf": (C.c_int, [_P, _P, C.c_char_p, C.POINTER(CallerParams), C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_char,
                                                 C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _P, _P]),
    "snpgpu_format_vcf_rows": (C.c_size_t, [_P, _P, C.c_uint32, _P, _P, _P, C.POINTER(C.c_char_p), C.c_int, C.c_char, _P, C.c_uint32, _P,
                                            C.c_size_t, C.POINTER(C.c_int32)]),
    "snpgpu_symbol_spill_read": (C.c_int, [_P, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    "snpgpu_symbol_spill_capacity": (C.c_uint32, [_P]),
    "snpgpu_siteset_line_offsets": (C.c_int, [_P, _P, _P]),
    "snpgpu_packed_row_bytes": (C.c_size_t, [C.c_uint32]),
    "snpgpu_pack_matrix_dev": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_size_t, _P]),
    "snpgpu_distance_packed_dev": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P]),
    "snpgpu_distance": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P]),
    "snpgpu_varscan_file": (C.c_int, [_P, C.c_char_p, _P, C.c_uint32, _P, C.POINTER(C.c_uint32), _P]),
    "snpgpu_varscan_files": (C.c_int, [_P, C.POINTER(C.c_char_p), C.c_uint32, _P, C.c_uint32, _P, _P, _P, _P]),
    "snpgpu_varscan_format_rows": (C.c_size_t, [_P, C.c_uint32, _P, C.c_uint64, _P, _P, C.c_size_t, C.POINTER(C.c_uint32)]),
    "snpgpu_fasta_scan": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "snpgpu_fasta_load": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint8, _P, _P, _P, _P]),
    "snpgpu_vcf_sites": (C.c_int, [C.c_char_p, C.c_uint64, _P, _P, C.POINTER(C.c_uint64), _P, C.c_uint64, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    "snpgpu_snplist_sites": (C.c_int, [C.c_char_p, C.c_uint64, _P, _P, C.POINTER(C.c_uint64), _P, C.c_uint64, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    "snpgpu_write_snplist": (C.c_int, [C.c_char_p, _P, _P, _P, C.c_uint64, _P, _P, _P, _P]),
    "snpgpu_write_distance_tsv": (C.c_int, [C.c_char_p, C.c_int, _P, _P, C.c_uint32, _P, C.c_uint64]),
    "snpgpu_dense_windows": (C.c_int, [_P, _P, _P, C.c_uint32, _P, _P, C.c_uint32, _P, _P, _P, C.POINTER(C.c_uint32)]),
    "snpgpu_merge_regions": (C.c_int, [_P, _P, _P, _P, C.c_uint32, _P, _P, _P, C.POINTER(C.c_uint32)]),
    "snpgpu_in_regions": (C.c_int, [_P, _P, _P, C.c_uint32, _P, _P, _P, C.c_uint32, _P]),
    "snpgpu_merge_sites": (C.c_int, [_P, _P, _P, C.c_size_t, _P, _P, _P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "snpgpu_dense_windows_dev": (C.c_int, [_P, _P, _P, C.c_uint32, C.c_uint32, _P, _P, C.c_uint32, _P, _P, _P, _P]),
    "snpgpu_merge_regions_dev": (C.c_int, [_P, _P, _P, _P, C.c_uint32, _P, _P, _P, _P]),
    "snpgpu_in_regions_dev": (C.c_int, [_P, _P, _P, C.c_uint32, _P, _P, _P, C.c_uint32, _P]),
    "snpgpu_merge_sites_dev": (C.c_int, [_P, _P, _P, C.c_uint32, _P, _P, _P, _P]),
    "snpgpu_comm_available": (C.c_int, []),
    "snpgpu_comm_version": (C.c_int, [C.POINTER(C.c_int)]),
    "snpgpu_comm_unique_id": (C.c_int, [_P]),
    "snpgpu_comm_init": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "snpgpu_comm_destroy": (None, [_P]),
    "snpgpu_comm_abort": (None, [_P]),
    "snpgpu_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "snpgpu_allgather": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "snpgpu_allgatherv": (C.c_int, [_P, _P, _P, _P, _P]),
    "snpgpu_alltoallv": (C.c_int, [_P, _P, _P, _P, _P]),
    "snpgpu_stream_wait": (C.c_int, [_P, C.c_uint32]),
    "snpgpu_tiles_gather_dev": (C.c_int, [_P, _P, C.c_uint32, _P, _P, C.c_uint32, _P]),
    "snpgpu_tiles_scatter_dev": (C.c_int, [_P, _P, _P, _P, C.c_uint32, _P, C.c_uint32]),
    "snpgpu_group_check_dev": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_uint32, C.c_uint32, _P]),
    "snpgpu_synth_reference_dev": (C.c_int, [_P, C.c_uint64, C.c_uint32, _P]),
    "snpgpu_synth_pileup_dev": (C.c_int, [_P, C.POINTER(SynthParams), _P, _P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
}

_lib = None


class SnpGpuError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "snpgpu error %d: %s" % (code, message))
        self.code = code


TORCH_FREE_OK = False          # set by the console script: a CLI process never shares device pointers with torch
loaded_without_torch = False


def load():
    """Load libsnpgpu.so.  torch is imported first so that the HIP runtime torch ships (same SONAME) is the one
    and only runtime in the process — device pointers of torch tensors are then valid for our kernels.  The console
    script skips that (its processes never touch torch, and the import costs seconds per sample process) unless torch
    is already loaded or SNPGPU_TORCH=1."""
    global _lib, loaded_without_torch
    if _lib is not None:
        return _lib
    path = os.environ.get("SNPGPU_LIB") or LIB_PATH          # (SNPGPU_LIB: the sanitizer builds of build.py, loaded by its sanitized_env)
    if not os.path.exists(path):
        raise ImportError("%s is missing: run `python -m snp_pipeline_amd.build` (there is no CPU fallback)" % path)
    import sys
    if "torch" in sys.modules or not TORCH_FREE_OK or os.environ.get("SNPGPU_TORCH") == "1":
        import torch  # noqa: F401  (side effect: loads libamdhip64)
    else:
        loaded_without_torch = True
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
