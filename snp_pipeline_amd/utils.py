"""Host-side conventions of the hot subcommands: logging banner, error/exit protocol, make-style freshness,
and the small text codecs (snplist, VCF CHROM/POS, FASTA) the five steps share.

Mirrors the subset of snppipeline/utils.py the hot path uses (names and behaviour kept so a maintainer can
diff them): verbose_print :49, print_log_header :85, print_arguments :130, global_error :542,
sample_error :575, handle_global_exception :629, handle_sample_exception :675, verify_*_input_files :754/:804,
target_needs_rebuild :977, write_list_of_snps :1056, read_snp_position_list :1073,
convert_vcf_file_to_snp_set :1113, sample_id_from_file :469.  No arithmetic of the path lives here.
"""
from __future__ import print_function

import os
import re
import platform
import sys
import time
import traceback

__version__ = "2.2.1"            # the reference version whose CLI/formats this build reproduces

log_verbosity = 0


def set_logging_verbosity(args):
    global log_verbosity
    log_verbosity = args.verbose


def verbose_print(*args):
    if log_verbosity > 0:
        print(*args)


def timestamp():
    return time.strftime('%Y-%m-%d %H:%M:%S', time.localtime())


def program_name():
    return os.path.basename(sys.argv[0])


def program_name_with_command():
    program = os.path.basename(sys.argv[0])
    if program == "cfsan_snp_pipeline" and len(sys.argv) > 1:
        program += " " + sys.argv[1]
    return program


def command_line_short():
    return "%s %s" % (program_name(), " ".join(sys.argv[1:]))


def command_line_long():
    return " ".join(sys.argv)


def _ram_mbytes():
    try:
        return os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") // (1024 * 1024)
    except (ValueError, OSError, AttributeError):
        return 0


def print_log_header(classpath=False):
    verbose_print("# Command           : %s" % command_line_long())
    verbose_print("# Working Directory : %s" % os.getcwd())
    env = os.environ.get
    pbs, sge, sge_task = env("PBS_JOBID"), env("JOB_ID"), env("SGE_TASK_ID")
    slurm_array, slurm_job, slurm_task = env("SLURM_ARRAY_JOB_ID"), env("SLURM_JOBID"), env("SLURM_ARRAY_TASK_ID")
    if sge_task == "undefined":
        sge_task = None
    if pbs:
        verbose_print("# Job ID            : %s" % pbs)
    elif sge and sge_task:
        verbose_print("# Job ID            : %s[%s]" % (sge, sge_task))
    elif sge:
        verbose_print("# Job ID            : %s" % sge)
    elif slurm_array and slurm_task:
        verbose_print("# Job ID            : %s[%s]" % (slurm_array, slurm_task))
    elif slurm_job:
        verbose_print("# Job ID            : %s" % slurm_job)
    verbose_print("# Hostname          : %s" % platform.node())
    verbose_print("# RAM               : %s MB" % format(_ram_mbytes(), ",d"))
    if classpath:
        verbose_print("# CLASSPATH         : %s" % os.environ.get("CLASSPATH"))
    verbose_print("# Python Version    : %s" % sys.version.replace("\n", " "))
    verbose_print("# Program Version   : %s %s" % (program_name_with_command(), __version__))
    verbose_print("")
    verbose_print("# %s %s" % (timestamp(), command_line_short()))


def print_arguments(args):
    verbose_print("Options:")
    options = vars(args)
    for key in sorted(options):
        if key in ("subparser_name", "func", "excepthook"):
            continue
        verbose_print("    %s=%s" % (key, options[key]))
    verbose_print("")


# ---- error / exit protocol (exit 100 = stop the pipeline, 98 = this sample failed, continue) -----------------
def _append_error_log(lines):
    path = os.environ.get("errorOutputFile")
    if path:
        with open(path, "a") as err_log:
            for line in lines:
                print(line, file=err_log)


def report_error(message):
    if message is not None:
        _append_error_log([message])
    sys.stdout.flush()
    if message:
        print(message, file=sys.stderr)


def global_error(message):
    lines = ["%s failed." % program_name_with_command()]
    if message:
        lines.append(message)
    lines.append("=" * 80)
    _append_error_log(lines)
    sys.stdout.flush()
    if message:
        print(message, file=sys.stderr)
    sys.exit(100)


def _stop_on_sample_error():
    v = os.environ.get("StopOnSampleError")
    return v is None or v == "true"


def sample_error(message, continue_possible=False):
    stop = _stop_on_sample_error()
    first = "%s failed." % program_name_with_command() if (stop or not continue_possible) else "%s" % program_name_with_command()
    _append_error_log([first, message, "=" * 80])
    sys.stdout.flush()
    print(message, file=sys.stderr)
    if stop:
        sys.exit(100)
    if not continue_possible:
        sys.exit(98)


def _log_exception(exc_type, exc_value, exc_traceback):
    """The part the two exception hooks of the reference share (utils.py:629-700): an entry in the error log, the trace on
    stderr — or, for a failed external program (subprocess.CalledProcessError: samtools in call_sites), its command line."""
    import subprocess
    external_program_command = exc_value.cmd if exc_type == subprocess.CalledProcessError else None
    path = os.environ.get("errorOutputFile")
    if path:
        entries = traceback.extract_tb(exc_traceback)
        file_name, line_number, function_name, code_text = entries[-1] if entries else ("?", 0, "?", "")
        head = ["Error detected while running %s." % program_name_with_command(), "", "The command line was:", "    %s" % command_line_short(), ""]
        if external_program_command:
            body = ["The error occured while running:", "    %s" % external_program_command]
        else:
            body = ["%s exception in function %s at line %d in file %s" % (exc_type.__name__, function_name, line_number, file_name), "    %s" % code_text]
        _append_error_log(head + body + ["=" * 80])
    sys.stdout.flush()
    if external_program_command:
        print("Error occured while running:", file=sys.stderr)
        print("    %s" % external_program_command, file=sys.stderr)
    else:
        traceback.print_exception(exc_type, exc_value, exc_traceback)


def handle_global_exception(exc_type, exc_value, exc_traceback):
    _log_exception(exc_type, exc_value, exc_traceback)
    sys.exit(100)


def handle_sample_exception(exc_type, exc_value, exc_traceback):
    _log_exception(exc_type, exc_value, exc_traceback)
    sys.exit(100 if _stop_on_sample_error() else 98)


def _dispatch_error(err_messages, error_handler, continue_possible):
    if error_handler not in ("global", "sample", None):
        raise ValueError("Invalid error_handler: %s" % repr(error_handler))
    if err_messages:
        text = "\n".join(err_messages)
        if error_handler == "global":
            global_error(text)
        elif error_handler == "sample":
            sample_error(text, continue_possible=continue_possible)
        else:
            report_error(text)
    return len(err_messages)


def verify_existing_input_files(error_prefix, file_list, error_handler=None, continue_possible=False):
    bad = ["%s %s does not exist." % (error_prefix, p) for p in file_list if not os.path.isfile(p)]
    return _dispatch_error(bad, error_handler, continue_possible)


def verify_non_empty_input_files(error_prefix, file_list, error_handler=None, continue_possible=False, empty_ok=False):
    bad = []
    for p in file_list:
        if not os.path.isfile(p):
            bad.append("%s %s does not exist." % (error_prefix, p))
        elif not empty_ok and os.path.getsize(p) == 0:
            bad.append("%s %s is empty." % (error_prefix, p))
    return _dispatch_error(bad, error_handler, continue_possible)


def target_needs_rebuild(source_files, target_file):
    """make-style freshness: rebuild when the target is missing/empty or any existing source is newer."""
    if not os.path.isfile(target_file) or os.path.getsize(target_file) == 0:
        return True
    target_mtime = os.stat(target_file).st_mtime
    return any(os.path.isfile(s) and os.stat(s).st_mtime > target_mtime for s in source_files)


def sample_id_from_file(file_path):
    return os.path.basename(os.path.dirname(os.path.abspath(file_path)))


# ---- text codecs ------------------------------------------------------------------------------------------------
def write_list_of_snps(file_path, keys, carrier_lists):
    """snplist.txt: ``chrom\\tpos\\tcount\\tname...``; keys already in (chrom, pos) order."""
    with open(file_path, "w") as f:
        for (chrom, pos), names in zip(keys, carrier_lists):
            f.write("%s\t%d\t%d\t%s\n" % (chrom, pos, len(names), "\t".join(names)))


def read_snp_position_list(snp_list_file_path):
    """[(chrom, pos)] in file order; a malformed line raises (ValueError) exactly where the reference does."""
    snp_list = list()
    with open(snp_list_file_path, "r") as snp_list_file_object:
        for line in snp_list_file_object:
            chrom, pos = line.split(None, 2)[0:2]            # (the first two fields only: a snplist line lists every carrier)
            snp_list.append((chrom, int(pos)))
    return snp_list


def read_vcf_sites(vcf_file_path):
    """(CHROM, POS) of every data record, in file order (what the reference reads through PyVCF3's Reader in
    utils.py:1127 and filter_regions.py:408-410), plus the raw lines.  Returns (header_lines, data_lines, sites)."""
    header, data, sites = [], [], []
    with open(vcf_file_path, "r") as f:
        for line in f:
            if line.startswith("#"):
                header.append(line)
                continue
            if not line.strip():
                continue
            if not header or all(h.startswith("##") for h in header):
                # No "#CHROM" line yet: PyVCF's Reader takes the first line that does not start with "##" as the column
                # header WHATEVER it says (its first byte is dropped like the '#', the rest split at TABs and runs of blanks),
                # and the Writer prints it back as '#' + TAB-joined columns.  The reference's own regression tests rest on
                # that: with "Dummy vcf content" as var.flt.vcf, filter_regions and merge_sites get as far as creating their
                # output files (regression_tests.sh:2030-2093, 2772-2800).
                header.append("#" + "\t".join(re.split("\t| +", line.strip()[1:])) + "\n")
                continue
            fields = line.split("\t", 2)
            if len(fields) < 2:
                fields = line.split(None, 2)
            sites.append((fields[0], int(fields[1])))
            data.append(line)
    return header, data, sites


def _site_arrays_from_tuples(sites):
    import numpy as np
    order, index = [], {}
    for c, _ in sites:
        if c not in index:
            index[c] = len(order)
            order.append(c)
    return (order, np.fromiter((index[c] for c, _ in sites), dtype=np.uint32, count=len(sites)),
            np.fromiter((p for _, p in sites), dtype=np.int64, count=len(sites)))


def _read_site_arrays(symbol, file_path, python_reader):
    """(contig names in order of first appearance, contig index per record, position per record) of a VCF or a snplist by
    the library's host code (csrc/vcf_in.hip); files outside its plain case go through `python_reader` (-> [(chrom, pos)]),
    which also raises what the reference raises for them."""
    import ctypes as C
    import numpy as np
    from . import _lib as L
    lib = L.load()
    fn = getattr(lib, symbol)
    cap = max(1024, os.path.getsize(file_path) // 24)
    names_cap, names_max = 1 << 16, 4096
    while True:
        pos = np.empty(cap, dtype=np.uint32)
        cidx = np.empty(cap, dtype=np.uint32)
        names = C.create_string_buffer(names_cap)
        off = np.zeros(names_max + 1, dtype=np.uint64)
        n, nn = C.c_uint64(), C.c_uint32()
        rc = fn(os.fsencode(file_path), cap, pos.ctypes.data, cidx.ctypes.data, C.byref(n), names, names_cap,
                off.ctypes.data, names_max, C.byref(nn))
        if rc == L.E_IO:
            raise IOError("cannot read %s" % file_path)
        if rc == L.E_UNSUPPORTED:
            return _site_arrays_from_tuples(python_reader(file_path))
        if rc == L.E_NOMEM:
            names_cap, names_max = names_cap * 8, names_max * 8
            continue
        if rc != 0:
            raise RuntimeError("%s failed (%d)" % (symbol, rc))
        if n.value > cap:
            cap = n.value
            continue
        raw = names.raw
        return ([raw[int(off[i]):int(off[i + 1])].decode("utf-8") for i in range(nn.value)], cidx[:n.value], pos[:n.value].astype(np.int64))


def read_vcf_site_arrays(vcf_file_path):
    """The CHROM / POS columns of read_vcf_sites as numpy arrays (see _read_site_arrays, snpgpu_vcf_sites)."""
    return _read_site_arrays("snpgpu_vcf_sites", vcf_file_path, lambda path: read_vcf_sites(path)[2])


def read_snp_position_arrays(snp_list_file_path):
    """read_snp_position_list as numpy arrays: 200 000 lines cost the per-sample process ~0.15 s of Python otherwise."""
    if os.path.getsize(snp_list_file_path) == 0:
        import numpy as np
        return [], np.zeros(0, np.uint32), np.zeros(0, np.int64)
    return _read_site_arrays("snpgpu_snplist_sites", snp_list_file_path, read_snp_position_list)


def convert_vcf_file_to_snp_set(vcf_file_path):
    return set(read_vcf_sites(vcf_file_path)[2])


_WS = bytes(range(9, 14)) + bytes(range(28, 33))          # what str.split() removes from ASCII text
_WS_TO_SPACE = bytes.maketrans(_WS, b" " * len(_WS))       # (bytes.split() does not know 0x1c-0x1f; the text-mode loops below do)


def fasta_records_ascii(path):
    """[(record id, sequence bytes)] of a FASTA file the way the line loops below see it, whole records at a time (a 5 Mbp
    reference is 83 000 lines): the id is the first word of the header, the sequence the data lines without any white space;
    text before the first header is ignored.  Returns None for a file that is not plain ASCII or uses lone CRs as line ends
    (the line loops then do the work, with text-mode decoding and universal newlines)."""
    with open(path, "rb") as f:
        data = f.read()
    if not data.isascii() or (b"\r" in data and b"\r" in data.replace(b"\r\n", b"")):
        return None
    out = []
    # a header is a line that STARTS with '>': a '>' elsewhere is sequence text
    start = 0 if data.startswith(b">") else data.find(b"\n>") + 1
    if start == 0 and not data.startswith(b">"):
        return out
    for chunk in data[start + 1:].split(b"\n>"):
        head, _, body = chunk.partition(b"\n")
        words = head.translate(_WS_TO_SPACE).split()
        out.append((words[0].decode("ascii") if words else "", body.translate(None, _WS)))
    return out


def read_fasta_lengths(path):
    """{record id: sequence length} as Bio.SeqIO.parse yields them (id = first word of the header)."""
    fast = fasta_records_ascii(path)
    if fast is not None:
        return {name: len(seq) for name, seq in fast}
    lengths = {}
    name, n = None, 0
    with open(path, "r") as f:
        for line in f:
            if line.startswith(">"):
                if name is not None:
                    lengths[name] = n
                words = line[1:].split()
                name, n = (words[0] if words else ""), 0
            elif name is not None:
                n += len("".join(line.split()))
    if name is not None:
        lengths[name] = n
    return lengths


def write_fasta_record(handle, record_id, sequence, width=60):
    """Bio.SeqIO's FASTA writer layout: ``>id`` then the sequence wrapped at 60 columns."""
    handle.write(">%s\n" % record_id)
    for i in range(0, len(sequence), width):
        handle.write(sequence[i:i + width] + "\n")


def private_temp_dir():
    """This user's private scratch directory (lock files): see _paths.private_dir."""
    from . import _paths
    return _paths.private_dir()


# ---- the per-sample metrics file (name=value properties, utils.py:323-380 of the reference reads it back) ---------------
def update_properties(prop_file_path, updates, keep_mtime=False):
    """Set ``name=value`` lines in a properties file, keeping every other line; the file is created when missing.  The regular and
    the preserved call_consensus job of one sample may get here at the same time — on one host or, in an HPC job array, on two —
    so the read-modify-write is made safe in two ways that need nothing but the file system the metrics file is on:
      * the new content goes to a temporary file beside it and takes its place with os.replace: a reader never sees half a file,
        and a crash leaves the old one;
      * the update runs under an exclusive flock ON THE METRICS FILE'S DIRECTORY ENTRY (a lock file named after it, beside it,
        removed afterwards — the sample directory keeps no file the reference's tools do not write), which two hosts sharing the
        file system both see; where that cannot be had (a read-only or lock-less file system) the per-user lock directory of this
        host serves (`_paths.private_dir`), and without any lock the update goes ahead: these are optional by-products, never a
        reason for the step to fail.
    keep_mtime: an existing file keeps its modification time, so that make-style consumers (collect_metrics decides per metric
    with target_needs_rebuild) do not take its OTHER values for fresh."""
    import fcntl
    import hashlib
    lock, beside = None, None
    try:
        beside = prop_file_path + ".lock"
        for _ in range(50):
            lock = open(beside, "a")
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:                                                # the holder before us removed the name: our lock is on a file nobody else will find
                same = os.fstat(lock.fileno()).st_ino == os.stat(beside).st_ino
            except OSError:
                same = False
            if same:
                break
            lock.close()
            lock = None
        if lock is None:
            raise OSError("the lock file keeps changing")
    except (OSError, IOError):
        if lock is not None:
            lock.close()
        lock, beside = None, None
        try:
            digest = hashlib.sha1(os.path.realpath(prop_file_path).encode("utf-8", "surrogateescape")).hexdigest()
            lock = open(os.path.join(private_temp_dir(), "metrics-%s.lock" % digest), "a")
            fcntl.flock(lock, fcntl.LOCK_EX)
        except (OSError, IOError):
            if lock is not None:
                lock.close()
            lock = None
    try:
        before = os.stat(prop_file_path) if (keep_mtime and os.path.isfile(prop_file_path)) else None
        _update_properties_unlocked(prop_file_path, updates)
        if before is not None:
            os.utime(prop_file_path, ns=(before.st_atime_ns, before.st_mtime_ns))
    finally:
        if lock is not None:
            try:
                if beside is not None:                          # (while the lock is held: whoever waits on it opens a new one afterwards)
                    try:
                        os.unlink(beside)
                    except OSError:
                        pass
                fcntl.flock(lock, fcntl.LOCK_UN)
            finally:
                lock.close()


def _update_properties_unlocked(prop_file_path, updates):
    lines = []
    if os.path.isfile(prop_file_path):
        with open(prop_file_path, "r") as f:
            lines = f.read().split("\n")
        if lines and lines[-1] == "":
            lines.pop()
    left = dict(updates)
    out = []
    for line in lines:
        name = line.split("=", 1)[0].strip() if "=" in line and not line.lstrip().startswith("#") else None
        if name in left:
            out.append("%s=%s" % (name, left.pop(name)))
        else:
            out.append(line)
    out.extend("%s=%s" % kv for kv in left.items())
    text = "\n".join(out) + "\n"
    tmp = "%s.tmp.%d" % (prop_file_path, os.getpid())
    try:                                                        # the new content beside the file, then in its place in one step
        with open(tmp, "w") as f:
            f.write(text)
        os.replace(tmp, prop_file_path)
    except (OSError, IOError):
        try:
            os.unlink(tmp)
        except OSError:
            pass
        with open(prop_file_path, "w") as f:                    # (a directory that takes no new file: in place, as the reference writes it)
            f.write(text)
