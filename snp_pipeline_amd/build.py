"""Build libsnpgpu.so (HIP kernels + C ABI) for gfx950, in-tree.

``python -m snp_pipeline_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles without a GPU.

``python -m snp_pipeline_amd.build --sanitize address`` (or ``thread``) builds a second library, ``lib/libsnpgpu_asan.so`` /
``lib/libsnpgpu_tsan.so``, whose HOST code — the threaded readers, the text parsers and writers of csrc/stream.hip, vcf_in.hip,
fasta_in.hip, tsv_out.hip, varscan_rows.hip, vcf_rows.hip: code that parses untrusted text — is instrumented with
AddressSanitizer + UndefinedBehaviorSanitizer (or ThreadSanitizer); the device code is compiled as always.  ``sanitized_env()``
gives the environment a Python process needs to load it (the sanitizer runtime preloaded, ``SNPGPU_LIB`` pointing at the
library); ``python -m snp_pipeline_amd.build --sanitize address --run <command ...>`` runs a command in it, e.g. the CPU test
suite or tools/fuzz_host.py.  tests/test_sanitized.py does that for a slice of both.
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libsnpgpu.so")
SOURCES = ["ctx.hip", "host_budget.hip", "scan.hip", "consensus.hip", "stream.hip", "varscan.hip", "varscan_rows.hip", "vcf_rows.hip", "tsv_out.hip", "fasta_in.hip", "vcf_in.hip", "distance.hip", "regions.hip", "synth.hip", "flows.hip", "lines_out.hip", "comm.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value",
         "-Wno-unused-result"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _flags():
    # SNPGPU_TUNING=1 builds the development instantiations of the scan kernel (per-wave time stamps, stream-only mode)
    # and their SNPGPU_SCAN_* environment knobs into a library for tools/; the product library never contains them.
    return FLAGS + (["-DSNPGPU_TUNING"] if os.environ.get("SNPGPU_TUNING") == "1" else [])


SANITIZERS = {
    # name: (suffix of the library, compile/link flags, runtime to preload, options of the runtime)
    "address": ("_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], "libclang_rt.asan-x86_64.so",
                {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:exitcode=97:verify_asan_link_order=0:protect_shadow_gap=0",
                 "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1:exitcode=97"}),
    "thread": ("_tsan", ["-fsanitize=thread"], "libclang_rt.tsan-x86_64.so",
               {"TSAN_OPTIONS": "exitcode=97:report_signal_unsafe=0:ignore_noninstrumented_modules=1"}),
}


def sanitized_lib(kind):
    return os.path.join(HERE, "lib", "libsnpgpu%s.so" % SANITIZERS[kind][0])


def _runtime_path(name):
    import glob
    found = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/" + name))
    if not found:
        raise RuntimeError("sanitizer runtime %s not found under /opt/rocm/lib/llvm" % name)
    return found[-1]


def sanitized_env(kind, base=None):
    """The environment in which a Python process loads the sanitized library instead of the product one."""
    env = dict(os.environ if base is None else base)
    _, _, runtime, options = SANITIZERS[kind]
    env["LD_PRELOAD"] = _runtime_path(runtime)
    env["SNPGPU_LIB"] = sanitized_lib(kind)
    for k, v in options.items():
        env.setdefault(k, v)
    return env


def build_sanitized(kind="address", verbose=True):
    """Compile every translation unit with the sanitizer on the host side and link lib/libsnpgpu_<kind>.so."""
    suffix, extra, _, _ = SANITIZERS[kind]
    out = sanitized_lib(kind)
    stamp_path = out + ".stamp"
    stamp = _stamp() + kind
    if os.path.exists(out) and os.path.exists(stamp_path) and open(stamp_path).read() == stamp:
        return out
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "lib", "obj" + suffix)
    os.makedirs(objdir, exist_ok=True)
    # -fno-gpu-sanitize: the device code stays as it is (device ASan needs xnack and another runtime); -O1 -g: readable reports
    flags = [f for f in FLAGS if f != "-O3"] + ["-O1", "-g", "-fno-omit-frame-pointer", "-fno-gpu-sanitize", "-shared-libsan"] + extra

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        r = subprocess.run([hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stdout))
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-sanitize", "-shared-libsan"] + extra + ["-o", out] + objs + ["-lpthread"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stdout)
    with open(stamp_path, "w") as f:
        f.write(stamp)
    if verbose:
        print("built", out, file=sys.stderr)
    return out


def _stamp():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)) + ["../../include/snpgpu.h"]:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode())
            h.update(f.read())
    h.update(" ".join(_flags()).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    """Compile every HIP translation unit and link the shared library.  Returns its path."""
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    stamp_path = LIB + ".stamp"
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_path) and open(stamp_path).read() == stamp:
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "lib", "obj")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc] + _flags() + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stdout))
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stdout)
    with open(stamp_path, "w") as f:
        f.write(stamp)
    if verbose:
        print("built", LIB, file=sys.stderr)
    return LIB


if __name__ == "__main__":
    if "--sanitize" in sys.argv:
        at = sys.argv.index("--sanitize")
        kind = sys.argv[at + 1] if len(sys.argv) > at + 1 and sys.argv[at + 1] in SANITIZERS else "address"
        build_sanitized(kind)
        if "--run" in sys.argv:
            cmd = sys.argv[sys.argv.index("--run") + 1:]
            sys.exit(subprocess.call(cmd, env=sanitized_env(kind)))
    else:
        build(force="--force" in sys.argv)
