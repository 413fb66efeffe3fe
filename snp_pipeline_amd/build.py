"""Build libsnpgpu.so (HIP kernels + C ABI) for gfx950, in-tree.

``python -m snp_pipeline_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles without a GPU.
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libsnpgpu.so")
SOURCES = ["ctx.hip", "scan.hip", "consensus.hip", "stream.hip", "varscan.hip", "varscan_rows.hip", "vcf_rows.hip", "tsv_out.hip", "fasta_in.hip", "vcf_in.hip", "distance.hip", "regions.hip", "synth.hip", "flows.hip"]
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
    build(force="--force" in sys.argv)
