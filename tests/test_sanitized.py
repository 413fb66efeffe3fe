"""The library's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer and ThreadSanitizer (race
detection / sanitizers), and bounded, seeded slices of the differential campaigns (tools/fuzz_*.py), so that neither is run by
hand only.

``python -m snp_pipeline_amd.build --sanitize address|thread`` compiles lib/libsnpgpu_asan.so / _tsan.so (host side instrumented,
device code as always); a child Python process loads it through SNPGPU_LIB with the sanitizer runtime preloaded
(build.sanitized_env).  A sanitizer report ends that process with exit code 97."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sanitized(kind, argv, timeout):
    from snp_pipeline_amd import build
    try:
        build.build_sanitized(kind, verbose=False)
        env = build.sanitized_env(kind)
    except RuntimeError as e:
        if "not found" in str(e):
            pytest.skip(str(e))
        raise
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable] + argv, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    report = "\n".join(ln for ln in (r.stdout + r.stderr).splitlines() if "Sanitizer" in ln or "runtime error" in ln or "WARNING: Thread" in ln)
    assert r.returncode == 0 and not report, (r.returncode, report or (r.stdout + r.stderr)[-3000:])
    return r.stdout


def test_host_parsers_and_writers_under_asan_and_ubsan():
    """The CPU tests of the library's text readers / writers (VCF, snplist, FASTA, TSV, var.flt.vcf rows, consensus files) in a
    process whose libsnpgpu is the ASan + UBSan build."""
    out = _sanitized("address", ["-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_host_cpu.py", "tests/test_abi.py",
                                 "-k", "library or varscan_host or abi or header or symbols or exports"], 900)
    assert " passed" in out and "failed" not in out


@pytest.mark.fuzz
def test_fuzz_host_slice_under_asan_and_ubsan():
    """20 seconds of tools/fuzz_host.py on fixed seeds — mutated VCF / snplist / FASTA files, consensus rows, TSVs against the
    Python loops — with the host code instrumented."""
    out = _sanitized("address", ["tools/fuzz_host.py", "20", "424242"], 600)
    assert "all agreed" in out


def test_threaded_host_code_under_tsan():
    """The host code that runs on several threads without a device — the FASTA loader by byte ranges, the TSV writer by row
    blocks, the consensus file writer's job queue — under ThreadSanitizer."""
    out = _sanitized("thread", ["-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_host_cpu.py",
                                "-k", "parallel or consensus_file_writer or fasta_matrix_loader or tsv_writer"], 900)
    assert " passed" in out and "failed" not in out


@pytest.mark.gpu
@pytest.mark.fuzz
@pytest.mark.parametrize("tool, seconds, seed, done", [
    ("fuzz_campaign.py", 45, 20260929, "all agreed"),           # kernels / per-sample commands vs the restatements
    ("fuzz_steps.py", 30, 20260929, "equal to the chain of restatements"),
    ("fuzz_jobs.py", 40, 20260929, "equal to the separate steps'"),
])
def test_campaign_slices_on_fixed_seeds(tool, seconds, seed, done):
    """Bounded slices of the three device campaigns on fixed seeds (they are timed: the number of cases depends on the box,
    the seeds do not)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(seconds), str(seed)], cwd=ROOT, capture_output=True, text=True,
                       timeout=seconds * 6 + 300)
    assert r.returncode == 0 and done in r.stdout, (r.stdout[-2500:], r.stderr[-2500:])
