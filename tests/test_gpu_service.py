"""The per-node service behind the per-sample CLI (snp_pipeline_amd/service.py): run.py's process array (run.py:704-718) with
SNPGPU_SERVICE set — thin clients, one server that keeps the device context — must write the same files, print the same log
and exit with the same codes as the in-process console script."""
import os
import subprocess
import sys
import time

import pytest

from oracle import fuzz
from oracle import pileup_oracle as po

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "bin", "cfsan_snp_pipeline")


def _fasta(name, seq):
    return ">%s\n" % name + "".join(seq[i:i + 60] + "\n" for i in range(0, len(seq), 60))


def test_call_consensus_through_the_service_equals_in_process(tmp_path):
    n = 9
    samples = []
    for k in range(n):
        data, _, sites = fuzz.synth_pileup(40 + k, genome_len=6000, n_sites=80)
        sdir = tmp_path / "samples" / ("s%d" % k)
        sdir.mkdir(parents=True)
        (sdir / "reads.all.pileup").write_bytes(data)
        samples.append((str(sdir), data, sites))
    sites = sorted(set(s for _, _, ss in samples for s in ss))
    snplist = str(tmp_path / "snplist.txt")
    with open(snplist, "w") as f:
        for c, p in sites:
            f.write("%s\t%d\t1\ts0\n" % (c.decode(), p))
    svc = str(tmp_path / "svc")
    env0 = {k: v for k, v in os.environ.items() if not k.startswith("SNPGPU_SERVICE")}
    env1 = dict(env0, SNPGPU_SERVICE=svc, SNPGPU_SERVICE_SPAWN="1", SNPGPU_SERVICE_IDLE="120")

    def cmd(sdir, out_name, extra=()):
        return [sys.executable, EXE, "call_consensus", "-f", "-l", snplist, "-o", os.path.join(sdir, out_name + ".fasta"), "-c", "0.7", "-D", "3",
                "--vcfFileName", out_name + ".vcf", "--vcfRefName", "ref.fasta"] + list(extra) + [os.path.join(sdir, "reads.all.pileup")]

    try:
        # the first client starts the server; outputs and log equal the in-process run's
        sdir, data, _ = samples[0]
        a = subprocess.run(cmd(sdir, "inproc"), env=env0, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
        b = subprocess.run(cmd(sdir, "served"), env=env1, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
        assert a.returncode == 0 and b.returncode == 0, (a.stderr[-2000:], b.stderr[-2000:])
        assert os.path.exists(os.path.join(svc, "dev0.sock"))
        want, _ = po.call_consensus_sites(data, sites, set(), po.CallerParams(0, 0.7, 3, 0, 0.0))
        for name in ("inproc", "served"):
            assert open(os.path.join(sdir, name + ".fasta")).read() == _fasta("s0", want.decode())
        assert open(os.path.join(sdir, "inproc.vcf")).read() == open(os.path.join(sdir, "served.vcf")).read()
        strip = lambda t: [ln for ln in t.splitlines() if not ln.startswith("# 20") and "inproc" not in ln and "served" not in ln]   # noqa: E731
        assert strip(a.stdout) == strip(b.stdout) and "call_consensus finished" in b.stdout
        # eight clients at once, each its own sample
        procs = [subprocess.Popen(cmd(s[0], "served"), env=env1, cwd=str(tmp_path), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE) for s in samples[1:]]
        assert [p.wait(timeout=300) for p in procs] == [0] * 8
        for k, (sdir, data, _) in enumerate(samples[1:], 1):
            want, _ = po.call_consensus_sites(data, sites, set(), po.CallerParams(0, 0.7, 3, 0, 0.0))
            assert open(os.path.join(sdir, "served.fasta")).read() == _fasta("s%d" % k, want.decode()), k
        # a malformed pileup: the reference's exception type in the client's error log, exit 98 / 100 by the client's StopOnSampleError
        bad = tmp_path / "samples" / "bad"
        bad.mkdir()
        (bad / "reads.all.pileup").write_bytes(b"chr\tnot_a_number\tA\t3\t...\tIII\n" + samples[0][1])
        for stop, code in (("false", 98), ("true", 100)):
            logs = {}
            for mode, env in (("inproc", env0), ("served", env1)):
                log = str(tmp_path / ("err_%s_%s.log" % (mode, stop)))
                r = subprocess.run(cmd(str(bad), mode), env=dict(env, errorOutputFile=log, StopOnSampleError=stop), capture_output=True, text=True,
                                   timeout=300, cwd=str(tmp_path))
                assert r.returncode == code, (mode, stop, r.returncode, r.stderr[-1500:])
                logs[mode] = [ln for ln in open(log).read().splitlines() if "inproc" not in ln and "served" not in ln]
            assert logs["inproc"] == logs["served"] and any("ValueError" in ln for ln in logs["served"])
        # the served process never loaded the HIP library or numpy: it is a thin client
        r = subprocess.run([sys.executable, "-X", "importtime", EXE] + cmd(samples[1][0], "served")[2:], env=env1, capture_output=True, text=True, timeout=300,
                           cwd=str(tmp_path))
        assert r.returncode == 0 and "numpy" not in r.stderr and "ctypes" not in r.stderr
        # a killed server leaves its socket file behind: the next client finds nobody there, starts a new server and is served by it
        import socket
        from snp_pipeline_amd import service
        subprocess.run([sys.executable, EXE, "serve", "--socketDir", svc, "--stop"], env=env0, capture_output=True, text=True, timeout=120)
        stale = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        for _ in range(200):                                        # (the stopped server unlinks its socket on the way out)
            if not os.path.exists(os.path.join(svc, "dev0.sock")):
                break
            time.sleep(0.05)
        stale.bind(os.path.join(svc, "dev0.sock"))
        stale.close()
        assert os.path.exists(os.path.join(svc, "dev0.sock")) and not service._probe(os.path.join(svc, "dev0.sock"))
        os.remove(os.path.join(samples[2][0], "served.fasta"))
        r = subprocess.run(cmd(samples[2][0], "served"), env=env1, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
        assert r.returncode == 0 and os.path.exists(os.path.join(samples[2][0], "served.fasta"))
        assert service._probe(os.path.join(svc, "dev0.sock"))       # a live server again
    finally:
        subprocess.run([sys.executable, EXE, "serve", "--socketDir", svc, "--stop"], env=env0, capture_output=True, text=True, timeout=120)
