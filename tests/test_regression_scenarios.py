"""The reference's shell regression suite (test/regression_tests.sh) for the hot-path subcommands, scenario by scenario, against
this build's console script run as a process: same set-up, same command lines, the same strings looked for (and looked for
in vain) in error.log and in the step's own log, the same exit codes under StopOnSampleError = true / false / unset.

Each scenario names the shell functions it mirrors.  Two departures, both forced by where the tests run:
* the suite runs as root, for whom ``chmod -w`` protects nothing: an unwritable output is a symbolic link into a directory
  that does not exist instead (it cannot be opened for writing, it can be removed), and the exception named in error.log may
  be FileNotFoundError beside the reference's ``IOError|PermissionError``;
* scenarios that need the aligners, samtools, the VarScan jar or a JVM (everything before call_sites' device pass) are out of
  scope; where the reference prepares a sample with index_ref / map_reads / call_sites, the bundled var.flt.vcf is copied in."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "bin", "cfsan_snp_pipeline")
STOP_MODES = (("true", 100), ("false", 0), (None, 100))          # StopOnSampleError -> exit code of a sample error
WRITE_ERRORS = "IOError|PermissionError|FileNotFoundError"


class Scenario(object):
    """A scratch copy of ``cfsan_snp_pipeline data lambdaVirusInputs`` with logs/ and error.log as run.py sets them up."""

    def __init__(self, tmp_path, stop):
        self.dir = str(tmp_path / ("stop_%s" % stop))
        self.logs = os.path.join(self.dir, "logs")
        os.makedirs(self.logs)
        os.makedirs(os.path.join(self.dir, "reference"))
        shutil.copy(os.path.join(ROOT, "tests", "golden", "fixtures", "lambdaVirus", "lambda_virus.fasta"), self.reference)
        for k in (1, 2, 3, 4):
            os.makedirs(self.sample(k))
        self.env = {k: v for k, v in os.environ.items() if not k.startswith("SNPGPU_SERVICE") and k not in ("StopOnSampleError", "errorOutputFile", "logDir")}
        self.env["errorOutputFile"] = self.error_log
        self.env["logDir"] = self.logs
        if stop is not None:
            self.env["StopOnSampleError"] = stop

    error_log = property(lambda self: os.path.join(self.dir, "error.log"))
    reference = property(lambda self: os.path.join(self.dir, "reference", "lambda_virus.fasta"))

    def path(self, *parts):
        return os.path.join(self.dir, *parts)

    def sample(self, k):
        return os.path.join(self.dir, "samples", "sample%d" % k)

    def write(self, rel, text):
        with open(self.path(rel), "w") as f:
            f.write(text)

    def list_samples(self, name):
        self.write(name, "".join(self.sample(k) + "\n" for k in (1, 2, 3, 4)))
        return self.path(name)

    def unwritable(self, rel):
        p = self.path(rel)
        if os.geteuid() == 0:
            os.symlink(os.path.join(self.dir, "no-such-directory", "file"), p)
        else:
            open(p, "w").close()
            os.chmod(p, 0o444)
        return p

    def remove(self, rel):
        p = self.path(rel)
        if os.path.islink(p):
            os.remove(p)
        elif os.path.exists(p):
            os.chmod(p, 0o644)
            os.remove(p)

    def run(self, *argv):
        """``cfsan_snp_pipeline <argv> &> log``: returns (exit code, the log's text)."""
        r = subprocess.run([sys.executable, EXE] + list(argv), env=self.env, cwd=self.dir, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        return r.returncode, r.stdout

    def errors(self):
        return open(self.error_log).read() if os.path.exists(self.error_log) else None

    def expected_results(self, fixture_trees, names):
        """Copy files of the bundled lambdaVirusExpectedResults tree into the samples (what the earlier steps would have left)."""
        root, _ = fixture_trees["lambdaVirus"]
        for k in (1, 2, 3, 4):
            for n in names:
                shutil.copy(os.path.join(root, "samples", "sample%d" % k, n), os.path.join(self.sample(k), n))


def check(text, contains=(), lacks=(), what=""):
    for s in contains:
        assert re.search(s, text) if s == WRITE_ERRORS else s in text, (what, "should contain", s, text[-1500:])
    for s in lacks:
        assert s not in text, (what, "should not contain", s, text[-1500:])


def each_stop_mode(tmp_path):
    for stop, code in STOP_MODES:
        yield Scenario(tmp_path, stop), code


# ---- global errors: the same under every StopOnSampleError ------------------------------------------------------------------
def test_missing_sample_directories_file_is_a_global_error(tmp_path):
    """tryFilterRegionsMissingSampleDirRaiseGlobalError, tryMergeSitesMissingSampleDirRaiseGlobalError,
    trySnpMatrixMissingSampleDirRaiseGlobalError (regression_tests.sh:2179, 2834, 3670)."""
    for sc, _ in each_stop_mode(tmp_path):
        dirs = sc.path("sampleDirectories.txt")
        for step, argv in (("filter_regions", [dirs, sc.reference]),
                           ("merge_sites", ["-o", sc.path("snplist.txt"), dirs, dirs + ".OrigVCF.filtered"]),
                           ("snp_matrix", ["-o", sc.path("snpma.fasta"), dirs])):
            rc, log = sc.run(step, *argv)
            assert rc == 100, (step, log[-1500:])
            check(sc.errors(), ["cfsan_snp_pipeline %s failed." % step, "File of sample directories %s does not exist" % dirs], what=step)
            check(log, ["File of sample directories %s does not exist" % dirs],
                  ["cfsan_snp_pipeline %s failed." % step, "cfsan_snp_pipeline %s finished" % step, "Use the -f option to force a rebuild"], what=step)
            os.remove(sc.error_log)


def test_filter_regions_missing_reference_and_outgroup_are_global_errors(tmp_path):
    """tryFilterRegionsMissingReferenceRaiseGlobalError, tryFilterRegionsMissingOutgroupRaiseGlobalError (:2231, :2288)."""
    for sc, _ in each_stop_mode(tmp_path):
        dirs = sc.list_samples("sampleDirectories.txt")
        for k in (1, 2, 3, 4):
            sc.write("samples/sample%d/var.flt.vcf" % k, "Dummy vcf content\n")
        rc, log = sc.run("filter_regions", dirs, sc.path("non-exist-reference"))
        assert rc == 100
        check(sc.errors(), ["cfsan_snp_pipeline filter_regions failed.", "Reference file %s does not exist" % sc.path("non-exist-reference")])
        check(log, ["Reference file %s does not exist" % sc.path("non-exist-reference")],
              ["cfsan_snp_pipeline filter_regions failed.", "cfsan_snp_pipeline filter_regions finished", "Use the -f option to force a rebuild"])
        os.remove(sc.error_log)
        rc, log = sc.run("filter_regions", "-g", sc.path("outgroup"), dirs, sc.reference)
        assert rc == 100
        check(sc.errors(), ["cfsan_snp_pipeline filter_regions failed.", "File of outgroup samples %s does not exist" % sc.path("outgroup")])
        check(log, ["File of outgroup samples %s does not exist" % sc.path("outgroup")],
              ["cfsan_snp_pipeline filter_regions failed.", "cfsan_snp_pipeline filter_regions finished", "Use the -f option to force a rebuild"])


def test_all_vcf_files_missing_is_a_global_error(tmp_path):
    """tryFilterRegionsMissingVcfRaiseGlobalError, tryMergeSitesMissingVcfRaiseGlobalError (:2345, :2886)."""
    for sc, _ in each_stop_mode(tmp_path):
        dirs = sc.list_samples("sampleDirList.txt")
        for step, argv in (("filter_regions", [dirs, sc.reference]),
                           ("merge_sites", ["-o", sc.path("snplist.txt"), dirs, sc.path("sampleDirectories.txt.OrigVCF.filtered")])):
            rc, log = sc.run(step, *argv)
            assert rc == 100, (step, log[-1500:])
            missing = ["VCF file %s/var.flt.vcf does not exist" % sc.sample(k) for k in (1, 2, 3, 4)]
            check(sc.errors(), ["cfsan_snp_pipeline %s failed." % step, "Error: all 4 VCF files were missing or empty"] + missing, what=step)
            check(log, ["Error: all 4 VCF files were missing or empty"] + missing,
                  ["cfsan_snp_pipeline %s failed." % step, "cfsan_snp_pipeline %s finished" % step, "Use the -f option to force a rebuild"], what=step)
            os.remove(sc.error_log)


def test_all_consensus_files_missing_is_a_global_error(tmp_path):
    """trySnpMatrixMissingConsensusRaiseGlobalError (:3722)."""
    for sc, _ in each_stop_mode(tmp_path):
        dirs = sc.list_samples("sampleDirList.txt")
        rc, log = sc.run("snp_matrix", "-o", sc.path("snpmap.fasta"), dirs)
        assert rc == 100
        missing = ["Consensus fasta file %s/consensus.fasta does not exist" % sc.sample(k) for k in (1, 2, 3, 4)]
        check(sc.errors(), ["cfsan_snp_pipeline snp_matrix failed.", "Error: all 4 consensus fasta files were missing or empty"] + missing)
        check(log, ["Error: all 4 consensus fasta files were missing or empty"] + missing,
              ["cfsan_snp_pipeline snp_matrix failed.", "cfsan_snp_pipeline snp_matrix finished", "Use the -f option to force a rebuild"])


def test_call_consensus_input_errors(tmp_path):
    """tryCallConsensusMissingSnpListRaiseGlobalError (:3102), tryCallConsensusMissingPileupRaiseSampleError (:3215),
    tryCallConsensusMissingExcludeRaiseSampleError (:3273): the two sample errors cannot be continued from — 100, or 98 when
    StopOnSampleError is false."""
    for sc, code in each_stop_mode(tmp_path):
        code = code or 98
        pileup = os.path.join(sc.sample(1), "reads.all.pileup")
        rc, log = sc.run("call_consensus", "-o", sc.path("consensus.fasta"), pileup)
        assert rc == 100
        both = ["cannot call consensus without the snplist file", "Snplist file snplist.txt does not exist"]
        check(sc.errors(), ["cfsan_snp_pipeline call_consensus failed."] + both)
        check(log, both, ["cfsan_snp_pipeline call_consensus failed.", "cfsan_snp_pipeline call_consensus finished", "Use the -f option to force a rebuild"])
        os.remove(sc.error_log)

        sc.write("snplist.txt", "fake snplist\n")
        rc, log = sc.run("call_consensus", "-l", sc.path("snplist.txt"), "-o", sc.path("consensus.fasta"), pileup)
        assert rc == code
        both = ["cannot call consensus without the pileup file", "Pileup file %s does not exist" % pileup]
        check(sc.errors(), ["cfsan_snp_pipeline call_consensus failed."] + both)
        check(log, both, ["cfsan_snp_pipeline call_consensus failed.", "cfsan_snp_pipeline call_consensus finished", "Use the -f option to force a rebuild"])
        os.remove(sc.error_log)

        sc.write("samples/sample1/reads.all.pileup", "fake pileup\n")
        exclude = os.path.join(sc.sample(1), "excludeFile.vcf")
        rc, log = sc.run("call_consensus", "-e", exclude, "-l", sc.path("snplist.txt"), "-o", sc.path("consensus.fasta"), pileup)
        assert rc == code
        both = ["cannot call consensus without the file of excluded positions", "Exclude file %s does not exist" % exclude]
        check(sc.errors(), ["cfsan_snp_pipeline call_consensus failed."] + both)
        check(log, both, ["cfsan_snp_pipeline call_consensus failed.", "cfsan_snp_pipeline call_consensus finished", "Use the -f option to force a rebuild"])


def test_call_consensus_corrupt_snplist_is_trapped(tmp_path):
    """tryCallConsensusCorruptSnplistTrap (:3047): the exception is raised in a function called read_snp_position_list and the
    trap's report says so; call_consensus has the per-sample hook: 100, or 98 when StopOnSampleError is false."""
    for sc, code in each_stop_mode(tmp_path):
        code = code or 98
        sc.write("snplist.txt", "Corrupt snplist content\n")
        sc.write("samples/sample1/reads.all.pileup", "Dummy pileup content\n")
        rc, log = sc.run("call_consensus", "-l", sc.path("snplist.txt"), "-o", sc.path("consensus.fasta"), os.path.join(sc.sample(1), "reads.all.pileup"))
        assert rc == code, log[-1500:]
        check(sc.errors(), ["Error detected while running cfsan_snp_pipeline call_consensus", "function read_snp_position_list at line"])
        check(log, [], ["Error detected while running cfsan_snp_pipeline call_consensus", "cfsan_snp_pipeline call_consensus finished",
                        "Use the -f option to force a rebuild"])


def test_snp_reference_input_errors(tmp_path):
    """trySnpReferenceMissingSnpListRaiseGlobalError (:3943), trySnpReferenceMissingReferenceRaiseGlobalError (:3997)."""
    for sc, _ in each_stop_mode(tmp_path):
        rc, log = sc.run("snp_reference", "-o", sc.path("referenceSNP.fasta"), sc.reference)
        assert rc == 100
        both = ["Snplist file snplist.txt does not exist", "cannot create the snp reference sequence without the snplist file"]
        check(sc.errors(), ["cfsan_snp_pipeline snp_reference failed."] + both)
        check(log, both, ["cfsan_snp_pipeline snp_reference failed.", "cfsan_snp_pipeline snp_reference finished", "Use the -f option to force a rebuild"])
        os.remove(sc.error_log)
        sc.write("snplist", "Dummy snplist content\n")
        os.remove(sc.reference)
        rc, log = sc.run("snp_reference", "-l", sc.path("snplist"), "-o", sc.path("referenceSNP.fasta"), sc.reference)
        assert rc == 100
        both = ["Reference file %s does not exist" % sc.reference, "cannot create the snp reference sequence without the reference fasta file"]
        check(sc.errors(), ["cfsan_snp_pipeline snp_reference failed."] + both)
        check(log, both, ["cfsan_snp_pipeline snp_reference failed.", "cfsan_snp_pipeline snp_reference finished", "Use the -f option to force a rebuild"])


def test_distance_input_errors(tmp_path):
    """tryDistanceMissingInputRaiseGlobalError (:4598), tryDistanceMissingOutputOptionsRaiseGlobalError (:4649)."""
    for sc, _ in each_stop_mode(tmp_path):
        snpma = sc.path("snpma.fasta")
        rc, log = sc.run("distance", "-p", "pp", "-m", "mm", snpma)
        assert rc == 100
        both = ["Error: cannot calculate sequence distances without the snp matrix file", "SNP matrix file %s does not exist" % snpma]
        check(sc.errors(), ["cfsan_snp_pipeline distance failed."] + both)
        check(log, both, ["cfsan_snp_pipeline distance failed", "cfsan_snp_pipeline distance finished", "Use the -f option to force a rebuild"])
        os.remove(sc.error_log)
        open(snpma, "w").close()
        rc, log = sc.run("distance", snpma)
        assert rc == 100
        check(sc.errors(), ["cfsan_snp_pipeline distance failed.", "Error: no output file specified"])
        check(log, ["Error: no output file specified"], ["cfsan_snp_pipeline distance failed", "cfsan_snp_pipeline distance finished",
                                                         "Use the -f option to force a rebuild"])


def test_call_sites_input_errors(tmp_path):
    """tryCallSitesMissingReferenceRaiseGlobalError (:1750), tryCallSitesMissingBamFileRaiseSampleError (:1808; not a
    continuable error: 100, or 98 when StopOnSampleError is false)."""
    for sc, code in each_stop_mode(tmp_path):
        code = code or 98
        os.remove(sc.reference)
        rc, log = sc.run("call_sites", sc.reference, "xxxx")
        assert rc == 100
        check(sc.errors(), ["cfsan_snp_pipeline call_sites failed", "Reference file %s does not exist" % sc.reference])
        check(log, ["Reference file %s does not exist" % sc.reference],
              ["cfsan_snp_pipeline call_sites failed", "cfsan_snp_pipeline call_sites finished", "Use the -f option to force a rebuild"])
        os.remove(sc.error_log)
        sc.write("reference/lambda_virus.fasta", ">x\nACGT\n")
        bam = os.path.join(sc.sample(1), "reads.sorted.deduped.indelrealigned.bam")
        rc, log = sc.run("call_sites", sc.reference, sc.sample(1))
        assert rc == code
        check(sc.errors(), ["cfsan_snp_pipeline call_sites failed", "Sample BAM file %s does not exist" % bam])
        check(log, ["Sample BAM file %s does not exist" % bam],
              ["cfsan_snp_pipeline call_sites failed", "cfsan_snp_pipeline call_sites finished", "Use the -f option to force a rebuild"])


# ---- traps around unwritable outputs ------------------------------------------------------------------------------------------
def _trap_unwritable(tmp_path, steps):
    for sc, _ in each_stop_mode(tmp_path):
        dirs = sc.list_samples("sampleDirectories.txt")
        for k in (1, 2, 3, 4):
            sc.write("samples/sample%d/consensus.fasta" % k, "Dummy content\n")
        sc.write("snplist.txt", "Dummy content\n")
        sc.write("snpma_in.fasta", "> Sequence\nACGT\n")
        table = {"snp_matrix": ("snpma.fasta", ["-o", sc.path("snpma.fasta"), dirs]),
                 "snp_reference": ("referenceSNP.fasta", ["-l", sc.path("snplist.txt"), "-o", sc.path("referenceSNP.fasta"), sc.reference]),
                 "distance": ("pairwise", ["-p", sc.path("pairwise"), sc.path("snpma_in.fasta")])}
        for step in steps:
            out, argv = table[step]
            sc.unwritable(out)
            rc, log = sc.run(step, *argv)
            assert rc == 100, (step, log[-1500:])
            check(sc.errors(), ["Error detected while running cfsan_snp_pipeline %s" % step, WRITE_ERRORS], what=step)
            check(log, [WRITE_ERRORS] if step == "distance" else [],
                  ["Error detected while running cfsan_snp_pipeline %s" % step, "cfsan_snp_pipeline %s finished" % step, "Use the -f option to force a rebuild"],
                  what=step)
            sc.remove(out)
            os.remove(sc.error_log)


def test_host_steps_trap_an_unwritable_output(tmp_path):
    """trySnpMatrixPermissionTrap (:3609), trySnpReferencePermissionTrap (:3885): the exception hook's report in error.log, not in
    the step's log."""
    _trap_unwritable(tmp_path, ("snp_matrix", "snp_reference"))


@pytest.mark.gpu
def test_distance_traps_an_unwritable_output(tmp_path):
    """tryDistancePermissionTrap (:4701): as above, and the traceback reaches the step's log through stderr."""
    _trap_unwritable(tmp_path, ("distance",))


@pytest.mark.gpu
def test_merge_sites_traps_an_unwritable_snplist(tmp_path):
    """tryMergeSitesPermissionTrap (:2772): four files that say "Dummy vcf content" are VCF files without records to PyVCF, so
    the step gets as far as its output file."""
    for sc, _ in each_stop_mode(tmp_path):
        dirs = sc.list_samples("sampleDirectories.txt")
        for k in (1, 2, 3, 4):
            sc.write("samples/sample%d/var.flt.vcf" % k, "Dummy vcf content\n")
        sc.unwritable("snplist.txt")
        rc, log = sc.run("merge_sites", "-o", sc.path("snplist.txt"), dirs, sc.path("sampleDirectories.txt.OrigVCF.filtered"))
        assert rc == 100, log[-1500:]
        check(sc.errors(), ["Error detected while running cfsan_snp_pipeline merge_sites", WRITE_ERRORS])
        check(log, [], ["Error detected while running cfsan_snp_pipeline merge_sites", "cfsan_snp_pipeline merge_sites finished", "Use the -f option to force a rebuild"])


@pytest.mark.gpu
def test_filter_regions_reports_an_unwritable_output_as_a_sample_error(tmp_path):
    """tryFilterRegionsPermissionTrap (:2030) and testFilterRegionsPermissionTrapNoStop (:2102): "Cannot create the file for
    preserved / removed SNPs" — fatal under StopOnSampleError true / unset, a logged sample error (exit 0, the step finishes)
    under false."""
    for sc, code in each_stop_mode(tmp_path):
        dirs = sc.list_samples("sampleDirectories.txt")
        for k in (1, 2, 3, 4):
            sc.write("samples/sample%d/var.flt.vcf" % k, "Dummy vcf content\n")
        for out, what in (("var.flt_preserved.vcf", "preserved"), ("var.flt_removed.vcf", "removed")):
            sc.unwritable("samples/sample1/" + out)
            rc, log = sc.run("filter_regions", dirs, sc.reference)
            assert rc == code, (what, rc, log[-1500:])
            if code == 100:
                check(sc.errors(), ["cfsan_snp_pipeline filter_regions failed.", "Cannot create the file for %s SNPs" % what], what=what)
                check(log, [], ["Error detected while running cfsan_snp_pipeline filter_regions", "cfsan_snp_pipeline filter_regions finished",
                                "Use the -f option to force a rebuild"], what=what)
            else:
                check(sc.errors(), ["cfsan_snp_pipeline filter_regions", "Cannot create the file for %s SNPs" % what], ["cfsan_snp_pipeline filter_regions failed."], what=what)
                check(log, ["cfsan_snp_pipeline filter_regions finished"],
                      ["Error detected while running cfsan_snp_pipeline filter_regions", "Use the -f option to force a rebuild"], what=what)
            sc.remove("samples/sample1/" + out)
            os.remove(sc.error_log)


# ---- sample errors: some inputs missing -----------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_some_vcf_files_missing_is_a_sample_error(tmp_path, fixture_trees):
    """tryFilterRegionsMissingVcfRaiseSampleError (:2406, :2455), tryMergeSitesMissingVcfRaiseSampleError (:2947 and its NoStop
    twin): sample1 has its var.flt.vcf, the other three have none."""
    root, _ = fixture_trees["lambdaVirus"]
    for sc, code in each_stop_mode(tmp_path):
        shutil.copy(os.path.join(root, "samples", "sample1", "var.flt.vcf"), os.path.join(sc.sample(1), "var.flt.vcf"))
        dirs = sc.list_samples("sampleDirList.txt")
        for step, argv in (("filter_regions", [dirs, sc.reference]),
                           ("merge_sites", ["-o", sc.path("snplist.txt"), dirs, sc.path("sampleDirectories.txt.OrigVCF.filtered")])):
            rc, log = sc.run(step, *argv)
            assert rc == code, (step, rc, log[-1500:])
            missing = ["VCF file %s/var.flt.vcf does not exist" % sc.sample(k) for k in (2, 3, 4)] + ["Error: 3 VCF files were missing or empty"]
            if code == 100:
                check(sc.errors(), ["cfsan_snp_pipeline %s failed." % step] + missing, what=step)
                check(log, missing, ["cfsan_snp_pipeline %s failed." % step, "cfsan_snp_pipeline %s finished" % step, "Use the -f option to force a rebuild"], what=step)
            else:
                check(sc.errors(), ["cfsan_snp_pipeline %s" % step] + missing, ["cfsan_snp_pipeline %s failed." % step], what=step)
                check(log, missing + ["cfsan_snp_pipeline %s finished" % step], ["cfsan_snp_pipeline %s failed." % step, "Use the -f option to force a rebuild"], what=step)
            os.remove(sc.error_log)


def test_some_consensus_files_missing_is_a_sample_error(tmp_path, fixture_trees):
    """trySnpMatrixMissingConsensusRaiseSampleError (:3788) and its NoStop twin."""
    for sc, code in each_stop_mode(tmp_path):
        sc.expected_results(fixture_trees, ["consensus.fasta"])
        dirs = sc.list_samples("sampleDirectories.txt")
        os.remove(os.path.join(sc.sample(1), "consensus.fasta"))
        os.remove(os.path.join(sc.sample(4), "consensus.fasta"))
        rc, log = sc.run("snp_matrix", "-o", sc.path("snpma.fasta"), dirs)
        assert rc == code, log[-1500:]
        missing = ["Consensus fasta file %s/consensus.fasta does not exist" % sc.sample(k) for k in (1, 4)] + ["Error: 2 consensus fasta files were missing or empty"]
        if code == 100:
            check(sc.errors(), ["cfsan_snp_pipeline snp_matrix failed."] + missing)
            check(log, missing, ["cfsan_snp_pipeline snp_matrix failed.", "cfsan_snp_pipeline snp_matrix finished", "Use the -f option to force a rebuild"])
        else:
            check(sc.errors(), ["cfsan_snp_pipeline snp_matrix"] + missing, ["cfsan_snp_pipeline snp_matrix failed."])
            check(log, missing + ["cfsan_snp_pipeline snp_matrix finished"], ["cfsan_snp_pipeline snp_matrix failed.", "Use the -f option to force a rebuild"])
            assert os.path.getsize(sc.path("snpma.fasta")) > 0


# ---- runs that succeed ------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_call_consensus_with_an_empty_snplist(tmp_path):
    """tryCallConsensusEmptySnpList (:3156): no error, non-empty consensus.fasta and consensus.vcf, "finished" in the log."""
    from oracle import fuzz
    for sc, _ in each_stop_mode(tmp_path):
        data, _, _ = fuzz.synth_pileup(3, genome_len=3000, n_sites=20)
        pileup = os.path.join(sc.sample(1), "reads.all.pileup")
        with open(pileup, "wb") as f:
            f.write(data)
        open(sc.path("snplist.txt"), "w").close()
        rc, log = sc.run("call_consensus", "-l", sc.path("snplist.txt"), "-o", os.path.join(sc.sample(1), "consensus.fasta"),
                         "--vcfFileName", os.path.join(sc.sample(1), "consensus.vcf"), pileup)
        assert rc == 0, log[-1500:]
        assert sc.errors() is None
        assert os.path.getsize(os.path.join(sc.sample(1), "consensus.fasta")) > 0 and os.path.getsize(os.path.join(sc.sample(1), "consensus.vcf")) > 0
        check(log, ["cfsan_snp_pipeline call_consensus finished"],
              ["cfsan_snp_pipeline call_consensus failed.", "cannot call consensus without the snplist file", "Snplist file %s is empty" % sc.path("snplist.txt"),
               "Use the -f option to force a rebuild"])


@pytest.mark.gpu
def test_filter_regions_partial_rebuild_and_outgroup(tmp_path, fixture_trees):
    """testFilterRegionsPartialRebuildModeAll / ModeEach (:2507, :2567), testFilterRegionsOutgroupModeAll / ModeEach (:2630,
    :2699): one output removed -> mode all rebuilds every sample to the bundled results, mode each processes that sample alone;
    an outgroup sample keeps all its SNPs (preserved = its var.flt.vcf, removed = header only)."""
    root, _ = fixture_trees["lambdaVirus"]
    day = 86400

    def age(path, days):
        t = os.stat(path).st_mtime - days * day
        os.utime(path, (t, t))

    def same_but_dates(a, b):
        strip = lambda p: [ln for ln in open(p) if not ln.startswith(("##fileDate", "##source"))]   # noqa: E731
        return strip(a) == strip(b)

    # mode all: remove one output, everything is rebuilt and equals the bundled expected results
    sc = Scenario(tmp_path, "true")
    sc.expected_results(fixture_trees, ["var.flt.vcf", "var.flt_preserved.vcf", "var.flt_removed.vcf"])
    dirs = sc.list_samples("sampleDirectories.txt")
    age(sc.reference, 12)
    for k in (1, 2, 3, 4):
        age(os.path.join(sc.sample(k), "var.flt.vcf"), 2)
        age(os.path.join(sc.sample(k), "var.flt_preserved.vcf"), 1)
        age(os.path.join(sc.sample(k), "var.flt_removed.vcf"), 1)
    os.remove(os.path.join(sc.sample(1), "var.flt_preserved.vcf"))
    rc, log = sc.run("filter_regions", "--window_size", "1000", "125", "15", "--max_snp", "3", "2", "1", "--mode", "all", dirs, sc.reference)
    assert rc == 0 and log and "already freshly built" not in log, log[-1500:]
    for k in (1, 2, 3, 4):
        for n in ("var.flt_preserved.vcf", "var.flt_removed.vcf"):
            assert same_but_dates(os.path.join(sc.sample(k), n), os.path.join(root, "samples", "sample%d" % k, n)), (k, n)

    # mode each: build all once, remove one output, only that sample is processed again
    sc = Scenario(tmp_path, "false")
    sc.expected_results(fixture_trees, ["var.flt.vcf"])
    dirs = sc.list_samples("sampleDirectories.txt")
    each = ["--window_size", "1000", "125", "15", "--max_snp", "3", "2", "1", "--mode", "each"]
    rc, log = sc.run("filter_regions", *(each + [dirs, sc.reference]))
    assert rc == 0, log[-1500:]
    age(sc.reference, 12)
    for k in (1, 2, 3, 4):
        age(os.path.join(sc.sample(k), "var.flt.vcf"), 2)
        for n in ("var.flt_preserved.vcf", "var.flt_removed.vcf"):
            age(os.path.join(sc.sample(k), n), 1)
            shutil.copy(os.path.join(sc.sample(k), n), os.path.join(sc.sample(k), n + ".save"))
    os.remove(os.path.join(sc.sample(1), "var.flt_preserved.vcf"))
    rc, log = sc.run("filter_regions", *(each + [dirs, sc.reference]))
    assert rc == 0 and "already freshly built" not in log, log[-1500:]
    check(log, ["Processing sample sample1"], ["Processing sample sample2", "Processing sample sample3", "Processing sample sample4"])
    for n in ("var.flt_preserved.vcf", "var.flt_removed.vcf"):
        assert same_but_dates(os.path.join(sc.sample(1), n), os.path.join(sc.sample(1), n + ".save"))

    # outgroup, mode all and mode each
    for mode in ("all", "each"):
        (tmp_path / mode).mkdir()
        sc = Scenario(tmp_path / mode, "false")
        sc.expected_results(fixture_trees, ["var.flt.vcf"])
        dirs = sc.list_samples("sampleDirectories.txt")
        plain = ["--edge_length", "500", "--window_size", "1000", "125", "15", "--max_snp", "3", "2", "1", "--mode", mode]
        rc, log = sc.run("filter_regions", *(plain + [dirs, sc.reference]))
        assert rc == 0, log[-1500:]
        for k in (1, 2, 3, 4):
            p = os.path.join(sc.sample(k), "var.flt_preserved.vcf")
            assert os.path.getsize(p) > 0
            shutil.copy(p, p + ".save")
            age(os.path.join(sc.sample(k), "var.flt.vcf"), 1)
        sc.write("outgroup.txt", "sample4\n")
        rc, log = sc.run("filter_regions", "--mode", mode, "--out_group", sc.path("outgroup.txt"), dirs, sc.reference)
        assert rc == 0 and "already freshly built" not in log and "cfsan_snp_pipeline filter_regions finished" in log, log[-1500:]
        for k in (1, 2, 3, 4):
            p = os.path.join(sc.sample(k), "var.flt_preserved.vcf")
            assert os.path.getsize(p) > 0
            if mode == "all" or k == 4:                     # (mode all: the outgroup's SNPs no longer count for anybody's regions)
                assert not same_but_dates(p, p + ".save"), (mode, k)
            else:
                assert same_but_dates(p, p + ".save"), (mode, k)
        out = sc.sample(4)
        assert open(os.path.join(out, "var.flt_preserved.vcf")).read() == open(os.path.join(out, "var.flt.vcf")).read()
        assert [ln for ln in open(os.path.join(out, "var.flt_removed.vcf")) if not ln.startswith("#")] == []


@pytest.mark.gpu
def test_merge_sites_excludes_samples_with_excessive_snps(tmp_path, fixture_trees):
    """testRunSnpPipelineExcessiveSnps (:6287), the merge_sites part: with --maxsnps 40 the bundled lambda samples 1 and 2 (46 and
    44 phase-1 SNPs) leave snplist.txt; of the preserved VCFs sample1 (32) stays and sample2 (41) goes."""
    sc = Scenario(tmp_path, "false")
    sc.expected_results(fixture_trees, ["var.flt.vcf", "var.flt_preserved.vcf"])
    dirs = sc.list_samples("sampleDirectories.txt")
    counts = {}
    for k in (1, 2):
        for n in ("var.flt.vcf", "var.flt_preserved.vcf"):
            counts[(k, n)] = sum(1 for ln in open(os.path.join(sc.sample(k), n)) if not ln.startswith("#"))
    assert counts == {(1, "var.flt.vcf"): 46, (2, "var.flt.vcf"): 44, (1, "var.flt_preserved.vcf"): 32, (2, "var.flt_preserved.vcf"): 41}
    rc, log = sc.run("merge_sites", "--maxsnps", "40", "-n", "var.flt.vcf", "-o", sc.path("snplist.txt"), dirs, dirs + ".OrigVCF.filtered")
    assert rc == 0 and "cfsan_snp_pipeline merge_sites finished" in log, log[-1500:]
    rc, log = sc.run("merge_sites", "--maxsnps", "40", "-n", "var.flt_preserved.vcf", "-o", sc.path("snplist_preserved.txt"), dirs, dirs + ".PresVCF.filtered")
    assert rc == 0 and "cfsan_snp_pipeline merge_sites finished" in log, log[-1500:]
    assert sc.errors() is None
    check(open(sc.path("snplist.txt")).read(), ["sample3", "sample4"], ["sample1", "sample2"])
    check(open(sc.path("snplist_preserved.txt")).read(), ["sample1", "sample3", "sample4"], ["sample2"])
    assert open(dirs + ".OrigVCF.filtered").read().split() == [sc.sample(3), sc.sample(4)]
    assert open(dirs + ".PresVCF.filtered").read().split() == [sc.sample(1), sc.sample(3), sc.sample(4)]


@pytest.mark.gpu
def test_the_steps_finish_when_no_sample_has_a_snp(tmp_path, fixture_trees):
    """testRunSnpPipelineZeroSnps (:5878), the hot steps: every PASS line deleted from the var.flt.vcf files -> empty snplist files,
    snpma.fasta and referenceSNP.fasta that are not empty and hold no line of bases, every step "finished", no error.log."""
    from oracle import fuzz
    sc = Scenario(tmp_path, "true")
    sc.expected_results(fixture_trees, ["var.flt.vcf"])
    for k in (1, 2, 3, 4):
        p = os.path.join(sc.sample(k), "var.flt.vcf")
        kept = [ln for ln in open(p) if "PASS" not in ln]                    # sed -i '/PASS/d'
        assert kept and all(ln.startswith("#") for ln in kept)
        open(p, "w").writelines(kept)
        with open(os.path.join(sc.sample(k), "reads.all.pileup"), "wb") as f:
            f.write(fuzz.synth_pileup(10 + k, genome_len=3000, n_sites=20)[0])
    dirs = sc.list_samples("sampleDirectories.txt")

    def step(name, *argv):
        rc, log = sc.run(name, *argv)
        assert rc == 0 and "cfsan_snp_pipeline %s finished" % name in log, (name, log[-1500:])

    step("filter_regions", dirs, sc.reference)
    for suffix, vcf, filt in (("", "var.flt.vcf", ".OrigVCF.filtered"), ("_preserved", "var.flt_preserved.vcf", ".PresVCF.filtered")):
        snplist = sc.path("snplist%s.txt" % suffix)
        step("merge_sites", "-n", vcf, "-o", snplist, dirs, dirs + filt)
        assert os.path.getsize(snplist) == 0
        for k in (1, 2, 3, 4):
            step("call_consensus", "-l", snplist, "-o", os.path.join(sc.sample(k), "consensus%s.fasta" % suffix),
                 "--vcfFileName", "consensus%s.vcf" % suffix, os.path.join(sc.sample(k), "reads.all.pileup"))
        step("snp_matrix", "-c", "consensus%s.fasta" % suffix, "-o", sc.path("snpma%s.fasta" % suffix), dirs + filt)
        step("snp_reference", "-l", snplist, "-o", sc.path("referenceSNP%s.fasta" % suffix), sc.reference)
        step("distance", "-p", sc.path("snp_distance_pairwise%s.tsv" % suffix), "-m", sc.path("snp_distance_matrix%s.tsv" % suffix), sc.path("snpma%s.fasta" % suffix))
        for n in ("snpma%s.fasta" % suffix, "referenceSNP%s.fasta" % suffix):
            lines = open(sc.path(n)).read().splitlines()
            assert lines and all(ln.startswith(">") for ln in lines), n
    assert sc.errors() is None


def test_already_fresh_outputs_are_left_alone(tmp_path, fixture_trees):
    """testAlreadyFreshOutputs (:6045), the hot steps: with every output newer than its inputs each step says, in the reference's
    words, that its output has already been freshly built, and touches nothing."""
    root, _ = fixture_trees["lambdaVirus"]
    sc = Scenario(tmp_path, "true")
    per_sample = ["var.flt.vcf", "var.flt_preserved.vcf", "var.flt_removed.vcf", "consensus.fasta", "consensus_preserved.fasta", "consensus.vcf",
                  "consensus_preserved.vcf"]
    sc.expected_results(fixture_trees, per_sample)
    top = ["snplist.txt", "snplist_preserved.txt", "snpma.fasta", "snpma_preserved.fasta", "referenceSNP.fasta", "referenceSNP_preserved.fasta",
           "snp_distance_pairwise.tsv", "snp_distance_matrix.tsv", "snp_distance_pairwise_preserved.tsv", "snp_distance_matrix_preserved.tsv"]
    for n in top:
        shutil.copy(os.path.join(root, n), sc.path(n))
    for k in (1, 2, 3, 4):
        sc.write("samples/sample%d/reads.sorted.deduped.indelrealigned.bam" % k, "Dummy\n")
        sc.write("samples/sample%d/reads.all.pileup" % k, "Dummy\n")
    dirs = sc.list_samples("sampleDirectories.txt")
    for filt in (".OrigVCF.filtered", ".PresVCF.filtered"):
        shutil.copy(dirs, dirs + filt)
    ages = [(15, [sc.reference]), (14, [dirs, dirs + ".OrigVCF.filtered", dirs + ".PresVCF.filtered"])]
    by_name = {"reads.sorted.deduped.indelrealigned.bam": 6, "reads.all.pileup": 5, "var.flt.vcf": 4, "var.flt_preserved.vcf": 3, "var.flt_removed.vcf": 3,
               "consensus.fasta": 1.5, "consensus_preserved.fasta": 1.5, "consensus.vcf": 1, "consensus_preserved.vcf": 1}
    for k in (1, 2, 3, 4):
        for n, days in by_name.items():
            ages.append((days, [os.path.join(sc.sample(k), n)]))
    ages += [(2, [sc.path("snplist.txt"), sc.path("snplist_preserved.txt")]), (0.8, [sc.path(n) for n in top if n.startswith(("snpma", "referenceSNP"))]),
             (0.5, [sc.path(n) for n in top if n.startswith("snp_distance")])]
    now = os.stat(sc.reference).st_mtime
    for days, paths in ages:
        for p in paths:
            os.utime(p, (now - days * 86400, now - days * 86400))
    outputs = [sc.path(n) for n in top] + [os.path.join(sc.sample(k), n) for k in (1, 2, 3, 4) for n in per_sample]
    before = {p: (os.stat(p).st_mtime_ns, open(p, "rb").read()) for p in outputs}
    fresh = "Use the -f option to force a rebuild"

    def step(name, says, *argv):
        rc, log = sc.run(name, *argv)
        assert rc == 0 and says in log and fresh in log, (name, log[-1500:])

    for k in (1, 2, 3, 4):
        rc, log = sc.run("call_sites", sc.reference, sc.sample(k))
        assert rc == 0, log[-1500:]
        check(log, ["Pileup file is already freshly created for sample%d.  %s." % (k, fresh), "VCF file is already freshly created for sample%d.  %s." % (k, fresh)])
    step("filter_regions", "All preserved and removed vcf files are already freshly built.  %s." % fresh, dirs, sc.reference)
    for suffix, vcf, filt in (("", "var.flt.vcf", ".OrigVCF.filtered"), ("_preserved", "var.flt_preserved.vcf", ".PresVCF.filtered")):
        snplist = sc.path("snplist%s.txt" % suffix)
        step("merge_sites", "snplist%s.txt has already been freshly built.  %s." % (suffix, fresh), "-n", vcf, "-o", snplist, dirs, dirs + filt)
        for k in (1, 2, 3, 4):
            extra = ["-e", os.path.join(sc.sample(k), "var.flt_removed.vcf")] if suffix else []
            step("call_consensus", "sample%d/consensus%s.fasta has already been freshly built.  %s." % (k, suffix, fresh), "-l", snplist, *(extra + [
                 "-o", os.path.join(sc.sample(k), "consensus%s.fasta" % suffix), "--vcfFileName", "consensus%s.vcf" % suffix, os.path.join(sc.sample(k), "reads.all.pileup")]))
        step("snp_matrix", "/snpma%s.fasta has already been freshly built.  %s." % (suffix, fresh), "-c", "consensus%s.fasta" % suffix, "-o", sc.path("snpma%s.fasta" % suffix),
             dirs + filt)
        step("snp_reference", "referenceSNP%s.fasta has already been freshly built.  %s." % (suffix, fresh), "-l", snplist, "-o", sc.path("referenceSNP%s.fasta" % suffix),
             sc.reference)
        step("distance", "have already been freshly built.  %s" % fresh, "-p", sc.path("snp_distance_pairwise%s.tsv" % suffix),
             "-m", sc.path("snp_distance_matrix%s.tsv" % suffix), sc.path("snpma%s.fasta" % suffix))
    assert sc.errors() is None
    assert {p: (os.stat(p).st_mtime_ns, open(p, "rb").read()) for p in outputs} == before
