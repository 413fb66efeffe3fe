"""The committed golden vectors are what the committed generator writes from the real reference.

Runs only where /root/reference exists (the build container): every slice of ``oracle/gen_golden.py`` is regenerated into a
scratch directory by importing the reference in place, and the decompressed JSON is compared with ``tests/golden/``; the
fixture archives are compared member by member (tar headers carry the checkout's mtimes).  Without this nothing would notice
the generator and its committed output drifting apart."""
import gzip
import io
import json
import lzma
import os
import subprocess
import sys
import tarfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "snppipeline")), reason="the reference checkout is not on this machine")


def _slices():
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("_gen_golden_names", os.path.join(ROOT, "oracle", "gen_golden.py"))
    src = open(spec.origin).read()
    # (the module inserts /root/reference into sys.path on import: read the table without importing it)
    import ast
    for node in ast.parse(src).body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") == "SLICES":
            return {k.value: v.elts[0].value for k, v in zip(node.value.keys, node.value.values)}
    raise AssertionError("no SLICES table in oracle/gen_golden.py")


@pytest.fixture(scope="module")
def regenerated(tmp_path_factory):
    out = tmp_path_factory.mktemp("golden_regen")
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_golden.py"), "--out", str(out)], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
    return str(out)


def test_every_committed_vector_file_has_a_slice():
    files = set(_slices().values())
    committed = {f for f in os.listdir(GOLD) if f.endswith(".json.gz")}
    assert files == committed, (sorted(files - committed), sorted(committed - files))


@pytest.mark.parametrize("name", sorted(_slices()) if os.path.isdir(os.path.join(REF, "snppipeline")) else [])
def test_slice_regenerates_identically(regenerated, name):
    fname = _slices()[name]
    new = json.loads(gzip.open(os.path.join(regenerated, fname)).read())
    old = json.loads(gzip.open(os.path.join(GOLD, fname)).read())
    assert new == old, "%s: the generator no longer writes the committed file; re-run oracle/gen_golden.py --only %s and look at the diff" % (fname, name)
    # mtime=0 in the gzip header and sorted keys: the bytes are reproducible too
    assert open(os.path.join(regenerated, fname), "rb").read() == open(os.path.join(GOLD, fname), "rb").read()


def test_fixture_trees_regenerate_identically(regenerated):
    for ds in ("lambdaVirus", "agona", "listeria"):
        def members(path):
            with tarfile.open(fileobj=io.BytesIO(lzma.decompress(open(path, "rb").read()))) as tar:
                return {m.name: tar.extractfile(m).read() for m in tar.getmembers() if m.isfile()}
        new, old = members(os.path.join(regenerated, "fixtures", ds, "expected.tar.xz")), members(os.path.join(GOLD, "fixtures", ds, "expected.tar.xz"))
        assert sorted(new) == sorted(old) and new == old, ds
        for extra in sorted(set(os.listdir(os.path.join(GOLD, "fixtures", ds))) - {"expected.tar.xz"}):
            assert open(os.path.join(regenerated, "fixtures", ds, extra), "rb").read() == open(os.path.join(GOLD, "fixtures", ds, extra), "rb").read(), (ds, extra)
