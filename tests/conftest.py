import gzip
import io
import json
import lzma
import os
import sys
import tarfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "fuzz: a bounded, seeded slice of one of the differential campaigns of tools/fuzz_*.py")


@pytest.fixture(scope="session", autouse=True)
def _library_is_built():
    """libsnpgpu.so compiled from the sources in the tree (a no-op when its content stamp is current), so that any subset
    of the tests finds it; the product itself never builds on demand and never falls back."""
    from snp_pipeline_amd import build
    build.build(verbose=False)


def load_golden(name):
    with gzip.open(os.path.join(GOLD, name), "rb") as f:
        return json.loads(f.read().decode())


@pytest.fixture(scope="session")
def pileup_vectors():
    return load_golden("pileup_vectors.json.gz")


@pytest.fixture(scope="session")
def steps_vectors():
    return load_golden("steps_vectors.json.gz")


@pytest.fixture(scope="session")
def longref_vectors():
    return load_golden("longref_vectors.json.gz")


def extract_fixture(dataset, dest):
    """Unpack tests/golden/fixtures/<dataset>/expected.tar.xz into dest; returns meta."""
    d = os.path.join(GOLD, "fixtures", dataset)
    with open(os.path.join(d, "expected.tar.xz"), "rb") as f:
        raw = lzma.decompress(f.read())
    with tarfile.open(fileobj=io.BytesIO(raw)) as tar:
        tar.extractall(dest)
    with open(os.path.join(d, "meta.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def fixture_trees(tmp_path_factory):
    out = {}
    for ds in ("lambdaVirus", "agona", "listeria"):
        dest = tmp_path_factory.mktemp(ds)
        meta = extract_fixture(ds, str(dest))
        out[ds] = (str(dest), meta)
    return out
