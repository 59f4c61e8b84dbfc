"""The silesia hook (BASELINE.json quotes the metric on silesia; the corpus is not available offline): a directory of files named
like the corpus' members goes file by file through the drop-in file API (fourMCcompressFilename / fourMcDecompressFileName,
native/4mc.h:36-41) and every .4mc file must equal, byte for byte (SHA-256), what the reference CLI built from the reference's own
sources (oracle/_ref/4mc_ref) writes for the same file; bench.py's loader takes the same directory.
With SILESIA_DIR set the real files are used.  Without it the path is exercised on a STAND-IN directory: twelve files with the
members' names, filled from the synthetic generator (tools/corpus.c) - so that the hook cannot rot on boxes without the corpus."""
import hashlib
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers
from helpers import B, corpus, pkg

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEMBERS = ["dickens", "mozilla", "mr", "nci", "ooffice", "osdb", "reymont", "samba", "sao", "webster", "x-ray", "xml"]


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 22), b""):
            h.update(chunk)
    return h.hexdigest()


@pytest.fixture(scope="module")
def silesia_dir(tmp_path_factory):
    d = os.environ.get("SILESIA_DIR")
    if d and os.path.isdir(d) and any(os.path.isfile(os.path.join(d, f)) for f in os.listdir(d)):
        return d, True
    d = tmp_path_factory.mktemp("silesia_standin")
    rng = np.random.default_rng(0x4D43)
    for i, name in enumerate(MEMBERS):                      # ragged sizes: less than a block, a block and a bit, several blocks
        n = int(rng.integers(200_000, 9_000_000)) if i % 3 else int(rng.integers(1000, B))
        corpus(((n + B - 1) // B) * B, first_block=i)[:n].tofile(os.path.join(d, name))
    return str(d), False


def test_every_member_through_the_file_api_equals_the_reference_cli(silesia_dir, tmp_path):
    d, real = silesia_dir
    ref = helpers.ref_cli()
    if ref is None:
        pytest.skip("oracle/_ref/4mc_ref is not built (needs /root/reference at build time)")
    p = pkg(); p.gpu_init()
    L = p.lib()
    names = sorted(f for f in os.listdir(d) if os.path.isfile(os.path.join(d, f)))
    assert names, d
    for name in names:
        src = os.path.join(d, name)
        mine, theirs, back = tmp_path / (name + ".4mc"), tmp_path / (name + ".ref.4mc"), tmp_path / (name + ".back")
        assert L.fourMCcompressFilename(0, 1, src.encode(), str(mine).encode(), 1) == 0, name
        r = subprocess.run([ref, "-1", "-f", src, str(theirs)], capture_output=True)
        assert r.returncode == 0, (name, r.stderr)
        assert _sha(mine) == _sha(theirs), (name, "real silesia" if real else "stand-in")
        assert L.fourMcDecompressFileName(0, 1, str(mine).encode(), str(back).encode()) == 0, name
        assert _sha(back) == _sha(src), name
        for f in (mine, theirs, back):
            os.unlink(f)


def test_bench_loader_takes_the_directory(silesia_dir, monkeypatch):
    d, real = silesia_dir
    monkeypatch.setenv("SILESIA_DIR", d)
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    data, nb, note = bench.load_corpus(helpers, 48, B)
    names = sorted(f for f in os.listdir(d) if os.path.isfile(os.path.join(d, f)))
    total = sum(os.path.getsize(os.path.join(d, f)) for f in names)
    assert nb == total // B and len(data) == nb * B and "SILESIA_DIR" in note
    first = np.fromfile(os.path.join(d, names[0]), dtype=np.uint8)
    assert np.array_equal(data[: min(len(first), len(data))], first[: min(len(first), len(data))])
