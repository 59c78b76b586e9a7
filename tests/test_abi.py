"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/spliser.h declares, the
BAM reader round-trips, and the compute entry points fail loudly without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from spliser_amd import native, samio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    native.build()
    return native.lib()


def test_library_exports_every_declared_symbol(built):
    header = open(os.path.join(ROOT, "include", "spliser.h")).read()
    declared = set(re.findall(r"\b(spl_[a-z0-9_]+)\s*\(", header))
    declared -= {"spl_status"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(built, name), "libspliser_hip.so does not export %s" % name
    assert set(native.EXPORTS) == declared
    assert built.spl_abi_version() == 1


def test_no_cpu_fallback_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(native.SpliserNativeError) as err:
        native.Context(0)
    assert "no CPU fallback" in str(err.value)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "spliser_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in text and "spliser_oracle" not in text and "import oracle" not in text \
                    and "from oracle" not in text, f


@pytest.mark.parametrize("kw", [dict(), dict(with_seq=True, unplaced=3), dict(long_cigar_tag=True, level=6)])
def test_bam_roundtrip(built, tmp_path, kw):
    names, sets = samio.read_sam(os.path.join(ROOT, "tests", "golden", "random_b", "reads.sam"))
    path = str(tmp_path / "x.bam")
    samio.write_bam(path, names, [10 ** 8] * len(names), [(c, sets[c]) for c in names if c in sets], **kw)
    for threads in (1, 4):
        bam = native.BamFile(path, threads=threads)
        assert bam.ref_names == names
        assert bam.n_records == sum(s.n for s in sets.values()) + kw.get("unplaced", 0)
        for c in names:
            got, want = bam.reads(c), sets[c]
            assert np.array_equal(got.pos, want.pos) and np.array_equal(got.flag, want.flag)
            assert np.array_equal(got.cig_off, want.cig_off) and np.array_equal(got.cigar, want.cigar)
            assert got.max_end == want.max_end
        bam.close()


def test_bam_zlib_path_matches_libdeflate(built, tmp_path, monkeypatch):
    # the inflate backend is chosen once per process: run the zlib path in a child
    import subprocess
    import sys
    names, sets = samio.read_sam(os.path.join(ROOT, "tests", "golden", "random_a", "reads.sam"))
    path = str(tmp_path / "x.bam")
    samio.write_bam(path, names, [10 ** 8] * len(names), [(c, sets[c]) for c in names if c in sets])
    code = ("import sys; sys.path.insert(0, %r); from spliser_amd import native; b = native.BamFile(%r, 2); "
            "print(b.n_records, int(b.reads('Chr1').pos.sum()))" % (ROOT, path))
    env = dict(os.environ, SPL_BAM_NO_LIBDEFLATE="1")
    out = subprocess.check_output([sys.executable, "-c", code], env=env).decode().split()
    assert int(out[0]) == sum(s.n for s in sets.values())
    assert int(out[1]) == int(sets["Chr1"].pos.astype(np.int64).sum())


def test_bam_errors_are_loud(built, tmp_path):
    with pytest.raises(native.SpliserNativeError) as e1:
        native.BamFile(os.path.join(ROOT, "tests", "golden", "kat1", "reads.sam"))
    assert "not BGZF" in str(e1.value)
    with pytest.raises(native.SpliserNativeError):
        native.BamFile(str(tmp_path / "missing.bam"))
    names, sets = samio.read_sam(os.path.join(ROOT, "tests", "golden", "random_a", "reads.sam"))
    good = str(tmp_path / "g.bam")
    samio.write_bam(good, names, [10 ** 8] * len(names), [(c, sets[c]) for c in names if c in sets])
    data = open(good, "rb").read()
    trunc = str(tmp_path / "t.bam")
    open(trunc, "wb").write(data[:-28])
    with pytest.raises(native.SpliserNativeError) as e2:
        native.BamFile(trunc)
    assert "truncated" in str(e2.value)
    corrupt = bytearray(data)
    corrupt[len(corrupt) // 2] ^= 0xFF
    bad = str(tmp_path / "c.bam")
    open(bad, "wb").write(bytes(corrupt))
    with pytest.raises(native.SpliserNativeError):
        native.BamFile(bad)


def test_importing_the_package_asks_for_eight_hardware_queues():
    """The HIP runtime deals streams out to four hardware queues unless told otherwise, and a `process` call has eight streams:
    the package says so before anything can start the runtime (spl_create's comment has the measurements); what the caller has
    set stands."""
    import subprocess
    import sys
    code = "import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); import spliser_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    assert subprocess.check_output([sys.executable, "-c", code], cwd=ROOT).decode().strip() == "8"
    code = "import os; os.environ['GPU_MAX_HW_QUEUES'] = '5'; import spliser_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    assert subprocess.check_output([sys.executable, "-c", code], cwd=ROOT).decode().strip() == "5"


def test_integration_md_quotes_the_stub():
    """INTEGRATION.md shows spliser_amd/refstub.py itself, not a paraphrase of it (tests/test_refstub_reference.py runs it on the
    reference's live objects in the build container)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "spliser_amd", "refstub.py")).read()
    code = src[src.index("import ctypes"):]
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    assert "```python\n" + code + "```" in doc
