"""The C-ABI library loads and exports every symbol include/bonsai_amd.h declares (no compute calls:
there is no GPU in the CPU test tier)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "bonsai_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bns_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from bonsai_amd.build import build_device_library
    build_device_library()
    import bonsai_amd
    return bonsai_amd.load()


def test_every_declared_symbol_is_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    # and the Python binding covers the same set
    assert sorted(lib._bns_signatures) == syms


def test_error_surface_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tier")
    h = C.c_void_p()
    rc = lib.bns_create(0, C.byref(h))
    assert rc == -8 and not h.value                        # BNS_ERR_NO_DEVICE, no context leaked
    assert lib.bns_strerror(rc) == b"no usable GPU device"
    assert lib.bns_strerror(0) == b"ok" and lib.bns_strerror(-12345) == b"unknown error"
    assert lib.bns_set_encoder(None, 31, None, 1, 1) == -1  # NULL context -> BNS_ERR_ARG, no crash
    assert lib.bns_version() >= 100
    import bonsai_amd
    with pytest.raises(bonsai_amd.BonsaiAmdError):
        bonsai_amd.Context(0)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import bonsai_amd._lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "SO", str(tmp_path / "nope.so"))
    with pytest.raises(L.BonsaiAmdError, match="no CPU fallback"):
        L.load()


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under bonsai_amd/ or include/ may reference it."""
    bad = []
    for base in ("bonsai_amd", "include"):
        for dp, dn, fn in os.walk(os.path.join(ROOT, base)):
            for f in fn:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".c", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"oracle_lib|liboracle|bns_oracle|bo_classify|libbns_ref", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_gpu_tier_never_needs_the_reference_build():
    """oracle/_ref/libbns_ref.so -- the reference's own code compiled in the build container -- travels to the GPU box for ONE user:
    bench.py's cpu_baseline leg.  No `-m gpu` test may load it -- directly, or through a bench.py it spawns (those pass --no-ref; round 5's
    driver record listed the file among the libraries the GPU tier had mapped: two tests ran bench.py's CPU leg) --: what the GPU tier
    checks against are the committed vectors.  (Tests that mention it are CPU-tier cross-checks that skip when it is absent, or scripts
    under tests/golden/ that made the vectors.)"""
    import ast
    here = os.path.join(ROOT, "tests")
    bad = []
    for f in sorted(os.listdir(here)):
        if not (f.startswith("test_") and f.endswith(".py")):
            continue
        src = open(os.path.join(here, f)).read()
        module_gpu = re.search(r"^pytestmark\s*=\s*pytest\.mark\.gpu", src, re.M) is not None
        tree = ast.parse(src)
        for node in tree.body:
            if not isinstance(node, ast.FunctionDef) or not node.name.startswith("test_"):
                continue
            marked = module_gpu or any("gpu" in ast.unparse(d) for d in node.decorator_list)
            body = ast.get_source_segment(src, node) or ""
            if marked and re.search(r"libbns_ref|\.ref\(\)|oracle/_ref", body):
                bad.append("%s::%s" % (f, node.name))
            # ... nor through bench.py, whose CPU leg opens the file when it is there: a GPU test that runs that leg passes --no-ref
            # (--no-cpu and --dry-run-world switch it off themselves)
            if marked and re.search(r"bench\.py|_bench\(", body) and "--cpu-sample" in body and "--no-ref" not in body:
                bad.append("%s::%s (bench.py's CPU leg without --no-ref)" % (f, node.name))
    assert not bad, bad
