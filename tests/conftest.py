import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # GPU tier: torch (device tensors, torch.distributed) must initialise ITS HIP runtime before the C-ABI library maps its own
    # copy of libamdhip64 -- in the other order torch finds "No HIP GPUs" (tests that use both import torch lazily).
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass


def pytest_sessionstart(session):
    """Safety net: the built artefacts are git-ignored; (re)build whatever is missing before any test imports them.
    (hipcc cross-compiles without a GPU; nothing here runs the hot path.)"""
    need = [os.path.join(ROOT, "bonsai_amd", "lib", "libbonsai_amd.so"), os.path.join(ROOT, "bonsai_amd", "lib", "libbns_host.so"),
            os.path.join(ROOT, "bonsai_amd", "bin", "bonsai"), os.path.join(ROOT, "bonsai_amd", "bin", "bns_api_check"),
            os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        import importlib.util
        spec = importlib.util.spec_from_file_location("graft_entry", os.path.join(ROOT, "__graft_entry__.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib


@pytest.fixture(scope="session")
def small_world(oracle):
    """Synthetic 6-leaf taxonomy + genomes with shared segments + the oracle-built k=31 db."""
    import synth
    return synth.make_world(oracle, seed=11, k=31, genome_len=6000)


@pytest.fixture(scope="session")
def gpu_ctx():
    import bonsai_amd
    ctx = bonsai_amd.Context(0)      # raises loudly when the HIP library or the GPU is missing
    yield ctx
    ctx.close()
