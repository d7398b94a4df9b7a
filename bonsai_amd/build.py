"""Build the gfx950 device library (bonsai_amd/lib/libbonsai_amd.so) with hipcc.

In-tree output so the .so travels with the repo snapshot to the GPU box.
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SRC = os.path.join(PKG, "csrc", "bns_api.hip")
SRC_INFLATE = os.path.join(PKG, "csrc", "bns_inflate.hip")      # BGZF members inflated on the device: a translation unit of its own
DEPS = [os.path.join(PKG, "csrc", f) for f in ("bns_api.hip", "bns_kernels.hip", "bns_kernels.hpp", "bns_device.hpp", "bns_inflate.hip", "bns_inflate.hpp", "bns_inflate_wave.hpp", "bns_ingest.hip", "bns_gzstream.hip")] + \
       [os.path.join(ROOT, "include", "bonsai_amd.h")]
OUT = os.path.join(PKG, "lib", "libbonsai_amd.so")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the gfx950 device library cannot be built")


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_device_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -O3 -shared -fPIC -> bonsai_amd/lib/libbonsai_amd.so"""
    if not force and not stale():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", SRC, SRC_INFLATE, "-o", OUT]
    if os.environ.get("BNS_ABLATION") == "1":          # profiling-only build with classify_kernel ablation switches
        cmd.insert(1, "-DBNS_ABLATION")
    for d in os.environ.get("BNS_EXTRA_DEFINES", "").split():   # profiling-only switches (e.g. BNS_PAD_VALU=64, tools/pad.sh)
        cmd.insert(1, "-D" + d)
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build_device_library(force=True, verbose=True))
