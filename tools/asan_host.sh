#!/bin/bash
# The host library under AddressSanitizer + UBSan (round 5; the round-4 advisor found pgzip's overflow this way): builds
# the host translation units / pgzip.cpp with -fsanitize=address,undefined into /tmp/asan/libbns_host.so, swaps it in for the CPU tier's host tests
# (tests/test_host.py, tests/test_pack.py) and puts the normal build back.  libstdc++ is preloaded beside libasan: python does not
# link it, and ASan's __cxa_throw interceptor needs it at start-up (every reader error is an exception).
set -eu
cd "$(dirname "$0")/.."
H=bonsai_amd/csrc/host
mkdir -p /tmp/asan
F="-O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer"
OBJS=""
for u in bns_host bns_reader bns_chunks bns_text_pipeline bns_bgzf_pipeline bns_gz_pipeline bns_dataset pgzip; do g++ $F -c $H/$u.cpp -o /tmp/asan/$u.o & OBJS="$OBJS /tmp/asan/$u.o"; done
wait
g++ $F -shared $H/bns_host_capi.cpp $OBJS -o /tmp/asan/libbns_host.so -Lbonsai_amd/lib -lbonsai_amd -lz -ldl -lpthread \
    -Wl,-rpath,$PWD/bonsai_amd/lib -Wl,-rpath,/opt/rocm/lib
cp bonsai_amd/lib/libbns_host.so /tmp/asan/libbns_host.orig.so
trap 'cp /tmp/asan/libbns_host.orig.so bonsai_amd/lib/libbns_host.so' EXIT
cp /tmp/asan/libbns_host.so bonsai_amd/lib/libbns_host.so
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)" ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_host.py tests/test_pack.py -x -q -m "not gpu" -p no:cacheprovider
