#!/bin/bash
# parse-only throughput on the box's host cores (uses the FASTQ tools/cli_bench.py leaves in /tmp/clibench)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
g++ -O2 -std=c++17 -Ibonsai_amd/csrc/host -Iinclude tools/micro/host_parse_bench.cpp bonsai_amd/csrc/host/bns_host.o -o /tmp/host_parse_bench \
    -Lbonsai_amd/lib -lbonsai_amd -lz -lpthread -Wl,-rpath,$PWD/bonsai_amd/lib -Wl,-rpath,/opt/rocm/lib || exit 1
[ -f /tmp/clibench/r.fq ] || python tools/cli_bench.py 4000000 > /dev/null 2>&1
for args in "" "" "1048576" "16777216"; do echo "block bytes: ${args:-default}"; /tmp/host_parse_bench /tmp/clibench/r.fq $args; done
