#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/r04_ab.sh r04c3 "libbonsai_amd_r03.so libbonsai_amd_v1.so libbonsai_amd.so" quick
