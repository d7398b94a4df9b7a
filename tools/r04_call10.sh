#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/r04_ab.sh r04c10 "libbonsai_amd_v2.so libbonsai_amd.so" full
