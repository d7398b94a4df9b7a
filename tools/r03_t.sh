#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 2400 python -m pytest tests/test_gpu_scale.py tests/test_gpu_properties.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert|^E " | head -30
