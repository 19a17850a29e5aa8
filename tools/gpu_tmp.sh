#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 python -u -m pytest tests/test_gpu_store.py -m gpu -x -q --timeout 120 --timeout-method thread -p no:cacheprovider 2>&1 | tail -30
