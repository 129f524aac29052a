#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -x -k "256_inducing" 2>&1 | tail -15
