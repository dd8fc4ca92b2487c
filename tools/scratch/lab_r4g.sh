#!/bin/bash
SPAL_FUZZ_SEEDS=60 timeout -k 10 900 python -m pytest tests/test_gpu_csc_coo.py -x -q -m gpu -k "randomised_bands" 2>&1 | tail -15
