#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_csr_spmv.py tests/test_gpu_csc_coo.py -x -q -m gpu -k "halves or handoff or csc" 2>&1 | tail -6
