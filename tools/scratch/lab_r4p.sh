#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_csr_spmv.py -x -q -m gpu -k "panels or wide_bands" 2>&1 | tail -8
