#!/bin/bash
timeout -k 10 500 python tools/lab_ab1.py "slide=-1" "rows_per_tile=32" "rows_per_tile=64,stream_row_max=1024" "slide=-1" ragged @rounds=3 2>&1 | grep -v amdgpu | cut -c1-330
