#!/bin/bash
timeout -k 10 600 python tools/lab_csc_stress.py 6000 2>&1 | grep -v amdgpu | tail -5
