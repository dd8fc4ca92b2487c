#!/bin/bash
timeout -k 10 800 python tools/lab_coo_stress.py 150 2>&1 | grep -v amdgpu | tail -4
