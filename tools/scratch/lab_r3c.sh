#!/bin/bash
timeout -k 10 300 python tools/lab_csc_capture.py 2>&1 | grep -v amdgpu
