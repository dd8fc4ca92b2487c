#!/bin/bash
timeout -k 10 300 python tools/lab_burst.py 2>&1 | grep -v amdgpu | tail -22
