#!/bin/bash
timeout -k 10 900 python tools/lab_place2.py 2>&1 | grep -v amdgpu | tail -8
