#!/bin/bash
timeout -k 10 500 python tools/lab_ab1.py "slide=-1,slide_on=1" "slide_on=0" "slide=-1,slide_on=1" f32 @rounds=3 2>&1 | grep -v amdgpu | cut -c1-300
