#!/bin/bash
export TMPDIR=/tmp
CWT_TABLES_VERBOSE=1 timeout 600 python tools/lab/mc_cprofile.py 2>&1 | grep -v amdgpu.ids | tail -70
