#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 200 python -u tools/stress.py --cases 160 --seed 77 --big 2>&1 | grep -v amdgpu.ids | tail -8
timeout 120 python -u tools/stress.py --cases 250 --seed 78 2>&1 | grep -v amdgpu.ids | tail -4
