#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 150 python -u tools/host_rate.py 2>&1 | grep -v amdgpu.ids
