#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 40 python tools/time_lanes.py --fmt r64 --ways 2 --encode 2>&1 | grep -v amdgpu.ids
timeout 40 python tools/time_lanes.py --fmt r64 --ways 2 --encode --chunk 1024 2>&1 | grep -v amdgpu.ids
timeout 40 python tools/time_lanes.py --fmt r64 --ways 2 --encode --chunk 256 2>&1 | grep -v amdgpu.ids
