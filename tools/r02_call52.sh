#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
( timeout 200 python -m pytest tests -m gpu -q -x --timeout 100 2>&1 | tail -4 ) > gpurun_out/c52_pytest.log
tail -2 gpurun_out/c52_pytest.log
timeout 100 python -u tools/host_rate.py 2>&1 | grep -v amdgpu.ids | head -6
