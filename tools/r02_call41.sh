#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c41; mkdir -p $OUT
( timeout 240 python -m pytest tests -m gpu -q -x --timeout 90 -k "lane or layout or stress or scale or chunked or two_way" 2>&1 | tail -6 ) > $OUT/pytest.log
tail -3 $OUT/pytest.log
for a in "--fmt r64 --ways 2 --chunk 4096" "--fmt r64 --ways 2 --chunk 2048" "--fmt r64 --ways 2 --chunk 1024"; do
  timeout 40 python tools/time_lanes.py $a --encode 2>&1 | grep -v amdgpu.ids
done
