#!/bin/bash
# last GPU check of round 2: the whole GPU suite + smoke on the final code, the default bench line
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
( timeout 200 python -m pytest tests -m gpu -q --timeout 100 2>&1 | tail -4 ) > gpurun_out/r02_final_pytest.log
tail -2 gpurun_out/r02_final_pytest.log
( timeout 60 python __graft_entry__.py smoke 2>&1 | tail -1 ) > gpurun_out/r02_final_smoke.log; cat gpurun_out/r02_final_smoke.log
timeout 120 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; cut -c1-400 gpurun_out/r02_bench.json
