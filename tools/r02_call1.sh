#!/bin/bash
# round 2, GPU call 1: parity of the new window / group kernel, A/B against round 1, chunk-size sweep, ubench2, trace
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c1; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest.log
B="python bench.py --no-cpu-baseline --no-configs --steps 20 --warmup 5"
pick='import json,sys; d=json.load(sys.stdin); print(d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], d["bit_exact_roundtrip"], d.get("clocks",{}).get("sclk_hz_measured"), d.get("clocks",{}).get("per_simd_clocks_per_round_of_64"))'
for r in 1 2 3; do
  echo "group $(timeout 120 $B 2>/dev/null | python -c "$pick")"
  echo "nogroup $(RANS_AMD_NO_GROUP=1 timeout 120 $B 2>/dev/null | python -c "$pick")"
  echo "r01 $(RANS_AMD_LIB=$PWD/build/libexp_r01.so timeout 120 $B 2>/dev/null | python -c "$pick")"
done > $OUT/ab.log 2>&1
for c in 4096 8192 16384 65536; do
  echo "chunk $c $(timeout 120 $B --chunk $c 2>/dev/null | python -c "$pick")"
done > $OUT/chunks.log 2>&1
for c in 16384 32768; do
  echo "r01 chunk $c $(RANS_AMD_LIB=$PWD/build/libexp_r01.so timeout 120 $B --chunk $c 2>/dev/null | python -c "$pick")"
done >> $OUT/chunks.log 2>&1
RANS_AMD_TRACE=$PWD/$OUT/trace.txt timeout 120 $B --steps 2 --warmup 1 > /dev/null 2>&1
python - <<'PY' > $OUT/trace_summary.txt 2>&1
import numpy as np
t=np.loadtxt("gpurun_out/c1/trace.txt")
st,en,xcc,cyc,rounds=t[:,1],t[:,2],t[:,3],t[:,4],t[:,5]
t0=st.min(); dur=(en.max()-t0)/100.0
print("waves",len(t),"kernel us",dur)
e=(en-t0)/100.0
for q in (0,1,5,25,50,75,95,99,100): print("end pct",q,round(np.percentile(e,q),1))
print("rounds/wave min/mean/max",rounds.min(),rounds.mean(),rounds.max())
print("cycles per round per wave", cyc.sum()/rounds.sum(), "sclk GHz", (cyc/(en-st)).mean()*0.1)
for x in range(8):
    m=xcc==x
    print("xcc",x,"waves",m.sum(),"last end",round(e[m].max(),1),"rounds",rounds[m].sum())
PY
rm -f $OUT/trace.txt
timeout 300 build/ubench2 > $OUT/ubench2.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err
ls -la $OUT
