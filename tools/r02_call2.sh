#!/bin/bash
# round 2, GPU call 2: pipelined chunk hand-over (k_decode_word64) vs serial vs round 1; chunk sizes; launch spans; sizes
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c3; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest.log
B="python bench.py --no-cpu-baseline --no-configs --steps 20 --warmup 5"
pick='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; print(d["ms_per_step"], r["kernel_ms_avg"], r.get("wave_span_ms_avg"), r["frac"], d["bit_exact_roundtrip"], r["kernel"], d.get("clocks",{}).get("sclk_hz_measured"), d.get("clocks",{}).get("per_simd_clocks_per_round_of_64"), d["config"]["compressed_bytes_per_symbol"])'
for r in 1 2 3; do
  echo "pipe $(timeout 120 $B 2>/dev/null | python -c "$pick")"
  echo "nopipe $(RANS_AMD_NO_PIPE=1 timeout 120 $B 2>/dev/null | python -c "$pick")"
  echo "r01 $(RANS_AMD_LIB=$PWD/build/libexp_r01.so timeout 120 $B 2>/dev/null | python -c "$pick")"
done > $OUT/ab.log 2>&1
for c in 4096 8192 16384 65536; do
  echo "pipe chunk $c $(timeout 120 $B --chunk $c 2>/dev/null | python -c "$pick")"
done > $OUT/chunks.log 2>&1
for n in 28 29 31; do
  echo "pipe log2n $n $(timeout 200 $B --log2n $n 2>/dev/null | python -c "$pick")"
done > $OUT/sizes.log 2>&1
echo "pipe steps100 $(timeout 200 $B --steps 100 2>/dev/null | python -c "$pick")" >> $OUT/sizes.log
RANS_AMD_TRACE=$PWD/$OUT/trace.txt timeout 120 $B --steps 2 --warmup 1 > /dev/null 2>&1
python - <<'PY' > $OUT/trace_summary.txt 2>&1
import numpy as np
t=np.loadtxt("gpurun_out/c3/trace.txt")
st,en,xcc,cyc,rounds=t[:,1],t[:,2],t[:,3],t[:,4],t[:,5]
t0=st.min(); dur=(en.max()-t0)/100.0
print("waves",len(t),"kernel us",dur)
e=(en-t0)/100.0
for q in (0,1,5,25,50,75,95,99,100): print("end pct",q,round(np.percentile(e,q),1))
print("start pct 100", round(((st-t0)/100.0).max(),1))
print("rounds/wave min/mean/max",rounds.min(),rounds.mean(),rounds.max())
print("cycles per round per wave", cyc.sum()/rounds.sum(), "sclk GHz", (cyc/(en-st)).mean()*0.1)
PY
rm -f $OUT/trace.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err
ls -la $OUT
