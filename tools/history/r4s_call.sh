# headline decode: dynamic chunk claims against static striding (65536 chunks over 8192 waves: exactly eight each), measure build
mkdir -p gpurun_out/r4s
export RANS_AMD_LIB=$GRAFT_REPO_ROOT/ryg_rans_amd/lib/libryg_rans_amd_measure.so
for i in 1 2 3; do
  for v in dyn static; do
    if [ $v = static ]; then export RANS_AMD_STATIC_SCHED=1; else unset RANS_AMD_STATIC_SCHED; fi
    python bench.py --measure --no-configs --no-cpu-baseline > gpurun_out/r4s/$v$i.json 2> gpurun_out/r4s/$v$i.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r4s/$v$i.json"))
print("$v $i kernel %.4f frac %.4f span %s probe min %.4f max %.4f exact %s" % (d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], d["roofline"].get("wave_span_ms_avg"), d["placement"]["probe_ms_min"], d["placement"]["probe_ms_max"], d["bit_exact_roundtrip"]))
PY
  done
done
