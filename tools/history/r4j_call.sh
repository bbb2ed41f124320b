mkdir -p gpurun_out/r4j
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4j/smoke.log 2>&1; tail -2 gpurun_out/r4j/smoke.log
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r4j/gpu_tests.log 2>&1; tail -6 gpurun_out/r4j/gpu_tests.log
timeout 600 python bench.py > gpurun_out/r4j/bench.json 2> gpurun_out/r4j/bench.err; tail -c 400 gpurun_out/r4j/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r4j/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_avg"], d["roofline"]["traffic"], d["bit_exact_roundtrip"])
for c in d["configs"]:
    print(c["name"][:44], "| dec", c["decode"]["ms_mean"], c["decode"]["frac"], "| enc", c["encode"]["ms_mean"], c["encode"]["frac"], "| encC", c["encode_compact"]["ms_mean"], "| cpu", (c.get("cpu_baseline") or {}).get("value"))
PY
