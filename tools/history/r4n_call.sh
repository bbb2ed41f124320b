# placement probe with the extension: five fresh processes on one box
mkdir -p gpurun_out/r4n
for i in 1 2 3 4 5; do
  python bench.py --no-configs --no-cpu-baseline > gpurun_out/r4n/b$i.json 2> gpurun_out/r4n/b$i.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r4n/b$i.json")); p=d["placement"]
print("run $i kernel %.4f frac %.4f cand %s chosen %s min %.4f max %.4f first %.4f" % (d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], p["candidates"], p["chosen"], p["probe_ms_min"], p["probe_ms_max"], p["probe_ms_first_pair"]))
PY
done
