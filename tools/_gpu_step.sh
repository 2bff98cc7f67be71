set -e -o pipefail
mkdir -p gpurun_out
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/s2_bench_b.json 2> gpurun_out/s2_bench_b.err || { tail -20 gpurun_out/s2_bench_b.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/s2_bench_b.json"))
print(round(d["ms_per_step"],2)); print(json.dumps(d["configs"], indent=1))
PY
