set -x
mkdir -p gpurun_out/r02c
timeout 900 python -m pytest tests/test_gpu_ntt.py -m gpu -q -x --durations=5 > gpurun_out/r02c/pytest_ntt.txt 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/r02c/pytest_ntt.txt
timeout 900 python bench.py 2> gpurun_out/r02c/bench.err | grep '^{"metric"' > gpurun_out/r02c/bench.json
python tools/bench_brief.py plain < gpurun_out/r02c/bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02c/bench.json"))
print(json.dumps({k:d[k] for k in ("value","ms_per_step")}))
print(json.dumps(d["roofline"],indent=0)[:1500])
print(json.dumps(d.get("cpu_baseline"),indent=0))
print(json.dumps(d["ntt"].get("cpu_baseline"),indent=0)); print(d["ntt"]["value"], d["ntt"]["roofline"]["frac"])
PY
tail -5 gpurun_out/r02c/bench.err
