set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ecntt.py -q 2>&1 | tail -3
timeout 300 python tools/perf_matrix.py ecntt 2>/dev/null | grep -v amdgpu.ids | tee $O/ecntt_lazy.txt
python bench.py 2>$O/bench.err | grep '^{"metric"' > $O/bench.json; echo "bench rc=$?"
python tools/bench_brief.py plain < $O/bench.json
python - <<'PY' $O/bench.json
import json,sys
d=json.load(open(sys.argv[1]))
print("traffic", d["roofline"].get("traffic"), d["roofline"].get("traffic_note"), "| ntt", d["ntt"]["roofline"].get("traffic"))
PY
