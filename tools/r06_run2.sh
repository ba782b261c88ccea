set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
( time timeout 2400 python -m pytest tests -q -m gpu -x --durations=15 ) > $O/gpu_pytest.txt 2>&1; echo "pytest rc=$?"
tail -30 $O/gpu_pytest.txt
python bench.py 2>$O/bench.err | grep '^{"metric"' > $O/bench.json; echo "bench rc=$?"; tail -3 $O/bench.err
python tools/bench_brief.py plain < $O/bench.json
python - <<'PY' $O/bench.json
import json,sys
d=json.load(open(sys.argv[1]))
for k in ("config3_shard","config4_shard","collectives"):
    print(k, json.dumps(d.get(k))[:1500])
print("ntt.alu", json.dumps(d["ntt"]["roofline"].get("alu"))[:800])
print("msm.alu", json.dumps(d["roofline"].get("alu"))[:600])
PY
timeout 600 python tools/perf_matrix.py precompute 2>/dev/null | grep -v amdgpu.ids > $O/precompute_perf.txt; cat $O/precompute_perf.txt
bash tools/r06_profile_shards.sh
for q in 16384 32768 65536; do echo "ECNTT_QUADS=$q"; ICICLE_HIP_ECNTT_QUADS=$q timeout 300 python tools/perf_matrix.py ecntt 2>/dev/null | grep -v amdgpu.ids; done > $O/ecntt_quads.txt; cat $O/ecntt_quads.txt
