#!/bin/bash
# Round-end measurement on the GPU box: the GPU test suite, the bench line (plain and under rocprofv3 --kernel-trace --stats),
# PMC traffic of the dominant kernels, the perf matrix and the criterion-shaped sweep. Outputs land in gpurun_out/final/
# (copied to profiles/ by hand afterwards). usage: tools/final_measure.sh [notests]
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/final"
mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
if [ "${1:-}" != "notests" ]; then
  timeout 2400 python -m pytest tests -q -m gpu --durations=12 > "$O/gpu_pytest.txt" 2>&1; echo "pytest rc=$?" | tee -a "$O/summary.txt"
fi
python bench.py 2>"$O/bench.err" | grep '^{"metric"' > "$O/bench.json"; echo "bench rc=$?" | tee -a "$O/summary.txt"
python tools/bench_brief.py plain < "$O/bench.json"
cd /tmp
rm -rf "$O/prof"
rocprofv3 --kernel-trace --stats -d "$O/prof" -o bench -- python "$R/bench.py" --no-cpu-baseline --no-shard-extras > "$O/bench_prof.log" 2>&1
grep '^{"metric"' "$O/bench_prof.log" > "$O/bench_prof.json"
DB=$(find "$O/prof" -name '*.db' | head -1)
[ -n "$DB" ] && python "$R/tools/rocpd_stats.py" "$DB" > "$O/kernel_stats.txt"
find "$O/prof" -name '*.db' -delete
head -16 "$O/kernel_stats.txt"
bash "$R/tools/pmc_traffic.sh" > "$O/pmc_traffic.txt" 2>&1; cat "$O/pmc_traffic.txt"
cd "$R"
{ python tools/perf_matrix.py; python tools/perf_matrix.py ntt; python tools/perf_matrix.py layouts; python tools/perf_matrix.py polymul; python tools/exp_ntt_rn.py; python tools/exp_msm_midsize.py; python tools/perf_matrix.py midsize; python tools/perf_matrix.py distributions; python tools/perf_matrix.py g2; python tools/perf_matrix.py scalar-ntt; python tools/perf_matrix.py ecntt; python tools/perf_matrix.py precompute; python tools/perf_matrix.py criterion; python tools/perf_matrix.py curves; python tools/ntt_gold_time.py; } 2>/dev/null | grep -v amdgpu.ids > "$O/perf_matrix.txt"
cat "$O/perf_matrix.txt"
tail -20 "$O/gpu_pytest.txt" 2>/dev/null
