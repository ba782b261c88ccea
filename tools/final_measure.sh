#!/bin/bash
# Round-end measurement on the GPU box: bench line (plain and under rocprofv3 --kernel-trace --stats), perf matrix.
# Outputs land in gpurun_out/ (copied to profiles/ by hand afterwards).
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out"
mkdir -p "$O"
cd "$R"
python bench.py 2>"$O/bench.err" | grep '^{"metric"' > "$O/bench.json"
python tools/bench_brief.py plain < "$O/bench.json"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/prof"
rocprofv3 --kernel-trace --stats -d "$O/prof" -o bench -- python "$R/bench.py" --no-cpu-baseline > "$O/bench_prof.log" 2>&1
grep '^{"metric"' "$O/bench_prof.log" > "$O/bench_prof.json"
DB=$(find "$O/prof" -name '*.db' | head -1)
[ -n "$DB" ] && python "$R/tools/rocpd_stats.py" "$DB" > "$O/kernel_stats.txt"
find "$O/prof" -name '*.db' -delete
head -16 "$O/kernel_stats.txt"
cd "$R"
{ python tools/perf_matrix.py; python tools/perf_matrix.py g2; python tools/perf_matrix.py scalar-ntt; python tools/perf_matrix.py host; } 2>/dev/null | grep -v amdgpu.ids > "$O/perf_matrix.txt"
cat "$O/perf_matrix.txt"
