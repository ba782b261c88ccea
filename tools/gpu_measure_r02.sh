# round-2 measurement bundle: bench line, rocprofv3 kernel stats of the same command, PMC traffic, perf matrix
set -x
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out/$1"; mkdir -p "$O"; cd "$R"
python bench.py 2>"$O/bench.err" | grep '^{"metric"' > "$O/bench.json"
python tools/bench_brief.py plain < "$O/bench.json"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/prof"
rocprofv3 --kernel-trace --stats -d "$O/prof" -o bench -- python "$R/bench.py" --no-cpu-baseline > "$O/bench_prof.log" 2>&1
grep '^{"metric"' "$O/bench_prof.log" > "$O/bench_prof.json"
DB=$(find "$O/prof" -name '*.db' | head -1)
[ -n "$DB" ] && python "$R/tools/rocpd_stats.py" "$DB" > "$O/kernel_stats.txt"
find "$O/prof" -name '*.db' -delete
head -24 "$O/kernel_stats.txt"
cd "$R"
bash tools/pmc_traffic.sh > "$O/pmc_traffic.txt" 2>&1
cat "$O/pmc_traffic.txt"
{ python tools/perf_matrix.py; python tools/perf_matrix.py g2; python tools/perf_matrix.py scalar-ntt; } 2>/dev/null | grep -v amdgpu.ids > "$O/perf_matrix.txt"
cat "$O/perf_matrix.txt"
