set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
{ nproc; free -g; df -h /dev/shm /tmp; } > $O/box.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_msm.py -q -x -k "precompute or table" > $O/precompute_pytest.txt 2>&1; echo "precompute pytest rc=$?"
tail -3 $O/precompute_pytest.txt
timeout 600 python tools/perf_matrix.py precompute 2>/dev/null | grep -v amdgpu.ids > $O/precompute_perf.txt; cat $O/precompute_perf.txt
( time timeout 2400 python -m pytest tests -q -m gpu -x --durations=25 ) > $O/gpu_pytest.txt 2>&1; echo "pytest rc=$?"
tail -40 $O/gpu_pytest.txt
