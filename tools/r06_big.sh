cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ntt_fullsize.py -q -k "four_step" 2>&1 | tail -3
python tools/perf_matrix.py ntt 2>/dev/null | grep -v amdgpu | grep "2^25\|2^26\|2^27\|2^24"
