set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; mkdir -p $O
( time timeout 2400 python -m pytest tests -q -m gpu -x --durations=20 ) > $O/gpu_pytest.txt 2>&1; echo "pytest rc=$?"
tail -40 $O/gpu_pytest.txt | cut -c1-600
