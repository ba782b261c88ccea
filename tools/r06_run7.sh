set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
( time timeout 2400 python -m pytest tests -q -m gpu -x --durations=20 ) > $O/gpu_pytest.txt 2>&1; echo "pytest rc=$?"
tail -42 $O/gpu_pytest.txt | cut -c1-1800
bash tools/exp_ntt_big.sh 2>/dev/null | grep -v amdgpu
cp gpurun_out/ntt_big_parts.txt $O/
