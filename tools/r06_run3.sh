set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O
timeout 600 python tools/ref_scaling.py 24 2>&1 | grep -v DEBUG | tee $O/ref_scaling.txt
timeout 900 python -m pytest tests/test_gpu_ntt_refspace.py -q -x 2>&1 | tail -15 | cut -c1-3000 | tee $O/refspace.txt
