set -x
mkdir -p gpurun_out/$1
timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/$1/pytest_full.txt 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/$1/pytest_full.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
