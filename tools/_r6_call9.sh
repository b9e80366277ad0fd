cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r6_call9_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6_call9_pytest.txt
tail -25 gpurun_out/r6_call9_pytest.txt
