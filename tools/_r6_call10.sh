cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_vm2.py tests/test_gpu_vm_slice.py -q -m gpu > gpurun_out/r6_call10_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6_call10_pytest.txt
tail -15 gpurun_out/r6_call10_pytest.txt
