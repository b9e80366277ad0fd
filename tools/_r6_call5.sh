cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r6_call5_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6_call5_pytest.txt
tail -5 gpurun_out/r6_call5_pytest.txt
FLOW_STRESS_TAG=r6_300 FLOW_STRESS_ENV=ZKHIP_SELF_CHECK=1 bash tools/flow_stress.sh 300 14
