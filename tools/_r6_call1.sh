cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_merkle_stress.py -x -q -m gpu > gpurun_out/r6_call1_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6_call1_pytest.txt
tail -5 gpurun_out/r6_call1_pytest.txt
FLOW_STRESS_TAG=r6a FLOW_STRESS_ENV=ZKHIP_SELF_CHECK=1 bash tools/flow_stress.sh 40 14
FLOW_STRESS_TAG=r6_early_retry FLOW_STRESS_LIB=test FLOW_STRESS_RETRY=1 FLOW_STRESS_ENV="ZKHIP_SELF_CHECK=1 ZKHIP_TREE_STORE_EARLY=1" bash tools/flow_stress.sh 120 14
