cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python3 tests/chip_fuzz.py 600 5000 2>&1 | tail -3 > gpurun_out/r6_chip_fuzz.txt
cat gpurun_out/r6_chip_fuzz.txt
