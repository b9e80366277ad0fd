cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_config_forms.py -x -q -m gpu -k "shared or fused" > gpurun_out/r6_call7_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6_call7_pytest.txt
tail -5 gpurun_out/r6_call7_pytest.txt
bash tools/profile_round6.sh > gpurun_out/r6_call7_profile.log 2>&1
tail -40 gpurun_out/r6_call7_profile.log | cut -c1-400
# the shared-rows form's L2 misses, for the write-up
export TMPDIR=/tmp
ZKHIP_JIT_SHARED=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r6prof/fs -o f --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-guest-flow --no-aggregate --inflight 1 > /dev/null 2> gpurun_out/r6prof/fs_err.txt
python3 - <<'PY'
import csv, glob
tot = 0; n = 0
for f in glob.glob("gpurun_out/r6prof/fs/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("quot_jit") and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n += 1
open("gpurun_out/r6prof/quot_jit_shared_fetch.txt", "w").write("quot_jit, shared-rows form, one proof: launches %d, FETCH_SIZE %.0f KiB (x 2 for full-line reads = %.2f GB)\n" % (n, tot, 2 * tot * 1024 / 1e9))
print(open("gpurun_out/r6prof/quot_jit_shared_fetch.txt").read())
PY
find gpurun_out/r6prof -name "*counter_collection.csv" -delete; find gpurun_out/r6prof -name "*kernel_trace.csv" -delete; find gpurun_out/r6prof -name "*agent_info.csv" -delete
