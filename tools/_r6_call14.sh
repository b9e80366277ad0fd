cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r6_fuzz.txt
: > $O
echo "== parity_fuzz, every constraint kernel compiled (instances eight at a time): 500 sets, heights 1 .. 256" >> $O
ZKHIP_FORCE_JIT=1 timeout 900 python3 tests/parity_fuzz.py 500 61000 0 8 2>&1 | tail -3 >> $O
echo "== parity_fuzz, compiled + the shared-rows form forced: 120 sets, heights 2^9 .. 2^13" >> $O
ZKHIP_FORCE_JIT=1 ZKHIP_JIT_SHARED=1 timeout 1200 python3 tests/parity_fuzz.py 120 62000 9 13 2>&1 | tail -3 >> $O
echo "== parity_fuzz, shared rows with 2 and 8 waves: 40 sets each" >> $O
ZKHIP_FORCE_JIT=1 ZKHIP_JIT_SHARED=1 ZKHIP_JIT_SHARED_WAVES=2 timeout 600 python3 tests/parity_fuzz.py 40 63000 9 12 2>&1 | tail -2 >> $O
ZKHIP_FORCE_JIT=1 ZKHIP_JIT_SHARED=1 ZKHIP_JIT_SHARED_WAVES=8 timeout 600 python3 tests/parity_fuzz.py 40 64000 9 12 2>&1 | tail -2 >> $O
echo "== chip_fuzz (device trace generators == oracle/tracegen.c): 400 rounds" >> $O
timeout 900 python3 tests/chip_fuzz.py 400 65000 2>&1 | tail -3 >> $O
cat $O
