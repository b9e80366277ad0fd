cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out /tmp/eb
O=gpurun_out/r6_call2.txt
: > $O
nproc >> $O; python3 -c "import ctypes; l=ctypes.CDLL('zkvm-prover_amd/libzkhip.so'); print('host_cpus', l.zkhip_host_cpus())" >> $O
g++ -O2 -std=c++17 -pthread -I include tools/exec_bench.cpp -o /tmp/exec_bench -Lzkvm-prover_amd -lzkhip -Wl,-rpath,$PWD/zkvm-prover_amd
python3 tools/make_guest_files.py /tmp/eb 2800000 > /dev/null
python3 - <<'PY'
import sys
sys.path.insert(0,'tools'); sys.path.insert(0,'tests')
import rv32_model as rv
from guest_bench2 import memsum_program
open('/tmp/eb/mem.elf','wb').write(rv.elf_bytes(memsum_program()))
open('/tmp/eb/mem512.in','wb').write((512).to_bytes(4,'little'))
PY
for t in 0 2 4 6 8 12; do echo "== exec_bench fib threads $t" >> $O; /tmp/exec_bench /tmp/eb/fib.elf /tmp/eb/stdin.bin 20 3 $t >> $O 2>&1; done
for t in 0 4 8 12; do echo "== exec_bench mem threads $t" >> $O; /tmp/exec_bench /tmp/eb/mem.elf /tmp/eb/mem512.in 20 3 $t >> $O 2>&1; done
for t in 0 2 4 6; do
  for g in "2800000 20" "512 20 mem" "8192 20 mixed"; do
    echo "== flow $g exec_threads $t" >> $O
    ZKHIP_EXEC_THREADS=$t ZKHIP_LANES=3 python3 tools/guest_bench2.py $g > /dev/null 2>&1
    ZKHIP_EXEC_THREADS=$t ZKHIP_LANES=3 python3 tools/guest_bench2.py $g >> $O 2>&1
  done
done
python -m pytest tests/test_gpu_merkle_stress.py tests/test_gpu_vm2.py -x -q -m gpu > gpurun_out/r6_call2_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6_call2_pytest.txt
tail -5 gpurun_out/r6_call2_pytest.txt
grep -v "^{\"total" $O | head -80
