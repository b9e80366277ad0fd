# Round 5: the chunk-like mixed guest under the reference's chunk-circuit configuration (tools/guest_bench2.py mixed): the run the bench line
# reports, then the same flow under rocprofv3 --kernel-trace --stats (three lanes, three node pipelines).  prove_cli is the program behind `--`
# (no launcher in between).  Writes gpurun_out/r5mixed19/.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5mixed19
mkdir -p $O /tmp/gm /tmp/om
export ZKHIP_LANES=3
python3 tools/guest_bench2.py 8192 ${FRAME:-20} mixed > $O/run1.json 2> $O/run1_err.txt
python3 tools/guest_bench2.py 8192 ${FRAME:-20} mixed > $O/run2.json 2> $O/run2_err.txt
python3 - <<'PY'
import os, sys
sys.path.insert(0, "tests")
import rv32_model as rv
from test_vm_cpu import chunk_circuit_toml, mixed_chunk_data, mixed_chunk_program
open("/tmp/gm/mixed.elf", "wb").write(rv.elf_bytes(mixed_chunk_program(), data=mixed_chunk_data()))
open("/tmp/gm/stdin.bin", "wb").write((8192).to_bytes(4, "little"))
open("/tmp/gm/openvm.toml", "w").write(chunk_circuit_toml((1, 0, 100, 16, 16)))
PY
rocprofv3 --kernel-trace --stats -d $O/p -o p --output-format csv -- ./zkvm-prover_amd/prove_cli prove-elf /tmp/gm/mixed.elf /tmp/gm/stdin.bin /tmp/om /tmp/gm/openvm.toml ${FRAME:-20} > $O/profiled.json 2> $O/profiled_err.txt
cp $(find $O/p -name "*kernel_stats.csv" | head -1) $O/mixed_flow_kernel_stats.csv
python3 - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r5mixed19"
f = glob.glob(O + "/p/**/*kernel_trace.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
    u = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: u += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    u += ce - cs
    open(O + "/gpu_busy.txt", "w").write("launches %d span_ms %.1f sum_kernel_ms %.1f gpu_busy_union_ms %.1f\n" % (len(rows), (t1 - t0) / 1e6, busy / 1e6, u / 1e6))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/run2.json | cut -c1-1500; cat $O/gpu_busy.txt; head -12 $O/mixed_flow_kernel_stats.csv | cut -c1-160
