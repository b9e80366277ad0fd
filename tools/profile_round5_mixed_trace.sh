# Round 5 (second session): the mixed guest at frame 2^19 with ONE lane and ONE node pipeline under rocprofv3 --kernel-trace; keeps a COMPACT per-launch
# trace (kernel id, start, end, queue, grid, workgroup) so that the time of each proof (segment / leaf / wrapper / internal) can be split by stage offline.
# Writes gpurun_out/r5mixed19_trace/.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5mixed19_trace
mkdir -p $O /tmp/gm /tmp/om
python3 - <<'PY'
import os, sys
sys.path.insert(0, "tests")
import rv32_model as rv
from test_vm_cpu import chunk_circuit_toml, mixed_chunk_data, mixed_chunk_program
open("/tmp/gm/mixed.elf", "wb").write(rv.elf_bytes(mixed_chunk_program(), data=mixed_chunk_data()))
open("/tmp/gm/stdin.bin", "wb").write((2048).to_bytes(4, "little"))
open("/tmp/gm/openvm.toml", "w").write(chunk_circuit_toml((1, 0, 100, 16, 16)))
PY
export ZKHIP_LANES=1 ZKHIP_AGG_SLOTS=1
./zkvm-prover_amd/prove_cli prove-elf /tmp/gm/mixed.elf /tmp/gm/stdin.bin /tmp/om /tmp/gm/openvm.toml ${FRAME:-20} > $O/warm.json 2> $O/warm_err.txt
rocprofv3 --kernel-trace --stats -d $O/p -o p --output-format csv -- ./zkvm-prover_amd/prove_cli prove-elf /tmp/gm/mixed.elf /tmp/gm/stdin.bin /tmp/om /tmp/gm/openvm.toml ${FRAME:-20} > $O/profiled.json 2> $O/profiled_err.txt
cp $(find $O/p -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r5mixed19_trace"
f = glob.glob(O + "/p/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
print(list(rows[0].keys()))
names = {}
t0 = min(int(r["Start_Timestamp"]) for r in rows)
with open(O + "/trace_compact.csv", "w") as w:
    w.write("kid,start_ns,dur_ns,queue,stream,grid,wg\n")
    for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
        k = names.setdefault(r["Kernel_Name"], len(names))
        w.write("%d,%d,%d,%s,%s,%s,%s\n" % (k, int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Queue_Id", ""), r.get("Stream_Id", ""), r.get("Grid_Size", ""), r.get("Workgroup_Size", "")))
with open(O + "/trace_names.csv", "w") as w:
    for n, k in names.items():
        w.write("%d,%s\n" % (k, n[:160].replace(",", ";")))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
gzip -f $O/trace_compact.csv
cut -c1-1500 $O/profiled.json
