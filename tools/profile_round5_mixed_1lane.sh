# Round 5: the mixed guest at frame 2^19 with ONE lane and ONE node pipeline under rocprofv3 --kernel-trace --stats: every launch in sequence, so
# a kernel's duration is its own (no sharing with other streams).  Writes gpurun_out/r5mixed19_1lane/.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5mixed19_1lane
mkdir -p $O /tmp/gm /tmp/om
python3 - <<'PY'
import os, sys
sys.path.insert(0, "tests")
import rv32_model as rv
from test_vm_cpu import chunk_circuit_toml, mixed_chunk_data, mixed_chunk_program
open("/tmp/gm/mixed.elf", "wb").write(rv.elf_bytes(mixed_chunk_program(), data=mixed_chunk_data()))
open("/tmp/gm/stdin.bin", "wb").write((4096).to_bytes(4, "little"))
open("/tmp/gm/openvm.toml", "w").write(chunk_circuit_toml((1, 0, 100, 16, 16)))
PY
export ZKHIP_LANES=1 ZKHIP_AGG_SLOTS=1
./zkvm-prover_amd/prove_cli prove-elf /tmp/gm/mixed.elf /tmp/gm/stdin.bin /tmp/om /tmp/gm/openvm.toml 19 > $O/warm.json 2> $O/warm_err.txt
rocprofv3 --kernel-trace --stats -d $O/p -o p --output-format csv -- ./zkvm-prover_amd/prove_cli prove-elf /tmp/gm/mixed.elf /tmp/gm/stdin.bin /tmp/om /tmp/gm/openvm.toml 19 > $O/profiled.json 2> $O/profiled_err.txt
cp $(find $O/p -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r5mixed19_1lane"
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
    # by grid size class: launches whose grid cannot fill 256 CUs
    small = [r for r in rows if int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])) < 256]
    open(O + "/gpu_busy.txt", "w").write("launches %d span_ms %.1f sum_kernel_ms %.1f gpu_busy_union_ms %.1f; launches of fewer than 256 workgroups: %d, %.1f ms\n" % (
        len(rows), (t1 - t0) / 1e6, busy / 1e6, u / 1e6, len(small), sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in small) / 1e6))
    print(open(O + "/gpu_busy.txt").read())
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cut -c1-600 $O/profiled.json
