# Round-5 kernel stats of the guest flow: one lane and one node pipeline (every launch in sequence), and the default three + three.
# Usage: bash tools/profile_round5_flow.sh [log_frame = 19] [fibonacci iterations = 1400000]
set -x
F=${1:-19}
N=${2:-1400000}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5flow$F
mkdir -p $O /tmp/gf /tmp/o
python3 tools/make_guest_files.py /tmp/gf $N
./zkvm-prover_amd/prove_cli prove-elf /tmp/gf/fib.elf /tmp/gf/stdin.bin /tmp/o - $F > /dev/null 2>&1
export ZKHIP_LANES=1 ZKHIP_AGG_SLOTS=1
rocprofv3 --kernel-trace --stats -d $O/a -o a --output-format csv -- ./zkvm-prover_amd/prove_cli prove-elf /tmp/gf/fib.elf /tmp/gf/stdin.bin /tmp/o - $F > $O/a.json 2> $O/a_err.txt
export ZKHIP_LANES=3 ZKHIP_AGG_SLOTS=3
rocprofv3 --kernel-trace --stats -d $O/b -o b --output-format csv -- ./zkvm-prover_amd/prove_cli prove-elf /tmp/gf/fib.elf /tmp/gf/stdin.bin /tmp/o - $F > $O/b.json 2> $O/b_err.txt
cp $(find $O/a -name "*kernel_stats.csv" | head -1) $O/a_kernel_stats.csv
cp $(find $O/b -name "*kernel_stats.csv" | head -1) $O/b_kernel_stats.csv
F=$F python3 - <<'PY'
import csv,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r5flow"+os.environ["F"]
for tag in "ab":
    f=glob.glob(O+"/%s/**/*kernel_trace.csv"%tag, recursive=True)
    if not f: continue
    rows=list(csv.DictReader(open(f[0])))
    t0=min(int(r["Start_Timestamp"]) for r in rows); t1=max(int(r["End_Timestamp"]) for r in rows)
    busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows)
    # union of busy intervals
    iv=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"])) for r in rows)
    u=0; cs,ce=iv[0]
    for s,e in iv[1:]:
        if s>ce: u+=ce-cs; cs,ce=s,e
        else: ce=max(ce,e)
    u+=ce-cs
    print(tag, "launches", len(rows), "span_ms", (t1-t0)/1e6, "sum_kernel_ms", busy/1e6, "gpu_busy_union_ms", u/1e6)
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/a.json | cut -c1-400; cat $O/b.json | cut -c1-400
