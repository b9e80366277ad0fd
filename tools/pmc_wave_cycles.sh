# Where the wave-cycles of the headline's kernels go (MI355X_MICROARCH.md: SQ_WAIT_ANY = parked at s_waitcnt / a barrier, SQ_WAIT_INST_ANY = issue stalls,
# SQ_ACTIVE_INST_ANY = issuing; the three are disjoint and add up to SQ_WAVE_CYCLES).  One proof at a time, every counter group in a run of its own
# (--pmc only beside --kernel-trace).  Writes gpurun_out/r5bpmc/wave_cycles.json.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5bpmc
mkdir -p $O
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-guest-flow --no-aggregate --inflight 1"
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $O/g$i -o g --output-format csv -- $B > $O/g${i}_bench.json 2> $O/g${i}_err.txt
done
python3 - <<'PY'
import csv, glob, json, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r5bpmc"
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(O + "/g*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": calls[k] += 1
out = {}
for k, c in tot.items():
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc <= 0: continue
    out[k] = {"launches": calls[k], "wave_cycles": wc}
    for n, v in c.items():
        if n != "SQ_WAVE_CYCLES": out[k][n] = v
    for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
        if n in c: out[k]["share_" + n] = round(c[n] / wc, 4)
top = dict(sorted(out.items(), key=lambda kv: -kv[1]["wave_cycles"])[:12])
json.dump(top, open(O + "/wave_cycles.json", "w"), indent=1)
for k, v in top.items():
    print(k[:50], {n: v[n] for n in v if n.startswith("share_")})
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
