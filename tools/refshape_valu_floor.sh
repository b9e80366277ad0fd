# The VALU floor of the reference-shaped 17-AIR proof: SQ_INSTS_VALU over a run of tools/refshape_bench.py with one
# proof in flight and 10 timed proofs (18 proofs in all: the verified one, the profiled one, five best-of runs, one warm-up, ten timed; the
# key generation's kernels are in the sum, a few per cent), priced like the headline's: x 2.823 cycles per wave-instruction / (1024 SIMDs x 2.4 GHz).  Writes
# gpurun_out/r5refshape/.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5refshape
mkdir -p $O
python3 tools/refshape_bench.py 0 3 12 > $O/refshape.txt 2> $O/refshape_err.txt
rocprofv3 --pmc SQ_INSTS_VALU -d $O/v -o v --output-format csv -- python3 tools/refshape_bench.py 0 1 10 > $O/under_pmc.txt 2> $O/under_pmc_err.txt
python3 - <<'PY'
import csv, glob, json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r5refshape"
tot, per = 0.0, {}
for f in glob.glob(O + "/v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "SQ_INSTS_VALU":
            v = float(r["Counter_Value"]); tot += v
            k = r["Kernel_Name"].split("(")[0]; per[k] = per.get(k, 0.0) + v
n_proofs = 18
floor_ms = tot / n_proofs * 2.823 / (1024 * 2.4e9) * 1e3
top = sorted(per.items(), key=lambda kv: -kv[1])[:8]
out = {"valu_wave_instr_per_proof": round(tot / n_proofs), "proofs_in_the_counted_run": n_proofs, "cycles_per_wave_instr_model": 2.823,
       "floor_ms_per_proof": round(floor_ms, 2), "top_kernels_share": {k: round(v / tot, 3) for k, v in top}}
open(O + "/valu_floor.json", "w").write(json.dumps(out, indent=1))
print(json.dumps(out))
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
tail -3 $O/refshape.txt
