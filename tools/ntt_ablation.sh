#!/bin/bash
# Where does the four-step NTT pass spend its time?  Builds three variants of libzkhip.so with parts of
# k_ntt_pass4_ct compiled out (-DNTT_ABL=1 no butterflies, 2 no tile loads, 3 no tile stores, 4 no LDS rounds, 5 no arithmetic at all; results are
# wrong by construction) and times the 2^22 x 300 LDE with each.  Run on the GPU box:
#   gpurun -- 'bash tools/ntt_ablation.sh'
set -e
cd "$(dirname "$0")/.."
CS=zkvm-prover_amd/csrc
OUT=${TMPDIR:-/tmp}/ntt_abl
mkdir -p "$OUT"
cp zkvm-prover_amd/libzkhip.so "$OUT/libzkhip_orig.so"
trap 'cp "$OUT/libzkhip_orig.so" zkvm-prover_amd/libzkhip.so' EXIT
echo "== full kernel"
python tools/stage_bench.py 22 300 2>&1 | grep -E "iter 2|ntt" | tail -3
for n in 1 2 3 4 5; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DNTT_ABL=$n -c $CS/ntt.hip -o "$OUT/ntt_abl$n.o"
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o zkvm-prover_amd/libzkhip.so $(ls $CS/*.o | grep -v '/ntt.o$') "$OUT/ntt_abl$n.o" -lhiprtc
  echo "== ablation $n"
  python tools/stage_bench.py 22 300 2>&1 | grep -E "iter 2|ntt" | tail -3
done
