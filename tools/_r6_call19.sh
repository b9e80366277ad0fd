cd ${GRAFT_REPO_ROOT:-.}
FLOW_STRESS_TAG=r6_early_300 FLOW_STRESS_LIB=test FLOW_STRESS_RETRY=1 FLOW_STRESS_ENV="ZKHIP_SELF_CHECK=1 ZKHIP_TREE_STORE_EARLY=1" bash tools/flow_stress.sh 300 14
python3 - <<'PY'
import json,subprocess,os
# is the test library really the one in use under LD_PRELOAD?  one flow's own line says so
os.makedirs('/tmp/pl',exist_ok=True)
subprocess.check_call(['python3','tools/make_guest_files.py','/tmp/pl','300000'],stdout=subprocess.DEVNULL)
open('/tmp/pl/stdin.bin','wb').write((30000).to_bytes(4,'little'))
env=dict(os.environ, LD_PRELOAD=os.getcwd()+'/zkvm-prover_amd/libzkhip_test.so', ZKHIP_TREE_STORE_EARLY='1')
r=subprocess.run(['./zkvm-prover_amd/prove_cli','prove-elf','/tmp/pl/fib.elf','/tmp/pl/stdin.bin','/tmp/pl','-','14'],env=env,capture_output=True,text=True)
l=json.loads(r.stdout.strip().splitlines()[-1]); print('library_has_test_kernels under LD_PRELOAD:', l.get('library_has_test_kernels'), 'verified', l.get('verified'))
PY
