cd ${GRAFT_REPO_ROOT:-.}
python3 -c "
import json, bench
print(json.dumps(bench.guest_flow_devices(1))[:1500])
print(json.dumps(bench.guest_flow_devices(2))[:600])
"
