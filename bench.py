#!/usr/bin/env python3
"""bench.py -- chunk STARK proofs/sec on MI355X (BASELINE.json metric).

A "step" is ONE full proof (trace commit -> LDE -> Merkle-Poseidon2 -> constraint/quotient ->
FRI fold loop -> queries) of the synthetic chunk-shaped workload of SURVEY.md 8(d) cfg #4:
a 2^22-row x 300-column degree-3 AIR (A) plus a 2^22 x 2 Fibonacci AIR (B), with the reference's
parameters (crates/circuits/chunk-circuit/openvm.toml:1-6: blow-up 2, 100 queries, PoW 16+16).
Traces are synthetic and already resident in HBM when the timed region starts.

N GPUs = N independent proof streams (segments shard across GPUs); every finished proof's 32-byte
trace commitment is all-gathered and its bytes gathered on rank 0 over RCCL -- posted ASYNCHRONOUSLY on a side stream
(zkvm-prover_amd/shard.py AsyncExchange), so ranks never run in lockstep; the timed region ends when every proof has
been proven AND exchanged.  `python bench.py --gpus N` without a launcher starts the N ranks itself (one child process
per GPU, before the parent touches the GPU); under `torch.distributed.run` it uses the ranks it is given.

Prints ONE JSON line (see the task contract) including `roofline` for the dominant kernel and
`cpu_baseline` (the optimised CPU prover oracle/fast proving the full instance of pipeline 0 on the host cores,
measured; rank 0, N=1; its proof bytes are compared with the GPU's).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-rows", type=int, default=22)
    ap.add_argument("--width", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aggregate", action="store_true",
                    help="skip folding the gathered proofs of the last step to one root on rank 0 (`aggregate`, after the timed region)")
    ap.add_argument("--no-guest-flow", action="store_true",
                    help="skip the ELF -> root proof run reported next to the headline (`guest_flow`, N = 1 only)")
    ap.add_argument("--commit-parts", type=int, default=-1,
                    help="pipelined trace commit: column blocks per commit (default 0 = off: it gains ~1 ms on one proof alone and loses 4 ms with three in flight, DESIGN.md 5)")
    ap.add_argument("--inflight", type=int, default=3,
                    help="independent proofs in flight per GPU, each on its own HIP stream (segments of a chunk are "
                         "independent): memory-bound stages of one overlap the VALU-bound hashing of the other")
    return ap.parse_args()


def visible_gpu_count():
    """GPUs this process would see, WITHOUT touching the HIP runtime (a launcher that has initialised the GPU must not
    fork+exec its ranks on this pool): KFD topology nodes that have SIMDs (CPU nodes have simd_count 0), cut down by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set.  ZKHIP_BENCH_DEVICE_COUNT overrides (tests)."""
    forced = os.environ.get("ZKHIP_BENCH_DEVICE_COUNT")
    if forced is not None:
        return int(forced)
    base = "/sys/class/kfd/kfd/topology/nodes"
    n = 0
    try:
        for node in os.listdir(base):
            try:
                with open(os.path.join(base, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                continue
    except OSError:
        return 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher: start one child per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    in its environment, as torch.distributed.run would), relay rank 0's JSON line, exit with the first non-zero code.
    The parent never imports torch and never calls into HIP (devices are counted from the KFD topology in sysfs), so the
    fork+exec of the ranks happens in a process that has not initialised the GPU."""
    import socket
    import subprocess

    n = args.gpus
    plumbing = os.environ.get("ZKHIP_BENCH_PLUMBING_ONLY") == "1"
    if not plumbing and os.environ.get("ZKHIP_BENCH_DRYRUN_1GPU") != "1":
        have = visible_gpu_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d but %d GPU(s) visible (ZKHIP_BENCH_DRYRUN_1GPU=1 runs all ranks on one "
                             "GPU as a plumbing check)\n" % (n, have))
            return 2
    assert "torch" not in sys.modules, "the launcher must not load torch / HIP before it starts its ranks"
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        for p in procs:
            code = p.wait()
            if code != 0 and rc == 0:
                rc = code
                for q in procs:  # a failed rank would leave the others waiting in a collective
                    if q.poll() is None:
                        q.terminate()
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    return rc


class StubProver:
    """ZKHIP_BENCH_PLUMBING_ONLY=1 (CPU test of the launcher / exchange / JSON plumbing; no GPU, no proving, no
    measurement): stands in for the device pipelines with fixed-size byte strings."""

    proof_size = 4096
    workspace_bytes = 0

    def __init__(self, rank):
        self.rank = rank

    def launch(self, i):
        pass

    def collect(self, i):
        import hashlib

        seed = hashlib.sha256(b"%d/%d" % (self.rank, i)).digest()
        return (seed * (self.proof_size // len(seed)))[:self.proof_size]


def cpu_baseline(args, params, airs, host_traces, pvs, gpu_proof):
    """The optimised CPU prover (oracle/fast: packed Montgomery AVX-512 / AVX2 Poseidon2 and NTTs, batch inversions,
    OpenMP over every stage; bit-exact against oracle/stark.c, tests/test_fast_oracle_cpu.py) proving the SAME instance
    as GPU pipeline 0 at FULL size on all host cores, measured -- no extrapolation.  Its proof must equal the GPU's."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ora

    cores = ora.effective_cpus()  # min(visible CPUs, cgroup quota); oracle_lib has set OMP_NUM_THREADS to it
    threads = int(os.environ.get("OMP_NUM_THREADS", cores))
    lanes = ora.fast_lib().fast_vector_lanes()
    inst = [dict(a, trace=t, pvs=np.ascontiguousarray(pv, dtype=np.uint32)) for a, t, pv in zip(airs, host_traces, pvs)]
    cap = len(gpu_proof) // 4 + 16
    ora.fast_lib().fast_warmup(max(a["log_height"] for a in airs) + params[0])  # twiddle tables: setup, like the GPU's keygen
    # first proof = warm-up (its workspace arena is mapped and page-faulted once, like the GPU key's workspace at keygen);
    # the second, steady-state proof is the one timed -- the GPU side is timed after warm-up too
    t0 = time.time()
    ora.fast_stark_prove(params, inst, cap_words=cap)
    dt_first = time.time() - t0
    t0 = time.time()
    proof = ora.fast_stark_prove(params, inst, cap_words=cap)
    dt = time.time() - t0
    same = proof.tobytes() == bytes(gpu_proof)
    return {"value": round(1.0 / dt, 5), "unit": "proofs/s", "cores": min(cores, threads), "kind": "port",
            "threads": threads, "host_cpus_visible": os.cpu_count(),
            "seconds_per_proof": round(dt, 3), "seconds_first_proof": round(dt_first, 3), "proof_bytes_equal_gpu": bool(same),
            "sample": "2^%d rows, measured: oracle/fast (optimised C restatement: %d-lane %s Montgomery, OpenMP x%d threads = the CPUs "
                      "this container's cgroup grants; the host shows more) "
                      "proved the full bench instance of pipeline 0 in %.2f s (second proof; workspace warm); proof bytes %s the GPU's; not the reference "
                      "Rust binary (unbuildable here)"
                      % (args.log_rows, lanes, "AVX-512" if lanes == 16 else "AVX2", threads, dt,
                         "EQUAL" if same else "DIFFER FROM")}


def _flow_runs(cmd, env):
    """warm-up run + three measured runs of one guest flow; returns (warm-up, the measured run of MEDIAN instructions per second, all three rates)"""
    import subprocess

    def rate(g):
        return g["total_cycles"] / ((g["segment_tracegen_and_proving_ms"] + g["aggregation_setup_wait_ms"] + g["aggregation_ms"]) / 1e3)

    runs = []
    for _ in range(4):
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-300:])
        runs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    measured = sorted(runs[1:], key=rate)
    # segment proofs made twice (FlowOptions::retry_segments) are counted over ALL four runs: a clean line says 0
    measured[1]["segments_retried_in_all_four_runs"] = sum(g.get("segments_retried", 0) for g in runs)
    return runs[0], measured[1], [round(rate(g)) for g in runs[1:]]


def launches_per_shape():
    """kernel launches of one segment proof per shape, from the newest committed per-proof trace of the mixed guest on ONE lane (rocprofv3
    --kernel-trace split into proofs by tools/trace_split_proofs.py; bench.py cannot count launches itself): {chips: launches}, with its source"""
    import re
    try:
        d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        name = sorted((n for n in os.listdir(d) if re.fullmatch(r"round\d+[a-z]?_launches_per_shape\.json", n)), reverse=True)[0]
        with open(os.path.join(d, name)) as f:
            out = json.load(f)
        out["source_file"] = "profiles/" + name
        return out
    except Exception as e:
        return {"note": "no profiles/roundNN_launches_per_shape.json: %r" % (e,)}


def guest_flow_mixed(log_frame=20):
    """A CHUNK-LIKE guest under the reference's chunk-circuit configuration: tools/guest_bench2.py `mixed` -- register
    loops, strided loads, Keccak-f, SHA-256, secp256k1 additions / doublings, modular and 256-bit arithmetic in phases (ratios in the tool's
    output), so that segments land in the lean 22-chip, the 26-chip (base + hashes) and the full 51-chip shapes.  Reported: instructions per
    second from the ELF to the verified root, the reference's own figure (MHz = cycles / 1e6 / seconds of proving, crates/prover/src/prover/
    mod.rs:358-366), segments per shape and milliseconds per segment proof per shape.  Child process before this one touches the GPU; one
    warm-up run, then the median of three."""
    import subprocess

    here = os.path.dirname(os.path.abspath(__file__))
    # (log_frame = 19: at frames of 2^20 instructions every segment of this guest spans a full phase and lands in the 51-chip shape; at 2^19
    # about half of them hold plain and hash phases only -- the 26-chip shape gets a number on the driver line again: VERDICT round 5, 9a)
    cmd = [sys.executable, os.path.join(here, "tools", "guest_bench2.py"), "8192", str(log_frame), "mixed"]
    try:
        first, g, rates = _flow_runs(cmd, dict(os.environ, ZKHIP_LANES="3"))
        secs = (g["segment_tracegen_and_proving_ms"] + g["aggregation_setup_wait_ms"] + g["aggregation_ms"]) / 1e3
        return {"metric": "guest_instructions_per_sec_elf_to_verified_root", "value": round(g["total_cycles"] / secs), "unit": "instr/s",
                "prove_speed_mhz": g.get("prove_speed_mhz"), "instructions": g["total_cycles"], "segments": g["segments"], "aggregation_levels": g["levels"],
                "chips_per_shape": g.get("chips_per_shape"), "segments_per_shape": g.get("segments_per_shape"),
                "ms_per_segment_proof_per_shape": g.get("ms_per_segment_proof_per_shape"), "instructions_per_segment_per_shape": g.get("instructions_per_segment_per_shape"),
                "execution_ms": g["execution_ms"], "segments_ms": g["segment_tracegen_and_proving_ms"], "tree_tail_ms": g["aggregation_setup_wait_ms"] + g["aggregation_ms"],
                "process_wall_s": g["wall_s"], "process_wall_s_first_run_on_this_box": first["wall_s"], "instr_per_s_of_the_three_measured_runs": rates,
                "leaf_circuits_at_setup": g.get("leaf_circuits_at_setup"), "leaf_circuits_on_demand": g.get("leaf_circuits_on_demand"), "verified": g["verified"], "guest": g.get("guest"),
                "segments_retried": g.get("segments_retried"), "segments_retried_in_all_four_runs": g.get("segments_retried_in_all_four_runs"),
                "log_frame": log_frame, "launches_per_segment_proof_per_shape": launches_per_shape(),
                "command": "ZKHIP_LANES=3 python tools/guest_bench2.py 8192 %d mixed" % log_frame}
    except Exception as e:   # a reported extra, never a gate
        return {"value": None, "note": "failed: %r" % (e,)}


def guest_flow(chunk_config=False, memory_bound=False):
    """The path AROUND the headline kernel, reported beside it (never `value`): a Fibonacci guest of 16.8 M instructions through
    `prove_cli prove-elf` (tools/guest_bench2.py) -- segmenting executor, 22 chips per segment as ONE statement, device trace generation,
    segment proofs, aggregation tree on verifier circuits, ONE self-verified root proof, at the reference's FRI parameters, in frames of
    2^20 instructions (round 5: `profiles/round05_frame_sweep.txt` -- 3.3 / 5.6 / 8.2 / 10.5 / 10.7 M instr/s at 2^16 .. 2^20 while the executor
    bound the flow; its second session, with smaller tree nodes and a faster executor: 17.3 at 2^19, 21.6 at 2^20 -- `profiles/round05b_frame20.txt`;
    rounds 3 - 4 measured at 2^17).  The rate depends on the guest's length: after the executor's last instruction the last segment proofs and the
    tree's last three levels take ~270 ms whatever the length (`profiles/round05_guest_length.txt`: 14 / 18.3 M instr/s at 8.4 / 33.6 M
    instructions).  memory_bound: the guest that sweeps a 64 KiB array 512 times (33.7 M instructions) (a load every fourth instruction).  Runs as a
    CHILD process BEFORE this process touches the GPU (a process that has initialised HIP must not start programs); four times: the
    first run pays what a fresh box pays (page-in, key cache), the median of the other three is reported.  chunk_config: the same guest under the reference's
    chunk-circuit openvm.toml (crates/circuits/chunk-circuit/openvm.toml: 51 chips in the full set; a Fibonacci guest's segments carry
    the 22 base chips -- per-proof chip presence)."""
    import subprocess

    here = os.path.dirname(os.path.abspath(__file__))
    cmd = [sys.executable, os.path.join(here, "tools", "guest_bench2.py")] + (["512", "20", "mem"] if memory_bound else ["2800000", "20"] + (["chunk"] if chunk_config else []))
    env = dict(os.environ, ZKHIP_LANES="3")
    try:
        first, g, rates = _flow_runs(cmd, env)
        secs = (g["segment_tracegen_and_proving_ms"] + g["aggregation_setup_wait_ms"] + g["aggregation_ms"]) / 1e3
        return {"metric": "guest_instructions_per_sec_elf_to_verified_root", "value": round(g["total_cycles"] / secs), "unit": "instr/s",
                "instructions": g["total_cycles"], "segments": g["segments"], "aggregation_levels": g["levels"], "lanes": g["lanes"],
                "execution_ms": g["execution_ms"], "segments_ms": g["segment_tracegen_and_proving_ms"],
                "tree_tail_ms": g["aggregation_setup_wait_ms"] + g["aggregation_ms"], "process_wall_s": g["wall_s"],
                "process_wall_s_first_run_on_this_box": first["wall_s"], "instr_per_s_of_the_three_measured_runs": rates, "root_proof_bytes": g["root_proof_bytes"], "verified": g["verified"],
                "chips_per_shape": g.get("chips_per_shape"), "segments_per_shape": g.get("segments_per_shape"),
                "instr_per_s_segments_only": g.get("instr_per_s_segments_only"),
                "instr_per_s_segments_plus_executor_busy_time": g.get("instr_per_s_segments_plus_executor_busy_time"),
                "prove_speed_mhz": g.get("prove_speed_mhz"), "log_frame": g.get("log_frame"), "guest": g.get("guest"),
                "segments_retried": g.get("segments_retried"), "segments_retried_in_all_four_runs": g.get("segments_retried_in_all_four_runs"),
                "command": "ZKHIP_LANES=3 python " + " ".join(["tools/guest_bench2.py"] + cmd[2:])}
    except Exception as e:   # a reported extra, never a gate
        return {"value": None, "note": "failed: %r" % (e,)}


def guest_flow_devices(n, plumbing=False):
    """SURVEY.md 8(e)(ii) beside 8(e)(i): ONE guest task over the N GPUs of the node (`FlowOptions::devices`, ZKHIP_DEVICES=0..N-1): the
    Fibonacci guest with N times the one-GPU flow's instructions (weak, like the headline), three segment lanes and three node pipelines
    per device, all fed by the one parallel executor (a metered pass + record passes, include/zkhip_vm_exec.hpp).  Reported: instructions
    per second from the ELF to the verified root, segment proofs per device, tree nodes per device slot and the executor's share (busy
    time of its slowest stage over the segment phase) -- on eight GPUs the executor is what a single task can become bound by.  The
    reference proves a batch's chunks one after the other (crates/integration/src/testers/batch.rs:97-107).  Runs as a child process of a
    process that has not touched the GPU: the launcher before it starts its ranks, or rank 0 under torch.distributed.run while the other
    ranks wait for its flag file.  One run (bounded by ZKHIP_BENCH_DEVICES_TIMEOUT, 420 s): setup is outside the flow's timers."""
    import subprocess

    devs = ",".join(str(d) for d in range(n))
    here = os.path.dirname(os.path.abspath(__file__))
    iters = 2800000 * n
    command = "ZKHIP_DEVICES=%s ZKHIP_LANES=3 python tools/guest_bench2.py %d 20" % (devs, iters)
    if plumbing:
        return {"plumbing_only": True, "devices": list(range(n)), "value": None, "command": command}
    try:
        # ONE run, bounded: the rate comes from the flow's own timers (segment phase + tree; the lanes' keys are built before they start), so no
        # warm-up run is needed -- and a task over eight devices that has never run on such a node must not hold the bench line for long
        runs = []
        r = subprocess.run([sys.executable, os.path.join(here, "tools", "guest_bench2.py"), str(iters), "20"], capture_output=True, text=True,
                           env=dict(os.environ, ZKHIP_DEVICES=devs, ZKHIP_LANES="3"), timeout=int(os.environ.get("ZKHIP_BENCH_DEVICES_TIMEOUT", "420")))
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-300:])
        runs.append(json.loads(r.stdout.strip().splitlines()[-1]))
        g = runs[-1]
        secs = (g["segment_tracegen_and_proving_ms"] + g["aggregation_setup_wait_ms"] + g["aggregation_ms"]) / 1e3
        lanes = g.get("lanes") or 3
        per_lane = g.get("segments_per_lane") or []
        per_dev = [sum(per_lane[d * lanes:(d + 1) * lanes]) for d in range(n)] if len(per_lane) == n * lanes else None
        threads = max(1, g.get("executor_record_threads") or 1)
        stage = max(g.get("executor_metered_pass_busy_ms") or 0, (g.get("executor_record_passes_busy_ms_sum") or 0) / threads, g.get("executor_memory_tree_busy_ms") or 0)
        return {"metric": "guest_instructions_per_sec_elf_to_verified_root_one_task_over_n_gpus", "value": round(g["total_cycles"] / secs), "unit": "instr/s",
                "devices": list(range(n)), "instructions": g["total_cycles"], "segments": g["segments"], "segments_per_device": per_dev,
                "tree_nodes_per_device_slot": g.get("tree_nodes_per_device_slot"), "aggregation_levels": g["levels"],
                "segments_ms": g["segment_tracegen_and_proving_ms"], "tree_tail_ms": g["aggregation_setup_wait_ms"] + g["aggregation_ms"],
                "feeding_thread_waited_for_executor_ms": g["execution_ms"], "executor_record_threads": g.get("executor_record_threads"),
                "executor_metered_pass_busy_ms": g.get("executor_metered_pass_busy_ms"), "executor_record_passes_busy_ms_sum": g.get("executor_record_passes_busy_ms_sum"),
                "executor_share_of_segment_phase": round(stage / max(1, g["segment_tracegen_and_proving_ms"]), 3),
                "segments_retried": g.get("segments_retried"), "process_wall_s_setup_included": g["wall_s"],
                "verified": g["verified"], "command": command}
    except Exception as e:   # a reported extra, never a gate
        return {"value": None, "devices": list(range(n)), "note": "failed: %r" % (e,), "command": command}


def devices_flow_before_the_ranks_touch_the_gpu(args, plumbing):
    """N > 1: the one-task-over-N-GPUs flow runs ONCE, while no rank uses a GPU: in the launcher (`python bench.py --gpus N`) before it
    starts its ranks -- handed to them in ZKHIP_BENCH_DEVICES_FLOW --, or, under torch.distributed.run, in rank 0 before it loads torch,
    the other ranks waiting for its flag file (named after MASTER_PORT) before they load theirs.  Returns the block (rank 0) or None."""
    if args.no_guest_flow or args.gpus < 2:
        return None
    if "ZKHIP_BENCH_DEVICES_FLOW" in os.environ:
        return json.loads(os.environ["ZKHIP_BENCH_DEVICES_FLOW"]) if os.environ.get("RANK", "0") == "0" else None
    if os.environ.get("ZKHIP_BENCH_DRYRUN_1GPU") == "1":
        return None
    flag = "/tmp/zkhip_bench_devices_%s.json" % os.environ.get("MASTER_PORT", "0")
    if os.environ.get("RANK", "0") == "0":
        block = guest_flow_devices(args.gpus, plumbing)
        with open(flag + ".tmp", "w") as f:
            json.dump(block, f)
        os.replace(flag + ".tmp", flag)
        return block
    t0 = time.time()
    while not os.path.exists(flag) and time.time() - t0 < 600:
        time.sleep(0.2)
    return None


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if not args.no_guest_flow and os.environ.get("ZKHIP_BENCH_DRYRUN_1GPU") != "1":
            os.environ["ZKHIP_BENCH_DEVICES_FLOW"] = json.dumps(guest_flow_devices(args.gpus, os.environ.get("ZKHIP_BENCH_PLUMBING_ONLY") == "1"))
        sys.exit(spawn_ranks(args))
    plumbing = os.environ.get("ZKHIP_BENCH_PLUMBING_ONLY") == "1"
    guest_devices = devices_flow_before_the_ranks_touch_the_gpu(args, plumbing) if "torch" not in sys.modules else None
    guest = guest_chunk = guest_mixed = guest_mixed19 = guest_mem = None
    # (child processes: only before torch / HIP are loaded here, and never under a profiler -- its preloaded library has initialised the GPU
    # before this program starts; the profiling recipes pass --no-cpu-baseline or --no-guest-flow, either of which skips the guest flows)
    if args.gpus == 1 and "WORLD_SIZE" not in os.environ and not plumbing and not args.no_guest_flow and not args.no_cpu_baseline and "torch" not in sys.modules:
        guest = guest_flow()
        guest_chunk = guest_flow(chunk_config=True)
        guest_mixed = guest_flow_mixed()
        guest_mixed19 = guest_flow_mixed(19)
        guest_mem = guest_flow(memory_bound=True)
    import numpy as np
    import torch
    import torch.distributed as dist

    import zkvm_prover_amd as z
    from zkvm_prover_amd import air, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ZKHIP_BENCH_DRYRUN_1GPU=1: exercise the N-rank code path on a box with ONE GPU (every rank on cuda:0, gloo
    # collectives on host tensors).  A plumbing check only -- never a measurement.
    dry = world > 1 and os.environ.get("ZKHIP_BENCH_DRYRUN_1GPU") == "1"
    if plumbing:
        return plumbing_run(args, world, rank, guest_devices)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if dry:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            # RCCL for device tensors, gloo for host tensors in the same group: the exchange (32-byte commitments, ~1 MB
            # proofs) runs over RCCL/xGMI; should RCCL not come up on this node the same exchange runs over gloo on host
            # copies and the JSON line says so (the proofs are independent: the exchange is not on the proving path).
            import datetime

            dist.init_process_group("cpu:gloo,cuda:nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    coll_dev = torch.device("cpu") if dry else dev
    coll_note = None
    if world > 1 and not dry:
        ok_flag = torch.ones(1, device="cpu")
        try:
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError("RCCL all-reduce returned %r" % probe.item())
        except Exception as e:  # noqa: BLE001
            ok_flag[0] = 0
            coll_note = "gloo on host copies (RCCL unavailable on this node: %s)" % (repr(e)[:120])
        dist.all_reduce(ok_flag, op=dist.ReduceOp.MIN)  # gloo: every rank takes the same path
        if int(ok_flag.item()) == 0:
            coll_dev = torch.device("cpu")
            coll_note = coll_note or "gloo on host copies (RCCL unavailable on another rank)"
    n_pipe = max(1, args.inflight)
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_pipe)]
    ctxs = []
    for st in streams:
        with torch.cuda.stream(st):
            ctxs.append(z.Context(dev.index))  # binds the context to this HIP stream
            cp = max(0, args.commit_parts)
            ctxs[-1].set_commit_pipeline(cp)
    ctx = ctxs[0]

    params = z.DEFAULT_PARAMS
    log_n, width = args.log_rows, args.width
    sa_kwargs = dict(width=width, n_free=max(4, width // 5), n_bool=min(16, max(1, width // 20)),
                     n_boundary=min(8, max(1, width // 40)), seed=0)
    sa = air.SyntheticAir(**sa_kwargs)
    fa = air.fibonacci_air()
    airs = [dict(program=sa.program(), log_height=log_n, width=width, n_pvs=sa.n_pvs),
            dict(program=fa.program(), log_height=log_n, width=2, n_pvs=3)]
    # independent instance per rank and per pipeline (different witness seed), same proving-key shape
    pipes = []
    for i, c in enumerate(ctxs):
        with torch.cuda.stream(streams[i]):
            tr, pv = sa.gen_trace(log_n, seed=1000 + rank * n_pipe + i, xp="torch", device=dev)
            d_trace = tr.reshape(-1).contiguous()
            del tr
            c._check(c.lib.zkhip_to_monty(c.h, d_trace.data_ptr(), d_trace.numel()))
            ftr_np, fpv = air.fibonacci_trace(log_n, a0=rank * n_pipe + i, b0=1)
            d_ftrace = c.upload(ftr_np.reshape(-1))
            pipes.append(dict(ctx=c, pk=z.ProvingKey(c, params, airs), traces=[d_trace, d_ftrace], pvs=[pv, fpv]))
    torch.cuda.synchronize()
    pk = pipes[0]["pk"]

    def launch(i):
        p = pipes[i % n_pipe]
        p["pk"].prove_async(p["traces"], p["pvs"])

    def collect(i):
        # waits for proof i and copies it to the host (part of the step)
        return pipes[i % n_pipe]["pk"].fetch()

    # the one exchange step of the sharded path (zkvm-prover_amd/shard.py): 32-byte trace commitments to every
    # rank, proofs to rank 0, over RCCL/xGMI -- posted asynchronously per proof, completed before the clock stops
    xch = shard.AsyncExchange(device=coll_dev)
    xstat = {"wait_s": 0.0, "posted": 0}

    def run(n_steps):
        # keep n_pipe proofs in flight: proof i+n_pipe is enqueued as soon as proof i has been fetched
        last = (0, None)
        xstat["posted"] = 0
        for i in range(min(n_pipe, n_steps)):
            launch(i)
        for i in range(n_steps):
            last = (i, collect(i))
            if i + n_pipe < n_steps:
                launch(i + n_pipe)
            if world > 1:
                xch.post(last[1])
                xstat["posted"] += 1
        if world > 1:
            tw = time.perf_counter()
            commits, proofs = xch.wait()
            xstat["wait_s"] = time.perf_counter() - tw
            assert all(len(c) == world for c in commits) and (rank != 0 or all(len(pr) == world for pr in proofs))
            if rank == 0 and proofs:
                xstat["last_gathered"] = proofs[-1]
        return last

    # setup, not warm-up: the first proof of a proving key grows its context's scratch buffers and loads its code
    # objects; do that once per pipeline so that the W warm-up steps and the K timed steps see steady state whatever
    # W is (with 3 pipelines a warm-up of 2 would otherwise leave one of them cold)
    for i in range(n_pipe):
        launch(i)
        collect(i)
    def barrier():
        # on the backend the exchange uses (dist.barrier() would pick RCCL even when the exchange fell back to gloo)
        t = torch.zeros(1, device=coll_dev)
        dist.all_reduce(t)
        if t.is_cuda:
            torch.cuda.synchronize()

    run(args.warmup)
    for c in ctxs:
        c.profile_reset()
    ctx.profile_enable(True)  # per-kernel HIP events on pipeline 0's stream
    if world > 1:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last_i, last = run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        barrier()
    t1 = time.perf_counter()
    ctx.profile_enable(False)
    stats_timed = ctx.profile_read()
    # per-stage breakdown: two more proofs on pipeline 0 ALONE (outside the timed region), so that each
    # kernel's HIP-event time is its own duration, not its wait behind the other pipeline's kernels
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(2):
        launch(0)
        collect(0)
    ctx.profile_enable(False)
    stats_alone = {k: (v[0], v[1]) for k, v in ctx.profile_read().items()}
    # the exchange alone (outside the timed region): one blocking all-gather + gather of a finished proof
    exchange_alone_ms = None
    if world > 1:
        barrier()
        te = time.perf_counter()
        for _ in range(4):
            shard.exchange(last, device=coll_dev)
        exchange_alone_ms = (time.perf_counter() - te) / 4 * 1e3
    pvs = pipes[last_i % n_pipe]["pvs"]
    last_gathered = xstat.get("last_gathered")
    steps_profiled = len(range(0, args.steps, n_pipe))  # proofs that ran on pipeline 0
    dt = t1 - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    stats = stats_timed

    # every rank checks its own last proof with the host verifier (outside the timed region)
    ok = z.verify(params, airs, pvs, last) == 0
    if world > 1:
        okt = torch.tensor([1 if ok else 0], device=coll_dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())

    if rank == 0:
        n, M = 1 << log_n, 1 << (log_n + 1)
        # dominant kernel: Poseidon2 row sponge over the main LDE (+ the narrow quotient/FRI trees,
        # which share the kernel name but are <1% of its time)
        name = "poseidon2_hash_rows" if "poseidon2_hash_rows" in stats else (max(stats, key=lambda k: stats[k][1]) if stats else None)
        roof = None
        if name:
            launches, total_ms = stats[name]
            per_step_ms = total_ms / steps_profiled
            if name == "poseidon2_hash_rows":
                # algorithmic bytes per proof: read every committed LDE cell once, write one digest per row
                alg = 4 * M * (width + 2) + 32 * M + (4 * M * 16 + 32 * M)
                note = "row sponge, main+quotient commits"
            elif name.startswith("ntt_dif_pass"):
                alg = 8 * M * (width + 2 + 16) // (2 if name.endswith("inv") else 1)
                note = "one read + one write of the transformed matrices"
            else:
                alg = None
                note = ""
            if alg:
                ach = alg / (per_step_ms * 1e-3) / 1e9
                roof = {"bound": "hbm", "kernel": name,
                        "kernel_symbol": {"poseidon2_hash_rows": "zk::k_hash_rows"}.get(name, name),  # name in profiles/*_kernel_stats.csv
                        "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                        "launches_per_step": launches / steps_profiled, "ms_per_step": round(per_step_ms, 3),
                        "algorithmic_bytes_per_step": alg, "note": note}
                # HBM traffic of this kernel from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
                # corrected as MI355X_MICROARCH.md prescribes; bench.py cannot collect counters itself)
                try:
                    # (the NEWEST committed pass: profiles/roundNN[x]_pmc_traffic.json, highest round first -- VERDICT round 5, 9b)
                    import re as _re
                    pmc_file = sorted((n for n in os.listdir(os.path.join(ROOT, "profiles")) if _re.fullmatch(r"round\d+[a-z]?_pmc_traffic\.json", n)), reverse=True)[0]
                    with open(os.path.join(ROOT, "profiles", pmc_file)) as f:
                        pmc = json.load(f)["kernels"]
                    key = {"poseidon2_hash_rows": "zk::k_hash_rows"}.get(name)
                    if key and log_n == 22 and width == 300:
                        roof["traffic"] = round(pmc[key]["hbm_bytes_per_proof_corrected"] / pmc[key]["launches_per_proof"])
                        roof["traffic_note"] = "bytes per launch, profiles/" + pmc_file
                        roof["algorithmic_bytes_per_launch"] = alg // 2
                except Exception:
                    pass
                if name == "poseidon2_hash_rows":
                    perms = M * ((width + 7) // 8 + 1 + 2)
                    rate = perms / (per_step_ms * 1e-3)
                    roof["valu"] = {"perms_per_s": round(rate, 0),
                                    "note": "the kernel is integer-VALU bound (~570 Montgomery products and ~1300 modular add/sub per "
                                            "permutation, 6.1k VALU instructions), not HBM bound; frac above is vs HBM as BASELINE asks"}
                    # ABSOLUTE VALU roofline from the committed counter pass (profiles/round04_pmc_valu.json, tools/pmc_valu3.py):
                    # SQ_INSTS_VALU per launch x cycles per wave-instruction of the kernel's ISA mix (tools/isa_mix.py: multiply-class
                    # share x 4.2 + the rest x 2.2 cycles, costs measured by tools/ubench_valu.hip) / (1024 SIMDs x 2.4 GHz), against
                    # THIS run's HIP-event time per launch -- no rate of the kernel itself enters the floor
                    try:
                        pv_name, pv3, pv_stale, pv_note = valu_counts()
                        kv = pv3["kernels"]["zk::k_hash_rows"]
                        if log_n == 22 and width == 300:
                            ms_launch = per_step_ms / (launches / steps_profiled)
                            roof["valu"].update({
                                "wave_instr_per_launch": kv["valu_wave_instr_per_launch"],
                                "cycles_per_wave_instr_model": kv["isa_mix"]["cycles_per_wave_instruction_model"],
                                "multiply_class_fraction": kv["isa_mix"]["multiply_class_fraction"],
                                "floor_ms_per_launch": kv["valu_roofline_ms_per_launch"],
                                "measured_ms_per_launch": round(ms_launch, 3),
                                "frac_of_valu_peak": round(kv["valu_roofline_ms_per_launch"] / ms_launch, 4),
                                "clock_ghz_under_pmc": kv.get("clock_ghz_from_grbm_gui_active"),
                                "issue_efficiency_at_measured_clock_under_pmc": kv.get("issue_efficiency_at_measured_clock"),
                                "stale": pv_stale, "code_check": pv_note,
                                "source": "profiles/%s (instruction counts and ISA mix of the body whose hash is checked above)" % pv_name})
                            if pv_stale:   # the counts describe another body of the kernel: no fraction from them
                                roof["valu"].pop("frac_of_valu_peak", None)
                    except Exception as e:
                        roof["valu"]["counts_note"] = "failed: %r" % (e,)
        # the memory-side kernels against the same HBM peak (algorithmic bytes / measured time)
        others = {}
        cols_all = width + 2 + 16  # trace columns + quotient-chunk columns that go through the LDE
        alg_of = {
            "ntt_pass_fwd": 8 * M * cols_all,            # 2^b coset transforms: 1 read + 1 write of the LDE
            "ntt_pass_inv": 8 * n * cols_all,
            "reduced_openings": 4 * M * cols_all + 16 * M,
            "open_col_reduce": 4 * n * cols_all,
        }
        for k, alg_b in alg_of.items():
            if k in stats_alone and stats_alone[k][1] > 0:
                ms = stats_alone[k][1] / 2
                others[k] = {"algorithmic_bytes_per_step": alg_b, "ms_per_step": round(ms, 3),
                             "achieved_GBps": round(alg_b / (ms * 1e-3) / 1e9, 1),
                             "frac_of_hbm_peak": round(alg_b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        # whole-proof VALU roofline from the committed counter pass (tools/pmc_valu3.py): one proof's VALU wave-instructions priced at
        # the cycles per instruction of the row-hash kernel's ISA mix on 1024 SIMDs at 2.4 GHz, against the measured time per proof
        valu_roof = None
        if log_n == 22 and width == 300:
            try:   # a reported extra after the timed region: a missing or stale committed profile must not lose the headline line
                pmc_file, pv, pv_stale, pv_note = valu_counts()
                # the totals from the per-kernel entries (round 3's file has no summed keys and the block came out null on the driver's line)
                kern = pv["kernels"]
                total_instr = pv.get("total_valu_wave_instr_per_proof") or sum(k["launches_per_proof"] * k["valu_wave_instr_per_launch"] for k in kern.values())
                cpi = kern["zk::k_hash_rows"]["isa_mix"]["cycles_per_wave_instruction_model"]
                floor_ms = pv.get("total_valu_roofline_ms_per_proof") or total_instr * cpi / (pv["peak"]["simds"] * pv["peak"]["clock_hz"]) * 1e3
                # ... and every kernel priced with ITS OWN instruction mix where one is committed (profiles/roundNN_isa_mix_by_kernel.json:
                # the transform passes and the constraint kernel carry more multiply-class instructions than the sponge -- VERDICT round 5, 9c)
                own_floor_ms = own_note = None
                try:
                    import re as _re2
                    mix_file = sorted((n for n in os.listdir(os.path.join(ROOT, "profiles")) if _re2.fullmatch(r"round\d+[a-z]?_isa_mix_by_kernel\.json", n)), reverse=True)[0]
                    with open(os.path.join(ROOT, "profiles", mix_file)) as f:
                        own = json.load(f)["cycles_per_wave_instruction_model"]
                    cyc = sum(k["launches_per_proof"] * k["valu_wave_instr_per_launch"] * own.get(name, cpi) for name, k in kern.items())
                    own_floor_ms = cyc / (pv["peak"]["simds"] * pv["peak"]["clock_hz"]) * 1e3
                    own_note = "profiles/%s: %s; every other kernel at the sponge's %.3f" % (mix_file, ", ".join("%s %.2f" % (n, c) for n, c in sorted(own.items()) if n in kern), cpi)
                except Exception as e:
                    own_note = "no per-kernel mix: %r" % (e,)
                ms = dt / args.steps * 1e3  # every rank proves `steps` proofs in dt
                valu_roof = {"valu_wave_instr_per_proof": int(total_instr), "cycles_per_wave_instr_model": cpi,
                             "floor_ms_per_proof": round(floor_ms, 2),
                             "ms_per_proof_per_gpu": round(ms, 2),
                             "frac": None if pv_stale else round(floor_ms / ms, 3),
                             "floor_ms_per_proof_each_kernel_at_its_own_mix": None if own_floor_ms is None else round(own_floor_ms, 2),
                             "frac_each_kernel_at_its_own_mix": None if (pv_stale or own_floor_ms is None) else round(own_floor_ms / ms, 3),
                             "own_mix_note": own_note,
                             "stale": pv_stale, "code_check": pv_note,
                             "hash_rows_share_of_valu_instr": round(2 * kern["zk::k_hash_rows"]["valu_wave_instr_per_launch"] / total_instr, 3),
                             "note": "profiles/%s: SQ_INSTS_VALU summed over the kernels of one proof x the model cycles per wave-instruction "
                                     "of the row-hash kernel's ISA mix / (1024 SIMDs x 2.4 GHz)" % pmc_file}
            except Exception as e:
                valu_roof = {"frac": None, "note": "failed: %r" % (e,)}
        # whole-proof HBM roofline (SURVEY.md 8(d) cfg #4: unfused per-stage algorithmic bytes of one proof): the LDE reads
        # the trace and writes the codeword, the row hash and the constraint kernel each read the codeword once, the
        # quotient is written once, the fold loop moves ~2 x 2 x M ext elements.  The proof is integer-VALU bound, so
        # this fraction is small by nature; it is reported because BASELINE.json asks for it.
        Wt = width + 2
        alg_proof = 4 * n * Wt + 4 * M * Wt + 4 * M * Wt + 4 * M * Wt + 16 * M + 2 * 2 * 16 * M
        per_gpu_s = dt / args.steps
        hbm_whole = {"bound": "hbm", "algorithmic_bytes_per_proof": alg_proof, "achieved": round(alg_proof / per_gpu_s / 1e9, 1),
                     "peak": 8000.0, "unit": "GB/s", "frac": round(alg_proof / per_gpu_s / 8e12, 4),
                     "floor_ms_at_6.3TBps": round(alg_proof / 6.3e12 * 1e3, 2)}
        out = {
            "metric": "chunk STARK proofs/sec (2^%d-row trace)" % log_n,
            "value": round(world * args.steps / dt, 4),
            "unit": "proofs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "full STARK proof: 2^%d x %d degree-3 AIR + 2^%d x 2 Fibonacci AIR, blow-up 2, "
                                   "%d queries, PoW %d+%d (SURVEY.md 8(d) cfg #4)"
                                   % (log_n, width, log_n, params[2], params[3], params[4]),
                       "proof_bytes": pk.proof_size, "verified": ok,
                       "resident_key_bytes_per_pipeline": pk.workspace_bytes,
                       "inflight_per_gpu": n_pipe,
                       "commit_pipeline_parts": max(0, args.commit_parts),
                       "exchange": None if world == 1 else {
                           "posted_per_rank": xstat["posted"], "mode": "asynchronous per proof (side stream), completed inside the timed region",
                           "ms_blocked_at_end_of_timed_region": round(xstat["wait_s"] * 1e3, 3),
                           "ms_one_blocking_exchange_alone": round(exchange_alone_ms, 3),
                           "bytes_per_proof": "32 x %d all-gather + %d gather to rank 0" % (world, pk.proof_size),
                           "backend": "gloo (dry run)" if dry else (coll_note or "RCCL (nccl backend), device tensors")},
                       "parallelism": "%d independent proof(s) in flight per GPU (one HIP stream each)" % n_pipe + (", RCCL all-gather of commitments + proof gather" if world > 1 else "")
                                      + (" [DRY RUN: all ranks on one GPU, gloo -- not a measurement]" if dry else "")},
            "roofline": roof,
            "roofline_other_kernels": others,
            "roofline_valu_whole_proof": valu_roof,
            "roofline_hbm_whole_proof": hbm_whole,
            "stage_ms_single_stream": {k: round(v[1] / 2, 3) for k, v in sorted(stats_alone.items(), key=lambda kv: -kv[1][1])},
            "stage_note": "per-kernel HIP-event times of one proof running alone (measured after the timed region); "
                          "`roofline` is from the timed region itself",
        }
        if not args.no_aggregate and not args.no_cpu_baseline:
            # "gather of encoded proofs to rank 0 ... aggregation-tree proving of the gathered proofs runs on rank 0": the
            # proofs the ranks made in the last step, folded to ONE root under one aggregation key (zkvm-prover_amd/aggregate.py: leaf
            # node(s) over <= 4 proofs, wrapped by the internal circuit).  After the timed region; never part of `value`.
            try:
                from zkvm_prover_amd import aggregate

                gathered = [last] if world == 1 else [bytes(b) for b in last_gathered]
                t_a = time.perf_counter()
                agg = aggregate.TreeAggregator(ctx, params, pk.verifying_airs())
                t_b = time.perf_counter()
                root, rpv, levels = agg.aggregate(gathered, [pvs] * len(gathered))
                t_c = time.perf_counter()
                root2, _, _ = agg.aggregate(gathered, [pvs] * len(gathered))
                t_d = time.perf_counter()
                out["aggregate"] = {"proofs_folded": len(gathered), "nodes_per_level": [len(l) for l in levels],
                                    "root_ms": round((t_d - t_c) * 1e3, 1), "root_ms_first": round((t_c - t_b) * 1e3, 1),
                                    "setup_s": round(t_b - t_a, 2), "circuit_build_s": round(agg.build_s, 2), "keygen_s": round(agg.keygen_s, 2),
                                    "leaf_circuit": {"gate_rows": agg.leaf.n_gates, "permutations": agg.leaf.n_perms, "log_heights": agg.leaf.log_heights()[:2]},
                                    "root_proof_bytes": len(root), "root_verified_under_one_key": bool(agg.verify_root(root, rpv)), "deterministic": root == root2,
                                    "note": "witness generation on the host + device traces + proof, per node; every rank proves the same synthetic instance"}
            except Exception as e:  # a reported extra, never a gate
                out["aggregate"] = {"root_ms": None, "note": "failed: %r" % (e,)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                # the instance of pipeline 0, fetched from the device: the CPU proves exactly what the GPU proved
                p0 = pipes[0]
                p0["pk"].prove_async(p0["traces"], p0["pvs"])
                gpu_proof0 = p0["pk"].fetch()
                with torch.cuda.stream(streams[0]):  # the context issues on this stream: clone and conversion must share it
                    host_traces = [p0["ctx"].download(t).reshape(a["width"], -1) for t, a in zip(p0["traces"], airs)]
                out["cpu_baseline"] = cpu_baseline(args, params, airs, host_traces, p0["pvs"], gpu_proof0)
                out["config"]["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 2)
            except Exception as e:  # the baseline is a reported number, never a gate
                out["cpu_baseline"] = {"value": None, "unit": "proofs/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        if guest is not None:
            out["guest_flow"] = guest
        if guest_chunk is not None:
            out["guest_flow_chunk_config"] = guest_chunk
        if guest_mixed is not None:
            out["guest_flow_mixed"] = guest_mixed
        if guest_mixed19 is not None:
            out["guest_flow_mixed_frame19"] = guest_mixed19
        if guest_mem is not None:
            out["guest_flow_memory_bound"] = guest_mem
        if guest_devices is not None:
            out["guest_flow_devices"] = guest_devices
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()



def valu_counts():
    """The committed VALU counter pass (profiles/roundNN_pmc_valu.json, newest first) and whether it still describes the code that runs:
    (file name, parsed file, stale flag, note).  The counts belong to ONE compiled body of zk::k_hash_rows; the pass stores that body's
    sha256 (tools/code_object_hash.py) and this recomputes it from the libzkhip.so that is loaded -- a kernel edited since the pass makes
    every figure derived from the file stale.  stale is None when the file carries no hash."""
    import re
    names = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"round\d+[a-z]?_pmc_valu\.json", f)), reverse=True)
    name = names[0] if names else None
    if name is None:
        return None, None, None, "no profiles/round*_pmc_valu.json"
    with open(os.path.join(ROOT, "profiles", name)) as f:
        pv = json.load(f)
    want = pv.get("kernels", {}).get("zk::k_hash_rows", {}).get("code_sha256")
    if not want:
        return name, pv, None, "profiles/%s carries no code hash: unchecked" % name
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import code_object_hash

    so = os.environ.get("ZKHIP_LIBRARY_FOR_HASH") or os.path.join(ROOT, "zkvm-prover_amd", "libzkhip.so")
    try:
        have, _ = code_object_hash.kernel_code_sha256(so, code_object_hash.HASH_ROWS)
    except Exception as e:
        return name, pv, True, "cannot hash zk::k_hash_rows in %s: %r" % (so, e)
    if have != want:
        return name, pv, True, "zk::k_hash_rows in the loaded library (sha256 %s...) is not the body profiles/%s was counted on (%s...)" % (have[:12], name, want[:12])
    return name, pv, False, "zk::k_hash_rows code sha256 %s... == profiles/%s" % (have[:12], name)


def plumbing_run(args, world, rank, guest_devices=None):
    """ZKHIP_BENCH_PLUMBING_ONLY=1: the launcher, the rendezvous, the asynchronous exchange and the JSON line with stub
    proofs over gloo -- what a box without a GPU can check (tests/test_bench_launcher_cpu.py).  Not a measurement."""
    import torch.distributed as dist

    from zkvm_prover_amd import shard

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    pr = StubProver(rank)
    xch = shard.AsyncExchange()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        xch.post(pr.collect(i))
    commits, proofs = xch.wait()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ok = all(len(c) == world for c in commits)
    folded = None
    if rank == 0:
        for i, prs in enumerate(proofs):
            ok &= [bytes(b) for b in prs] == [StubProver(r).collect(i) for r in range(world)]
            ok &= [bytes(c) for c in commits[i]] == [shard.commitment_of(b) for b in prs]
        # gather -> aggregate on rank 0 (the tree's shape with stub nodes: a node = the hash of its children)
        import hashlib

        from zkvm_prover_amd import aggregate

        root, levels = aggregate.fold_tree([bytes(b) for b in proofs[-1]], lambda g: hashlib.sha256(b"L" + b"".join(g)).digest(),
                                           lambda g, leaves: hashlib.sha256(b"I" + b"".join(g)).digest())
        folded = {"proofs_folded": world, "nodes_per_level": [len(l) for l in levels], "stub_root": root.hex()[:16]}
        print(json.dumps({"metric": "chunk STARK proofs/sec (2^%d-row trace)" % args.log_rows, "value": None,
                          "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / max(1, args.steps) * 1e3, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                          "config": {"workload": "PLUMBING ONLY: stub proofs, gloo, no GPU -- not a measurement",
                                     "exchange_ok": bool(ok)}, "aggregate": folded, "guest_flow_devices": guest_devices}))
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
