"""Segment-parallel proving across the GPUs of one node.

The path shards by independent units: continuation segments of one chunk / chunks of one batch
are independent STARKs (the reference proves them in a sequential loop on one device,
crates/integration/src/testers/batch.rs:97-107).  One process per GPU proves segments
`rank, rank + world, ...`; the only exchange step is latency-bound: an all-gather of the 32-byte
trace commitments (so every rank can derive the shared aggregation transcript) and a gather of the
~1 MB proofs to rank 0.  `torch.distributed` backend "nccl" is RCCL over xGMI on ROCm; the same code
runs over "gloo" on CPU tensors for tests.
"""
import torch
import torch.distributed as dist

COMMIT_OFFSET = 16  # bytes: proof header is 4 words, then the main-trace root (8 words)
COMMIT_BYTES = 32


def assign_segments(n_segments, world, rank):
    """Static round-robin: segment i -> rank i mod world."""
    return list(range(rank, n_segments, world))


def commitment_of(proof_bytes):
    return bytes(proof_bytes[COMMIT_OFFSET:COMMIT_OFFSET + COMMIT_BYTES])


def exchange(proof_bytes, device=None, group=None):
    """All-gathers the trace commitment of every rank's proof and gathers the proofs on rank 0.

    proof_bytes: bytes-like, same length on every rank (FRI proofs of one key are shape-static).
    Returns (commitments: list[bytes] of length world, proofs: list[bytes] on rank 0 else None)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [commitment_of(proof_bytes)], [bytes(proof_bytes)]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    proof_t = torch.frombuffer(bytearray(proof_bytes), dtype=torch.uint8).to(dev)
    root = proof_t[COMMIT_OFFSET:COMMIT_OFFSET + COMMIT_BYTES].clone()
    allr = torch.empty(world * COMMIT_BYTES, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(allr, root, group=group)
    gl = [torch.empty_like(proof_t) for _ in range(world)] if rank == 0 else None
    dist.gather(proof_t, gl, dst=0, group=group)
    commits = [bytes(allr[i * COMMIT_BYTES:(i + 1) * COMMIT_BYTES].cpu().numpy().tobytes()) for i in range(world)]
    proofs = [bytes(t.cpu().numpy().tobytes()) for t in gl] if rank == 0 else None
    return commits, proofs


class AsyncExchange:
    """The same exchange, posted without blocking the proving loop.

    `post(proof_bytes)` stages the proof on `device`, enqueues the all-gather of its commitment and the gather of the
    proof to rank 0 as asynchronous collectives (on a side stream for RCCL, so they never wait behind -- or hold up --
    the prover's streams) and returns at once; ranks do NOT run in lockstep: a collective completes whenever the last
    rank has posted its part.  `wait()` blocks until everything posted has completed and returns
    (commitments: list over posts of list[bytes] per rank, proofs: the same for proof bytes on rank 0, else None).
    """

    def __init__(self, device=None, group=None):
        self.group = group
        self.on = dist.is_initialized() and dist.get_world_size(group) > 1
        self.device = device if device is not None else torch.device("cpu")
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.stream = torch.cuda.Stream(device=self.device) if self.on and self.device.type == "cuda" else None
        self.pending = []

    def post(self, proof_bytes):
        if not self.on:
            self.pending.append((None, None, None, bytes(proof_bytes)))
            return
        host = torch.frombuffer(bytearray(proof_bytes), dtype=torch.uint8)
        if self.stream is not None:
            host = host.pin_memory()
            with torch.cuda.stream(self.stream):
                proof_t = host.to(self.device, non_blocking=True)
                work, allr, gl = self._collectives(proof_t)
        else:
            proof_t = host
            work, allr, gl = self._collectives(proof_t)
        self.pending.append((work, allr, gl, (host, proof_t)))

    def _collectives(self, proof_t):
        root = proof_t[COMMIT_OFFSET:COMMIT_OFFSET + COMMIT_BYTES].clone()
        allr = torch.empty(self.world * COMMIT_BYTES, dtype=torch.uint8, device=proof_t.device)
        w1 = dist.all_gather_into_tensor(allr, root, group=self.group, async_op=True)
        gl = [torch.empty_like(proof_t) for _ in range(self.world)] if self.rank == 0 else None
        w2 = dist.gather(proof_t, gl, dst=0, group=self.group, async_op=True)
        return (w1, w2), allr, gl

    def wait(self):
        commits, proofs = [], []
        for work, allr, gl, keep in self.pending:
            if work is None:
                commits.append([commitment_of(keep)])
                proofs.append([keep])
                continue
            for w in work:
                w.wait()
        if self.stream is not None:
            self.stream.synchronize()
        for work, allr, gl, keep in self.pending:
            if work is None:
                continue
            a = allr.cpu().numpy().tobytes()
            commits.append([a[i * COMMIT_BYTES:(i + 1) * COMMIT_BYTES] for i in range(self.world)])
            proofs.append([t.cpu().numpy().tobytes() for t in gl] if self.rank == 0 else None)
        self.pending = []
        return commits, (proofs if self.rank == 0 else None)
