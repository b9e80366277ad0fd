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
