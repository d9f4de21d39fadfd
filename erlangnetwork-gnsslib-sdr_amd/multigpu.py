"""Multi-GPU layout of the correlation path: channels shard across ranks (they
are independent given the shared IF stream) and the IF chunk of each
epoch batch is broadcast once from the ingest rank.  No reduction step exists:
per-channel results go straight to the host.  Works with any torch.distributed
backend ("nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests)."""


def shard_channels(nch, world, rank):
    """Indices of the channels rank `rank` owns: contiguous blocks, sizes differing by at most one
    (32 GPS SVs over 8 GPUs -> 4 each; 46 channels -> 6,6,6,6,6,6,5,5)."""
    base, extra = divmod(nch, world)
    lo = rank * base + min(rank, extra)
    return list(range(lo, lo + base + (1 if rank < extra else 0)))


def owner_of(ch, nch, world):
    for r in range(world):
        if ch in shard_channels(nch, world, r):
            return r
    raise ValueError(ch)


def broadcast_chunk(dist, ring_tensor, byte_lo, nbytes, src=0, async_op=False):
    """One broadcast of ring_tensor[byte_lo:byte_lo+nbytes] from the ingest rank into the same
    bytes of every rank's IF ring (the tensor the HBM ring was created on)."""
    return dist.broadcast(ring_tensor[byte_lo:byte_lo + nbytes], src=src, async_op=async_op)
