"""Multi-GPU layout of the correlation path: channels shard across ranks (they are independent given the
shared IF stream) and every IF chunk is broadcast once from the rank that owns the stream into
every rank's HBM ring.  No reduction step exists: per-channel results go straight to the host (or are
gathered on rank 0).  Works with any torch.distributed backend ("nccl" = RCCL over xGMI on the GPUs, "gloo"
in the CPU tests).

ShardedEngine is the driver: one instance per rank (one rank per GPU), built on an Engine-like object --
the ctypes Engine of this package on a GPU, or any stand-in with the same ring / channel / tracking methods
(the CPU tests use the oracle).

Ring schedule (a batch may read up to one code period -- plus code-rate drift -- past the end of its chunk,
so a chunk is never received into the slot next to the one being correlated):

    ring = 4 slots of `chunk` samples.  While batch k is correlated in slots k and k+1 (mod 4) -- reaching back
    into slot k-1 by whatever the code-rate drift has accumulated -- chunk k+2 is broadcast into slot k+2
    (mod 4).  The write position of the ring moves (ring_commit) only after the broadcast has completed."""


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


class ShardedEngine:
    """One rank's share of a multi-GPU receiver.

    engine       Engine-like object of this rank (ring_create / ring_commit / set_channels / trk_* / acq_*)
    ring_tensor  int8 torch tensor of 4*chunk*dtype bytes on this rank's device: the HBM ring (the engine's
                 ring is created on its memory, broadcasts land in it directly)
    channels     the receiver's full channel list (every rank passes the same list)
    strong       True: the list is sharded over the ranks (BASELINE configs[3]: 46 channels -> 6,6,6,6,6,6,5,5);
                 False: every rank tracks the whole list it was given (weak scaling: per-rank channel sets)
    dist         torch.distributed (initialised) or None for a single rank
    """

    SLOTS = 4

    def __init__(self, engine, ring_tensor, channels, chunk, dtype, dist=None, rank=0, world=1, strong=True, ftype=1,
                 src=0):
        self.eng, self.ring_t, self.dist = engine, ring_tensor, dist
        self.rank, self.world, self.src, self.ftype = rank, world, src, ftype
        self.chunk, self.dtype = int(chunk), int(dtype)
        self.chunk_bytes = self.chunk * self.dtype
        if ring_tensor.numel() != self.SLOTS * self.chunk_bytes:
            raise ValueError("ring tensor must hold four chunks")
        self.all_channels = list(channels)
        self.mine = shard_channels(len(channels), world, rank) if strong and world > 1 else list(range(len(channels)))
        self.channels = [self.all_channels[i] for i in self.mine]
        engine.ring_create(ftype, dtype, self.SLOTS * self.chunk, ring_tensor.data_ptr())
        if self.channels:
            engine.set_channels(self.channels)
        self.received = 0           # chunks complete in the ring
        self.pending = None         # (work handle, chunk index) of the broadcast in flight

    # -- IF stream ----------------------------------------------------------------------------------------
    def _slot(self, k):
        lo = (k % self.SLOTS) * self.chunk_bytes
        return self.ring_t[lo:lo + self.chunk_bytes]

    def feed(self, chunk_tensor=None, blocking=False, resident=False):
        """Chunk number `received (+1 if one is in flight)` of the stream: the owning rank passes its samples
        (int8 tensor of chunk*dtype bytes, any device), the others None.  Starts the broadcast and returns;
        the chunk counts as received -- and the ring's write position moves -- at the next feed() / wait().
        resident=True: the owning rank's slot already holds the chunk (a repeating synthetic stream)."""
        self.wait()
        k = self.received
        dst = self._slot(k)
        if self.rank == self.src and not resident:
            if chunk_tensor is None:
                raise ValueError("the rank that owns the stream must pass the chunk")
            dst.copy_(chunk_tensor.reshape(-1).to(dst.device), non_blocking=True)
        if self.dist is not None and self.world > 1:
            self.pending = (self.dist.broadcast(dst, src=self.src, async_op=True), k)
        else:
            self.pending = (None, k)
        if blocking:
            self.wait()

    def wait(self):
        """The broadcast in flight has landed: commit its chunk to the ring."""
        if self.pending is None:
            return
        work, k = self.pending
        if work is not None:
            work.wait()
        self.pending = None
        self.received = k + 1
        self.eng.ring_commit(self.ftype, self.chunk)

    # -- tracking -----------------------------------------------------------------------------------------
    def set_states(self, states_all):
        """states_all: one state dict per channel of the full list; this rank keeps its share."""
        if self.channels:
            self.eng.trk_set_state([states_all[i] for i in self.mine])

    def trk_run(self, nepoch):
        if self.channels:
            self.eng.trk_run(nepoch)

    def trk_fetch(self):
        """(channel indices of this rank, II, QQ, nsamp) of the last batch"""
        if not self.channels:
            return self.mine, None, None, None
        II, QQ, ns = self.eng.trk_fetch()
        return self.mine, II, QQ, ns

    def gather(self, local):
        """local: {channel index: anything picklable}; returns the merged dict on every rank."""
        if self.dist is None or self.world == 1:
            return dict(local)
        parts = [None] * self.world
        self.dist.all_gather_object(parts, local)
        merged = {}
        for p in parts:
            merged.update(p)
        return merged

    def step(self, nepoch, next_chunk=None, resident=False):
        """One batch in the steady state: the next chunk of the stream starts travelling (two chunks ahead of
        the batch), then the batch over the chunk(s) already in the ring is launched."""
        self.feed(next_chunk, resident=resident)
        self.trk_run(nepoch)
