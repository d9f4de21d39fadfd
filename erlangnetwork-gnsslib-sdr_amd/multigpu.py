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
    (mod 4).  The write position of the ring moves (ring_commit) only after the broadcast has completed.

A receiver with two front ends (ref frontend/stereo_L1G1.ini) has one such ring and one broadcast per front end."""


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


class _Stream:
    """One IF stream of the receiver on this rank: its HBM ring (a torch tensor the engine's ring is created on), the
    chunk schedule and -- when only some ranks own channels of this front end -- the process group it travels in."""

    def __init__(self, ftype, dtype, chunk, ring_tensor):
        self.ftype, self.dtype, self.chunk = int(ftype), int(dtype), int(chunk)
        self.chunk_bytes = self.chunk * self.dtype
        self.ring_t = ring_tensor
        self.received = 0           # chunks complete in the ring
        self.pending = None         # (work handle, chunk index) of the broadcast in flight
        self.group = None           # torch.distributed group of the ranks the stream goes to (None: all ranks)
        self.member = True          # this rank receives the stream


class ShardedEngine:
    """One rank's share of a multi-GPU receiver.

    engine       Engine-like object of this rank (ring_create / ring_commit / set_channels / trk_* / acq_*)
    ring_tensor  int8 torch tensor of 4*chunk*dtype bytes on this rank's device: the HBM ring of front end `ftype`
                 (the engine's ring is created on its memory, broadcasts land in it directly)
    channels     the receiver's full channel list (every rank passes the same list)
    strong       True: the list is sharded over the ranks (BASELINE configs[3]: 46 channels -> 6,6,6,6,6,6,5,5);
                 False: every rank tracks the whole list it was given (weak scaling: per-rank channel sets)
    dist         torch.distributed (initialised) or None for a single rank
    second       (ring_tensor2, chunk2, dtype2) of the receiver's second front end (ftype 2; ref
                 frontend/stereo_L1G1.ini: two streams with their own sample type), or None.  A stream travels only to
                 the ranks that own channels of its front end (plus the rank that ingests it): the channel list keeps a
                 front end's channels together, so with 32 GPS + 14 GLONASS channels over 8 ranks the GLONASS stream
                 goes to three of them.

    Stream order.  The ring memory is written on torch's current stream (the copy of the owner's chunk, the
    RCCL broadcast) and read on the engine's stream (the correlator kernels).  When those are two streams -- an
    Engine created with its own -- feed() first makes torch's stream wait for the engine's last launch (the slot it
    is about to overwrite may still be read by the batch in flight) and wait() makes the engine's stream wait for
    the broadcast before the write position moves.  With the engine on torch's current stream both are no-ops.
    """

    SLOTS = 4

    def __init__(self, engine, ring_tensor, channels, chunk, dtype, dist=None, rank=0, world=1, strong=True, ftype=1,
                 src=0, second=None):
        self.eng, self.dist = engine, dist
        self.rank, self.world, self.src = rank, world, src
        self.all_channels = list(channels)
        self.mine = shard_channels(len(channels), world, rank) if strong and world > 1 else list(range(len(channels)))
        self.channels = [self.all_channels[i] for i in self.mine]
        self.streams = {}
        specs = [(ftype, dtype, chunk, ring_tensor)]
        if second is not None:
            specs.append((2 if ftype == 1 else 1, second[2], second[1], second[0]))
        for ft, dt, ck, rt in specs:
            st = _Stream(ft, dt, ck, rt)
            if rt.numel() != self.SLOTS * st.chunk_bytes:
                raise ValueError("ring tensor must hold four chunks")
            self.streams[ft] = st
        # who needs which stream: the ranks that own a channel of the front end, and the rank that ingests it
        if dist is not None and world > 1 and len(self.streams) > 1 and strong:
            for ft, st in sorted(self.streams.items()):
                need = sorted({owner_of(i, len(channels), world) for i, c in enumerate(self.all_channels)
                               if getattr(c, "ftype", 1) == ft} | {src})
                st.member = rank in need
                # (every rank takes part in creating every group, members or not: torch.distributed's rule)
                st.group = dist.new_group(ranks=need) if len(need) < world else None
        for ft, st in self.streams.items():
            engine.ring_create(ft, st.dtype, self.SLOTS * st.chunk, st.ring_t.data_ptr())
        if self.channels:
            engine.set_channels(self.channels)
        # stream order between torch's stream and the engine's (see the class comment)
        self._cuda = None
        if getattr(ring_tensor, "is_cuda", False) and hasattr(engine, "stream_ptr"):
            import torch
            self._cuda = torch.cuda
            self._eng_stream = torch.cuda.ExternalStream(engine.stream_ptr(), device=ring_tensor.device)
            self._last_launch = None

    # compatibility with the single-stream attributes of the first version
    @property
    def received(self):
        return next(iter(self.streams.values())).received

    @property
    def chunk(self):
        return next(iter(self.streams.values())).chunk

    @property
    def ftype(self):
        return next(iter(self.streams.values())).ftype

    # -- IF streams ---------------------------------------------------------------------------------------
    def _slot(self, st, k):
        lo = (k % self.SLOTS) * st.chunk_bytes
        return st.ring_t[lo:lo + st.chunk_bytes]

    def _mark_launch(self):
        if self._cuda is not None:
            ev = self._cuda.Event()
            ev.record(self._eng_stream)
            self._last_launch = ev

    def feed(self, chunk_tensor=None, blocking=False, resident=False, ftype=None):
        """Chunk number `received (+1 if one is in flight)` of stream `ftype` (default: the first): the owning rank
        passes its samples (int8 tensor of chunk*dtype bytes, any device), the others None.  Starts the broadcast and
        returns; the chunk counts as received -- and the ring's write position moves -- at the next feed() / wait().
        resident=True: the owning rank's slot already holds the chunk (a repeating synthetic stream)."""
        st = self.streams[ftype if ftype is not None else next(iter(self.streams))]
        self.wait(st.ftype)
        k = st.received
        if not st.member:
            st.pending = (None, k)
            return
        dst = self._slot(st, k)
        if self._cuda is not None and self._last_launch is not None:
            self._cuda.current_stream().wait_event(self._last_launch)     # the batch in flight may still read this slot
        if self.rank == self.src and not resident:
            if chunk_tensor is None:
                raise ValueError("the rank that owns the stream must pass the chunk")
            dst.copy_(chunk_tensor.reshape(-1).to(dst.device), non_blocking=True)
        if self.dist is not None and self.world > 1:
            st.pending = (self.dist.broadcast(dst, src=self.src, group=st.group, async_op=True), k)
        else:
            st.pending = (None, k)
        if blocking:
            self.wait(st.ftype)

    def wait(self, ftype=None):
        """The broadcasts in flight (of one stream, or of all) have landed: commit their chunks to the rings."""
        for st in ([self.streams[ftype]] if ftype is not None else list(self.streams.values())):
            if st.pending is None:
                continue
            work, k = st.pending
            if work is not None:
                work.wait()
            st.pending = None
            st.received = k + 1
            if self._cuda is not None and st.member:
                ev = self._cuda.Event()
                ev.record(self._cuda.current_stream())
                self._eng_stream.wait_event(ev)                           # the engine reads the chunk only after it has landed
            self.eng.ring_commit(st.ftype, st.chunk)

    # -- acquisition --------------------------------------------------------------------------------------
    def acq_run(self, wrpos=0):
        """Cold search of this rank's channels over what its rings hold (ref src/sdracq.c:14-62 per channel)."""
        if self.channels:
            self.eng.acq_run(wrpos)
            self._mark_launch()

    def acq_fetch(self):
        """(channel indices of this rank, their acquisition results)"""
        return self.mine, (self.eng.acq_fetch() if self.channels else [])

    def trk_start_from_acq(self):
        """Device-side hand-over of the acquired channels to tracking (ref src/sdracq.c:51-55)."""
        if self.channels:
            self.eng.trk_start_from_acq()

    # -- tracking -----------------------------------------------------------------------------------------
    def set_states(self, states_all):
        """states_all: one state dict per channel of the full list; this rank keeps its share."""
        if self.channels:
            self.eng.trk_set_state([states_all[i] for i in self.mine])

    def trk_run(self, nepoch):
        if self.channels:
            self.eng.trk_run(nepoch)
            self._mark_launch()

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

    def step(self, nepoch, next_chunk=None, resident=False, next_chunk2=None):
        """One batch in the steady state: the next chunk of every stream starts travelling (two chunks ahead of
        the batch), then the batch over the chunk(s) already in the rings is launched."""
        fts = list(self.streams)
        self.feed(next_chunk, resident=resident, ftype=fts[0])
        if len(fts) > 1:
            self.feed(next_chunk2, resident=resident, ftype=fts[1])
        self.trk_run(nepoch)
