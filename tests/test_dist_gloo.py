"""N > 1 path: the package's ShardedEngine (erlangnetwork-gnsslib-sdr_amd/multigpu.py, the driver bench.py uses
with the nccl = RCCL backend) on two ranks.

  CPU (gloo):  an oracle-backed stand-in for the Engine, so the driver's own logic is what is tested: strong
               sharding of the channel list, the four-slot ring with the broadcast two chunks ahead of the
               batch, write-position commits, result gather -- against one single-process oracle run.
  GPU (-m gpu): the real Engine on both ranks (one GPU shared on a single-GPU box), same comparison against a
               single-process Engine run."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nsamples, prns, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ctypes as C
    import torch
    import torch.distributed as dist
    import gnsscorr_loader
    gnsscorr_loader.load()
    import importlib
    mg = importlib.import_module("erlangnetwork_gnsslib_sdr_amd.multigpu")
    import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ring = torch.zeros(nsamples * 2, dtype=torch.int8)
    if rank == 0:
        rng = np.random.default_rng(99)
        ring.copy_(torch.from_numpy(rng.integers(-60, 61, size=nsamples * 2, dtype=np.int8)))
    # one broadcast per epoch batch: two chunks here
    half = nsamples  # bytes per chunk (nsamples*2 bytes total)
    mg.broadcast_chunk(dist, ring, 0, half, src=0)
    w = mg.broadcast_chunk(dist, ring, half, half, src=0, async_op=True)
    w.wait()
    data = ring.numpy()
    mine = mg.shard_channels(len(prns), world, rank)
    res = {}
    r = orc.make_ring(data, nsamples, nsamples)
    for i in mine:
        o = orc.make_chan(prns[i], dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3)
        o.carrfreq, o.codefreq, o.remcode, o.remcarr = 500.0 * (i + 1), o.crate + 0.1 * i, 0.1 * i + 0.05, 0.2 * i
        b = 10 + i
        rows = []
        for e in range(3):
            orc.lib().orc_sdrtracking(C.byref(o), C.byref(r), b)
            rows.append([o.II[t] for t in range(5)] + [o.QQ[t] for t in range(5)])
            b += o.currnsamp
        res[i] = rows
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        merged = {}
        for g in gathered:
            merged.update(g)
        out_q.put((merged, data.copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharding_helpers(gc):
    import importlib
    mg = importlib.import_module("erlangnetwork_gnsslib_sdr_amd.multigpu")
    for nch, world in ((32, 8), (46, 8), (5, 2), (3, 4), (32, 1)):
        shards = [mg.shard_channels(nch, world, r) for r in range(world)]
        assert sorted(sum(shards, [])) == list(range(nch))
        assert max(map(len, shards)) - min(map(len, shards)) <= 1
        for r, s in enumerate(shards):
            assert all(mg.owner_of(c, nch, world) == r for c in s)
    assert [len(mg.shard_channels(46, 8, r)) for r in range(8)] == [6, 6, 6, 6, 6, 6, 5, 5]


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process(gc, orc):
    import ctypes as C
    import torch.multiprocessing as mp
    world, nsamples, prns = 2, 16 * 4096, [2, 5, 9, 14, 30]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nsamples, prns, q)) for r in range(world)]
    for p in procs:
        p.start()
    merged, data = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(merged) == list(range(len(prns)))
    ring = orc.make_ring(data, nsamples, nsamples)
    for i, prn in enumerate(prns):
        o = orc.make_chan(prn, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3)
        o.carrfreq, o.codefreq, o.remcode, o.remcarr = 500.0 * (i + 1), o.crate + 0.1 * i, 0.1 * i + 0.05, 0.2 * i
        b = 10 + i
        for e in range(3):
            orc.lib().orc_sdrtracking(C.byref(o), C.byref(ring), b)
            assert merged[i][e] == [o.II[t] for t in range(5)] + [o.QQ[t] for t in range(5)]
            b += o.currnsamp


# ---------------------------------------------------------------------------------------------------------
# ShardedEngine
# ---------------------------------------------------------------------------------------------------------
CHUNK_EPOCHS = 3
CHUNK = CHUNK_EPOCHS * 16368                 # samples per chunk = what a batch consumes (periods start mid-chunk: every batch straddles two)
NCHUNK = 5
PRNS = [2, 5, 9, 14, 30, 31, 7]


def _stream():
    rng = np.random.default_rng(4242)
    return rng.integers(-70, 71, size=(NCHUNK * CHUNK, 2), dtype=np.int8)


def _states(nch):
    return [dict(carrfreq=700.0 * (i - 3), codefreq=1.023e6 + 0.3 * i, remcode=0.01 * i, remcarr=0.3 * i if i % 2 else -50.0 * i,
                 buffloc=40 + 11 * i) for i in range(nch)]


class OracleEngine:
    """Engine look-alike on the CPU oracle: just enough of the interface for ShardedEngine."""

    def __init__(self, orc):
        self.orc, self.wrpos, self.chans, self.st = orc, 0, [], []

    def ring_create(self, ftype, dtype, ringlen, devmem):
        self.ringlen, self.ptr = ringlen, devmem

    def ring_commit(self, ftype, n):
        self.wrpos += n

    def set_channels(self, chans):
        self.chans = chans

    def trk_set_state(self, states):
        self.st = [dict(s) for s in states]

    def trk_run(self, nepoch):
        import ctypes as C
        orc = self.orc
        ring = orc.Ring()
        ring.buff, ring.ringlen, ring.wrpos = self.ptr, self.ringlen, self.wrpos
        self.out = []
        for prn, s in zip(self.chans, self.st):
            o = orc.make_chan(prn, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3)
            o.carrfreq, o.codefreq, o.remcode, o.remcarr = s["carrfreq"], s["codefreq"], s["remcode"], s["remcarr"]
            rows = []
            for _ in range(nepoch):
                orc.lib().orc_sdrtracking(C.byref(o), C.byref(ring), s["buffloc"])
                assert o.flagtrk == 1, "the batch ran ahead of the ring's write position"
                rows.append([o.II[t] for t in range(5)] + [o.QQ[t] for t in range(5)])
                s["buffloc"] += o.currnsamp
            s["remcode"], s["remcarr"] = o.remcode, o.remcarr
            self.out.append(rows)

    def trk_fetch(self):
        a = np.array(self.out)
        return a[:, :, :5], a[:, :, 5:], None


def _drive(se, torch, stream, nch, batches):
    """The steady-state schedule of ShardedEngine.step(): two chunks in the ring before the first batch."""
    def chunk(k):
        return torch.from_numpy(stream[k * CHUNK:(k + 1) * CHUNK].reshape(-1)) if se.rank == 0 and k < NCHUNK else None
    se.feed(chunk(0))
    se.feed(chunk(1))
    se.wait()
    se.set_states(_states(nch))
    got = {i: [] for i in se.mine}
    for k in range(batches):
        if k + 2 < NCHUNK:
            se.step(CHUNK_EPOCHS, chunk(k + 2))
        else:
            se.wait()
            se.trk_run(CHUNK_EPOCHS)
        mine, II, QQ, _ = se.trk_fetch()
        for j, i in enumerate(mine):
            got[i] += [list(II[j, e]) + list(QQ[j, e]) for e in range(CHUNK_EPOCHS)]
    return se.gather(got)


def _drive_nofetch(se, torch, stream, nch, batches):
    """The same schedule with no fetch (= no synchronisation) between the batches: only stream order keeps a chunk from
    being overwritten while the batch in flight reads it, and a batch from reading a chunk that has not landed."""
    def chunk(k):
        return torch.from_numpy(stream[k * CHUNK:(k + 1) * CHUNK].reshape(-1)) if se.rank == 0 and k < NCHUNK else None
    se.feed(chunk(0))
    se.feed(chunk(1))
    se.wait()
    se.set_states(_states(nch))
    for k in range(batches):
        if k + 2 < NCHUNK:
            se.step(CHUNK_EPOCHS, chunk(k + 2))
        else:
            se.wait()
            se.trk_run(CHUNK_EPOCHS)
    mine, II, QQ, _ = se.trk_fetch()
    return se.gather({i: [list(II[j, e]) + list(QQ[j, e]) for e in range(CHUNK_EPOCHS)] for j, i in enumerate(mine)})


def _sharded_worker(rank, world, port, use_gpu, out_q, nofetch=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import gnsscorr_loader
    gc = gnsscorr_loader.load()
    import importlib
    mg = importlib.import_module("erlangnetwork_gnsslib_sdr_amd.multigpu")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    stream = _stream()
    if use_gpu:
        torch.cuda.set_device(0)
        ring_t = torch.zeros(4 * CHUNK * 2, dtype=torch.int8, device="cuda:0")
        eng = gc.Engine(0)
        chans = [gc.Channel(p, dtype=2, f_if=0.0) for p in PRNS]
    else:
        import oracle as orc
        ring_t = torch.zeros(4 * CHUNK * 2, dtype=torch.int8)
        eng = OracleEngine(orc)
        chans = PRNS
    se = mg.ShardedEngine(eng, ring_t, chans, CHUNK, 2, dist=dist, rank=rank, world=world, strong=True)
    assert se.mine == mg.shard_channels(len(PRNS), world, rank)
    merged = (_drive_nofetch if nofetch else _drive)(se, torch, stream, len(PRNS), NCHUNK - 1)
    if rank == 0:
        out_q.put(merged)
    dist.barrier()
    dist.destroy_process_group()


def _reference(orc):
    import ctypes as C
    stream = _stream()
    n = stream.shape[0]
    ring = orc.make_ring(stream, n, n)
    ref = {}
    for i, (prn, s) in enumerate(zip(PRNS, _states(len(PRNS)))):
        o = orc.make_chan(prn, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3)
        o.carrfreq, o.codefreq, o.remcode, o.remcarr = s["carrfreq"], s["codefreq"], s["remcode"], s["remcarr"]
        b, rows = s["buffloc"], []
        for _ in range((NCHUNK - 1) * CHUNK_EPOCHS):
            orc.lib().orc_sdrtracking(C.byref(o), C.byref(ring), b)
            rows.append([o.II[t] for t in range(5)] + [o.QQ[t] for t in range(5)])
            b += o.currnsamp
        ref[i] = rows
    return ref


def _run_two_ranks(use_gpu, nofetch=False):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, use_gpu, q, nofetch)) for r in range(2)]
    for p in procs:
        p.start()
    merged = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return merged


@pytest.mark.timeout(300)
def test_sharded_engine_two_ranks_gloo_cpu(gc, orc):
    merged = _run_two_ranks(False)
    assert merged == _reference(orc)


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_sharded_engine_two_ranks_real_engine(gc, orc):
    merged = _run_two_ranks(True)
    assert merged == _reference(orc)


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_sharded_engine_two_ranks_no_fetch_between_batches(gc, orc):
    """The engine on a stream of its own (not torch's current one), four batches back to back without a fetch: the last
    batch's sums must still be the single-process ones (ShardedEngine orders the two streams with events)."""
    merged = _run_two_ranks(True, nofetch=True)
    ref = _reference(orc)
    assert merged == {i: rows[-CHUNK_EPOCHS:] for i, rows in ref.items()}


# ---------------------------------------------------------------------------------------------------------
# two front ends, acquisition on the shard
# ---------------------------------------------------------------------------------------------------------
class _Ch:
    """channel stand-in for the oracle-backed engine: PRN + front end"""

    def __init__(self, prn, ftype):
        self.prn, self.ftype = prn, ftype


class OracleEngine2(OracleEngine):
    """Two rings; acquisition replaced by a cheap deterministic probe of the ring contents (the driver's logic is what
    the CPU test is about: which rank holds which stream, which channels, and that the hand-over reaches tracking)."""

    def __init__(self, orc):
        super().__init__(orc)
        self.rings = {}

    def ring_create(self, ftype, dtype, ringlen, devmem):
        self.rings[ftype] = dict(ringlen=ringlen, ptr=devmem, wrpos=0)

    def ring_commit(self, ftype, n):
        self.rings[ftype]["wrpos"] += n

    def _bytes(self, ftype, lo, n):
        import ctypes as C
        r = self.rings[ftype]
        return np.ctypeslib.as_array((C.c_int8 * (2 * r["ringlen"])).from_address(r["ptr"]))[2 * lo:2 * (lo + n)]

    def acq_run(self, wrpos=0):
        self.acq = []
        for c in self.chans:
            win = self._bytes(c.ftype, 100 * c.prn, 64).astype(np.int64)
            self.acq.append(dict(flagacq=1, buffloc=int(50 + 7 * c.prn), acqfreq=float(200 * (int(win.sum()) % 11 - 5)),
                                 probe=int((win * np.arange(1, 129)).sum())))

    def acq_fetch(self):
        return self.acq

    def trk_start_from_acq(self):
        self.st = [dict(carrfreq=a["acqfreq"], codefreq=1.023e6, remcode=0.0, remcarr=0.0, buffloc=a["buffloc"]) for a in self.acq]

    def trk_run(self, nepoch):
        import ctypes as C
        orc = self.orc
        self.out = []
        for c, s in zip(self.chans, self.st):
            r = self.rings[c.ftype]
            ring = orc.Ring()
            ring.buff, ring.ringlen, ring.wrpos = r["ptr"], r["ringlen"], r["wrpos"]
            o = orc.make_chan(c.prn, dtype=2, f_if=0.0, corrn=2, corrd=3, corrp=3)
            o.carrfreq, o.codefreq, o.remcode, o.remcarr = s["carrfreq"], s["codefreq"], s["remcode"], s["remcarr"]
            rows = []
            for _ in range(nepoch):
                orc.lib().orc_sdrtracking(C.byref(o), C.byref(ring), s["buffloc"])
                assert o.flagtrk == 1, "the batch ran ahead of the ring's write position"
                rows.append([o.II[t] for t in range(5)] + [o.QQ[t] for t in range(5)])
                s["buffloc"] += o.currnsamp
            self.out.append(rows)


C2_CHUNK = 2 * 16368
C2_CHANS = [(2, 1), (5, 1), (9, 1), (14, 1), (30, 1), (31, 1), (7, 2), (11, 2), (19, 2)]     # (prn, front end)


def _two_streams():
    rng = np.random.default_rng(777)
    return (rng.integers(-70, 71, size=(4 * C2_CHUNK, 2), dtype=np.int8),
            rng.integers(-40, 41, size=(4 * C2_CHUNK, 2), dtype=np.int8))


def _two_stream_worker(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import gnsscorr_loader
    gnsscorr_loader.load()
    import importlib
    mg = importlib.import_module("erlangnetwork_gnsslib_sdr_amd.multigpu")
    import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s1, s2 = _two_streams()
    r1 = torch.zeros(4 * C2_CHUNK * 2, dtype=torch.int8)
    r2 = torch.zeros(4 * C2_CHUNK * 2, dtype=torch.int8)
    chans = [_Ch(p, f) for p, f in C2_CHANS]
    se = mg.ShardedEngine(OracleEngine2(orc), r1, chans, C2_CHUNK, 2, dist=dist, rank=rank, world=world, strong=True,
                          second=(r2, C2_CHUNK, 2))
    for k in range(4):
        se.feed(torch.from_numpy(s1[k * C2_CHUNK:(k + 1) * C2_CHUNK].reshape(-1)) if rank == 0 else None, ftype=1)
        se.feed(torch.from_numpy(s2[k * C2_CHUNK:(k + 1) * C2_CHUNK].reshape(-1)) if rank == 0 else None, ftype=2)
    se.wait()
    se.acq_run()
    mine, acq = se.acq_fetch()
    se.trk_start_from_acq()
    se.trk_run(3)
    _, II, QQ, _ = se.trk_fetch()
    local = {i: dict(acq=a, rows=[list(II[j, e]) + list(QQ[j, e]) for e in range(3)]) for j, (i, a) in enumerate(zip(mine, acq))}
    merged = se.gather(local)
    # which streams reached this rank (a rank outside a stream's group keeps its ring at zero)
    have = se.gather({-1 - rank: (bool(r1.any()), bool(r2.any()), se.streams[1].member, se.streams[2].member)})
    if rank == 0:
        out_q.put((merged, {k: v for k, v in have.items() if k < 0}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_engine_two_streams_three_ranks_gloo_cpu(gc, orc):
    """Nine channels on two front ends, strong-sharded over three ranks: every stream travels only to the ranks that own
    channels of its front end (and the ingesting rank 0); acquisition, hand-over and tracking run on the shard and equal
    one single-process run."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_stream_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    merged, have = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # rank 0 ingests both; rank 1 owns front-end-1 channels only, rank 2 front-end-2 channels only
    assert have[-1] == (True, True, True, True)
    assert have[-2] == (True, False, True, False)
    assert have[-3] == (False, True, False, True)
    # single-process reference with the same stand-in
    import torch
    s1, s2 = _two_streams()
    ref_eng = OracleEngine2(orc)
    t1, t2 = torch.from_numpy(s1.reshape(-1).copy()), torch.from_numpy(s2.reshape(-1).copy())
    ref_eng.ring_create(1, 2, 4 * C2_CHUNK, t1.data_ptr())
    ref_eng.ring_create(2, 2, 4 * C2_CHUNK, t2.data_ptr())
    ref_eng.ring_commit(1, 4 * C2_CHUNK)
    ref_eng.ring_commit(2, 4 * C2_CHUNK)
    ref_eng.set_channels([_Ch(p, f) for p, f in C2_CHANS])
    ref_eng.acq_run()
    acq = ref_eng.acq_fetch()
    ref_eng.trk_start_from_acq()
    ref_eng.trk_run(3)
    assert sorted(merged) == list(range(len(C2_CHANS)))
    for i in range(len(C2_CHANS)):
        assert merged[i]["acq"] == acq[i], i
        assert merged[i]["rows"] == ref_eng.out[i], i


def _config3_worker(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hashlib
    import torch
    import torch.distributed as dist
    import gnsscorr_loader
    gc = gnsscorr_loader.load()
    import importlib
    mg = importlib.import_module("erlangnetwork_gnsslib_sdr_amd.multigpu")
    synth = importlib.import_module("erlangnetwork_gnsslib_sdr_amd.synth")
    import full_size_inputs as fs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    d1, d2, _, _ = fs.two_streams(gc, synth) if rank == 0 else (None, None, None, None)
    n = fs.C3_MS * fs.NS
    chunk = n // 4
    r1 = torch.zeros(4 * chunk * 2, dtype=torch.int8, device="cuda:0")
    r2 = torch.zeros(4 * chunk * 2, dtype=torch.int8, device="cuda:0")
    chans = fs.config3_channels(gc)
    se = mg.ShardedEngine(gc.Engine(0), r1, chans, chunk, 2, dist=dist, rank=rank, world=world, strong=True, second=(r2, chunk, 2))
    for k in range(4):
        se.feed(torch.from_numpy(d1[k * chunk:(k + 1) * chunk].reshape(-1).copy()) if rank == 0 else None, ftype=1)
        se.feed(torch.from_numpy(d2[k * chunk:(k + 1) * chunk].reshape(-1).copy()) if rank == 0 else None, ftype=2)
    se.wait()
    import json
    f = json.load(open(os.path.join(ROOT, "tests", "golden", "full_size.json")))["config3"]
    se.acq_run(f["wrpos"])
    mine, acq = se.acq_fetch()
    se.set_states([dict(carrfreq=0.0, codefreq=c.crate, remcode=0.5, remcarr=0.0, buffloc=100) for c in chans])
    se.trk_start_from_acq()
    se.trk_run(f["epochs"])
    _, II, QQ, ns = se.trk_fetch()
    fin = se.eng.trk_get_state()
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    local = {}
    for j, i in enumerate(mine):
        local[i] = dict(acq=acq[j], ns=ns[j].tolist(), II=sha(II[j]), QQ=sha(QQ[j]), II_last=II[j, -1].tolist(), QQ_last=QQ[j, -1].tolist(),
                        final=fin[j])
    merged = se.gather(local)
    if rank == 0:
        out_q.put(merged)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_config3_46_channels_sharded_over_two_ranks(gc):
    """BASELINE configs[3] through the multi-GPU driver: 32 GPS + 14 GLONASS channels strong-sharded over two ranks (one GPU
    shared on a single-GPU box), both IF streams broadcast from rank 0, acquisition + hand-over + 20 periods on the shards:
    the fixture of the single-process run (tests/golden/full_size.json, config3) channel for channel."""
    import json
    import torch.multiprocessing as mp
    f = json.load(open(os.path.join(ROOT, "tests", "golden", "full_size.json")))["config3"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_config3_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged = q.get(timeout=500)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(merged) == list(range(46))
    for i, v in enumerate(f["channels"]):
        a, g = merged[i]["acq"], v["acq"]
        assert (a["flagacq"], a["iters"], a["acqcodei"], a["freqi"], a["buffloc"]) == (g["flagacq"], g["iters"], g["acqcodei"], g["freqi"], g["buffloc"]), i
        assert abs(a["peakr"] - g["peakr"]) <= 1e-4 * abs(g["peakr"]) and abs(a["cn0"] - g["cn0"]) <= 1e-4 * abs(g["cn0"]), i
        if not v["trk"]:
            continue
        t, m = v["trk"], merged[i]
        assert m["ns"] == t["ns"] and m["II"] == t["II_sha256"] and m["QQ"] == t["QQ_sha256"], i
        assert m["II_last"] == t["II_last"] and m["QQ_last"] == t["QQ_last"], i
        assert m["final"]["remcode"] == t["final"]["remcode"] and m["final"]["remcarr"] == t["final"]["remcarr"], i
        assert m["final"]["buffloc"] == t["final"]["buffloc"], i
