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


def _sharded_worker(rank, world, port, use_gpu, out_q):
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
    merged = _drive(se, torch, stream, len(PRNS), NCHUNK - 1)
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


def _run_two_ranks(use_gpu):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, use_gpu, q)) for r in range(2)]
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
