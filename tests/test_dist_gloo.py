"""N > 1 path on CPU: two gloo ranks shard the channels, rank 0 broadcasts the IF chunk, every rank
correlates its shard (the oracle stands in for the kernels here), results are gathered and must
equal the single-process answer.  Exercises erlangnetwork-gnsslib-sdr_amd/multigpu.py, the same
helpers bench.py uses with the nccl (RCCL) backend."""
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
