"""The N>1 path on CPU: world_size 2, gloo.  Model blob broadcast from rank 0, contiguous shards of the global
synthetic image stream, per-rank class ids (computed here by the ORACLE, standing in for the GPU kernel — this is
a test of the host-side sharding/collective logic only), all-reduced digest == single-process digest."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

import util
from bitnetmcu_amd import dist, synth, Model, DIST_U


def test_shard_ranges_cover_exactly():
    for n in (0, 1, 7, 10**8, 10**8 + 3):
        for world in (1, 2, 3, 8):
            r = [dist.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = util.load_golden_model("fc_4bitsym_64") if rank == 0 else None
        model = dist.broadcast_model(model, src=0)
        blob = model.to_blob()
        first, last = dist.shard_range(n, rank, world)
        x = synth.images(first, last - first, DIST_U)
        cls = util.OracleModel(model).infer(x)
        vec = np.zeros(11, np.uint64)
        vec[0] = synth.class_digest(cls, first)
        vec[1:] = np.bincount(cls, minlength=10)
        t = torch.from_numpy(vec.view(np.int64).copy())
        dist.allreduce_digest(t)
        out_q.put((rank, len(blob), t.numpy().view(np.uint64).tolist()))
    finally:
        td.destroy_process_group()


def test_two_rank_broadcast_shard_allreduce():
    n, world = 3001, 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=300) for _ in range(world))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    model = util.load_golden_model("fc_4bitsym_64")
    cls = util.OracleModel(model).infer(synth.images(0, n, DIST_U))
    want = [synth.class_digest(cls, 0)] + np.bincount(cls, minlength=10).tolist()
    for rank, blob_len, vec in res:
        assert blob_len == len(model.to_blob())
        assert vec == want, rank


def test_combine_digests_is_order_independent():
    rng = np.random.default_rng(1)
    cls = rng.integers(0, 10, 1000).astype(np.uint32)
    whole = synth.class_digest(cls, 0)
    parts = [np.array([synth.class_digest(cls[a:b], a)], np.uint64) for a, b in ((0, 300), (300, 301), (301, 1000))]
    assert int(dist.combine_digests(parts[::-1])[0]) == whole


def test_a_process_group_that_does_not_come_up_ends_with_a_reason_not_a_hang():
    """rank 0 of a two-rank group whose second rank never starts.  init_process_group_or_exit must end the
    process within the timeout with exit code 3 and ONE line on stderr that names the backend and where it was stuck."""
    import subprocess
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = ("import sys; sys.path.insert(0, %r); from bitnetmcu_amd import dist; "
            "dist.init_process_group_or_exit('gloo', 0, 2, timeout_s=4.0); print('came up')" % util.REPO)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
    assert time.time() - t0 < 60
    assert "came up" not in r.stdout
    lines = [l for l in r.stderr.splitlines() if l.startswith("bitnetmcu_amd.dist:")]
    assert len(lines) == 1 and "'gloo' process group did not come up" in lines[0] and "rank 0 of 2" in lines[0], r.stderr[-800:]
