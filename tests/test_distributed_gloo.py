"""CPU, world_size 2, gloo: the row-sharding / exchange helpers of the N>1 path."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from autognothi_amd import distributed as D
    try:
        n_inputs, k, p = 5, 4, 7
        masks = torch.arange(n_inputs * k * p).reshape(n_inputs * k, p)
        mine, lo, hi = D.shard_rows(masks, n_inputs, k)
        assert mine.shape[0] == (hi - lo) * k and torch.equal(mine, masks[lo * k:hi * k])
        counts = [(D.shard_range(n_inputs, r, world)[1] - D.shard_range(n_inputs, r, world)[0]) * k for r in range(world)]
        v_local = mine.float() * 2.0
        v_all = D.gather_rows(v_local, counts)
        assert torch.equal(v_all, masks.float() * 2.0)
        # gradient all-reduce in buckets: two params, tiny bucket to force >1 collective
        a = torch.nn.Parameter(torch.zeros(10)); b = torch.nn.Parameter(torch.zeros(3, 5))
        a.grad = torch.full((10,), float(rank + 1)); b.grad = torch.full((3, 5), float(10 * (rank + 1)))
        calls = D.allreduce_grads([a, b], average=True, bucket_bytes=48)
        assert calls == 2
        assert torch.allclose(a.grad, torch.full((10,), 1.5)) and torch.allclose(b.grad, torch.full((3, 5), 15.0))
        tot = D.reduce_scalars([1.0 + rank, 2.0], torch.device("cpu"))
        assert tot == [3.0, 4.0]
        # fewer inputs than ranks: the K masks of every input are split across ranks (SURVEY 8e fallback)
        n1, k1 = 1, 5
        m1 = torch.arange(n1 * k1 * p).reshape(n1 * k1, p)
        rows, mode, klo, khi = D.shard_auto(m1, n1, k1)
        assert mode == "mask" and rows.shape[0] == n1 * (khi - klo)
        assert torch.equal(rows, m1.view(n1, k1, p)[:, klo:khi].reshape(-1, p))
        back = D.gather_masks_within_inputs(rows.float() * 3.0, n1, k1)
        assert torch.equal(back, m1.float() * 3.0)
        n2, k2 = 3, 4   # several inputs, still split by mask: interleaving back into input-major order
        m2 = torch.arange(n2 * k2 * 2).reshape(n2 * k2, 2)
        r2, lo2, hi2 = D.shard_masks_within_inputs(m2, n2, k2)
        assert torch.equal(D.gather_masks_within_inputs(r2, n2, k2), m2)
        # the Shapley-loss partial sums over k are additive across ranks
        part = float((r2.double() ** 2).sum())
        assert abs(D.reduce_scalars([part], torch.device("cpu"))[0] - float((m2.double() ** 2).sum())) < 1e-9
        assert D.shard_auto(masks, n_inputs, k)[1] == "input"
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Array("i", [0] * world)
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert list(out) == [1, 1]


def test_shard_range_covers_everything():
    from autognothi_amd import distributed as D
    for n in (1, 7, 8, 9, 64):
        for w in (1, 2, 8):
            spans = [D.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
