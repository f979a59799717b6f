"""CPU: the N>1 path -- shard assignment and the detections gather -- on world_size-2 (and 3) gloo."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_shard_range_partitions_exactly():
    from codetr.sharding import shard_range

    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _fake_detections(idx, k=5):
    """deterministic per-image block so the gathered order can be verified"""
    base = torch.arange(k * 6, dtype=torch.float32).view(k, 6)
    return torch.stack([base + 1000.0 * i for i in idx]) if len(idx) else torch.zeros(0, k, 6)


def _worker(rank, world, port, n_items, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "co-detr-tensorrt_amd"))
    from codetr.sharding import gather_detections, pack_detections, shard_range, unpack_detections

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s, e = shard_range(n_items, rank, world)
        local = _fake_detections(list(range(s, e)))
        boxes, scores, labels = unpack_detections(local)
        packed = pack_detections(boxes, scores, labels.long())
        full = gather_detections(packed, n_items)
        ok = torch.equal(full, _fake_detections(list(range(n_items))))
        # timing reduction used by bench.py: MAX over ranks
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, bool(ok), float(t.item()), tuple(full.shape)))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n_items", [(2, 8), (2, 5), (3, 7)])
def test_gather_detections_gloo(world, n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, tmax, shape in results:
        assert ok, f"rank {rank} gathered a wrong tensor"
        assert tmax == float(world) and shape == (n_items, 5, 6)


def test_single_process_is_identity():
    from codetr.sharding import gather_detections

    x = torch.randn(3, 300, 6)
    assert gather_detections(x) is x
