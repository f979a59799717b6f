"""Image sharding across the GPUs of one node.

Every stage of the path is per image, so N ranks = N independent replicas with replicated weights; the only
exchange is the gather of the final detections ``[B_local, K, 6]`` (x1, y1, x2, y2, score, label) -- 7.2 KB per
image at K = 300 -- done with one ``all_gather_into_tensor`` (RCCL when the backend is ``nccl``; the same code
runs on ``gloo`` for the CPU tests).  The reference has no distributed code at all (single process, single
device: export.py:229-233)."""
import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous balanced shard [start, stop) of `n_items` for `rank`: the first n % world ranks get one more."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def pack_detections(boxes, scores, labels):
    """(boxes [B,K,4], scores [B,K], labels [B,K]) -> one fp32 tensor [B,K,6] for the gather."""
    return torch.cat((boxes.float(), scores.float().unsqueeze(-1), labels.float().unsqueeze(-1)), -1)


def unpack_detections(packed):
    return packed[..., :4], packed[..., 4], packed[..., 5].long()


def gather_detections(local, n_items=None, out=None):
    """All ranks contribute their ``[B_local, K, 6]`` block; every rank receives ``[n_items, K, 6]`` in global
    image order.  With uneven shards the blocks are padded to the largest shard for the collective and the
    padding is dropped afterwards.  No-op without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    b_local, k, c = local.shape
    if n_items is None:
        n_items = b_local * world
    b_max = -(-n_items // world)
    if b_local < b_max:
        local = torch.cat((local, local.new_zeros(b_max - b_local, k, c)), 0)
    if out is None or out.shape[0] != b_max * world:
        out = local.new_empty(b_max * world, k, c)
    if local.is_cuda and dist.get_backend() == "gloo":
        # test configuration only (several ranks sharing one GPU, CPU collective): gloo gathers host tensors
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local.contiguous().cpu())
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, local.contiguous())
    if b_max * world == n_items:
        return out
    keep = []
    for r in range(world):
        s, e = shard_range(n_items, r, world)
        keep.append(out[r * b_max: r * b_max + (e - s)])
    return torch.cat(keep, 0)
