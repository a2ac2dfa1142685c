"""Block-row sharding of a tiled image across the GPUs of one node.

Units are independent *groups* of 8 blocks, so the search needs no collective; the only
exchange is gathering the packed output.  One process per GPU; `torch.distributed` with the
`nccl` backend is RCCL over xGMI on ROCm (gloo on CPU for the tests).
"""
import torch
import torch.distributed as dist

GROUP = 8


def shard_block_rows(block_rows, blocks_per_row, rank, world):
    """[first_block, last_block) of rank's shard: whole block rows, group aligned.

    Rows are dealt in contiguous ranges; when `blocks_per_row` is not a multiple of 8 the
    boundary is moved to the next group boundary so that no reference group is split."""
    total = block_rows * blocks_per_row
    lo = (block_rows * rank // world) * blocks_per_row
    hi = (block_rows * (rank + 1) // world) * blocks_per_row
    lo = min(total, (lo + GROUP - 1) // GROUP * GROUP)
    hi = total if rank == world - 1 else min(total, (hi + GROUP - 1) // GROUP * GROUP)
    return lo, hi


def encode_sharded(encode, blocks, block_rows, blocks_per_row, bytes_per_block, group=None):
    """Every rank encodes its shard of `blocks` (a tensor holding the WHOLE tiled image, or at
    least this rank's range) with `encode(tensor[n,...]) -> uint8 tensor[n, bytes_per_block]`
    and the packed blocks of all ranks are gathered on every rank.  Returns the full output."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_block_rows(block_rows, blocks_per_row, rank, world)
    local = encode(blocks[lo:hi]) if hi > lo else torch.empty((0, bytes_per_block), dtype=torch.uint8, device=blocks.device)
    if world == 1:
        return local
    ranges = [shard_block_rows(block_rows, blocks_per_row, r, world) for r in range(world)]
    sizes = [b - a for a, b in ranges]
    if len(set(sizes)) == 1:
        out = torch.empty((sum(sizes), bytes_per_block), dtype=torch.uint8, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    # ragged shards: pad to the largest, gather once, trim
    m = max(sizes)
    padded = torch.zeros((m, bytes_per_block), dtype=torch.uint8, device=local.device)
    padded[:local.shape[0]] = local
    out = torch.empty((world * m, bytes_per_block), dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * m:r * m + sizes[r]] for r in range(world)])
