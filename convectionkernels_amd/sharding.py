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


def pipelined_gather_steps(num_steps, encode_step, outs, gathered, group=None):
    """Run `num_steps` encode steps whose packed output is gathered on every rank, with the gather of step i overlapping
    the encode of step i + 1: `encode_step(i, out)` fills `out` (= outs[i & 1]) on the current stream, the all-gather
    into gathered[i & 1] is issued asynchronously on the backend's stream, and a buffer pair is reused only after the
    gather that read it has completed.  Falls back to blocking gathers on a backend without asynchronous ones.
    Returns True when the gathers were overlapped.  With one process it just runs the steps."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    pending = []
    overlapped = world > 1
    for i in range(num_steps):
        buf = i & 1
        if len(pending) >= 2:
            pending.pop(0).wait()  # the current stream waits for the gather that last read outs[buf]
        encode_step(i, outs[buf])
        if world > 1:
            if overlapped:
                try:
                    pending.append(dist.all_gather_into_tensor(gathered[buf], outs[buf], group=group, async_op=True))
                    continue
                except Exception:  # noqa -- no asynchronous gathers: serialise
                    overlapped = False
            dist.all_gather_into_tensor(gathered[buf], outs[buf], group=group)
    for work in pending:
        work.wait()
    return overlapped


def shard_ranges(block_rows, blocks_per_row, world):
    """[first_block, last_block) of every rank's shard."""
    return [shard_block_rows(block_rows, blocks_per_row, r, world) for r in range(world)]


def gather_to_root(local, ranges, full, root=0, group=None, async_op=False):
    """The one exchange of the path: the packed blocks of every shard go to `root` -- grouped point-to-point
    sends (ncclSend / ncclRecv inside one group with the `nccl` = RCCL backend, each peer over its own xGMI link; no ring,
    nothing travels to the other ranks).  `root` receives every shard straight into its slice full[lo:hi] of the output;
    its own shard is copied there unless `local` already is that slice.  Ragged shards need no padding.
    Returns the list of pending works when `async_op` (wait on them before reusing `local` / reading `full`)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = ranges[rank]
    if world == 1:
        if full is not None and hi > lo and local.data_ptr() != full[lo:hi].data_ptr():
            full[lo:hi].copy_(local)
        return []
    # P2POp takes GLOBAL ranks as peers; `ranges` and `root` are indexed by the rank inside `group`
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    ops = []
    if rank == root:
        for r, (a, b) in enumerate(ranges):
            if r != root and b > a:
                ops.append(dist.P2POp(dist.irecv, full[a:b], peer(r), group))
        if hi > lo and local.data_ptr() != full[lo:hi].data_ptr():
            full[lo:hi].copy_(local)
    elif hi > lo:
        ops.append(dist.P2POp(dist.isend, local, peer(root), group))
    works = dist.batch_isend_irecv(ops) if ops else []
    if async_op:
        return works
    for w in works:
        w.wait()
    return []


def pipelined_steps(num_steps, encode_step, exchange, after_exchange=None):
    """`num_steps` steps of encode + exchange with two buffer sets: `encode_step(i, buf)` fills buffer set `buf` = i & 1
    on the current stream, `exchange(i, buf)` starts the exchange that reads it and returns its pending works; a buffer
    set is reused only after the exchange that read it has completed, so the exchange of step i overlaps the search of
    step i + 1.  `after_exchange(i, buf)` (optional) is called once step i's exchange has been waited for -- before the
    buffer set is handed to step i + 2 -- e.g. to check what arrived.  All works are waited for before returning."""
    pending = []
    for i in range(num_steps):
        buf = i & 1
        if len(pending) >= 2:
            j, works = pending.pop(0)
            for w in works:
                w.wait()
            if after_exchange is not None:
                after_exchange(j, j & 1)
        encode_step(i, buf)
        pending.append((i, exchange(i, buf)))
    for j, works in pending:
        for w in works:
            w.wait()
        if after_exchange is not None:
            after_exchange(j, j & 1)
