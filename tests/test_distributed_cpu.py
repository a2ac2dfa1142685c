"""world_size-2 gloo test of the N>1 path: block-row sharding never splits a group and the
gathered output equals the single-process output.  The per-shard encoder here is the CPU
oracle (test infrastructure) -- the GPU run uses the HIP path with the same sharding code."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, torch.distributed as dist
from convectionkernels_amd import sharding, api
from oracle import pyref
import content
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
orc = pyref.OracleLib()
rcp = np.array([1.0] + [1.0 / i for i in range(1, 17)], np.float32)
opt = pyref.make_options()
plan = np.frombuffer(api.BC7EncodingPlan().tobytes(), np.uint8).copy()
def encode(t):
    return torch.from_numpy(orc.encode_bc7(t.numpy(), opt, plan, rcp, 2))
for rows, per_row in ((6, 8), (5, 16), (3, 24), (7, 8)):
    blocks = content.mixed_ldr_blocks(5, rows * per_row // 8 + 1)[:rows * per_row]
    full = sharding.encode_sharded(encode, torch.from_numpy(blocks), rows, per_row, 16)
    if rank == 0:
        np.save(os.path.join(os.environ["OUT_DIR"], "out_%%d_%%d.npy" %% (rows, per_row)), full.numpy())
# the bench's step loop: gather of step i overlapped with the encode of step i + 1, two buffer pairs
outs = [torch.zeros((8, 16), dtype=torch.uint8) for _ in range(2)]
gathered = [torch.zeros((world * 8, 16), dtype=torch.uint8) for _ in range(2)]
log = []
def encode_step(i, out):
    out.fill_(rank * 16 + i)
    log.append(i)
overlapped = sharding.pipelined_gather_steps(5, encode_step, outs, gathered)
expect4 = [r * 16 + 4 for r in range(world) for _ in range(8)]
expect3 = [r * 16 + 3 for r in range(world) for _ in range(8)]
assert log == [0, 1, 2, 3, 4] and gathered[0][:, 0].tolist() == expect4 and gathered[1][:, 0].tolist() == expect3, (gathered[0][:, 0], gathered[1][:, 0])
if rank == 0:
    open(os.path.join(os.environ["OUT_DIR"], "pipeline_ok"), "w").write("overlapped=%%s" %% overlapped)
dist.destroy_process_group()
''' % (ROOT, ROOT)


def test_shard_ranges_are_group_aligned_and_cover():
    from convectionkernels_amd import sharding
    for rows, per_row in ((4096, 4096), (6, 8), (5, 16), (3, 12), (7, 4), (1, 8)):
        for world in (1, 2, 3, 4, 8):
            prev = 0
            for r in range(world):
                lo, hi = sharding.shard_block_rows(rows, per_row, r, world)
                assert lo == prev and lo % 8 == 0 and hi >= lo
                prev = hi
            assert prev == rows * per_row
    # 16384^2 over 8 GPUs: 512 block rows of 4096 blocks each per rank
    assert sharding.shard_block_rows(4096, 4096, 3, 8) == (3 * 512 * 4096, 4 * 512 * 4096)


def test_two_rank_gloo_matches_single_process(tmp_path, oracle_lib):
    import content
    from convectionkernels_amd import api
    from oracle import pyref
    env = dict(os.environ, OUT_DIR=str(tmp_path), MASTER_ADDR="127.0.0.1")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29731", str(script)], env=env,
                          timeout=600)
    rcp = np.array([1.0] + [1.0 / i for i in range(1, 17)], np.float32)
    plan = np.frombuffer(api.BC7EncodingPlan().tobytes(), np.uint8).copy()
    for rows, per_row in ((6, 8), (5, 16), (3, 24), (7, 8)):
        blocks = content.mixed_ldr_blocks(5, rows * per_row // 8 + 1)[:rows * per_row]
        n = blocks.shape[0] // 8 * 8
        exp = oracle_lib.encode_bc7(blocks[:n], pyref.make_options(), plan, rcp, 4)
        got = np.load(tmp_path / ("out_%d_%d.npy" % (rows, per_row)))
        assert got.shape[0] == rows * per_row
        assert (got[:n] == exp).all()
    assert open(tmp_path / "pipeline_ok").read() == "overlapped=True"


SUBGROUP_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from convectionkernels_amd import sharding
dist.init_process_group("gloo")
rank = dist.get_rank()
g = dist.new_group([1, 2])  # group rank 0 = global rank 1: P2POp peers are GLOBAL ranks
if rank in (1, 2):
    gr = dist.get_rank(g)
    ranges = [(0, 8), (8, 24)]
    lo, hi = ranges[gr]
    local = torch.full((hi - lo, 16), 10 + gr, dtype=torch.uint8)
    full = torch.zeros((24, 16), dtype=torch.uint8) if gr == 0 else None
    log = []
    sharding.pipelined_steps(3, lambda i, buf: None,
                             lambda i, buf: sharding.gather_to_root(local, ranges, full, root=0, group=g, async_op=True),
                             after_exchange=lambda i, buf: log.append((i, buf)))
    assert log == [(0, 0), (1, 1), (2, 0)], log
    if gr == 0:
        assert full[:8].eq(10).all() and full[8:].eq(11).all(), full[:, 0]
        open(os.path.join(os.environ["OUT_DIR"], "subgroup_ok"), "w").write("ok")
dist.barrier()
dist.destroy_process_group()
''' % (ROOT,)


def test_gather_to_root_inside_a_subgroup(tmp_path):
    """gather_to_root with group != None: peers are translated to global ranks (three gloo ranks, group = ranks 1 and 2),
    and pipelined_steps reports every finished exchange to `after_exchange` in order"""
    env = dict(os.environ, OUT_DIR=str(tmp_path), MASTER_ADDR="127.0.0.1")
    script = tmp_path / "sub.py"
    script.write_text(SUBGROUP_WORKER)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
                           "--master-addr", "127.0.0.1", "--master-port", "29733", str(script)], env=env, timeout=600)
    assert open(tmp_path / "subgroup_ok").read() == "ok"


EIGHT_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from convectionkernels_amd import sharding
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
assert world == 8
def pattern(lo, hi, step):
    # 16 bytes per block: a function of the GLOBAL block number and the step, so a misplaced, stale or missing shard shows
    i = torch.arange(lo, hi, dtype=torch.int32)[:, None]
    k = torch.arange(16, dtype=torch.int32)[None, :]
    return (i * (2 * k + 3) + (i >> 9) + 17 * step + k).to(torch.uint8)
# (block rows, blocks per row): the real table of BASELINE config 5 (16384^2 pixels = 4096 x 4096 blocks, 2 097 152 blocks =
# 32 MiB of packed output per rank, 224 MiB into rank 0 per step) and ragged tables: fewer rows than ranks, rows that are
# not a multiple of the group size, an empty shard
for rows, per_row, steps in ((4096, 4096, 2), (37, 24, 3), (5, 12, 3), (3, 8, 2), (4093, 20, 2)):
    ranges = sharding.shard_ranges(rows, per_row, world)
    total = rows * per_row
    assert ranges[0][0] == 0 and ranges[-1][1] == total and all(a %% 8 == 0 for a, _ in ranges)
    lo, hi = ranges[rank]
    full = [torch.zeros((total, 16), dtype=torch.uint8) for _ in range(2)] if rank == 0 else [None, None]
    outs = [f[lo:hi] for f in full] if rank == 0 else [torch.zeros((hi - lo, 16), dtype=torch.uint8) for _ in range(2)]
    seen = []
    def encode_step(i, buf):
        outs[buf].copy_(pattern(lo, hi, i))
    def exchange(i, buf):
        return sharding.gather_to_root(outs[buf], ranges, full[buf], root=0, async_op=True)
    def after(i, buf):
        if rank == 0:
            seen.append((i, bool((full[buf] == pattern(0, total, i)).all())))
    sharding.pipelined_steps(steps, encode_step, exchange, after_exchange=after)
    if rank == 0:
        assert seen == [(i, True) for i in range(steps)], (rows, per_row, seen)
dist.barrier()
if rank == 0:
    open(os.path.join(os.environ["OUT_DIR"], "eight_ok"), "w").write("ok")
dist.destroy_process_group()
''' % (ROOT,)


def test_eight_rank_gather_with_the_config5_shard_table(tmp_path):
    """World size 8 (gloo): the exchange of `bench.py --gpus 8` with the real shard table of the 16384^2 image -- every rank
    sends its 32 MiB of packed blocks to rank 0, two buffer sets, the gather of step i checked after step i + 1 was
    queued -- and with ragged tables (fewer block rows than ranks, an empty shard, rows that are no multiple of 8 blocks)"""
    env = dict(os.environ, OUT_DIR=str(tmp_path), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    script = tmp_path / "eight.py"
    script.write_text(EIGHT_WORKER)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                           "--master-addr", "127.0.0.1", "--master-port", "29735", str(script)], env=env, timeout=900)
    assert open(tmp_path / "eight_ok").read() == "ok"
