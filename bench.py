#!/usr/bin/env python3
"""Benchmark of the BC7 hot path (BASELINE.json).  Rank 0 prints ONE short JSON line (< 4 KB: the contract's fields, `roofline`,
`cpu_baseline`, one entry per other BASELINE config) as the last line of stdout and writes every leg in full to bench_detail.json.

    python bench.py --gpus N --steps K --warmup W
    N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
                bench.py --gpus N --steps K --warmup W
            (`python bench.py --gpus N` without a launcher starts exactly that itself)

N = 1  -- BASELINE configs[1]: EncodeBC7, BC7EncodingPlan() + Options(), 4096x4096 SplitMix64 random RGBA (seed 2),
          PixelBlocks resident in HBM.  A step = one encode of the image.  Besides the headline the line carries
          `roofline` (HIP events on the launch stream), `sustained` (>= 3 s of back-to-back encodes + the shader clock),
          `cpu_baseline` (the real reference from oracle/_ref on this box's host cores, 1 thread and all threads, driven by
          std::thread, and the count of GPU != CPU blocks), `configs` (one measurement per other BASELINE config: 1, 2b, 3,
          4, 5a, 5b), `content_families` (the same kernel on eight kinds of content) and `host_path` (the host-pointer
          entry point, PCIe included).
N > 1  -- BASELINE configs[4]: ONE 16384x16384 image (seed 5), block rows dealt to the ranks, every rank
          generates and encodes only its shard, the packed blocks are gathered on rank 0 (grouped RCCL send/recv over
          xGMI, overlapped with the next step's search) and rank 0 checks the SHA-256 of the whole output against the
          reference's.  "scaling": "strong".  (`--workload config5` runs the same image on one GPU.)
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
# algorithmic (compulsory) bytes per block, SURVEY.md 8(d): PixelBlock in + packed block out
ALGO_BYTES = {"bc7": 80, "bc1": 72, "bc6hu": 144, "etc2rgba": 80}
# VALU issue peak: MI355X_MICROARCH.md ("Wave scheduling"): a wave64 VALU instruction occupies its SIMD for 2 cycles, i.e.
# 256 CUs x 4 SIMDs x clock / 2 wave-instructions per second.  Measured (tools/valu_fma_probe.hip, profiles/r06/
# valu_peak_reconciled.md): 2.2-2.3 cycles for f32 add / sub / mul / fma whose sources sit in different VGPR banks; min / max,
# 24-bit multiplies, conversions, v_perm, dot products, packed f32 and any three-source instruction with two sources in one
# bank take 4.3 -- so no kernel with such instructions can reach frac = 1, and the line does not pretend to know how close it could get.
VALU_CYCLES_PER_INST = 2.0
SHADER_CLOCK_HZ = 2.4e9


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=("auto", "config2", "config5"), default="auto",
                    help="auto: config2 (4096^2, BASELINE configs[1]) on one GPU, config5 (one 16384^2 image, block-row sharded) on several")
    ap.add_argument("--size", type=int, default=0, help="image edge in pixels (default: the workload's)")
    ap.add_argument("--opaque", action="store_true", help="variant 2b: alpha forced to 255 (modes 0-3 run)")
    ap.add_argument("--no-cpu", action="store_true", help="skip every CPU leg")
    ap.add_argument("--no-extra", action="store_true", help="headline only (profiling runs): no sustained / configs / families / host path")
    ap.add_argument("--exhaustive", action="store_true",
                    help="evaluate every candidate like the reference (default: exact branch-and-bound, same output)")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the multi-rank plumbing (gloo, a stand-in encoder, no throughput reported); tests only")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# launching
# ---------------------------------------------------------------------------------------------------------------------
def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run with N ranks."""
    if not args.dry_run:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d requested but %d GPU(s) visible; refusing to run a smaller job under that name\n" % (args.gpus, have))
            sys.exit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def golden_hashes():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "config_hashes.json")))


def golden_rcp(h):
    return np.array(h["rcp_hex"], np.uint32).view(np.float32)


def sha256(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()



# ---------------------------------------------------------------------------------------------------------------------
# output: ONE short JSON line (the driver's parser reads the last stdout line; round 5's 21 KB line was not parsed) + a
# side file with every leg in full
# ---------------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 4096
DTYPE = "f32+u16"  # bit-exact emulation of the reference's SSE2 lanes (binary32 + 16-bit wrapping integers)


def _r(x, digits=5):
    """floats to `digits` significant figures (the side file keeps them in full)"""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    return x


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if isinstance(d, dict) and k in d}


def compact_line(full):
    """The driver-facing line: the contract's fields, `roofline`, `cpu_baseline`, and one short entry per other BASELINE
    config -- nothing else.  Always shorter than LINE_LIMIT (tests/test_bench_cli.py checks it on a maximal synthetic result)."""
    out = _pick(full, ("metric", "value", "unit", "gpixel_per_s", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                       "scaling", "vs_baseline", "dtype", "data"))
    cfg = full.get("config", {})
    out["config"] = {k: cfg[k] for k in ("workload", "blocks", "blocks_per_rank", "exchange", "search") if k in cfg}
    roof = full.get("roofline")
    if roof:
        r = _pick(roof, ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "valu_insts_per_block",
                         "avg_waves_per_simd", "counters_from"))
        hbm = roof.get("hbm", roof)
        r["hbm"] = _pick(hbm, ("achieved", "peak", "unit", "frac", "traffic", "traffic_bytes_per_launch", "traffic_over_algorithmic"))
        if "note" in roof and roof["bound"] == "hbm":
            r["note"] = roof["note"][:120]
        out["roofline"] = r
    cpu = full.get("cpu_baseline")
    if cpu:
        c = _pick(cpu, ("value", "unit", "cores", "kind", "gpu_mismatching_blocks", "blocks_checked"))
        c["sample"] = str(cpu.get("sample", ""))[:200]
        if "one_thread" in cpu:
            c["one_thread"] = _pick(cpu["one_thread"], ("value", "blocks"))
        if "host" in cpu and "cpu_model" in cpu["host"]:
            c["cpu_model"] = cpu["host"]["cpu_model"][:48]
        out["cpu_baseline"] = c
    for k in ("bit_exact_vs_cpu", "scale_base_mblocks_s", "rank0_search_mblocks_s", "dry_run", "value_note", "profile_note"):
        if k in full:
            out[k] = _r(full[k])
    if "output_check" in full:
        out["output_check"] = _pick(full["output_check"], ("matches_reference", "matches_single_process", "steps_checked", "every_step_identical"))
    if "sustained" in full:
        out["sustained_mblocks_s"] = _r(full["sustained"]["mblocks_s"])
    if "mblocks_s" in full.get("two_streams", {}):
        out["two_streams_mblocks_s"] = _r(full["two_streams"]["mblocks_s"])
    if "exhaustive_search" in full:
        out["exhaustive_identical_output"] = full["exhaustive_search"].get("identical_output")
    if "configs" in full:
        out["configs"] = {}
        for name, e in full["configs"].items():
            s = {"mblocks_s": _r(e["mblocks_s"], 4), "frac": _r(e["roofline"].get("frac"), 3), "bound": e["roofline"].get("bound")}
            if "sha256_matches_reference" in e:
                s["sha_ok"] = e["sha256_matches_reference"]
            if "cpu_baseline" in e:
                s["cpu_mblocks_s"] = _r(e["cpu_baseline"]["value"], 3)
                s["cpu_mismatch"] = e["cpu_baseline"]["gpu_mismatching_blocks"]
            out["configs"][name] = s
    for fam in ("content_families", "bc6h_content_families"):
        if fam in full:
            out[fam] = {k: _r(v["mblocks_s"], 4) for k, v in full[fam].items()}
            bad = sum(v.get("mismatches_vs_cpu", 0) for v in full[fam].values())
            out[fam + "_mismatches"] = bad
    if "detail_file" in full:
        out["detail"] = full["detail_file"]
    line = json.dumps(out, separators=(",", ":"))
    # belt and braces: drop the optional parts, last first, rather than ever print a line the driver cannot read
    for k in ("bc6h_content_families", "content_families", "configs", "output_check"):
        if len(line) < LINE_LIMIT:
            break
        out.pop(k, None)
        out.pop(k + "_mismatches", None)
        line = json.dumps(out, separators=(",", ":"))
    assert len(line) < LINE_LIMIT, len(line)
    return line


def emit(full):
    """Write every leg in full to bench_detail.json (next to this script, and under gpurun_out/ when that exists), then print the
    short line as the LAST line of stdout."""
    names = [os.path.join(ROOT, "bench_detail.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        names.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    full["detail_file"] = "bench_detail.json"
    for n in names:
        try:
            with open(n, "w") as f:
                json.dump(full, f, indent=1)
        except OSError as e:
            sys.stderr.write("bench.py: cannot write %s: %r\n" % (n, e))
    sys.stdout.flush()
    print(compact_line(full), flush=True)

# ---------------------------------------------------------------------------------------------------------------------
# N > 1 (and --workload config5): one image, block-row sharded, gather to rank 0
# ---------------------------------------------------------------------------------------------------------------------
def stand_in_encoder(blocks_u8):
    """--dry-run only: a deterministic function of each PixelBlock (NOT an encoder) so the shard / gather plumbing can
    be checked end to end on CPU."""
    b = blocks_u8.reshape(blocks_u8.shape[0], 64).to(dtype=__import__("torch").int32)
    return ((b[:, 0:16] * 3 + b[:, 16:32] * 5 + b[:, 32:48] * 7 + b[:, 48:64] * 11) & 0xFF).to(dtype=__import__("torch").uint8)


def preflight(torch, dist, dev, rank, world, backend):
    """Before anything is timed: one 1 MiB grouped send/recv round to rank 0 -- the exchange pattern of the timed steps
    (sharding.gather_to_root: batch_isend_irecv, every peer over its own link) -- with the payload checked, and an
    all-reduce of the verdict.  A broken link, a rank on the wrong device or an IPC problem (HSA_ENABLE_IPC_MODE_LEGACY)
    stops the run here with a message instead of a hang or a wrong hash twenty steps later."""
    from convectionkernels_amd import sharding
    n = 1 << 16  # 64 Ki blocks x 16 B = 1 MiB per rank
    ranges = [(r * n, (r + 1) * n) for r in range(world)]
    local = torch.full((n, 16), (rank * 29 + 7) & 0xFF, dtype=torch.uint8, device=dev)
    full = torch.zeros((world * n, 16), dtype=torch.uint8, device=dev) if rank == 0 else None
    ok = 1
    t0 = time.perf_counter()
    try:
        sharding.gather_to_root(local, ranges, full, root=0)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        if rank == 0:
            want = torch.tensor([(r * 29 + 7) & 0xFF for r in range(world)], dtype=torch.uint8, device=dev)
            ok = int(bool((full.view(world, -1) == want[:, None]).all()))
    except Exception as e:  # noqa
        sys.stderr.write("bench.py preflight: rank %d: %s exchange failed: %r\n" % (rank, backend, e))
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) != 1:
        if rank == 0:
            sys.stderr.write("bench.py preflight: the 1 MiB %s gather to rank 0 over %d ranks failed or delivered wrong bytes; "
                             "nothing was timed (check HSA_ENABLE_IPC_MODE_LEGACY=0, one GPU per rank, MASTER_ADDR=127.0.0.1)\n" % (backend, world))
        dist.destroy_process_group()
        sys.exit(3)
    if rank == 0:
        sys.stderr.write("bench.py preflight: %s gather of %d x 1 MiB to rank 0 ok (%.0f ms incl. communicator set-up)\n"
                         % (backend, world - 1, (time.perf_counter() - t0) * 1e3))


def run_sharded(args):
    import torch
    import torch.distributed as dist
    from convectionkernels_amd import sharding, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks\n" % (args.gpus, world))
        sys.exit(2)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.dry_run:
        dev = torch.device("cpu")
        if world > 1:
            dist.init_process_group("gloo")
    else:
        if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
            sys.stderr.write("bench.py: rank %d has no GPU %d\n" % (rank, local_rank))
            sys.exit(2)
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if world > 1:
            dist.init_process_group("nccl", device_id=dev)
    n_ranks = dist.get_world_size() if world > 1 else 1
    assert n_ranks == args.gpus
    if n_ranks > 1:
        preflight(torch, dist, dev, rank, n_ranks, "gloo" if args.dry_run else "RCCL")

    size = args.size or 16384
    seed = 5
    block_rows, blocks_per_row = size // 4, size // 4
    ranges = sharding.shard_ranges(block_rows, blocks_per_row, n_ranks)
    lo, hi = ranges[rank]
    total = block_rows * blocks_per_row
    # every rank generates only its own pixel rows of the image (SplitMix64 is counter based)
    row0, row1 = lo // blocks_per_row * 4, hi // blocks_per_row * 4
    assert lo % blocks_per_row == 0 and hi % blocks_per_row == 0, "config widths are multiples of 32 pixels: whole block rows"
    shard = synth.tile_blocks(synth.image_rgba8_rows(seed, size, size, row0, row1, opaque=args.opaque))
    d_in = torch.from_numpy(shard).to(dev)
    nloc = hi - lo

    h = golden_hashes()
    if args.dry_run:
        ctx = None
        encode = lambda out: out.copy_(stand_in_encoder(d_in))
    else:
        from convectionkernels_amd import api
        ctx = api.Context(dev.index)
        ctx.set_exhaustive(args.exhaustive)
        # the whole-image hash was made with the generating host's RCPPS table: use that table on every rank
        ctx.set_rcp_table(golden_rcp(h))
        opt, plan = api.Options(), api.BC7EncodingPlan()
        encode = lambda out: ctx.encode_bc7(d_in, opt, plan, out=out)

    # two buffer sets; rank 0 encodes straight into its slice of the gathered image
    if rank == 0:
        full = [torch.empty((total, 16), dtype=torch.uint8, device=dev) for _ in range(2)]
        outs = [f[lo:hi] for f in full]
    else:
        full = [None, None]
        outs = [torch.empty((nloc, 16), dtype=torch.uint8, device=dev) for _ in range(2)]

    def sync():
        if n_ranks > 1:
            dist.barrier()
        if not args.dry_run:
            torch.cuda.synchronize()

    def exchange(i, buf):
        return sharding.gather_to_root(outs[buf], ranges, full[buf], root=0, async_op=True)

    # Every timed step's gathered image is checked, not only the last one: once the gather of step i has landed, rank 0
    # takes one wrap-around 64-bit sum per shard slice of the gathered buffer (a misplaced, missing or stale shard changes
    # its slice's sum) into row i of `sums`; after the run every row must equal the last step's, whose whole 256 MiB are
    # SHA-256-checked against the reference's.  On the GPU the sums run on a side stream next to the following step's search.
    sums = torch.zeros((max(1, args.steps), n_ranks), dtype=torch.int64, device=dev) if rank == 0 else None
    side = None if (args.dry_run or rank != 0) else torch.cuda.Stream(device=dev)
    side_done = [None, None]

    def check_step(i, buf):
        if rank != 0:
            return
        def take():
            for r, (a, b) in enumerate(ranges):
                if b > a:
                    sums[i, r] = full[buf][a:b].reshape(-1).view(torch.int64).sum()
        if side is None:
            take()
            return
        side.wait_stream(torch.cuda.current_stream())  # the gather's completion was queued on the current stream
        with torch.cuda.stream(side):
            take()
            ev = torch.cuda.Event()
            ev.record(side)
        side_done[buf] = ev

    def before_reuse(buf):
        if side_done[buf] is not None:
            torch.cuda.current_stream().wait_event(side_done[buf])
            side_done[buf] = None

    if not args.dry_run:
        preheat(torch, lambda: encode(outs[0]))  # untimed: the shader clock has ramped when the warm-up steps start
    sharding.pipelined_steps(args.warmup, lambda i, buf: encode(outs[buf]), exchange)
    evs = None if args.dry_run else [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def encode_step(i, buf):
        before_reuse(buf)
        if evs:
            evs[i][0].record()
        encode(outs[buf])
        if evs:
            evs[i][1].record()

    sync()
    t0 = time.perf_counter()
    sharding.pipelined_steps(args.steps, encode_step, exchange, after_exchange=check_step)
    if side is not None:
        torch.cuda.current_stream().wait_stream(side)
    sync()
    elapsed = time.perf_counter() - t0
    if n_ranks > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    result = None
    if rank == 0:
        last = full[(args.steps - 1) & 1] if args.steps > 0 else full[(args.warmup - 1) & 1]
        got = last.cpu().numpy()
        digest = sha256(got)
        check = {"sha256": digest}
        if args.steps > 0:
            sm = sums.cpu().numpy()
            same = int((sm == sm[args.steps - 1]).all(axis=1).sum())
            check["steps_checked"] = int(args.steps)
            check["steps_equal_to_last"] = same
            check["every_step_identical"] = same == args.steps
            check["per_step_check"] = "64-bit sum per shard slice of every step's gathered image, compared with the last step's (which is hashed whole)"
        if args.dry_run:
            whole = torch.from_numpy(synth.tile_blocks(synth.image_rgba8(seed, size, size, opaque=args.opaque)))
            check["matches_single_process"] = bool((stand_in_encoder(whole).numpy() == got).all())
        elif size == 16384 and not args.opaque:
            check["reference"] = h["config5_bc7_16384_seed5"]
            check["matches_reference"] = digest == h["config5_bc7_16384_seed5"]
        k_ms = None if evs is None else float(np.mean([a.elapsed_time(b) for a, b in evs]))
        mblocks = total * args.steps / elapsed / 1e6 if args.steps else 0.0
        result = {
            "metric": "bc7_encode_default_plan_throughput",
            "value": None if args.dry_run else mblocks,
            "unit": "Mblocks/s",
            "gpixel_per_s": None if args.dry_run else mblocks * 16.0 / 1e3,
            "n_gpus": n_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(1, args.steps) * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": DTYPE,
            "data": "synthetic",
            "config": {
                "workload": "EncodeBC7, BC7EncodingPlan() + Options(), ONE %dx%d SplitMix64 random RGBA image%s, seed 5, %d blocks, block rows "
                            "[k*%d/N, (k+1)*%d/N) on rank k (BASELINE configs[4], variant 5a)" % (size, size, " alpha=255" if args.opaque else "", total, block_rows, block_rows),
                "blocks_per_rank": [b - a for a, b in ranges],
                "exchange": "gather of the packed blocks to rank 0 (%s grouped send/recv), overlapped with the next step's search" % ("gloo" if args.dry_run else "RCCL") if n_ranks > 1 else "none",
                "backend": "none" if n_ranks == 1 else dist.get_backend(),
                "search": "exhaustive" if args.exhaustive else "exact branch-and-bound (output bit-identical)",
            },
            "output_check": check,
        }
        if args.dry_run:
            result["dry_run"] = True
        else:
            # rank 0's launch: the per-block counters of the headline profile (same kernel, same kind of content) when that
            # profile belongs to the loaded library
            pmc = profiled_counters(api.library_source_sha256())
            if pmc and not pmc.get("stale"):
                result["roofline"] = roofline_block("bc7", nloc, k_ms, "cvttmi_bc7_kernel", insts_per_block=pmc["valu_insts_per_wave"] / 16.0,
                                                    hbm_bytes_per_block=pmc["hbm_bytes_per_launch"] / float(pmc["blocks"]),
                                                    waves_per_simd=pmc.get("avg_waves_per_simd"), source=pmc["source"])
            else:
                result["roofline"] = roofline_block("bc7", nloc, k_ms, "cvttmi_bc7_kernel")
            result["roofline"]["note"] = "rank 0's shard (%d blocks) per launch; per-block counters of the 4096^2 profile of the same kernel" % nloc
            result["rank0_search_mblocks_s"] = nloc / k_ms / 1e3
            if not args.no_cpu:
                # the CPU path on a bounded sample of rank 0's shard (the golden RCPPS table is in use on every rank)
                opt_b = np.frombuffer(api.Options().tobytes(), np.uint8).copy()
                plan_b = np.frombuffer(api.BC7EncodingPlan().tobytes(), np.uint8).copy()
                result["cpu_baseline"] = cpu_baseline("bc7", shard, got[lo:hi], opt_b, plan_b, golden_rcp(h), budget_one=2.0, budget_all=6.0)
                result["cpu_baseline"]["host"] = host_info()
                result["bit_exact_vs_cpu"] = result["cpu_baseline"]["gpu_mismatching_blocks"] == 0
        emit(result)
    if n_ranks > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


# ---------------------------------------------------------------------------------------------------------------------
# N = 1: BASELINE configs[1] + the per-config, sustained, content and host-path legs
# ---------------------------------------------------------------------------------------------------------------------
def usable_cores():
    """host cores this process may actually use: the affinity set, capped by the cgroup CPU quota (a container with 256
    visible CPUs and cpu.max = "1600000 100000" gets 16 cores' worth of time, however many threads it starts)"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def host_info():
    info = {"usable_cores": usable_cores(), "affinity_cpus": len(os.sched_getaffinity(0)), "os_cpu_count": os.cpu_count()}
    try:
        info["cgroup_cpu_max"] = open("/sys/fs/cgroup/cpu.max").read().strip()
    except OSError:
        pass
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                info["cpu_model"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return info


def cpu_baseline(fmt, blocks, gpu_out, opt_bytes, plan_bytes, rcp, budget_one=2.0, budget_all=4.0):
    """The CPU path on a bounded sample of the same workload, one thread and all usable threads, checked against the GPU
    output block by block.  `kind` "reference" = the unmodified reference (oracle/_ref, threads driven by std::thread inside
    the shim); "port" = the C restatement when the reference build did not travel."""
    from oracle import pyref
    threads = usable_cores()
    res = {"unit": "Mblocks/s", "cores": threads}
    if pyref.RefLib.available(fast=True) and (pyref.RefLib(fast=True).probe_rcp() == rcp).all():
        ref = pyref.RefLib(fast=True)
        res["kind"] = "reference"
        o1, d1, s1 = ref.encode_mt(fmt, blocks, opt_bytes, plan_bytes, threads=1, budget_s=budget_one, chunk_blocks=64)
        oa, da, sa = ref.encode_mt(fmt, blocks, opt_bytes, plan_bytes, threads=threads, budget_s=budget_all, chunk_blocks=64)
    else:
        orc = pyref.OracleLib()
        res["kind"] = "port"
        enc = {"bc7": lambda b, t: orc.encode_bc7(b, opt_bytes, plan_bytes, rcp, t),
               "bc1": lambda b, t: orc.encode_bc1(b, opt_bytes, rcp, t),
               "bc6hu": lambda b, t: orc.encode_bc6h(b, opt_bytes, False, rcp, t),
               "etc2rgba": lambda b, t: orc.encode_etc2(b, opt_bytes, 1, t)}[fmt]
        n1 = min(blocks.shape[0], 2048)
        t0 = time.perf_counter(); o1 = enc(blocks[:n1], 1); s1 = time.perf_counter() - t0; d1 = n1
        na = min(blocks.shape[0], max(2048, int(n1 / s1 * budget_all * threads * 0.5) // 64 * 64))
        t0 = time.perf_counter(); oa = enc(blocks[:na], threads); sa = time.perf_counter() - t0; da = na
    bad = int((oa[:da] != gpu_out[:da]).any(axis=1).sum()) + int((o1[:d1] != gpu_out[:d1]).any(axis=1).sum())
    res["value"] = da / sa / 1e6
    res["one_thread"] = {"value": d1 / s1 / 1e6, "unit": "Mblocks/s", "blocks": d1, "seconds": s1}
    res["per_thread_kblocks_s"] = da / sa / threads / 1e3
    res["sample"] = "first %d blocks of the same input on %d threads in %.1f s (chunks of 64 blocks claimed in order); first %d blocks on 1 thread in %.1f s" % (da, threads, sa, d1, s1)
    res["gpu_mismatching_blocks"] = bad
    res["blocks_checked"] = max(da, d1)
    return res


PREHEAT_S = 0.4


def preheat(torch, fn, seconds=PREHEAT_S):
    """Untimed, before the W warm-up steps: encodes for `seconds` of wall time, so that the shader clock -- idle a moment ago, and
    five 1.5 ms warm-up steps do not wake it -- has ramped when the K timed steps start (round 3 saw 614-678 Mblocks/s for the
    same library depending on how long the device had been idle; `sustained`, 3 s of back-to-back encodes, was always 687-689).
    The line says that it was done (`preheat_s`).  CVTTMI_BENCH_PREHEAT_S=0 turns it off."""
    try:
        seconds = float(os.environ.get("CVTTMI_BENCH_PREHEAT_S", seconds))
    except ValueError:
        pass
    if seconds <= 0:
        return 0.0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
    return time.perf_counter() - t0


def timed_encode(torch, fn, reps):
    """min / mean kernel time of `fn` over `reps` launches, HIP events on the current (= launch) stream."""
    fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    return float(np.min(ms)), float(np.mean(ms))


def read_sclk_mhz():
    """shader clock now, from rocm-smi (None when the tool or the field is missing)"""
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out[out.index("{"):])
        for card in d.values():
            for k, v in card.items():
                if "sclk" in k.lower():
                    return float(str(v).strip("()").lower().replace("mhz", ""))
    except Exception:  # noqa
        return None
    return None


VALU_PEAK = 256 * 4 * SHADER_CLOCK_HZ / VALU_CYCLES_PER_INST  # wave-instructions per second, whole chip


def format_counters(src_sha):
    """PMC figures per format from the newest committed profiles/rNN/formats_summary.json (tools/profile_formats.sh: SQ
    counters, FETCH_SIZE and WRITE_SIZE in passes of their own, at the BASELINE config sizes), used only when that profile
    was taken with the library that is loaded now (its build identity, api.library_source_sha256).  Returns
    {fmt: {valu_wave_insts_per_block, hbm_bytes_per_block, avg_waves_per_simd, kernels}} or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "formats_summary.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        if d.get("source_sha256") != src_sha:
            return {"stale": True, "source": os.path.relpath(files[-1], ROOT)}
        blocks = {e["fmt"]: e["blocks"] for e in d.get("fmt_bench", [])}
        out = {"source": os.path.relpath(files[-1], ROOT)}
        for fmt, kernels in d.items():
            if not isinstance(kernels, list) or not kernels or "derived" not in kernels[0] or fmt not in blocks:
                continue
            insts = sum(k["derived"]["valu_insts_per_wave"] * k["grid"] / 64.0 for k in kernels if "derived" in k)
            hbm = [k["hbm"]["bytes_corrected"] for k in kernels if "hbm" in k]
            dur = sum(k["dur_us"] for k in kernels)
            out[fmt] = {"valu_wave_insts_per_block": insts / blocks[fmt],
                        "hbm_bytes_per_block": (sum(hbm) / blocks[fmt]) if len(hbm) == len(kernels) else None,
                        "avg_waves_per_simd": sum(k["derived"]["avg_waves_per_simd"] * k["dur_us"] for k in kernels if "derived" in k) / dur,
                        "kernels": [k["kernel"] for k in kernels], "profiled_blocks": blocks[fmt],
                        }
        return out
    except Exception:  # noqa
        return None


def roofline_block(fmt, nblk, k_ms, kernel, insts_per_block=None, hbm_bytes_per_block=None, waves_per_simd=None, source=None):
    """The roofline of one kernel launch.  The search kernels are VALU-issue bound (72-144 algorithmic bytes per block against
    thousands of instructions), so when the PMC profile of this very library is at hand `bound` is "valu": achieved =
    wave-level VALU instructions per second (SQ_INSTS_VALU of the profile, per block, x the blocks of this launch / the kernel
    time measured here with HIP events), peak = one wave64 VALU instruction per SIMD per 2 cycles (MI355X_MICROARCH.md).
    The HBM view -- SURVEY 8(d)'s algorithmic bytes per block over the same time, against 8 TB/s, and `traffic`, the
    FETCH_SIZE / WRITE_SIZE bytes of the profile over the same time -- is always there under `hbm`; without a matching
    profile it is all there is, and `bound` says "hbm" with a note."""
    t = k_ms * 1e-3
    a = ALGO_BYTES[fmt] * nblk / t / 1e9
    hbm = {"achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS, "bytes_per_block": ALGO_BYTES[fmt], "traffic": None}
    if hbm_bytes_per_block is not None:
        hbm["traffic"] = hbm_bytes_per_block * nblk / t / 1e9
        hbm["traffic_bytes_per_launch"] = hbm_bytes_per_block * nblk
        hbm["traffic_over_algorithmic"] = hbm_bytes_per_block / float(ALGO_BYTES[fmt])
    if insts_per_block is None:
        r = dict(hbm)
        r.update({"bound": "hbm", "kernel": kernel, "kernel_ms": k_ms,
                  "note": "no PMC profile of this library build: only the HBM view; the kernel is VALU-issue bound (DESIGN.md 4)"})
        return r
    rate = insts_per_block * nblk / t
    r = {"bound": "valu", "achieved": rate, "peak": VALU_PEAK, "unit": "wave-instructions/s", "frac": rate / VALU_PEAK,
         "traffic": hbm["traffic"], "kernel": kernel, "kernel_ms": k_ms, "valu_insts_per_block": insts_per_block,
         "avg_waves_per_simd": waves_per_simd, "hbm": hbm, "counters_from": source,
         "peak_source": "MI355X_MICROARCH.md: one wave64 VALU instruction per SIMD per 2 cycles, 1024 SIMDs at %.1f GHz" % (SHADER_CLOCK_HZ / 1e9)}
    if hbm.get("traffic_over_algorithmic") is not None:
        r["traffic_over_algorithmic"] = hbm["traffic_over_algorithmic"]
    return r


def profiled_counters(lib_sha):
    """PMC figures of the headline kernel from the newest committed rocprofv3 summary (profiles/rNN/summary.json, written by
    tools/profile_round.sh: separate --pmc passes, FETCH_SIZE corrected as the microarch guide prescribes) -- used only when
    that profile was taken with the library that is loaded now: its build identity (api.library_source_sha256, a hash of
    the sources and flags that does not depend on the build directory)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "summary.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        if d.get("source_sha256") != lib_sha:
            return {"stale": True, "source": os.path.relpath(files[-1], ROOT)}
        sq = d["pmc_sq"][0]  # the dominant kernel (tools/summarize_pmc.py sorts by total duration)
        return {"source": os.path.relpath(files[-1], ROOT),
                "kernel": sq["kernel"],
                "hbm_bytes_per_launch": d["hbm_traffic_bytes_per_launch"]["bytes_corrected"],
                "valu_insts_per_wave": sq["derived"]["valu_insts_per_wave"],
                "avg_waves_per_simd": sq["derived"].get("avg_waves_per_simd(WAVE_CYCLES*4/simd_cycles)"),
                "blocks": int(sq["grid"]) // 4}
    except Exception:  # noqa
        return None


def run_single(args):
    import torch
    from convectionkernels_amd import api, synth

    if not torch.cuda.is_available():
        sys.stderr.write("bench.py: no GPU visible (the product has no CPU path)\n")
        sys.exit(2)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ctx = api.Context(0)
    ctx.set_exhaustive(args.exhaustive)
    rcp = ctx.get_rcp_table()  # this box's host RCPPS: "bit-exact vs the CPU path on the same box"
    h = golden_hashes()
    rcp_gold = golden_rcp(h)
    same_lut = bool((rcp == rcp_gold).all())
    opt, plan = api.Options(), api.BC7EncodingPlan()
    opt_b = np.frombuffer(opt.tobytes(), np.uint8).copy()
    plan_b = np.frombuffer(plan.tobytes(), np.uint8).copy()

    size = args.size or 4096
    img = synth.image_rgba8(2, size, size, opaque=args.opaque)
    blocks = synth.tile_blocks(img)
    nblk = blocks.shape[0]
    d_in = torch.from_numpy(blocks).to(dev)
    d_out = torch.empty((nblk, 16), dtype=torch.uint8, device=dev)

    preheat_s = preheat(torch, lambda: ctx.encode_bc7(d_in, opt, plan, out=d_out))
    for _ in range(args.warmup):
        ctx.encode_bc7(d_in, opt, plan, out=d_out)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        evs[i][0].record()
        ctx.encode_bc7(d_in, opt, plan, out=d_out)
        evs[i][1].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    k_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    mblocks = nblk * args.steps / elapsed / 1e6
    achieved = ALGO_BYTES["bc7"] * nblk / (k_ms * 1e-3) / 1e9
    out_host = d_out.cpu().numpy()

    result = {
        "metric": "bc7_encode_default_plan_throughput",
        "value": mblocks,
        "unit": "Mblocks/s",
        "gpixel_per_s": mblocks * 16.0 / 1e3,
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": DTYPE,
        "data": "synthetic",
        "config": {
            "workload": "EncodeBC7, BC7EncodingPlan() + Options(), %dx%d SplitMix64 random RGBA%s, seed 2, %d blocks resident in HBM "
                        "(BASELINE configs[1])" % (size, size, " alpha=255" if args.opaque else "", nblk),
            "flags": "0x%x" % opt.flags, "refineRoundsBC7": opt.refineRoundsBC7, "exchange": "none",
            "search": "exhaustive (every candidate evaluated, as the reference does)" if args.exhaustive else
                      "exact branch-and-bound (output bit-identical)",
            "blocks": nblk,
        },
        "roofline": roofline_block("bc7", nblk, k_ms, "cvttmi_bc7_kernel"),
        "burst_value": mblocks,
        "preheat_s": preheat_s,
    }

    lib_sha = api.library_source_sha256()
    result["library_source_sha256"] = lib_sha
    result["kernel_object_sha256"] = api.library_fatbin_sha256()
    pmc = profiled_counters(lib_sha)
    if pmc and pmc.get("stale"):
        result["profile_note"] = "%s was taken with a library built from other sources or flags: traffic / valu_issue omitted" % pmc["source"]
    elif pmc and pmc["blocks"] == nblk and not args.exhaustive and not args.opaque:
        waves = (nblk + 15) // 16
        result["roofline"] = roofline_block("bc7", nblk, k_ms, "cvttmi_bc7_kernel", insts_per_block=pmc["valu_insts_per_wave"] * waves / float(nblk),
                                            hbm_bytes_per_block=pmc["hbm_bytes_per_launch"] / float(nblk),
                                            waves_per_simd=pmc.get("avg_waves_per_simd"), source=pmc["source"])
        result["roofline"]["note"] = ("kernel_ms brackets the launches of one encode on its stream (search + hand-over launch + commit); "
                                      "%d algorithmic bytes per block, so the HBM fraction (`hbm`) is small by construction" % ALGO_BYTES["bc7"])
        result["valu_issue"] = {"insts_per_wave": pmc["valu_insts_per_wave"], "achieved": result["roofline"]["achieved"], "unit": "wave-instructions/s",
                                "peak": VALU_PEAK, "frac": result["roofline"]["frac"], "avg_waves_per_simd": pmc.get("avg_waves_per_simd"),
                                "source": pmc["source"]}

    if not args.no_extra and not args.exhaustive:
        # ---- sustained rate: >= 3 s of back-to-back encodes, shader clock sampled while they run
        target_s = 3.0
        per = max(1, int(0.25 / max(1e-4, k_ms * 1e-3)))
        clock = {}

        def sampler():
            time.sleep(1.0)
            clock["sclk_mhz_under_load"] = read_sclk_mhz()

        th = threading.Thread(target=sampler)
        th.start()
        n_done = 0
        torch.cuda.synchronize()
        ts = time.perf_counter()
        while time.perf_counter() - ts < target_s:
            for _ in range(per):
                ctx.encode_bc7(d_in, opt, plan, out=d_out)
            n_done += per
            torch.cuda.synchronize()
        sus_s = time.perf_counter() - ts
        th.join()
        sus = nblk * n_done / sus_s / 1e6
        result["sustained"] = {"mblocks_s": sus, "seconds": sus_s, "encodes": n_done, "clock_ghz": (clock.get("sclk_mhz_under_load") or 0) / 1e3 or None,
                               "ratio_to_burst": sus / mblocks}
        if sus < 0.95 * mblocks:
            result["value"] = sus
            result["gpixel_per_s"] = sus * 16.0 / 1e3
            result["value_note"] = "sustained rate (more than 5 % below the K-step burst) reported as value"

        # ---- two callers: the same image encoded alternately by two contexts on two streams (what a pipeline that has the next
        # texture ready does).  One encode is three launches in stream order, and the drain of its last wave generation and
        # its 25 us hand-over launch leave SIMDs idle that another stream's encode fills.  NOT the headline value: a step of
        # the contract is one encode on one stream.
        try:
            ctx2 = api.Context(dev.index or 0)
            ctx2.set_rcp_table(rcp)
            s_a, s_b = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
            d_out2 = torch.empty_like(d_out)
            pairs = max(4, args.steps)

            def both():
                ctx.encode_bc7(d_in, opt, plan, out=d_out, stream=s_a.cuda_stream)
                ctx2.encode_bc7(d_in, opt, plan, out=d_out2, stream=s_b.cuda_stream)
            for _ in range(3):
                both()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(pairs):
                both()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            same = bool((d_out2.cpu().numpy() == out_host).all()) and bool((d_out.cpu().numpy() == out_host).all())
            result["two_streams"] = {"mblocks_s": 2 * pairs * nblk / dt / 1e6, "encodes": 2 * pairs, "identical_output": same,
                                     "note": "two contexts, two streams, the same 4096^2 image alternately; not the headline"}
            del ctx2
        except Exception as e:  # an extra leg must not take the line down
            result["two_streams"] = {"error": repr(e)[:200]}

        # ---- the same workload with pruning off, for reference (not the headline value)
        ctx.set_exhaustive(True)
        ex_min, _ = timed_encode(torch, lambda: ctx.encode_bc7(d_in, opt, plan, out=d_out), 1)
        result["exhaustive_search"] = {"value": nblk / ex_min / 1e3, "unit": "Mblocks/s", "kernel_ms": ex_min,
                                       "identical_output": bool((d_out.cpu().numpy() == out_host).all())}
        ctx.set_exhaustive(False)

    if not args.no_cpu:
        result["cpu_baseline"] = cpu_baseline("bc7", blocks, out_host, opt_b, plan_b, rcp, budget_one=3.0, budget_all=8.0)
        result["cpu_baseline"]["host"] = host_info()
        result["bit_exact_vs_cpu"] = result["cpu_baseline"]["gpu_mismatching_blocks"] == 0

    if not args.no_extra and not args.exhaustive and not args.opaque and size == 4096:
        # (the host-path leg first: after the 16384^2 legs have churned a few GB of host memory it measures 30 % lower)
        hp = host_path_leg(torch, api, ctx, blocks, out_host, opt, plan)
        if hp:
            result["host_path"] = hp
        fmtc = format_counters(lib_sha)
        if fmtc and fmtc.get("stale"):
            result["formats_profile_note"] = "%s was taken with a library built from other sources or flags: per-config counters omitted" % fmtc["source"]
            fmtc = None
        result["configs"] = per_config_legs(torch, api, synth, ctx, dev, h, rcp, rcp_gold, same_lut, args, fmtc)
        result["content_families"] = family_legs(torch, api, synth, ctx, dev, rcp, args.no_cpu)
        result["bc6h_content_families"] = bc6h_family_legs(torch, api, synth, ctx, dev, rcp, args.no_cpu)
        # the one-GPU rate on the workload the N > 1 runs shard (one 16384^2 image): the base of a strong-scaling curve
        base = result["configs"].get("5a_bc7_16384")
        if base:
            result["scale_base"] = {"mblocks_s": base["mblocks_s"], "kernel_ms": base["kernel_ms"], "n_gpus": 1,
                                    "workload": "config 5a: the 16384x16384 image of the N > 1 runs on one GPU (same job as `bench.py --gpus N`, N = 1)",
                                    "sha256_matches_reference": base.get("sha256_matches_reference")}
            result["scale_base_mblocks_s"] = base["mblocks_s"]
        db = dropin_8block_leg()
        if db:
            result["dropin_8block"] = db
    emit(result)
    return result


def per_config_legs(torch, api, synth, ctx, dev, h, rcp, rcp_gold, same_lut, args, fmtc=None):
    """One measurement per BASELINE config besides the headline: kernel time (events), the roofline figure with SURVEY 8(d)'s
    bytes per block, the SHA-256 of the whole output against the reference's (made with the golden RCPPS table) and the CPU
    reference on a bounded sample with this box's table."""
    legs = {}

    def leg(name, fmt, blocks, encode, hash_key, opt, plan=None, reps=2, cpu=True, prof=None):
        d_in = torch.from_numpy(blocks).to(dev)
        n = blocks.shape[0]
        out = encode(d_in, None)
        ms_min, ms_mean = timed_encode(torch, lambda: encode(d_in, out), reps)
        host = out.cpu().numpy()
        # prof: the profiled format(s) whose counters describe this leg, most specific first
        c = next((fmtc[k] for k in ([prof] if isinstance(prof, str) else (prof or [])) if fmtc and isinstance(fmtc.get(k), dict)), None)
        if c:
            roof = roofline_block(fmt, n, ms_min, "+".join(c["kernels"]), insts_per_block=c["valu_wave_insts_per_block"],
                                  hbm_bytes_per_block=c["hbm_bytes_per_block"], waves_per_simd=c["avg_waves_per_simd"], source=fmtc["source"])
            if c["profiled_blocks"] != n:
                roof["counters_note"] = "per-block counters of the same kernel on %d blocks of the same kind of content" % c["profiled_blocks"]
        else:
            roof = roofline_block(fmt, n, ms_min, fmt)
        e = {"mblocks_s": n / ms_min / 1e3, "gpixel_per_s": n * 16 / ms_min / 1e6, "kernel_ms": ms_min, "blocks": n, "roofline": roof}
        if hash_key in h:
            if same_lut:
                e["sha256_matches_reference"] = sha256(host) == h[hash_key]
            else:
                ctx.set_rcp_table(rcp_gold)
                e["sha256_matches_reference"] = sha256(encode(d_in, None).cpu().numpy()) == h[hash_key]
                ctx.set_rcp_table(rcp)
        if cpu and not args.no_cpu:
            ob = np.frombuffer(opt.tobytes(), np.uint8).copy()
            pb = None if plan is None else np.frombuffer(plan.tobytes(), np.uint8).copy()
            e["cpu_baseline"] = cpu_baseline(fmt, blocks, host, ob, pb, rcp, budget_one=1.5, budget_all=2.5)
        legs[name] = e
        del d_in, out
        torch.cuda.empty_cache()

    o, p = api.Options(), api.BC7EncodingPlan()
    ultra = api.Options(flags=api.Flags.Ultra)
    leg("1_bc1_256", "bc1", synth.tile_blocks(synth.image_rgba8(1, 256, 256)), lambda t, out: ctx.encode_bc1(t, o, out=out), "config1_bc1_256_seed1", o, reps=5, prof="bc1")
    leg("2b_bc7_4096_opaque", "bc7", synth.tile_blocks(synth.image_rgba8(2, 4096, 4096, opaque=True)),
        lambda t, out: ctx.encode_bc7(t, o, p, out=out), "config2b_bc7_4096_seed2_opaque", o, p, prof="bc7o")
    leg("3_bc6hu_4096", "bc6hu", synth.tile_blocks(synth.image_f16bits(3, 4096, 4096)), lambda t, out: ctx.encode_bc6h(t, o, signed=False, out=out),
        "config3_bc6hu_4096_seed3", o, reps=2, prof="bc6hu")
    leg("4_etc2rgba_4096", "etc2rgba", synth.tile_blocks(synth.image_rgba8(4, 4096, 4096)), lambda t, out: ctx.encode_etc2_rgba(t, o, out=out),
        "config4_etc2rgba_4096_seed4", o, prof="etc2rgba")
    big = synth.tile_blocks(synth.image_rgba8(5, 16384, 16384))
    leg("5a_bc7_16384", "bc7", big, lambda t, out: ctx.encode_bc7(t, o, p, out=out), "config5_bc7_16384_seed5", o, p, prof=["bc7c5", "bc7"])
    leg("5b_bc7_16384_ultra", "bc7", big, lambda t, out: ctx.encode_bc7(t, ultra, p, out=out), "config5b_bc7_16384_seed5_ultra", ultra, p, reps=1, prof=["bc7c5u", "bc7u"])
    return legs


def family_legs(torch, api, synth, ctx, dev, rcp, no_cpu, n=1 << 20, check=1 << 15):
    """EncodeBC7 (default plan) on eight kinds of content: the pruning, and with it the rate, depends on the content.
    Every family's output is compared with the CPU path on its first `check` blocks (`mismatches_vs_cpu`)."""
    res = {}
    opt, plan = api.Options(), api.BC7EncodingPlan()
    ob = np.frombuffer(opt.tobytes(), np.uint8).copy()
    pb = np.frombuffer(plan.tobytes(), np.uint8).copy()
    cpu = None
    if not no_cpu:
        from oracle import pyref
        if pyref.RefLib.available(fast=True) and (pyref.RefLib(fast=True).probe_rcp() == rcp).all():
            ref = pyref.RefLib(fast=True)
            cpu = ("reference", lambda b: ref.encode_mt("bc7", b, ob, pb, threads=usable_cores(), budget_s=60.0, chunk_blocks=64)[0])
        else:
            orc = pyref.OracleLib()
            cpu = ("port", lambda b: orc.encode_bc7(b, ob, pb, rcp, usable_cores()))
    for name, b in synth.content_families(n).items():
        t = torch.from_numpy(b).to(dev)
        out = ctx.encode_bc7(t, opt, plan)
        ms, _ = timed_encode(torch, lambda: ctx.encode_bc7(t, opt, plan, out=out), 2)
        res[name] = {"mblocks_s": n / ms / 1e3, "kernel_ms": ms}
        if cpu:
            got = out[:check].cpu().numpy()
            exp = cpu[1](np.ascontiguousarray(b[:check]))
            res[name]["mismatches_vs_cpu"] = int((got != exp[:check]).any(axis=1).sum())
            res[name]["blocks_checked"] = int(check)
            res[name]["cpu_kind"] = cpu[0]
        del t, out
    return res


def bc6h_family_legs(torch, api, synth, ctx, dev, rcp, no_cpu, n=1 << 19, check=1 << 14):
    """EncodeBC6HU on three kinds of HDR content: the search skips what the delta coding of the end points rules out, so
    its rate depends on how close together a block's end points are (noise = BASELINE config 3: nearly everything is ruled
    out at the higher precisions; smooth content: nearly nothing is).  Each family is compared with the CPU path on its
    first `check` blocks."""
    res = {}
    opt = api.Options()
    ob = np.frombuffer(opt.tobytes(), np.uint8).copy()
    cpu = None
    if not no_cpu:
        from oracle import pyref
        if pyref.RefLib.available(fast=True) and (pyref.RefLib(fast=True).probe_rcp() == rcp).all():
            ref = pyref.RefLib(fast=True)
            cpu = ("reference", lambda b: ref.encode_mt("bc6hu", b, ob, None, threads=usable_cores(), budget_s=60.0, chunk_blocks=64)[0])
        else:
            orc = pyref.OracleLib()
            cpu = ("port", lambda b: orc.encode_bc6h(b, ob, False, rcp, usable_cores()))
    for name, b in synth.hdr_content_families(n).items():
        t = torch.from_numpy(b).to(dev)
        out = ctx.encode_bc6h(t, opt, signed=False)
        ms, _ = timed_encode(torch, lambda: ctx.encode_bc6h(t, opt, signed=False, out=out), 2)
        res[name] = {"mblocks_s": n / ms / 1e3, "kernel_ms": ms}
        if cpu:
            got = out[:check].cpu().numpy()
            exp = cpu[1](np.ascontiguousarray(b[:check]))
            res[name]["mismatches_vs_cpu"] = int((got != exp[:check]).any(axis=1).sum())
            res[name]["blocks_checked"] = int(check)
            res[name]["cpu_kind"] = cpu[0]
        del t, out
    return res


def dropin_8block_leg():
    """The UNMODIFIED caller's convention: one cvtt::Kernels::Encode* call per 8 blocks (reference API.cpp:41-54, caller loop
    etc2packer.cpp:215-281), through include/cvtt/ConvectionKernels.h -- a PCIe round trip and a one-wave launch per call.
    Timed by tools/dropin_bench (built by __graft_entry__.build()), 1 and 16 caller threads; the CPU reference's time per
    call on one thread stands beside it."""
    exe = os.path.join(ROOT, "convectionkernels_amd", "lib", "dropin_bench")
    if not os.path.exists(exe):
        return None
    try:
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "convectionkernels_amd", "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
        out = subprocess.run([exe, "0.5"], capture_output=True, text=True, timeout=120, env=env).stdout
        res = json.loads(out[out.index("{"):])
    except Exception as e:  # noqa
        return {"error": repr(e)}
    try:
        from oracle import pyref
        if pyref.RefLib.available(fast=True):
            ref = pyref.RefLib(fast=True)
            rng = np.random.default_rng(7)
            blocks = rng.integers(0, 256, (4096, 16, 4), dtype=np.uint8)
            ob = np.frombuffer(ref.default_options(), np.uint8).copy()
            pb = np.frombuffer(ref.default_plan(), np.uint8).copy()
            cpu = {}
            cpu16 = {}
            for fmt, n in (("bc7", 1024), ("bc1", 4096), ("etc2rgba", 512)):
                t0 = time.perf_counter()
                ref.encode_mt(fmt, blocks[:n], ob, pb if fmt == "bc7" else None, threads=1, budget_s=30.0, chunk_blocks=64)
                cpu[fmt] = (time.perf_counter() - t0) / (n / 8) * 1e6
                # the caller's own threading model on the CPU: 16 workers, 8 blocks per call
                nt = min(16, usable_cores())
                t0 = time.perf_counter()
                _, done, _ = ref.encode_mt(fmt, blocks[:n * 4], ob, pb if fmt == "bc7" else None, threads=nt, budget_s=30.0, chunk_blocks=8)
                cpu16[fmt] = done / 8 / (time.perf_counter() - t0)
            res["cpu_reference_calls_per_s_all_threads"] = cpu16
            res["cpu_reference_us_per_call_one_thread"] = cpu
            for fmt in cpu:
                if fmt in res and "us_per_call_1_thread" in res[fmt]:
                    res[fmt]["slower_than_cpu_reference_per_call"] = res[fmt]["us_per_call_1_thread"] > cpu[fmt]
                    res[fmt]["calls_16_threads_over_1_thread"] = res[fmt]["calls_per_s_16_threads"] / res[fmt]["calls_per_s_1_thread"]
                    res[fmt]["calls_16_threads_over_cpu_reference_all_threads"] = res[fmt]["calls_per_s_16_threads"] / cpu16[fmt]
    except Exception as e:  # noqa
        res["cpu_error"] = repr(e)
    res["note"] = ("8 blocks per call cannot fill a GPU: concurrent calls of one kind are coalesced into one launch (cxx_api.cpp), a "
                   "single caller thread pays a PCIe round trip and a one-wave launch per call; throughput needs the *Batch entry points")
    return res


def host_path_leg(torch, api, ctx, blocks, expect, opt, plan):
    """The drop-in's host-pointer entry point (cvttmi_encode_bc7, what cvtt::Kernels::EncodeBC7Batch calls): host memory
    in, host memory out, PCIe both ways inside the timed region, chunks pipelined with the search (shim.cpp hostPipeline).
    Twice: ordinary pageable numpy arrays (staged through pinned buffers, one CPU copy each way) and page-locked arrays
    (cvttmi_host_alloc; transferred in place).  Never the headline value."""
    n = blocks.shape[0]
    pcie_bound = 63e9 / ALGO_BYTES["bc7"] / 1e6
    res = {"pcie_bound_mblocks_s": pcie_bound, "note": "host to host, %d B per block over PCIe (63 GB/s spec)" % ALGO_BYTES["bc7"]}
    try:
        def run(src, dst):
            ctx.encode_bc7(src, opt, plan, out=dst)
            ts = []
            for _ in range(20):
                t0 = time.perf_counter()
                ctx.encode_bc7(src, opt, plan, out=dst)
                ts.append(time.perf_counter() - t0)
            return min(ts)
        out = np.empty((n, 16), np.uint8)
        s = run(blocks, out)
        res["pageable"] = {"mblocks_s": n / s / 1e6, "seconds": s, "identical_output": bool((out == expect).all())}
        pin_in = ctx.host_empty(blocks.shape, np.uint8)
        pin_in[...] = blocks
        pin_out = ctx.host_empty((n, 16), np.uint8)
        s = run(pin_in, pin_out)
        res["pinned"] = {"mblocks_s": n / s / 1e6, "seconds": s, "identical_output": bool((pin_out == expect).all())}
        res["mblocks_s"] = res["pinned"]["mblocks_s"]
    except Exception as e:  # noqa
        res["error"] = repr(e)
    return res


def main(argv=None):
    args = parse_args(argv)
    launched = "WORLD_SIZE" in os.environ
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and not launched:
        self_spawn(args)
    if launched and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s\n" % (args.gpus, os.environ["WORLD_SIZE"]))
        sys.exit(2)
    workload = args.workload
    if workload == "auto":
        workload = "config2" if args.gpus == 1 else "config5"
    if args.gpus > 1 and workload != "config5":
        sys.exit("bench.py: several GPUs run config5 (one image, block-row sharded); config2 is the one-GPU workload")
    if workload == "config5" or args.dry_run:
        return run_sharded(args)
    return run_single(args)


if __name__ == "__main__":
    main()
