#!/usr/bin/env python3
"""bench.py -- the rModel draw path on MI355X: Mtris/s + ms/frame on the headline scene
(1 000 000 triangles, 64 bones, 1920x1080; BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one frame: clear (fused) -> geometry -> bin -> tile raster/shade -> framebuffer complete in
HBM (N > 1: after the RCCL all-gather of the colour shards).  All inputs (vertex/index buffers, bone
palette, transforms) are resident in HBM before the timed region.  One process per GPU; N > 1 shards
the 16x16-pixel bins over the ranks (bin % N == rank) and exchanges colour with one all-gather.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HBM-bound accounting, hipEvent
timing on the library's own stream) and `cpu_baseline` (the CPU oracle -- kind "port": the
reference has no CPU path at all -- on a bounded sample of the same workload, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# roofline.traffic: HBM bytes per launch of the hot kernels from rocprofv3 PMC passes run separately (`--pmc FETCH_SIZE`,
# then `--pmc WRITE_SIZE`; KB units; FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md
# "HBM").  The numbers live in a tracked file, profiles/pmc_traffic.json, written by tools/profile_round.sh together
# with a hash of the kernel sources they were measured on; a build whose sources hash differently reports null.
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def kernel_source_hash():
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "mt_renderer_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "mt_renderer_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic():
    """{kernel name: HBM bytes per launch} for this build's kernels, or {} when the tracked measurement is of other sources"""
    try:
        d = json.load(open(PMC_TRAFFIC_FILE))
    except (OSError, ValueError):
        return {}
    return d.get("bytes_per_launch", {}) if d.get("kernel_source_hash") == kernel_source_hash() else {}


def algorithmic_bytes(md, width, height, npalettes, nbones=64):
    """B = V*stride + I*2 + W*H*(4+4) + N_pal*J*64 (+ T_unique, 0 for the debug-id shader), and its
    split over the two kernels: geometry reads vertices + indices + palettes, the tile kernel writes the framebuffer."""
    from mt_renderer_amd.scene import unpack_primitive
    v = i = 0
    for p in range(md.nprims):
        f = unpack_primitive(md.prims[p])
        v += f["vertex_num"] * f["vertex_stride"]
        i += f["index_num"] * 2
    geom, tile = v + i + npalettes * nbones * 64, width * height * 8
    return geom + tile, geom, tile


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--verify", action="store_true", help="N > 1: compare the gathered frame with an unsharded render")
    args = ap.parse_args()

    # N > 1: more hardware queues than ROCm's default 4, so that the exchange (public stream + RCCL's stream) does not
    # share a queue with the library's three render streams; read by the HIP runtime when it starts, i.e. set it first
    # MTR_BENCH_FORCE_DIST=1: run the N > 1 code path (process group, sharding calls, collective) with a world of one
    # rank -- the only way to exercise it with the real nccl backend on a single GPU (tools/probe/bench_world1_nccl.sh)
    force_dist = os.environ.get("MTR_BENCH_FORCE_DIST") == "1"
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or force_dist:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    sharded = world > 1 or force_dist
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    # MTR_BENCH_BACKEND=gloo is a single-GPU rehearsal of the N > 1 path (all ranks share cuda:0, the gather is staged
    # through host memory): it checks the plumbing, not the speed.  The driver's multi-GPU runs use nccl (= RCCL).
    backend = os.environ.get("MTR_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count()) if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)

    from mt_renderer_amd import api, scene

    W, H = args.width, args.height
    md = scene.headline_model()
    ntris = md.input_triangles()
    palette = scene.bone_palette()
    M = scene.to_f32_colmajor(scene.headline_transform(W, H))

    stream = torch.cuda.Stream()
    dev = api.Device(dev_index, stream=stream.cuda_stream)
    model = api.Model.new(dev, md)
    model.set_palette(palette)

    # N > 1: each rank renders bins (bin % N == rank), packs them bin-major (csrc/k_shard.hip), one RCCL
    # all-gather over xGMI exchanges W*H*4/N bytes per rank, one unpack kernel rebuilds the linear frame
    # The exchange of a frame is pack -> all-gather -> unpack on the device's public stream (which the library makes wait
    # for each frame), while later frames render on the library's internal streams.  That only overlaps if the runtime
    # gives those streams separate hardware queues: with ROCm's default of 4 the exchange shared a queue with the
    # render streams and a frame took 0.089 ms instead of 0.057 (tools/probe/nccl_one_rank.py), hence
    # GPU_MAX_HW_QUEUES=8 above.  Rotating several exchange streams was measured too and is worse (more queues to share).
    shard = gathered = final = None
    own = None  # (map, param, band rows) of the sharded frames
    if sharded:
        # Ownership: equal bands of bin rows (MTR_OWN_BANDS), so that a rank only processes the geometry that can reach
        # its band (the library culls the rest before any vertex work).  MTR_BENCH_OWNERSHIP=bands-balanced cuts the rows
        # into `world` bands of equal WEIGHT instead, from one unsharded calibration frame (every rank renders it, reads
        # the queue length of each bin; the counts are exact integers, so every rank derives the same bands without
        # talking to the others).  On this 53 us frame balanced bands do not pay (worst rank, one GPU standing in for
        # every rank, profiles/r02_c_shard_cost_v2.txt: 45.0 / 32.7 / 30.3 us per frame with equal bands at N = 2 / 4 / 8,
        # 46.9 / 33.1 / 32.6 balanced) and they make the all-gather block -- the LARGEST share, padded -- 1.5x larger at
        # N = 8 (13 of 68 rows instead of 8.5); on C4 / C5 they do (C5 at N = 8: 255 -> 219 us).
        # MTR_BENCH_OWNERSHIP=interleaved|bands|bands-balanced|supertiles overrides.
        from mt_renderer_amd import sharding
        kind = os.environ.get("MTR_BENCH_OWNERSHIP", "bands")
        if kind == "interleaved":
            own = (api.OWN_INTERLEAVED, 0, None)
        elif kind == "supertiles":
            own = (api.OWN_SUPERTILES, 3, None)
        elif kind == "bands":
            own = (api.OWN_BANDS, 0, sharding.equal_bands(H, world))
        else:
            fr = api.Frame(dev, W, H); model.render(fr, M); fr.end()
            entries, _ = fr.bin_counts()
            fr.close()
            nbx, nby, _ = sharding.grid(W, H)
            own = (api.OWN_BANDS, 0, sharding.balanced_bands(entries.reshape(nby, nbx).sum(axis=1).astype(np.float64) + 8.0 * nbx, world))
    rc = None  # direct RCCL communicator (mt_renderer_amd/rccl.py), or None: torch.distributed's collective
    xthread = False  # the exchange runs on the library's exchange thread
    xstream = None
    lane2 = None  # second exchange lane: (communicator, send, gathered, final, stream)
    if sharded:
        nbytes = api.shard_bytes_map(W, H, world, *own)
        shard = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        gathered = torch.empty(nbytes * world, dtype=torch.uint8, device="cuda")
        final = torch.empty(W * H * 4, dtype=torch.uint8, device="cuda")
        if backend == "nccl" and os.environ.get("MTR_BENCH_TORCH_COLLECTIVE") != "1":
            # ncclAllGather called straight from ctypes: ~3 us of host time per frame instead of ~20-30 us for
            # dist.all_gather_into_tensor, which matters once a rank's share of the frame is below 30 us of GPU time.
            # Every rank must take the same path: agree on it with a MIN all-reduce after each step that can fail.
            def all_ok(ok):
                t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                return bool(t.item())
            import threading
            from mt_renderer_amd import rccl

            def make_comm(what):
                """a communicator of this job's ranks, or None on EVERY rank if any step failed on any of them"""
                try:
                    c = rccl.Rccl()
                except Exception as e:  # noqa: BLE001
                    print(f"[rank {rank}] direct RCCL unavailable ({e}); {what}", file=sys.stderr)
                    c = None
                if not all_ok(c is not None):
                    return None
                uid = None
                if rank == 0:
                    try:
                        uid = c.unique_id()
                    except Exception as e:  # noqa: BLE001
                        print(f"[rank 0] ncclGetUniqueId failed ({e}); {what}", file=sys.stderr)
                box = [uid]
                dist.broadcast_object_list(box, src=0)
                ok = False
                if box[0] is not None:
                    # ncclCommInitRank is a blocking collective: if it cannot complete, end the run instead of hanging
                    guard = threading.Timer(180.0, lambda: (print(f"[rank {rank}] ncclCommInitRank did not return in 180 s",
                                                                  file=sys.stderr, flush=True), os._exit(5)))
                    guard.daemon = True
                    guard.start()
                    try:
                        c.init(box[0], world, rank)
                        ok = True
                    except Exception as e:  # noqa: BLE001
                        print(f"[rank {rank}] ncclCommInitRank failed ({e}); {what}", file=sys.stderr)
                    guard.cancel()
                return c if all_ok(ok) else None

            rc = make_comm("using torch.distributed")
            # With the direct communicator the whole exchange of a frame (pack -> ncclAllGather -> unpack -> destroy) moves
            # to the library's exchange thread (include/mtr.h: mtr_device_exchange_start): a rank's loop then costs the
            # host ~28-31 us per frame instead of ~43-51 (tools/probe/exchange_thread.py), which is what bounds N > 1 on this
            # frame.  MTR_BENCH_EXCHANGE_THREAD=0 keeps everything on one thread.
            if rc is not None and os.environ.get("MTR_BENCH_EXCHANGE_THREAD") != "0":
                xstream = torch.cuda.Stream()
                started = False
                try:
                    dev.exchange_start(rc.allgather_addr, rc.comm_handle, rccl.ncclUint8, shard.data_ptr(), shard.numel(),
                                       gathered.data_ptr(), final.data_ptr(), world, xstream.cuda_stream)
                    started = True
                except Exception as e:  # noqa: BLE001
                    print(f"[rank {rank}] exchange thread unavailable ({e}); exchanging on the render thread", file=sys.stderr)
                if all_ok(started):
                    xthread = True
                elif started:
                    dev.exchange_stop()
            # A lane is an in-order stream: it completes one (pack + all-gather + unpack) latency per frame.  A second lane
            # with its own communicator takes every other frame, so two collectives are in flight -- but it is one more
            # busy stream, and on the one GPU where it can be measured (world of one rank, the GPU rendering whole frames)
            # the extra stream costs more than it hides: 0.091 ms per frame against 0.057 with one lane, for any
            # GPU_MAX_HW_QUEUES from 8 to 24.  Off by default; MTR_BENCH_EXCHANGE_LANES=2 turns it on for an experiment on
            # a real multi-GPU node, where the all-gather's latency may be what bounds the exchange stream.
            want_lanes = int(os.environ.get("MTR_BENCH_EXCHANGE_LANES", "1"))
            if xthread and want_lanes > 1:
                rc2 = make_comm("one exchange lane")
                if rc2 is not None:
                    added = False
                    try:
                        lane2 = (rc2, torch.empty_like(shard), torch.empty_like(gathered), torch.empty_like(final), torch.cuda.Stream())
                        dev.exchange_add_lane(rc2.comm_handle, lane2[1].data_ptr(), lane2[2].data_ptr(), lane2[3].data_ptr(),
                                              lane2[4].cuda_stream)
                        added = True
                    except Exception as e:  # noqa: BLE001
                        print(f"[rank {rank}] second exchange lane unavailable ({e})", file=sys.stderr)
                    if not all_ok(added):
                        # the lane count must be the same on every rank: without agreement, go back to the render thread
                        dev.exchange_stop()
                        xthread = False
                        lane2 = None

    # the timed loop body with every argument marshalled once (api.FrameLoop): begin -> set_shard -> draw -> submit
    # [-> hand-over to the exchange thread] -> destroy, the same C calls as the classes make
    loop = api.FrameLoop(dev, W, H, model=model, view_proj=M, shard=(rank, world) + own if sharded else None, exchange=xthread)

    def one_frame(check=False):
        if not check and (xthread or not sharded):
            loop.run(1)
            return
        fr = api.Frame(dev, W, H)
        if sharded:
            fr.set_shard(rank, world, *own)
        model.render(fr, M)
        if check or not xthread:
            fr.submit()
        if check:
            fr.wait()  # grows the bin queues if needed and validates device flags
        if xthread:
            fr.submit_exchange()  # the exchange thread packs, gathers, unpacks and destroys the frame
            return
        if sharded:
            fr.pack_color_shard(shard.data_ptr(), shard.numel())
            if rc is not None:
                rc.all_gather_u8(shard.data_ptr(), gathered.data_ptr(), shard.numel(), stream.cuda_stream)
            with torch.cuda.stream(stream):
                if rc is not None:
                    pass
                elif backend == "nccl":
                    dist.all_gather_into_tensor(gathered, shard)
                else:  # rehearsal: host-staged gather
                    stream.synchronize()
                    host = torch.empty(gathered.numel(), dtype=torch.uint8)
                    dist.all_gather_into_tensor(host, shard.cpu())
                    gathered.copy_(host)
            fr.unpack_color_shards(gathered.data_ptr(), final.data_ptr())
        fr.close()

    def sync():
        if xthread:
            dev.exchange_drain()  # every handed-over frame has been issued
        torch.cuda.synchronize()
        dev.synchronize()  # raises if a frame nobody waited for overflowed its bin queues (it would be missing triangles)
        if sharded:
            dist.barrier()
            torch.cuda.synchronize()

    one_frame(check=True)
    # Device warm-up, untimed and independent of --warmup: the GPU's power management raises its clocks some 40 ms
    # after sustained load begins (one ~35 ms stall, then 52 us per frame instead of 58: tools/probe/hiccup.py), so a
    # short run would time the transition instead of the steady state.  0.3 s of the same frames first.
    if sharded:
        for _ in range(3000 if backend == "nccl" else 20):  # a fixed count: every rank must make the same number of collective calls
            one_frame()
    else:
        t_ramp = time.perf_counter() + 0.3
        while time.perf_counter() < t_ramp:
            for _ in range(50):
                one_frame()
    for _ in range(args.warmup):
        one_frame()
    sync()
    t0 = time.perf_counter()
    if xthread or not sharded:
        loop.run(args.steps)
    else:
        for _ in range(args.steps):
            one_frame()
    sync()
    dt = time.perf_counter() - t0
    if sharded:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if args.verify and sharded:
        fr = api.Frame(dev, W, H)
        model.render(fr, M)
        fr.end()
        ref = fr.color()
        fr.close()
        sync()
        got = final.cpu().numpy().reshape(H, W, 4)
        ok = bool((got == ref).all())
        if lane2 is not None:  # odd frames went through the second lane into its own destination
            ok = ok and bool((lane2[3].cpu().numpy().reshape(H, W, 4) == ref).all())
        print(f"[rank {rank}] verify gathered frame == unsharded frame: {ok}", file=sys.stderr, flush=True)
        if not ok:
            sys.exit(3)
    ms_per_step = dt * 1e3 / args.steps
    mtris = ntris / (ms_per_step * 1e-3) / 1e6

    # ---- roofline: per-stage hipEvent timing on the library's streams, separate from the timed region ----
    # (1) Same submission pattern as the timed region (frames in flight on the library's internal streams, nothing waited
    # until several later frames are queued), with hipEvents around every kernel of every frame: the averages are the
    # kernels' launch durations WHILE OVERLAPPING, which is what rocprofv3 --kernel-trace --stats reports for the
    # same command (profiles/).  (2) The same kernels one frame at a time, nothing else on the GPU: 60 frames, of which
    # the per-kernel stand-alone durations and the SURVEY 8(d) latency are the medians -- ms/frame from
    # mtr_frame_begin (clear fused into the tile kernel) to the framebuffer complete in HBM, by the events that bracket
    # the frame's first and last kernel, and by the host's clock around begin .. wait.
    dev.set_profiling(True)
    stage_ms = {k: 0.0 for k in api.STAGE_NAMES}
    nprof = max(8, min(100, args.steps))
    depth = 6
    stats = None
    inflight = []
    shard_args = (rank, world) + own if sharded else None

    def new_frame():
        fr = api.Frame(dev, W, H)
        if sharded:
            fr.set_shard(*shard_args)
        model.render(fr, M)
        return fr

    def retire(fr):
        nonlocal stats
        fr.wait()
        for k, v in fr.timings_ms().items():
            stage_ms[k] += v / nprof
        stats = fr.stats()
        fr.close()

    for _ in range(nprof):
        fr = new_frame()
        fr.submit()
        inflight.append(fr)
        if len(inflight) > depth:
            retire(inflight.pop(0))
    while inflight:
        retire(inflight.pop(0))
    serial = {k: [] for k in api.STAGE_NAMES}
    lat_gpu, lat_host = [], []
    for it in range(65):
        t1 = time.perf_counter()
        fr = new_frame()
        fr.end()
        t2 = time.perf_counter()
        tm = fr.timings_ms()
        fr.close()
        if it >= 5:
            for k, v in tm.items():
                serial[k].append(v)
            lat_gpu.append(sum(tm.values()))
            lat_host.append((t2 - t1) * 1e3)
    dev.set_profiling(False)
    med = lambda xs: float(np.median(xs))
    stage_ms_serial = {k: med(v) for k, v in serial.items()}
    alg_frame, alg_geom, alg_tile = algorithmic_bytes(md, W, H, 1)
    direct = stats["binning"] == 1
    vis = stats["tile_kernel"] == 2
    names = {"geom": "k_geom<%d, %s>" % (((2 if vis else 1) if direct else 0), "true" if sharded and own[0] != api.OWN_INTERLEAVED else "false"),
             "scan": "k_scan", "fill": "k_fill", "tile": ("k_tile_vis<false, %d>" % (2 if stats["shard_bins"] > 4096 else (4 if stats["shard_bins"] > 1536 else 8))) if vis else "k_tile<false>"}
    alg = {"geom": alg_geom, "tile": alg_tile}
    traffic_by_kernel = pmc_traffic() if world == 1 else {}
    kernels = {}
    for st in ("geom", "tile"):
        k = {"kernel": names[st], "algorithmic_bytes_per_launch": alg[st], "ms_overlapped": round(stage_ms[st], 5),
             "ms_standalone": round(stage_ms_serial[st], 5), "traffic": traffic_by_kernel.get(names[st])}
        k["frac_overlapped"] = round(alg[st] / (stage_ms[st] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        k["frac_standalone"] = round(alg[st] / (stage_ms_serial[st] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        kernels[st] = k
    dom = max(("geom", "tile"), key=lambda k: stage_ms[k])
    achieved = alg[dom] / (stage_ms[dom] * 1e-3) / 1e9
    # achieved / frac: the dominant kernel's OWN algorithmic bytes over its average launch duration in the pipelined
    # pattern of the timed region (what rocprofv3 --stats of this command reports); `kernels` has both kernels, each
    # also against its stand-alone duration; `frame_*`: the whole frame's algorithmic bytes over the measured ms_per_step
    roofline = {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": kernels[dom]["traffic"],
                "algorithmic_bytes_per_launch": alg[dom], "kernel_ms": round(stage_ms[dom], 5),
                "kernels": kernels,
                "stage_ms": {k: round(v, 5) for k, v in stage_ms.items()},
                "stage_ms_serial": {k: round(v, 5) for k, v in stage_ms_serial.items()},
                "frame_algorithmic_bytes": alg_frame,
                "frame_gbps": round(alg_frame / (ms_per_step * 1e-3) / 1e9, 3),
                "frame_frac": round(alg_frame / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "traffic_source": ("profiles/pmc_traffic.json (kernel sources %s)" % kernel_source_hash()) if traffic_by_kernel else None}
    if direct:  # single-pass binning: k_scan / k_fill are not launched at all
        roofline["stage_note"] = "single-pass binning: no k_scan / k_fill launch (reported as 0)" + \
                                 ("; sharded frames: the geom stage includes the culling kernels" if sharded else "")
    latency = {"ms_per_frame_latency": round(med(lat_gpu), 5), "frames": len(lat_gpu), "definition": "SURVEY 8(d): one frame at a time, first kernel start to framebuffer complete in HBM (hipEvents on the frame's stream), median",
               "ms_per_frame_latency_host_clock": round(med(lat_host), 5), "mtris_per_s_at_latency": round(ntris / (med(lat_gpu) * 1e-3) / 1e6, 2)}

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        om = orc.OracleModel(md)
        # the GPU box gives one-GPU jobs a 16-CPU share; more OpenMP threads than that only spin
        threads = max(1, min(len(os.sched_getaffinity(0)), 16))
        f = orc.OracleFrame(W, H)
        f.draw(om, M, palette, nthreads=threads)  # untimed: OpenMP pool start-up, page faults
        f.close()
        frames, t_cpu = 0, 0.0
        while t_cpu < args.cpu_seconds and frames < 64:
            f = orc.OracleFrame(W, H)
            t1 = time.perf_counter()
            f.draw(om, M, palette, nthreads=threads)
            t_cpu += time.perf_counter() - t1
            frames += 1
            f.close()
        cpu_baseline = {"value": round(ntris * frames / t_cpu / 1e6, 3), "unit": "Mtris/s", "cores": threads, "kind": "port",
                        "sample": f"{frames} full frames of the same 1M-triangle scene through the CPU oracle "
                                  f"(oracle/mtr_oracle.c, OpenMP row bands; the reference has no CPU path)"}

    if rank == 0:
        out = {
            "metric": "Mtris/sec, 1M-tri 64-bone skinned scene @1920x1080", "value": round(mtris, 2), "unit": "Mtris/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "headline: 20 x mesh50k primitives = 1,000,000 strip triangles, 506,520 vertices x 24 B, "
                                   "64-bone palette, debug-id shader, %dx%d" % (W, H),
                       "triangles_per_frame": ntris,
                       "sharding": ({api.OWN_INTERLEAVED: "bins %% %d" % world, api.OWN_BANDS: "bands of bin rows %s over %d ranks, geometry culled per rank" % (list(map(int, own[2])) if own[2] is not None else "equal", world),
                                     api.OWN_SUPERTILES: "super-tiles of %d x %d bins over %d ranks, geometry culled per rank" % (1 << own[1], 1 << own[1], world)}[own[0]]) if sharded else "none",
                       "collective": ((("ncclAllGather (exchange thread, 2 lanes)" if lane2 is not None else "ncclAllGather (exchange thread)") if xthread else "ncclAllGather (ctypes)") if rc is not None else "torch.distributed all_gather_into_tensor") if sharded else "none"},
            "value_is": "pipelined throughput: frames submitted back to back, three in flight on the library's streams; the per-frame latency is in `latency`",
            "latency": latency,
            "frame_stats": stats, "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out))
    model.close()
    dev.close()
    if sharded:
        if rc is not None:
            torch.cuda.synchronize()
            rc.close()
            if lane2 is not None:
                lane2[0].close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
