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
# HBM bytes per launch of the two heaviest kernels on the headline scene, from rocprofv3 PMC passes run separately
# (`--pmc FETCH_SIZE`, then `--pmc WRITE_SIZE`; KB units; FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B,
# MI355X_MICROARCH.md "HBM").  Raw per-kernel means are committed in profiles/ (r01_d_pmc_headline.csv).
TRAFFIC_TILE_VIS = 55451238     # k_tile_vis<false>: 2 x 18507.9 KB fetched + 17135.8 KB written (framebuffer = 16.6 MB)
TRAFFIC_GEOM_DIRECT = 61785498  # k_geom<2>:         2 x 8939.4 KB fetched + 42458.6 KB written (records, bin queues, scratch)


def algorithmic_bytes(md, width, height, npalettes, nbones=64):
    """B = V*stride + I*2 + W*H*(4+4) + N_pal*J*64 (+ T_unique, 0 for the debug-id shader)."""
    from mt_renderer_amd.scene import unpack_primitive
    v = i = 0
    for p in range(md.nprims):
        f = unpack_primitive(md.prims[p])
        v += f["vertex_num"] * f["vertex_stride"]
        i += f["index_num"] * 2
    return v + i + width * height * 8 + npalettes * nbones * 64


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
    rc = None  # direct RCCL communicator (mt_renderer_amd/rccl.py), or None: torch.distributed's collective
    xthread = False  # the exchange runs on the library's exchange thread
    xstream = None
    lane2 = None  # second exchange lane: (communicator, send, gathered, final, stream)
    if sharded:
        nbytes = int(api.lib.mtr_shard_bytes(W, H, world))
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

    def one_frame(check=False):
        fr = api.Frame(dev, W, H)
        if sharded:
            fr.set_shard(rank, world)
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
            dev.unpack_color_shards(gathered.data_ptr(), world, W, H, final.data_ptr())
        fr.close()

    def sync():
        if xthread:
            dev.exchange_drain()  # every handed-over frame has been issued
        torch.cuda.synchronize()
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

    # ---- roofline: per-stage hipEvent timing on the library's stream, separate from the timed region ----
    # Same submission pattern as the timed region (frames in flight on the library's internal streams, nothing waited
    # until several later frames are queued), with hipEvents around every kernel of every frame: the averages are the
    # kernels' launch durations WHILE OVERLAPPING, which is what rocprofv3 --kernel-trace --stats reports for the
    # same command (profiles/).  stage_ms_serial: the same kernels one frame at a time (nothing else on the GPU).
    dev.set_profiling(True)
    stage_ms = {k: 0.0 for k in api.STAGE_NAMES}
    nprof = max(8, min(100, args.steps))
    depth = 6
    stats = None
    inflight = []

    def retire(fr):
        nonlocal stats
        fr.wait()
        for k, v in fr.timings_ms().items():
            stage_ms[k] += v / nprof
        stats = fr.stats()
        fr.close()

    for _ in range(nprof):
        fr = api.Frame(dev, W, H)
        if sharded:
            fr.set_shard(rank, world)
        model.render(fr, M)
        fr.submit()
        inflight.append(fr)
        if len(inflight) > depth:
            retire(inflight.pop(0))
    while inflight:
        retire(inflight.pop(0))
    stage_ms_serial = {k: 0.0 for k in api.STAGE_NAMES}
    for _ in range(10):
        fr = api.Frame(dev, W, H)
        if sharded:
            fr.set_shard(rank, world)
        model.render(fr, M)
        fr.end()
        for k, v in fr.timings_ms().items():
            stage_ms_serial[k] += v / 10
        fr.close()
    dev.set_profiling(False)
    dom = max(stage_ms, key=lambda k: stage_ms[k])
    alg_bytes = algorithmic_bytes(md, W, H, 1)
    achieved = alg_bytes / (stage_ms[dom] * 1e-3) / 1e9
    kname = {"geom": "k_geom<%d>" % ((2 if stats["tile_kernel"] == 2 else 1) if stats["binning"] == 1 else 0), "scan": "k_scan", "fill": "k_fill",
             "tile": "k_tile_vis<false>" if stats["tile_kernel"] == 2 else "k_tile<false>"}[dom]
    # HBM bytes of that kernel per launch from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes,
    # FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md); measured offline, see profiles/README.md
    traffic = {"k_tile_vis<false>": TRAFFIC_TILE_VIS, "k_geom<2>": TRAFFIC_GEOM_DIRECT}.get(kname) if world == 1 else None
    roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(stage_ms[dom], 5),
                "stage_ms": {k: round(v, 5) for k, v in stage_ms.items()},
                "stage_ms_serial": {k: round(v, 5) for k, v in stage_ms_serial.items()},
                "frame_gbps": round(alg_bytes / (ms_per_step * 1e-3) / 1e9, 3)}
    if stats["binning"] == 1:  # single-pass binning: k_scan / k_fill are not launched at all
        roofline["stage_note"] = "single-pass binning: no k_scan / k_fill launch; their entries are the gap between two timing events recorded back to back"

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
                       "triangles_per_frame": ntris, "sharding": "bins %% %d" % world if sharded else "none",
                       "collective": ((("ncclAllGather (exchange thread, 2 lanes)" if lane2 is not None else "ncclAllGather (exchange thread)") if xthread else "ncclAllGather (ctypes)") if rc is not None else "torch.distributed all_gather_into_tensor") if sharded else "none"},
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
