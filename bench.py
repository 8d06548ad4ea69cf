#!/usr/bin/env python3
"""bench.py -- the rModel draw path on MI355X: Mtris/s + ms/frame on the headline scene
(1 000 000 triangles, 64 bones, 1920x1080; BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one frame: clear (fused) -> geometry -> bin -> tile raster/shade -> framebuffer complete in
HBM (N > 1: after the RCCL all-gather of the colour shards).  All inputs (vertex/index buffers, bone
palette, transforms) are resident in HBM before the timed region.  One process per GPU; N > 1 shards
the 16x16-pixel bins over the ranks (bands of bin rows; a rank culls the geometry that cannot reach its
band) and exchanges colour with one all-gather.

The timed region is EXACTLY `--steps` frames between two (barrier + synchronize) pairs, as the driver's
contract says; it is repeated R = ceil(0.2 s / (steps x estimated frame time)) times and `ms_per_step` is
the MEDIAN repetition (a 20-step region lasts 1 ms: one sample of it is mostly noise).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HBM-bound accounting, hipEvent
timing on the library's own stream) and `cpu_baseline` (the CPU oracle -- kind "port": the
reference has no CPU path at all -- on a bounded sample of the same workload, N = 1 only).  N > 1
(or --configs) adds `configs`: BASELINE.json's multi-GPU configs C4 (128 instances, 3840x2160) and C5
(1024 instances, 64 BC7 textures, 3840x2160) sharded over the N ranks, next to the same scene rendered
unsharded in the same run, and efficiency = t1 / (N x tN).
"""
import argparse
import json
import math
import os
import sys
import threading
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


class Watchdog:
    """N > 1: a rank that stops making progress (a peer died, a collective that never completes) must end the job with
    a message that says where, not sit in the driver's clock until it is killed.  One timer per phase; when it fires the
    process reports its phase and how far it got, and exits -- a fresh exit (os._exit), never a re-exec."""

    def __init__(self, rank, enabled):
        self.rank, self.enabled, self.timer, self.progress = rank, enabled, None, 0

    def arm(self, label, seconds):
        self.disarm()
        if not self.enabled:
            return
        def fire():
            print(f"[rank {self.rank}] WATCHDOG: no completion of '{label}' after {seconds:.0f} s (frames issued by this rank in the "
                  f"phase: {self.progress}); a peer rank failed or a collective cannot complete -- exiting", file=sys.stderr, flush=True)
            os._exit(6)
        self.progress = 0
        self.timer = threading.Timer(seconds, fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--verify", action="store_true", help="N > 1: compare every gathered frame (headline, C4, C5) with an unsharded render")
    ap.add_argument("--configs", choices=["auto", "on", "off"], default="auto",
                    help="time BASELINE configs C4 / C5 too (auto: when N > 1, where they are defined)")
    ap.add_argument("--config-steps", type=int, default=100, help="frames per timed region of a C4 / C5 leg")
    ap.add_argument("--config-instances", type=int, default=0, help="rehearsals: C5 with this many instances instead of 1024 (C4: a quarter)")
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
    watchdog = Watchdog(rank, sharded)
    wd_seconds = float(os.environ.get("MTR_BENCH_WATCHDOG_S", "240"))

    from mt_renderer_amd import api, scene, sharding

    def all_ok(ok):
        """the same answer on every rank: did the step succeed on ALL of them (MIN all-reduce)"""
        if not sharded:
            return bool(ok)
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    W, H = args.width, args.height
    md = scene.headline_model()
    ntris = md.input_triangles()
    palette = scene.bone_palette()
    M = scene.to_f32_colmajor(scene.headline_transform(W, H))

    stream = torch.cuda.Stream()
    dev = api.Device(dev_index, stream=stream.cuda_stream)

    class Workload:
        """one scene resident in HBM: a single model (headline) or an instanced batch (C4 / C5)"""

        def __init__(self, name, w, h, mdata, *, view, pal=None, model_mats=None, palettes=None, tex_override=None):
            self.name, self.w, self.h, self.md = name, w, h, mdata
            self.model = api.Model.new(dev, mdata)
            self.batch = None
            self.view = view
            if model_mats is not None:
                self.batch = api.Batch(dev, self.model, model_mats, palettes, tex_override)
                self.ntris = mdata.input_triangles() * int(model_mats.shape[0])
            else:
                self.model.set_palette(pal)
                self.ntris = mdata.input_triangles()

        def draw(self, fr):
            if self.batch is not None:
                fr.draw_batch(self.batch, self.view)
            else:
                self.model.render(fr, self.view)

        def frame_loop(self, shard, exchange):
            return api.FrameLoop(dev, self.w, self.h, model=None if self.batch is not None else self.model, batch=self.batch,
                                 view_proj=self.view, shard=shard, exchange=exchange)

        def close(self):
            if self.batch is not None:
                self.batch.close()
            self.model.close()

    headline = Workload("headline", W, H, md, view=M, pal=palette)

    def choose_ownership(work, kind):
        """(map, param, band rows) of the sharded frames of a workload.  Bands of bin rows, so that a rank only processes
        the geometry that can reach its band (the library culls the rest before any vertex work).  "bands-balanced" cuts
        the rows into `world` bands of equal WEIGHT from one unsharded calibration frame (every rank renders it and reads
        the queue length of each bin; the counts are exact integers, so every rank derives the same bands without talking
        to the others).  On the 53 us headline frame balanced bands do not pay (profiles/r02_c_shard_cost_v2.txt) and make
        the all-gather block -- the LARGEST share, padded -- 1.5x larger at N = 8; on C4 / C5 they do."""
        if kind == "interleaved":
            return (api.OWN_INTERLEAVED, 0, None)
        if kind == "supertiles":
            return (api.OWN_SUPERTILES, 3, None)
        if kind == "bands":
            return (api.OWN_BANDS, 0, sharding.equal_bands(work.h, world))
        fr = api.Frame(dev, work.w, work.h)
        work.draw(fr)
        fr.end()
        entries, _ = fr.bin_counts()
        fr.close()
        nbx, nby, _ = sharding.grid(work.w, work.h)
        return (api.OWN_BANDS, 0, sharding.balanced_bands(entries.reshape(nby, nbx).sum(axis=1).astype(np.float64) + 8.0 * nbx, world))

    # ---- the communicator: ncclAllGather called straight from ctypes (~3 us of host time per frame instead of ~20-30 us
    # for dist.all_gather_into_tensor, which matters once a rank's share of the frame is below 30 us of GPU time).
    # Every rank must take the same path: agree on it with a MIN all-reduce after each step that can fail.
    rc = None  # direct RCCL communicator (mt_renderer_amd/rccl.py), or None: torch.distributed's collective
    want_xthread = False
    rccl = None
    if sharded and backend == "nccl" and os.environ.get("MTR_BENCH_TORCH_COLLECTIVE") != "1":
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
        # host ~28-31 us per frame instead of ~43-51 (tools/probe/exchange_thread.py), which is what bounds N > 1 on the
        # headline frame.  MTR_BENCH_EXCHANGE_THREAD=0 keeps everything on one thread.
        want_xthread = rc is not None and os.environ.get("MTR_BENCH_EXCHANGE_THREAD") != "0"
    rc2 = None  # communicator of a second exchange lane (MTR_BENCH_EXCHANGE_LANES=2; an experiment for real multi-GPU nodes:
    # on the one GPU where it can be measured the extra stream costs more than it hides, 0.091 vs 0.057 ms per frame)
    if want_xthread and int(os.environ.get("MTR_BENCH_EXCHANGE_LANES", "1")) > 1:
        rc2 = make_comm("one exchange lane")

    class Exchange:
        """buffers + (optionally) the library's exchange thread for the sharded frames of ONE workload under ONE map"""

        def __init__(self, work, own):
            self.work, self.own = work, own
            self.nbytes = api.shard_bytes_map(work.w, work.h, world, *own)
            self.shard = torch.empty(self.nbytes, dtype=torch.uint8, device="cuda")
            self.gathered = torch.empty(self.nbytes * world, dtype=torch.uint8, device="cuda")
            self.final = torch.zeros(work.w * work.h * 4, dtype=torch.uint8, device="cuda")
            self.xthread, self.lane2, self.xstream = False, None, None
            if want_xthread:
                self.xstream = torch.cuda.Stream()
                started = False
                try:
                    dev.exchange_start(rc.allgather_addr, rc.comm_handle, rccl.ncclUint8, self.shard.data_ptr(), self.shard.numel(),
                                       self.gathered.data_ptr(), self.final.data_ptr(), world, self.xstream.cuda_stream)
                    started = True
                except Exception as e:  # noqa: BLE001
                    print(f"[rank {rank}] exchange thread unavailable ({e}); exchanging on the render thread", file=sys.stderr)
                if all_ok(started):
                    self.xthread = True
                elif started:
                    dev.exchange_stop()
                if self.xthread and rc2 is not None:
                    added = False
                    try:
                        self.lane2 = (rc2, torch.empty_like(self.shard), torch.empty_like(self.gathered), torch.zeros_like(self.final), torch.cuda.Stream())
                        dev.exchange_add_lane(rc2.comm_handle, self.lane2[1].data_ptr(), self.lane2[2].data_ptr(), self.lane2[3].data_ptr(),
                                              self.lane2[4].cuda_stream)
                        added = True
                    except Exception as e:  # noqa: BLE001
                        print(f"[rank {rank}] second exchange lane unavailable ({e})", file=sys.stderr)
                    if not all_ok(added):
                        # the lane count must be the same on every rank: without agreement, go back to the render thread
                        dev.exchange_stop()
                        self.xthread, self.lane2 = False, None
            torch.cuda.synchronize()  # the buffers exist (torch's stream made them) before another stream touches them

        def inline(self, fr):
            """pack -> all-gather -> unpack on the device's public stream (no exchange thread)"""
            fr.pack_color_shard(self.shard.data_ptr(), self.shard.numel())
            if rc is not None:
                rc.all_gather_u8(self.shard.data_ptr(), self.gathered.data_ptr(), self.shard.numel(), stream.cuda_stream)
            else:
                with torch.cuda.stream(stream):
                    if backend == "nccl":
                        dist.all_gather_into_tensor(self.gathered, self.shard)
                    else:  # rehearsal: host-staged gather
                        stream.synchronize()
                        host = torch.empty(self.gathered.numel(), dtype=torch.uint8)
                        dist.all_gather_into_tensor(host, self.shard.cpu())
                        self.gathered.copy_(host)
            fr.unpack_color_shards(self.gathered.data_ptr(), self.final.data_ptr())

        def close(self):
            if self.xthread:
                dev.exchange_stop()
                self.xthread = False

    pending_error = []  # what fence() caught: check() reports it (after the clock has been read)

    def fence(ex):
        """what brackets a timed region, and nothing else: every frame issued so far has been handed to the GPU and has left it,
        on every rank (barrier + torch.cuda.synchronize(), as the driver's contract says)"""
        try:
            if ex is not None and ex.xthread:
                dev.exchange_drain()  # every handed-over frame has been issued; raises the exchange thread's first error
        except api.MtrError as e:
            pending_error.append(e)
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
            torch.cuda.synchronize()

    def check(what):
        """no rank is in error: a rank whose exchange or whose frames failed still took part in every collective (the library
        sends a clear-colour shard), reports here, and ALL ranks leave together -- nobody is left waiting for a peer that has
        gone.  Outside the timed regions: the agreement is a collective of its own."""
        err = pending_error.pop() if pending_error else None
        pending_error.clear()
        try:
            dev.synchronize()  # raises if a frame nobody waited for overflowed its bin queues (it would be missing triangles)
        except api.MtrError as e:
            err = err or e
        if sharded:
            if not all_ok(err is None):
                print(f"[rank {rank}] {what}: {'FAILED: ' + str(err) if err else 'ok here, another rank failed'}; every rank exits", file=sys.stderr, flush=True)
                watchdog.disarm()
                sys.exit(4)
        elif err is not None:
            raise err

    def sync(ex, what):
        fence(ex)
        check(what)

    def run_frames(work, ex, loop, n, shard_args):
        """n frames, submitted back to back, nothing waited for"""
        if ex is None or ex.xthread:
            loop.run(n)
            watchdog.progress += n
            return
        for _ in range(n):
            fr = api.Frame(dev, work.w, work.h)
            fr.set_shard(*shard_args)
            work.draw(fr)
            fr.submit()
            ex.inline(fr)
            fr.close()
            watchdog.progress += 1

    def checked_frame(work, ex, shard_args):
        """one frame with a host-side wait: grows the bin queues if needed and validates the device flags"""
        fr = api.Frame(dev, work.w, work.h)
        if shard_args:
            fr.set_shard(*shard_args)
        work.draw(fr)
        fr.submit()
        fr.wait()
        if ex is None:
            fr.close()
        elif ex.xthread:
            fr.submit_exchange()
        else:
            ex.inline(fr)
            fr.close()

    def time_leg(work, ex, shard_args, steps, warmup, ramp_frames, ramp_seconds, min_reps=1, budget_s=0.2):
        """(ms per step: median repetition, every repetition's ms per step, repetitions).  A timed region = exactly `steps`
        frames between two sync()s; max over the ranks per region."""
        loop = work.frame_loop(shard_args, ex is not None and ex.xthread)
        watchdog.arm(f"{work.name}: first frames + clock ramp", wd_seconds)
        # waited frames first: one that overflows the per-bin queue bound is re-run there and the bound doubles for the frames
        # that follow (1024 -> 4096 entries per bin on C5: three doublings at most before the un-waited frames start).  A
        # fixed count: every rank makes the same collective calls.
        for _ in range(4):
            checked_frame(work, ex, shard_args)
        # Device warm-up, untimed and independent of --warmup: the GPU's power management raises its clocks some 40 ms
        # after sustained load begins (one ~35 ms stall, then 52 us per frame instead of 58: tools/probe/hiccup.py), so a
        # short run would time the transition instead of the steady state.
        sync(ex, f"{work.name} first frame")
        t_r0 = time.perf_counter()
        # sharded: a fixed count, every rank must make the same number of collective calls
        run_frames(work, ex, loop, ramp_frames, shard_args)
        n_ramp = ramp_frames
        if not sharded:
            t_ramp = time.perf_counter() + ramp_seconds
            while time.perf_counter() < t_ramp:
                run_frames(work, ex, loop, 50, shard_args)
                n_ramp += 50
        sync(ex, f"{work.name} clock ramp")
        est = (time.perf_counter() - t_r0) / max(1, n_ramp)  # seconds per frame, roughly
        reps = max(min_reps, min(400, math.ceil(budget_s / max(1e-6, steps * est))))
        if sharded:  # the same count on every rank
            t = torch.tensor([reps], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            reps = int(t.item())
        watchdog.arm(f"{work.name}: warm-up + {reps} timed regions of {steps} frames", wd_seconds)
        run_frames(work, ex, loop, warmup, shard_args)
        sync(ex, f"{work.name} warm-up")
        dts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            run_frames(work, ex, loop, steps, shard_args)
            fence(ex)  # barrier + synchronize: the end of the timed region
            dts.append(time.perf_counter() - t0)
            check(f"{work.name} timed region")
        watchdog.disarm()
        if sharded:
            t = torch.tensor(dts, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = [float(v) for v in t.cpu()]
        per_step = sorted(d * 1e3 / steps for d in dts)
        return float(np.median(per_step)), per_step, reps

    def verify_leg(work, ex):
        """the gathered frame(s) this rank holds == the same scene rendered unsharded on this rank, bit for bit"""
        fr = api.Frame(dev, work.w, work.h)
        work.draw(fr)
        fr.end()
        ref = fr.color()
        fr.close()
        torch.cuda.synchronize()
        got = ex.final.cpu().numpy().reshape(work.h, work.w, 4)
        ok = bool((got == ref).all())
        if ex.lane2 is not None:  # odd frames went through the second lane into its own destination
            ok = ok and bool((ex.lane2[3].cpu().numpy().reshape(work.h, work.w, 4) == ref).all())
        print(f"[rank {rank}] verify {work.name}: gathered frame == unsharded frame: {ok}", file=sys.stderr, flush=True)
        return ok

    # ---- headline leg (the metric) ----
    own = choose_ownership(headline, os.environ.get("MTR_BENCH_OWNERSHIP", "bands")) if sharded else None
    shard_args = (rank, world) + own if sharded else None
    ex = Exchange(headline, own) if sharded else None
    xthread = bool(ex and ex.xthread)
    lane2 = ex.lane2 if ex else None
    ms_per_step, per_step, reps = time_leg(headline, ex, shard_args, args.steps, args.warmup,
                                           (3000 if backend == "nccl" else 20) if sharded else 50, 0.3)
    verified = {}
    if args.verify and sharded:
        verified["headline"] = verify_leg(headline, ex)
        if not all_ok(verified["headline"]):
            sys.exit(3)
    if ex is not None:
        ex.close()
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

    def new_frame():
        fr = api.Frame(dev, W, H)
        if sharded:
            fr.set_shard(*shard_args)
        headline.draw(fr)
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
    # kernel names as rocprofv3 prints them: k_geom<queue builder, culled launch, waves per SIMD it is built for (7: a draw of
    # fewer than 65536 geometry waves)>, k_tile_vis<textured, waves per bin, order lists, quad walk>.  The timed region's frames
    # share the GPU with their neighbours and run the <.., false> instantiation; a frame that has the GPU to itself (the latency
    # figure, stage_ms_serial) runs <.., true>, which walks bboxes in 2 x 2 quads: shorter alone, slower among other kernels
    vis_waves = 2 if stats["shard_bins"] > 4096 else (4 if stats["shard_bins"] > 1536 else 8)
    names = {"geom": "k_geom<%d, %s, %d>" % (((2 if vis else 1) if direct else 0), "true" if sharded and own[0] != api.OWN_INTERLEAVED else "false",
                                             7 if stats["chunks"] < 65536 else 8),
             "scan": "k_scan", "fill": "k_fill", "tile": ("k_tile_vis<false, %d, false, false>" % vis_waves) if vis else "k_tile<false>"}
    alg = {"geom": alg_geom, "tile": alg_tile}
    traffic_by_kernel = pmc_traffic() if world == 1 else {}
    kernels = {}
    for st in ("geom", "tile"):
        k = {"kernel": names[st], "algorithmic_bytes_per_launch": alg[st], "ms_overlapped": round(stage_ms[st], 5),
             "ms_standalone": round(stage_ms_serial[st], 5), "traffic": traffic_by_kernel.get(names[st])}
        if st == "tile" and vis:
            k["kernel_standalone"] = "k_tile_vis<false, %d, false, true>" % vis_waves
        k["frac_overlapped"] = round(alg[st] / (stage_ms[st] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        k["frac_standalone"] = round(alg[st] / (stage_ms_serial[st] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        kernels[st] = k
    dom = max(("geom", "tile"), key=lambda k: stage_ms[k])
    achieved = alg[dom] / (stage_ms[dom] * 1e-3) / 1e9
    # achieved / frac: the dominant kernel's OWN algorithmic bytes over its average launch duration in the pipelined
    # pattern of the timed region (what rocprofv3 --stats of this command reports); frac_standalone: the same bytes over
    # the kernel's duration with nothing else on the GPU; `kernels` has both kernels; `frame_*`: the whole frame's
    # algorithmic bytes over the measured ms_per_step
    roofline = {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "frac_standalone": kernels[dom]["frac_standalone"],
                "traffic": kernels[dom]["traffic"],
                "algorithmic_bytes_per_launch": alg[dom], "kernel_ms": round(stage_ms[dom], 5), "kernel_ms_standalone": round(stage_ms_serial[dom], 5),
                "note": "kernel_ms > ms_per_step is stream concurrency, not an inconsistency: three frames' kernels are co-resident "
                        "on three streams, so a launch lasts longer than a step; frac uses that overlapped duration (what rocprofv3 "
                        "--stats of this command reports), frac_standalone the kernel alone on the GPU (the visibility kernel then runs "
                        "its quad-walk instantiation, kernels.tile.kernel_standalone)",
                "kernels": kernels,
                "stage_ms": {k: round(v, 5) for k, v in stage_ms.items()},
                "stage_ms_serial": {k: round(v, 5) for k, v in stage_ms_serial.items()},
                "frame_algorithmic_bytes": alg_frame,
                "frame_gbps": round(alg_frame / (ms_per_step * 1e-3) / 1e9, 3),
                "frame_frac": round(alg_frame / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "residency_note": "the bench re-renders one frame whose ~80 MB of traffic fit the 256 MB Infinity Cache: `traffic` counts "
                                  "fabric requests, not DRAM; FETCH_SIZE x2 (gfx950 correction for wide streaming reads) overstates 32-byte record gathers",
                "traffic_source": ("profiles/pmc_traffic.json (kernel sources %s)" % kernel_source_hash()) if traffic_by_kernel else None}
    if direct:  # single-pass binning: k_scan / k_fill are not launched at all
        roofline["stage_note"] = "single-pass binning: no k_scan / k_fill launch (reported as 0)" + \
                                 ("; sharded frames: the geom stage includes the culling kernels" if sharded else "")
    latency = {"ms_per_frame_latency": round(med(lat_gpu), 5), "frames": len(lat_gpu), "definition": "SURVEY 8(d): one frame at a time, first kernel start to framebuffer complete in HBM (hipEvents on the frame's stream), median",
               "ms_per_frame_latency_host_clock": round(med(lat_host), 5), "mtris_per_s_at_latency": round(ntris / (med(lat_gpu) * 1e-3) / 1e6, 2)}

    # ---- BASELINE configs C4 / C5: multi-GPU by definition ("sharded across 2/4/8 GPUs", "8 GPUs") ----
    configs = None
    if args.configs == "on" or (args.configs == "auto" and sharded):
        configs = {}
        n5 = args.config_instances or 1024
        n4 = max(2, n5 // 8)

        def lat(n):  # 1024 -> 32 x 32, 128 -> 16 x 8; powers of two in between for rehearsals
            nx = 1 << (n.bit_length() // 2)
            return nx, max(1, n // nx)
        cw, chh = 3840, 2160
        vp4k = scene.to_f32_colmajor(scene.reference_view_proj(cw, chh))
        for cname in ("C4", "C5"):
            nx, ny = lat(n4 if cname == "C4" else n5)
            mats, pals = scene.instance_lattice(nx, ny)
            ninst = nx * ny
            if cname == "C4":
                cmd = scene.mesh50k()
                work = Workload("C4", cw, chh, cmd, view=vp4k, model_mats=mats, palettes=pals)
                desc = f"{ninst} instanced mesh50k ({ninst * 50000} triangles), 64 bones each, debug-id shader, {cw}x{chh}"
            else:
                ntex = max(1, ninst // 16)
                texs = [scene.random_bc7_texture(1024, 1024, seed=200 + i, opaque_modes_only=True) for i in range(ntex)]
                cmd = scene.mesh50k(textured=True, textures=texs)
                work = Workload("C5", cw, chh, cmd, view=vp4k, model_mats=mats, palettes=pals, tex_override=[i // 16 for i in range(ninst)])
                desc = f"{ninst} instanced mesh50k ({ninst * 50000} triangles), {ntex} BC7 1024x1024 albedo textures (opaque set), textured shader, {cw}x{chh}"
            steps_c = args.config_steps
            # the same scene unsharded, on every rank at once (each on its own GPU): t1 (max over ranks, as every leg)
            ramp_c = (200 if backend == "nccl" else 4) if sharded else 20
            t1_ms, t1_all, reps1 = time_leg(work, None, None, steps_c, 20 if backend == "nccl" or not sharded else 2, ramp_c, 0.1, min_reps=3, budget_s=0.3)
            entry = {"workload": desc, "triangles_per_frame": work.ntris, "steps": steps_c,
                     "unsharded": {"ms_per_frame": round(t1_ms, 5), "mtris_per_s": round(work.ntris / (t1_ms * 1e-3) / 1e6, 1), "repetitions": reps1,
                                   "where": "same run, same build: every rank renders it at the same time on its own GPU, the slowest counts"}}
            if sharded:
                own_c = choose_ownership(work, os.environ.get("MTR_BENCH_CONFIG_OWNERSHIP", "bands-balanced"))
                sh_c = (rank, world) + own_c
                ex_c = Exchange(work, own_c)
                tn_ms, tn_all, repsn = time_leg(work, ex_c, sh_c, steps_c, 20 if backend == "nccl" else 2, ramp_c, 0.0, min_reps=3, budget_s=0.3)
                if args.verify:
                    verified[cname] = verify_leg(work, ex_c)
                    if not all_ok(verified[cname]):
                        sys.exit(3)
                ex_c.close()
                entry["sharded"] = {"ms_per_frame": round(tn_ms, 5), "mtris_per_s": round(work.ntris / (tn_ms * 1e-3) / 1e6, 1), "n_gpus": world,
                                    "repetitions": repsn, "ms_per_frame_min_max": [round(tn_all[0], 5), round(tn_all[-1], 5)],
                                    "bands": list(map(int, own_c[2])) if own_c[2] is not None else None,
                                    "ownership": os.environ.get("MTR_BENCH_CONFIG_OWNERSHIP", "bands-balanced"),
                                    "definition": "frames submitted back to back; a frame is complete when the gathered 4K colour buffer is on every rank (after the all-gather and the unpack); max over ranks"}
                entry["efficiency"] = round(t1_ms / (world * tn_ms), 4)
                entry["efficiency_definition"] = "t1 / (N x tN): unsharded ms/frame over N times the sharded ms/frame, same run"
            configs[cname] = entry
            work.close()
        if backend != "nccl" and sharded:
            configs["note"] = "rehearsal backend (gloo through host memory, every rank on one GPU): plumbing only, the times mean nothing"

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
            "repetitions": reps, "ms_per_step_min_max": [round(per_step[0], 5), round(per_step[-1], 5)],
            "ms_per_step_is": "the median of `repetitions` timed regions of exactly `steps` frames each (every region bracketed by barrier + synchronize)",
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
        if configs is not None:
            out["configs"] = configs
        if verified:
            out["verified"] = verified
        print(json.dumps(out))
    headline.close()
    dev.close()
    if sharded:
        if rc is not None:
            torch.cuda.synchronize()
            rc.close()
            if rc2 is not None:
                rc2.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
