"""Per-rank cost of sharded frames on ONE GPU (no exchange), multi-GPU v2: ownership map x world size x EVERY rank,
frames in flight (the bench's submission pattern).  A job of N ranks runs at the pace of its slowest rank, so the
figure that matters is max over ranks; `bound` is what 0.85 per-GPU efficiency leaves a rank: t(N=1) / (0.85 N).

    python tools/probe/shard_cost_v2.py [headline|c4|c5|all] [out.json]
    MTR_PROBE_MAPS=bands-balanced,bands-equal   restricts the ownership maps; MTR_PROBE_WORLDS=8 the world sizes
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from mt_renderer_amd import api, scene, sharding

what = sys.argv[1] if len(sys.argv) > 1 else "headline"
out_path = sys.argv[2] if len(sys.argv) > 2 else None
dev = api.Device(0)
results = {}


def measure(make_frame, nframes, warm_s=0.25):
    for _ in range(4):  # waited frames: a queue overflow is handled (and the bound raised) before the un-waited ones follow
        fr = make_frame(); fr.submit(); fr.wait(); st = fr.stats(); fr.close()
    t_end = time.perf_counter() + warm_s
    while time.perf_counter() < t_end:
        for _ in range(20):
            fr = make_frame(); fr.submit(); fr.close()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nframes):
        fr = make_frame(); fr.submit(); fr.close()
    torch.cuda.synchronize()
    dev.synchronize()
    dt = (time.perf_counter() - t0) / nframes
    dev.set_profiling(True)
    acc = {}
    for _ in range(6):
        fr = make_frame(); fr.end()
        for k, v in fr.timings_ms().items():
            acc[k] = acc.get(k, 0.0) + v / 6
        fr.close()
    dev.set_profiling(False)
    st["geom_us"], st["tile_us"] = acc["geom"] * 1e3, acc["tile"] * 1e3
    return dt, st


def row_weights(make_unsharded, W, H):
    """entries per bin row of the unsharded frame + a constant per bin: what balanced bands equalise"""
    fr = make_unsharded(); fr.submit(); fr.wait()
    e, _ = fr.bin_counts()
    fr_entries = fr.stats()["bin_entries"]
    fr.close()
    nbx, nby, _ = sharding.grid(W, H)
    rows = e.reshape(nby, nbx).sum(axis=1)
    print(f"  calibration frame: {int(e.sum())} entries in the bin queues (frame statistics: {fr_entries}), rows with entries {int((rows > 0).sum())} of {nby}", flush=True)
    return rows.astype(np.float64) + 8.0 * nbx


def run(name, W, H, draw, nframes):
    def unsharded():
        fr = api.Frame(dev, W, H); draw(fr); return fr
    t1, st1 = measure(unsharded, nframes)
    print(f"{name}: N=1 {t1*1e6:.1f} us/frame, {st1['tris_setup']} triangles set up, {st1['bin_entries']} entries, serial geom {st1['geom_us']:.1f} tile {st1['tile_us']:.1f} us", flush=True)
    rw = row_weights(unsharded, W, H)
    res = {"n1_us": t1 * 1e6, "maps": {}}
    only_maps = [m for m in os.environ.get("MTR_PROBE_MAPS", "").split(",") if m]
    worlds = [int(w) for w in os.environ.get("MTR_PROBE_WORLDS", "2,4,8").split(",")]
    for map_name, own_map, param, balanced in (("interleaved", sharding.INTERLEAVED, 0, False), ("bands-equal", sharding.BANDS, 0, False),
                                               ("bands-balanced", sharding.BANDS, 0, True), ("supertiles8", sharding.SUPERTILES, 3, False)):
        if only_maps and map_name not in only_maps:
            continue
        for world in worlds:
            bands = sharding.balanced_bands(rw, world) if balanced else None
            per_rank = []
            for rank in range(world):
                def sharded():
                    fr = api.Frame(dev, W, H); fr.set_shard(rank, world, own_map, param, bands); draw(fr); return fr
                t, st = measure(sharded, max(20, nframes // 2), warm_s=0.1)
                per_rank.append((t * 1e6, st["tris_setup"], st["chunks_culled"], st["chunks"], st["shard_bins"], st["geom_us"], st["tile_us"]))
            worst = max(p[0] for p in per_rank)
            bound = t1 * 1e6 / (0.85 * world)
            print(f"  {map_name:15s} N={world}: worst rank {worst:8.1f} us (mean {np.mean([p[0] for p in per_rank]):8.1f}), bound {bound:8.1f} us -> "
                  f"{worst / bound:5.2f}x the bound, efficiency ceiling {t1 * 1e6 / (world * worst):.2f}; "
                  f"kept chunks {[round(1 - p[2] / max(p[3], 1), 2) for p in per_rank]} serial geom/tile us {[(round(p[5]), round(p[6])) for p in per_rank]}" + (f" bands {bands}" if bands else ""), flush=True)
            res["maps"].setdefault(map_name, {})[str(world)] = {"worst_us": worst, "bound_us": bound, "ratio": worst / bound,
                                                                "efficiency_ceiling": t1 * 1e6 / (world * worst), "bands": bands,
                                                                "per_rank": [dict(us=p[0], tris_setup=p[1], chunks_culled=p[2], chunks=p[3], bins=p[4], geom_us=p[5], tile_us=p[6]) for p in per_rank]}
    results[name] = res


if what in ("headline", "all"):
    W, H = 1920, 1080
    md = scene.headline_model(); pal = scene.bone_palette(); M = scene.to_f32_colmajor(scene.headline_transform(W, H))
    model = api.Model.new(dev, md); model.set_palette(pal)
    run("headline 1M tris 1080p", W, H, lambda fr: model.render(fr, M), 4000)
    model.close()
if what in ("c4", "all"):
    W, H = 3840, 2160
    vp = scene.to_f32_colmajor(scene.reference_view_proj(W, H))
    mats, pals = scene.instance_lattice(16, 8)
    m = api.Model.new(dev, scene.mesh50k()); batch = api.Batch(dev, m, mats, pals, None)
    run("C4 128 inst 4K", W, H, lambda fr: fr.draw_batch(batch, vp), 1000)
    batch.close(); m.close()
if what in ("c5", "all"):
    W, H = 3840, 2160
    vp = scene.to_f32_colmajor(scene.reference_view_proj(W, H))
    mats, pals = scene.instance_lattice(32, 32)
    texs = [scene.random_bc7_texture(1024, 1024, seed=200 + i, opaque_modes_only=True) for i in range(64)]
    m = api.Model.new(dev, scene.mesh50k(textured=True, textures=texs)); batch = api.Batch(dev, m, mats, pals, [i // 16 for i in range(1024)])
    run("C5 1024 inst BC7 4K", W, H, lambda fr: fr.draw_batch(batch, vp), 200)
    batch.close(); m.close()
dev.close()
if out_path:
    with open(out_path, "w") as f:
        json.dump(results, f, indent=1)
