"""CPU, world_size 2, gloo: the N > 1 path of bench.py -- render own bins, pack bin-major, all-gather,
unpack -- for every ownership map (interleaved bins, bands of bin rows, super-tiles), with the oracle standing in
for the renderer (the HIP kernels are checked against the same index math in tests/test_gpu_sharding.py and
tests/test_gpu_cases.py::test_sharded_bins_and_pack_unpack)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mt_renderer_amd import scene, sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, own, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=16, cols=24)
    f = orc.OracleFrame(w, h)
    f.draw(orc.OracleModel(md), scene.to_f32_colmajor(scene.headline_transform(w, h)), scene.bone_palette())
    full = f.color()
    own_px = sharding.owner_map(w, h, world, *own) == rank
    mine = np.where(own_px[..., None], full, 0).astype(np.uint8)  # a rank only has its own bins
    shard = torch.from_numpy(sharding.pack_shard(mine, rank, world, *own).reshape(-1).copy())
    assert shard.numel() == sharding.shard_bytes(w, h, world, *own)
    gathered = torch.empty(shard.numel() * world, dtype=torch.uint8)
    dist.all_gather_into_tensor(gathered, shard)
    out = sharding.unpack_shards(gathered.numpy().reshape(world, -1, sharding.BIN, sharding.BIN, 4), w, h, *own)
    q.put((rank, bool((out == full).all()), int(own_px.sum())))
    dist.barrier()
    dist.destroy_process_group()


OWNERSHIPS = {"interleaved": (sharding.INTERLEAVED, 0, None), "bands": (sharding.BANDS, 0, None),
              "bands-uneven": (sharding.BANDS, 0, [0, 2, 8]), "supertiles": (sharding.SUPERTILES, 1, None)}


@pytest.mark.parametrize("own", list(OWNERSHIPS), ids=list(OWNERSHIPS))
@pytest.mark.parametrize("size", [(200, 120), (256, 144)])
def test_two_rank_gather_rebuilds_the_frame(size, own):
    w, h = size
    own = OWNERSHIPS[own]
    if own[2] is not None:
        own = (own[0], own[1], [0, 2, (h + 15) // 16])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, w, h, own, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert sum(n for _, _, n in res) == w * h


def test_shard_index_math_roundtrip():
    rng = np.random.default_rng(1)
    for (w, h, world) in [(64, 48, 2), (333, 171, 3), (100, 37, 8), (16, 16, 4)]:
        img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        nby = (h + 15) // 16
        for own in [(sharding.INTERLEAVED, 0, None), (sharding.BANDS, 0, None), (sharding.SUPERTILES, 0, None), (sharding.SUPERTILES, 2, None),
                    (sharding.BANDS, 0, sharding.balanced_bands(rng.uniform(size=nby) ** 4, world))]:
            shards = np.stack([sharding.pack_shard(img, r, world, *own) for r in range(world)])
            assert shards.shape[1] * shards.shape[2] * shards.shape[3] * 4 == sharding.shard_bytes(w, h, world, *own)
            assert (sharding.unpack_shards(shards, w, h, *own) == img).all()
            owners = sharding.owner_map(w, h, world, *own)
            assert owners.min() >= 0 and owners.max() < world
