"""CPU, world_size 2, gloo: the N > 1 path of bench.py -- render own bins, pack bin-major, all-gather,
unpack -- with the oracle standing in for the renderer (the HIP kernels are checked against the same
index math in tests/test_gpu_sharding.py)."""
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


def _worker(rank, world, port, w, h, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=16, cols=24)
    f = orc.OracleFrame(w, h)
    f.draw(orc.OracleModel(md), scene.to_f32_colmajor(scene.headline_transform(w, h)), scene.bone_palette())
    full = f.color()
    own = sharding.owner_map(w, h, world) == rank
    mine = np.where(own[..., None], full, 0).astype(np.uint8)  # a rank only has its own bins
    shard = torch.from_numpy(sharding.pack_shard(mine, rank, world).reshape(-1).copy())
    gathered = torch.empty(shard.numel() * world, dtype=torch.uint8)
    dist.all_gather_into_tensor(gathered, shard)
    out = sharding.unpack_shards(gathered.numpy().reshape(world, -1, sharding.BIN, sharding.BIN, 4), w, h)
    q.put((rank, bool((out == full).all()), int(own.sum())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("size", [(200, 120), (256, 144)])
def test_two_rank_gather_rebuilds_the_frame(size):
    w, h = size
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, w, h, q)) for r in range(2)]
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
        shards = np.stack([sharding.pack_shard(img, r, world) for r in range(world)])
        assert shards.shape[1] * shards.shape[2] * shards.shape[3] * 4 == sharding.shard_bytes(w, h, world)
        assert (sharding.unpack_shards(shards, w, h) == img).all()
