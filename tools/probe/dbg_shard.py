import sys, numpy as np
sys.path.insert(0, '.')
from mt_renderer_amd import api, scene, sharding
from tests.helpers import render_gpu
from tests.test_gpu_sharding import _scene
dev = api.Device(0)
w, h = 333, 171
draws_all = _scene(w, h)
for di in range(3):
    draws = [draws_all[di]]
    full = render_gpu(dev, w, h, draws, tile_mode=api.TILE_AUTO)
    for own_map in (0, 1, 2):
        for world in (2, 8):
            owner = sharding.owner_map(w, h, world, own_map, 0, None)
            for cull in (False, True):
                dev.set_culling(cull)
                bad = []
                for rank in range(world):
                    part = render_gpu(dev, w, h, draws, shard=(rank, world, own_map, 0, None), tile_mode=api.TILE_AUTO)
                    own = owner == rank
                    nbad = int((part[0][own] != full[0][own]).any(axis=-1).sum())
                    bad.append((nbad, part[2]["chunks_culled"], part[2]["chunks"], part[2]["tris_setup"]))
                print(f"draw {di} map {own_map} world {world} cull {cull}: {bad}", flush=True)
dev.close()
