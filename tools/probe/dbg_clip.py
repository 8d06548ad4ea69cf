import sys, numpy as np
sys.path.insert(0, '.')
from mt_renderer_amd import api, scene
from tests.helpers import render_gpu, render_oracle
from tests.pixel_scenes import pixel_model
from tests.test_gpu_clipping import M_W
dev = api.Device(0)
for nv in (20, 40, 64, 80):
    verts = [(40000.0 * np.cos(np.radians(125.0 * i)), 40000.0 * np.sin(np.radians(125.0 * i)), 1.0) for i in range(nv)]
    md = pixel_model([dict(verts=verts, indices=list(range(len(verts))), topology=scene.TOPO_STRIP, debug_id=2)])
    md.prim_states = np.array([(0, 1, 1, 1)], dtype=np.uint8)
    draws = [dict(md=md, M=M_W)]
    ref = render_oracle(128, 96, draws)
    try:
        g = render_gpu(dev, 128, 96, draws, tile_mode=api.TILE_AUTO)
        print(nv, "gpu", g[2]["tris_setup"], g[2]["bin_entries"], "oracle", ref[2]["tris_setup"], "same", bool((g[0] == ref[0]).all()))
    except api.MtrError as e:
        print(nv, "error", e, "oracle", ref[2]["tris_setup"])
