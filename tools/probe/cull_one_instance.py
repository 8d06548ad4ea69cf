"""chunk culling of ONE instance of the C5 lattice that straddles a band edge: what the chunk test keeps, as a batch
draw of one and as a single model draw, against the ideal from its actual vertex positions (computed on the host)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mt_renderer_amd import api, scene, sharding
W, H = 3840, 2160
dev = api.Device(0)
md = scene.mesh50k()
mats, pals = scene.instance_lattice(32, 32)
vp = scene.reference_view_proj(W, H)
m = api.Model.new(dev, md)
for k in (500, 501, 300, 700):
    one = api.Batch(dev, m, mats[k:k + 1], pals[k:k + 1], None)
    for rank in (3, 4):
        fr = api.Frame(dev, W, H); fr.set_shard(rank, 8, sharding.BANDS); fr.draw_batch(one, scene.to_f32_colmajor(vp)); fr.end()
        st = fr.stats(); fr.close()
        M = scene.to_f32_colmajor(vp @ mats[k].reshape(4, 4).T.astype(np.float64))
        m.set_palette(pals[k])
        fr = api.Frame(dev, W, H); fr.set_shard(rank, 8, sharding.BANDS); m.render(fr, M); fr.end()
        st2 = fr.stats(); fr.close()
        print(f"instance {k} rank {rank}: batch-of-one culled {st['chunks_culled']}/{st['chunks']} setup {st['tris_setup']}; single model culled {st2['chunks_culled']}/{st2['chunks']} setup {st2['tris_setup']}")
    one.close()
m.close(); dev.close()
