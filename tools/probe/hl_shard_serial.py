"""the headline frame as rank r of 8 (equal bands), one frame at a time: under tools/probe/prof_probe.sh the kernel trace
gives the stand-alone duration of the culling, geometry and tile kernels of a sharded frame"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mt_renderer_amd import api, scene, sharding
W, H = 1920, 1080
dev = api.Device(0)
md = scene.headline_model(); pal = scene.bone_palette(); M = scene.to_f32_colmajor(scene.headline_transform(W, H))
model = api.Model.new(dev, md); model.set_palette(pal)
for rank in (1, 3, 6):
    for i in range(100):
        fr = api.Frame(dev, W, H); fr.set_shard(rank, 8, sharding.BANDS); model.render(fr, M); fr.end(); fr.close()
model.close(); dev.close()
