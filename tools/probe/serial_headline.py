"""the headline frame, one at a time (each waited for): under tools/probe/prof_probe.sh the kernel trace gives the
stand-alone duration of every kernel, without the gaps hipEvents add around them"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mt_renderer_amd import api, scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, H = 1920, 1080
dev = api.Device(0)
md = scene.headline_model(); pal = scene.bone_palette(); M = scene.to_f32_colmajor(scene.headline_transform(W, H))
model = api.Model.new(dev, md); model.set_palette(pal)
for i in range(n):
    fr = api.Frame(dev, W, H); model.render(fr, M); fr.end(); fr.close()
model.close(); dev.close()
