"""Does initialising the nccl (RCCL) process group, without issuing any collective, slow the render loop down?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
from mt_renderer_amd import api, scene
W, H = 1920, 1080
torch.cuda.set_device(0)
if os.environ.get("INIT") == "1":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
    if os.environ.get("TOUCH") == "1":
        t = torch.zeros(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
md = scene.headline_model(); pal = scene.bone_palette(); M = scene.to_f32_colmajor(scene.headline_transform(W, H))
stream = torch.cuda.Stream()
dev = api.Device(0, stream=stream.cuda_stream); model = api.Model.new(dev, md); model.set_palette(pal)
def one():
    fr = api.Frame(dev, W, H); model.render(fr, M); fr.submit(); fr.close()
fr = api.Frame(dev, W, H); model.render(fr, M); fr.end(); fr.close()
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    for _ in range(50): one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(1000): one()
torch.cuda.synchronize()
print(f"INIT={os.environ.get('INIT')} TOUCH={os.environ.get('TOUCH')}: {(time.perf_counter()-t0)*1e3:.1f} us/frame")
