"""The N > 1 frame loop of bench.py (set_shard -> render -> pack -> all_gather_into_tensor -> unpack) with the REAL nccl
(= RCCL) backend and a world of ONE rank: no second GPU is needed, but process-group creation, the collective call on
the library's public stream and the stream ordering around it are the ones the driver's multi-GPU runs will use.
usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 tools/probe/nccl_one_rank.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.distributed as dist
from mt_renderer_amd import api, scene

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
NODIST = os.environ.get("NODIST") == "1"
if not NODIST:
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0), rank=rank, world_size=world)
W, H = 1920, 1080
md = scene.headline_model(); pal = scene.bone_palette(); M = scene.to_f32_colmajor(scene.headline_transform(W, H))
stream = torch.cuda.Stream()
dev = api.Device(0, stream=stream.cuda_stream)
model = api.Model.new(dev, md); model.set_palette(pal)
nbytes = int(api.lib.mtr_shard_bytes(W, H, world))
MODE = os.environ.get("MODE", "full")
NX = int(os.environ.get("NX", "4"))  # exchange contexts (1 = everything on the device's public stream, as a baseline)
final = torch.empty(W * H * 4, dtype=torch.uint8, device="cuda")
xstreams = [torch.cuda.Stream() for _ in range(NX)]
shards = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(NX)]
gathereds = [torch.empty(nbytes * world, dtype=torch.uint8, device="cuda") for _ in range(NX)]
shard, gathered = shards[0], gathereds[0]
count = [0]
rc = None
if MODE == "direct":
    from mt_renderer_amd import rccl
    rc = rccl.Rccl()
    box = [rc.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    rc.init(box[0], world, rank)

def one_frame():
    fr = api.Frame(dev, W, H)
    if os.environ.get("NOSHARD") != "1": fr.set_shard(rank, world)
    model.render(fr, M); fr.submit()
    x = count[0] % NX; count[0] += 1
    if MODE == "render":
        pass
    elif MODE == "nogather":
        fr.pack_color_shard(shards[x].data_ptr(), nbytes, stream=xstreams[x].cuda_stream)
        dev.unpack_color_shards(shards[x].data_ptr() if world == 1 else gathereds[x].data_ptr(), world, W, H, final.data_ptr(), stream=xstreams[x].cuda_stream)
    elif MODE == "copygather":  # the collective replaced by a device copy on the same stream
        fr.pack_color_shard(shards[x].data_ptr(), nbytes, stream=xstreams[x].cuda_stream)
        with torch.cuda.stream(xstreams[x]):
            gathereds[x].copy_(shards[x], non_blocking=True)
        dev.unpack_color_shards(gathereds[x].data_ptr(), world, W, H, final.data_ptr(), stream=xstreams[x].cuda_stream)
    elif MODE == "direct":  # ncclAllGather through ctypes on the public stream
        fr.pack_color_shard(shards[0].data_ptr(), nbytes)
        rc.all_gather_u8(shards[0].data_ptr(), gathereds[0].data_ptr(), nbytes, stream.cuda_stream)
        dev.unpack_color_shards(gathereds[0].data_ptr(), world, W, H, final.data_ptr())
    elif NX == 1:
        fr.pack_color_shard(shards[0].data_ptr(), nbytes)
        with torch.cuda.stream(stream):
            dist.all_gather_into_tensor(gathereds[0], shards[0])
        dev.unpack_color_shards(gathereds[0].data_ptr(), world, W, H, final.data_ptr())
    else:
        fr.pack_color_shard(shards[x].data_ptr(), nbytes, stream=xstreams[x].cuda_stream)
        with torch.cuda.stream(xstreams[x]):
            dist.all_gather_into_tensor(gathereds[x], shards[x])
        dev.unpack_color_shards(gathereds[x].data_ptr(), world, W, H, final.data_ptr(), stream=xstreams[x].cuda_stream)
    fr.close()

t_end = time.perf_counter() + 0.4  # past the GPU's clock ramp (tools/probe/hiccup.py)
while time.perf_counter() < t_end:
    for _ in range(50): one_frame()
torch.cuda.synchronize(); (None if NODIST else dist.barrier()); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(1000): one_frame()
torch.cuda.synchronize(); (None if NODIST else dist.barrier()); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 1000
# host time of each step of the loop (no device sync inside: pure submission cost)
acc = {"frame": 0.0, "pack": 0.0, "gather": 0.0, "unpack": 0.0}
for _ in range(500):
    a = time.perf_counter()
    fr = api.Frame(dev, W, H); fr.set_shard(rank, world); model.render(fr, M); fr.submit()
    b = time.perf_counter()
    fr.pack_color_shard(shard.data_ptr(), shard.numel())
    c = time.perf_counter()
    with torch.cuda.stream(stream):
        dist.all_gather_into_tensor(gathered, shard)
    d = time.perf_counter()
    dev.unpack_color_shards(gathered.data_ptr(), world, W, H, final.data_ptr())
    fr.close()
    e = time.perf_counter()
    acc["frame"] += b - a; acc["pack"] += c - b; acc["gather"] += d - c; acc["unpack"] += e - d
torch.cuda.synchronize()
print("host us per call:", {k: round(v / 500 * 1e6, 1) for k, v in acc.items()})
fr = api.Frame(dev, W, H); model.render(fr, M); fr.end(); ref = fr.color(); fr.close()
ok = bool((final.cpu().numpy().reshape(H, W, 4) == ref).all())
print(f"MODE={MODE} NX={NX} nccl world={world}: {dt*1e6:.1f} us/frame with pack + all_gather + unpack; gathered frame == direct frame: {ok}")
None if NODIST else dist.destroy_process_group()
sys.exit(0 if ok else 3)
