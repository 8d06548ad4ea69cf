"""Does a second host thread for the exchange (pack -> ncclAllGather -> unpack -> frame destroy) raise the frame rate of
a rank whose loop is bound by host submission time?  One GPU, real RCCL communicator of world size 1, but the frame is
rendered as shard 0 of FAKE (default 8) so the GPU has only a rank's share of the work (~28 us) and the host is the limit.
usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 tools/probe/exchange_thread.py"""
import os, sys, time, threading, collections
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
from mt_renderer_amd import api, scene, rccl

torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29545")
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
FAKE = int(os.environ.get("FAKE", "8"))
W, H = 1920, 1080
md = scene.headline_model(); pal = scene.bone_palette(); M = scene.to_f32_colmajor(scene.headline_transform(W, H))
stream = torch.cuda.Stream(); xstream = torch.cuda.Stream()
dev = api.Device(0, stream=stream.cuda_stream)
model = api.Model.new(dev, md); model.set_palette(pal)
nbytes = int(api.lib.mtr_shard_bytes(W, H, FAKE))
shard = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
gathered = torch.zeros(nbytes * FAKE, dtype=torch.uint8, device="cuda")
final = torch.empty(W * H * 4, dtype=torch.uint8, device="cuda")
rc = rccl.Rccl(); rc.init(rc.unique_id(), 1, 0)


def render():
    fr = api.Frame(dev, W, H); fr.set_shard(0, FAKE); model.render(fr, M); fr.submit()
    return fr


def render_x():
    fr = api.Frame(dev, W, H); fr.set_shard(0, FAKE); model.render(fr, M); fr.submit_exchange()


def exchange(fr, s):
    fr.pack_color_shard(shard.data_ptr(), nbytes, stream=s)
    rc.all_gather_u8(shard.data_ptr(), gathered.data_ptr(), nbytes, s)
    dev.unpack_color_shards(gathered.data_ptr(), FAKE, W, H, final.data_ptr(), stream=s)
    fr.close()


class Worker:
    def __init__(self, depth=8):
        self.q = collections.deque(); self.items = threading.Semaphore(0); self.space = threading.Semaphore(depth)
        self.idle = threading.Event(); self.idle.set(); self.pending = 0; self.lock = threading.Lock()
        self.t = threading.Thread(target=self.run, daemon=True); self.t.start()

    def run(self):
        torch.cuda.set_device(0)
        while True:
            self.items.acquire()
            fr = self.q.popleft()
            if fr is None:
                return
            exchange(fr, xstream.cuda_stream)
            with self.lock:
                self.pending -= 1
                if self.pending == 0:
                    self.idle.set()
            self.space.release()

    def put(self, fr):
        self.space.acquire()
        with self.lock:
            self.pending += 1
            self.idle.clear()
        self.q.append(fr); self.items.release()

    def drain(self):
        self.idle.wait()

    def stop(self):
        self.drain(); self.q.append(None); self.items.release(); self.t.join()


def timed(label, step, drain):
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        for _ in range(50): step()
    drain(); torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(2000): step()
        t1 = time.perf_counter()
        drain(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{label}: {1e6*(t2-t0)/2000:.1f} us/frame (host loop {1e6*(t1-t0)/2000:.1f})", flush=True)


timed("render only            ", lambda: render().close(), lambda: None)
timed("same thread, public str", lambda: exchange(render(), stream.cuda_stream), lambda: None)
timed("same thread, xstream   ", lambda: exchange(render(), xstream.cuda_stream), lambda: None)
w = Worker()
timed("worker thread, xstream ", lambda: w.put(render()), w.drain)
w.stop()
# the same hand-over in the library: a C++ thread, no interpreter lock involved
dev.exchange_start(rc.allgather_addr, rc.comm_handle, rccl.ncclUint8, shard.data_ptr(), nbytes, gathered.data_ptr(), final.data_ptr(),
                   FAKE, xstream.cuda_stream)
timed("library thread, xstream", lambda: render_x(), dev.exchange_drain)
dev.exchange_stop()
# the gathered frame's own shard must be what an unsharded render gives for those bins
fr = api.Frame(dev, W, H); model.render(fr, M); fr.end(); ref = fr.color(); fr.close()
got = final.cpu().numpy().reshape(H, W, 4)
nbx = (W + 15) // 16
import numpy as np
ok = True
for b in range(0, nbx * ((H + 15) // 16), FAKE):
    x, y = (b % nbx) * 16, (b // nbx) * 16
    ok = ok and bool((got[y:y + 16, x:x + 16] == ref[y:y + 16, x:x + 16]).all())
print("own bins identical to the unsharded frame:", ok)
dist.destroy_process_group()
sys.exit(0 if ok else 3)
