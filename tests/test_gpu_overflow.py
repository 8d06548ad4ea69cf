"""Frames that overflow a bounded per-bin queue and are NEVER waited for (ADVICE high):
  * submit + destroy: the tile kernels must not render from the incomplete queues (every bin stays at the clear colour,
    nothing out of bounds is read) and the next call that can return an error reports MTR_E_OVERFLOW, once; later frames
    fit (the bound was raised) and are bit-exact against the oracle;
  * submit_exchange: the exchange thread re-runs the frame through the exact two-pass queues BEFORE packing, so the
    gathered frame equals the unsharded render."""
import ctypes as C

import numpy as np
import pytest

from mt_renderer_amd import scene
from tests.helpers import assert_same, render_gpu, render_oracle
from tests.pixel_scenes import pixel_model, pixel_to_ndc_matrix


def _dense_bin_scene(size=32, quads=400, seed=7):
    rng = np.random.default_rng(seed)
    verts, idx = [], []
    for q in range(quads):  # 2 * quads triangles inside one 16x16 bin
        x0, y0, sz = rng.uniform(1, 10), rng.uniform(1, 10), rng.uniform(1.5, 4.0)
        b = 4 * q
        verts += [(x0, y0, .5 - q * 1e-4), (x0, y0 + sz, .5 - q * 1e-4), (x0 + sz, y0, .5 - q * 1e-4), (x0 + sz, y0 + sz, .5 - q * 1e-4)]
        idx += [b, b + 1, b + 2, b + 3, 0xFFFF]
    md = pixel_model([dict(verts=verts, indices=idx, topology=scene.TOPO_STRIP, debug_id=4)])
    return md, pixel_to_ndc_matrix(size, size)


@pytest.mark.gpu
@pytest.mark.parametrize("tile_mode", ["auto", "ordered"])
def test_unwaited_overflow_is_latched_never_a_wrong_frame(tile_mode):
    from mt_renderer_amd import api
    md, M = _dense_bin_scene()
    ref = render_oracle(32, 32, [dict(md=md, M=M)])
    with api.Device(0) as dev:
        dev.set_tile_mode(api.TILE_ORDERED if tile_mode == "ordered" else api.TILE_AUTO)
        dev.set_binning(True, 64)
        model = api.Model.new(dev, md)
        fr = api.Frame(dev, 32, 32)
        model.render(fr, M)
        fr.submit()
        # what a consumer of the device memory would see: not a frame with some triangles missing, but the clear colour
        import torch
        nbytes = int(api.lib.mtr_shard_bytes(32, 32, 1))
        packed = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        fr.pack_color_shard(packed.data_ptr(), nbytes)  # the frame's own bins, bin-major, on the device's public stream
        torch.cuda.synchronize()
        assert (packed.cpu().numpy() == 255).all(), "an overflowed frame must be left at the clear colour, not rendered from incomplete queues"
        fr.close()  # never waited
        with pytest.raises(api.MtrError, match="overflowed its bin queues"):
            dev.synchronize()
        dev.synchronize()  # reported once
        # the bound was doubled; keep going until the scene fits (each overflowing, un-waited frame doubles it again)
        errors = 0
        for _ in range(8):
            try:
                fr = api.Frame(dev, 32, 32)
            except api.MtrError:
                errors += 1
                continue
            model.render(fr, M)
            fr.submit()
            fr.close()
            try:
                dev.synchronize()
            except api.MtrError:
                errors += 1
        assert 1 <= errors <= 6
        fr = api.Frame(dev, 32, 32)
        model.render(fr, M)
        fr.submit()
        fr.wait()
        st = fr.stats()
        assert st["binning"] == 1, "by now the single-pass queues hold the scene"
        assert_same((fr.color(), fr.depth(), st), ref, "after the latch")
        fr.close()
        dev.synchronize()
        model.close()


@pytest.mark.gpu
def test_overflow_reported_by_next_frame_begin():
    from mt_renderer_amd import api
    md, M = _dense_bin_scene()
    with api.Device(0) as dev:
        dev.set_binning(True, 64)
        model = api.Model.new(dev, md)
        fr = api.Frame(dev, 32, 32); model.render(fr, M); fr.submit(); fr.close()
        import torch
        torch.cuda.synchronize()
        with pytest.raises(api.MtrError, match="never waited for"):
            api.Frame(dev, 32, 32)
        fr = api.Frame(dev, 32, 32)  # reported once
        fr.close()
        model.close()


@pytest.mark.gpu
def test_exchange_thread_reruns_an_overflowed_frame():
    """world 2, one GPU standing in for both ranks one after the other; qcap 64 against a bin with 800 triangles"""
    import torch
    from mt_renderer_amd import api
    W = H = 48
    md, M = _dense_bin_scene(size=W, quads=300)
    ref = render_oracle(W, H, [dict(md=md, M=M)])
    xs = torch.cuda.Stream()
    with api.Device(0) as dev:
        model = api.Model.new(dev, md)
        world = 2
        nbytes = int(api.lib.mtr_shard_bytes(W, H, world))
        # a send buffer LARGER than the shard: the all-gather count must still be the shard size (ADVICE: stride mismatch)
        shard = torch.zeros(nbytes + 4096, dtype=torch.uint8, device="cuda")
        gathered = torch.zeros(nbytes * world, dtype=torch.uint8, device="cuda")
        final = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
        state = {"rank": 0, "counts": set()}

        @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p)
        def fake_allgather(send, recv, count, dtype, comm, stream):
            state["counts"].add(count)
            with torch.cuda.stream(xs):
                r = state["rank"]
                gathered[r * count:(r + 1) * count].copy_(shard[:count], non_blocking=True)
            return 0

        dev.exchange_start(C.cast(fake_allgather, C.c_void_p).value, 0, 1, shard.data_ptr(), shard.numel(), gathered.data_ptr(),
                           final.data_ptr(), world, xs.cuda_stream)
        dev.set_binning(True, 64)
        for r in range(world):
            state["rank"] = r
            for _ in range(3):
                fr = api.Frame(dev, W, H); fr.set_shard(r, world); model.render(fr, M)
                fr.submit_exchange()
            dev.exchange_drain()
            torch.cuda.synchronize()
        assert state["counts"] == {nbytes}
        got = final.cpu().numpy().reshape(H, W, 4)
        assert (got == ref[0]).all(), "the gathered frame must hold every triangle"
        dev.synchronize()  # nothing latched: the exchange thread re-ran the frames itself
        dev.exchange_stop()
        model.close()
