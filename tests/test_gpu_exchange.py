"""The device's exchange thread (include/mtr.h: mtr_device_exchange_start / mtr_frame_submit_exchange): sharded frames
handed to a second host thread that packs, calls the host's all-gather, unpacks and destroys them.  The all-gather here
is a callback that copies the rank's block into its place of the gathered buffer (one GPU stands in for both ranks, one
after the other); the frame rebuilt from the two shards must equal the unsharded render bit for bit."""
import ctypes as C

import numpy as np
import pytest

W, H = 200, 120  # 13 x 8 bins, odd split over two ranks


def _scene():
    from mt_renderer_amd import scene
    md = scene.mesh50k()
    return md, scene.bone_palette(), scene.to_f32_colmajor(scene.headline_transform(W, H))


@pytest.mark.gpu
def test_exchange_thread_rebuilds_the_unsharded_frame():
    import torch
    from mt_renderer_amd import api
    md, pal, M = _scene()
    xs = torch.cuda.Stream()
    with api.Device(0) as dev:
        model = api.Model.new(dev, md)
        model.set_palette(pal)
        fr = api.Frame(dev, W, H); model.render(fr, M); fr.end(); ref = fr.color(); fr.close()
        world = 2
        nbytes = int(api.lib.mtr_shard_bytes(W, H, world))
        shard = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        gathered = torch.zeros(nbytes * world, dtype=torch.uint8, device="cuda")
        final = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
        state = {"rank": 0, "calls": 0, "fail": False, "per_lane": {}}

        @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p)
        def fake_allgather(send, recv, count, dtype, comm, stream):
            state["calls"] += 1
            if state["fail"]:
                return 7
            sh, ga, st, cm = lanes[send]
            assert recv == ga.data_ptr() and count == nbytes and dtype == 1 and comm == cm and stream == st.cuda_stream
            with torch.cuda.stream(st):
                r = state["rank"]
                ga[r * nbytes:(r + 1) * nbytes].copy_(sh, non_blocking=True)
            state["per_lane"][cm] = state["per_lane"].get(cm, 0) + 1
            return 0

        lanes = {shard.data_ptr(): (shard, gathered, xs, 1234)}

        fr = api.Frame(dev, W, H); fr.set_shard(0, world); model.render(fr, M)
        with pytest.raises(api.MtrError):  # no exchange thread yet: the handle is not consumed
            fr.submit_exchange()
        assert fr._h
        fr.close()
        dev.exchange_start(C.cast(fake_allgather, C.c_void_p).value, 1234, 1, shard.data_ptr(), nbytes, gathered.data_ptr(),
                           final.data_ptr(), world, xs.cuda_stream)
        with pytest.raises(api.MtrError):
            dev.exchange_start(C.cast(fake_allgather, C.c_void_p).value, 1234, 1, shard.data_ptr(), nbytes, gathered.data_ptr(),
                               final.data_ptr(), world, xs.cuda_stream)
        nframes = 24  # more than the hand-over queue holds: the render thread blocks and resumes
        for r in range(world):
            state["rank"] = r
            for _ in range(nframes):
                fr = api.Frame(dev, W, H); fr.set_shard(r, world); model.render(fr, M)
                fr.submit_exchange()
                assert fr._h is None
            dev.exchange_drain()
            torch.cuda.synchronize()
        assert state["calls"] == world * nframes
        got = final.cpu().numpy().reshape(H, W, 4)
        assert (got == ref).all()
        # a second lane (its own buffers, stream and "communicator"): frames alternate between the lanes
        xs2 = torch.cuda.Stream()
        shard2, gathered2, final2 = torch.zeros_like(shard), torch.zeros_like(gathered), torch.zeros_like(final)
        lanes[shard2.data_ptr()] = (shard2, gathered2, xs2, 5678)
        with pytest.raises(api.MtrError):  # a lane needs a stream and buffers of its own
            dev.exchange_add_lane(5678, shard2.data_ptr(), gathered2.data_ptr(), final2.data_ptr(), xs.cuda_stream)
        dev.exchange_add_lane(5678, shard2.data_ptr(), gathered2.data_ptr(), final2.data_ptr(), xs2.cuda_stream)
        final.zero_()
        state["per_lane"] = {}
        for r in range(world):
            state["rank"] = r
            for _ in range(nframes):
                fr = api.Frame(dev, W, H); fr.set_shard(r, world); model.render(fr, M)
                fr.submit_exchange()
            dev.exchange_drain()
            torch.cuda.synchronize()
        assert state["per_lane"] == {1234: nframes, 5678: nframes}
        assert (final.cpu().numpy().reshape(H, W, 4) == ref).all()
        assert (final2.cpu().numpy().reshape(H, W, 4) == ref).all()
        # a frame of another world size is refused before it reaches the thread
        fr = api.Frame(dev, W, H); fr.set_shard(0, 3); model.render(fr, M)
        with pytest.raises(api.MtrError):
            fr.submit_exchange()
        fr.close()
        # an all-gather that fails: reported by drain, once; the thread keeps consuming (and destroying) frames
        state["fail"] = True
        fr = api.Frame(dev, W, H); fr.set_shard(0, world); model.render(fr, M); fr.submit_exchange()
        with pytest.raises(api.MtrError, match="all-gather callback returned 7"):
            dev.exchange_drain()
        state["fail"] = False
        dev.exchange_drain()
        fr = api.Frame(dev, W, H); fr.set_shard(1, world); model.render(fr, M); fr.submit_exchange()
        dev.exchange_stop()
        torch.cuda.synchronize()
        dev.exchange_stop()  # idempotent
        model.close()
