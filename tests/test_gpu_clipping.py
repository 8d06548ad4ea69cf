"""-m gpu: guard-band clipping (SPEC.md 5.3): triangles with a vertex the rasteriser cannot project
(w -> 0+, w = 0, beyond +-2^20 px) are clipped against |x| <= 64 w, |y| <= 64 w instead of being dropped; HIP path vs
oracle, bit for bit."""
import numpy as np
import pytest

from mt_renderer_amd import scene
from tests.helpers import assert_same, render_gpu, render_oracle
from tests.pixel_scenes import pixel_model

pytestmark = pytest.mark.gpu

# clip = (x, y, 0.25 * pz, pz): the third position component IS w, depth is the constant 0.25
M_W = scene.to_f32_colmajor(np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0.25, 0], [0, 0, 1, 0]], dtype=np.float64))


def _tris(tris, did=3, topology=scene.TOPO_LIST):
    verts = [v for t in tris for v in t]
    return pixel_model([dict(verts=verts, indices=list(range(len(verts))), topology=topology, debug_id=did)])


@pytest.mark.parametrize("wc", [1e-6, 1e-12, 0.0, 3e-3])
def test_a_vertex_at_w_to_zero_keeps_the_visible_part(gpu_device, wc):
    # A, B on screen at w = 1; C at w -> 0+: x/w = 3e5 .. 3e11 NDC, far outside the +-2^20 px band (or no projection at all)
    md = _tris([[(-0.5, -0.5, 1.0), (0.3, 0.4, wc), (0.5, -0.5, 1.0)], [(-0.5, -0.5, 1.0), (0.5, -0.5, 1.0), (0.3, 0.4, wc)]])
    draws = [dict(md=md, M=M_W)]
    ref = render_oracle(128, 96, draws)
    assert (ref[0][..., :3] != 255).any(), "the on-screen part must be drawn"
    assert_same(render_gpu(gpu_device, 128, 96, draws), ref, f"w -> {wc}")


def test_vertices_far_outside_the_band_and_behind_the_eye(gpu_device):
    rng = np.random.default_rng(5)
    tris = []
    for _ in range(40):
        t = []
        for k in range(3):
            w = float(rng.choice([1.0, 0.5, 1e-5, 2e-7, -0.5, 0.0, 3.0]))
            t.append((float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-1.5, 1.5)), w))
        tris.append(t)
    # depth = 0.25 w: a vertex behind the eye (w < 0) has z < 0 and goes through the near clip first
    draws = [dict(md=_tris(tris, did=7), M=M_W)]
    ref = render_oracle(160, 120, draws)
    assert ref[2]["tris_setup"] > 10
    assert_same(render_gpu(gpu_device, 160, 120, draws), ref, "random huge triangles")


def test_a_floor_under_a_perspective_camera(gpu_device):
    """the practical case: a large ground quad whose near-clipped vertices project hundreds of thousands of pixels away
    (dropped before this round), plus a skinned mesh standing on it, sharded three ways as well"""
    from mt_renderer_amd import api, sharding
    w, h = 320, 200
    vp = scene.reference_view_proj(w, h)  # camera at (-5, 0, 1) looking down -z, near 0.01
    floor = pixel_model([dict(verts=[(-60.0, -0.8, 40.0), (50.0, -0.8, 40.0), (50.0, -0.8, -60.0), (-60.0, -0.8, -60.0)],
                              indices=[0, 1, 2, 0, 2, 3], debug_id=9),
                         dict(verts=[(-60.0, 2.0, 40.0), (50.0, 2.0, 40.0), (50.0, 2.0, -60.0), (-60.0, 2.0, -60.0)],
                              indices=[0, 2, 1, 0, 3, 2], debug_id=12)])
    mesh = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=12, cols=20)
    draws = [dict(md=floor, M=scene.to_f32_colmajor(vp)),
             dict(md=mesh, M=scene.to_f32_colmajor(scene.headline_transform(w, h)), palette=scene.bone_palette())]
    ref = render_oracle(w, h, draws)
    covered = (ref[0][..., :3] != 255).any(axis=-1).mean()
    assert covered > 0.5, covered  # floor + ceiling fill most of the frame
    full = render_gpu(gpu_device, w, h, draws)
    assert_same(full, ref, "floor")
    owner = sharding.owner_map(w, h, 3, sharding.BANDS)
    for rank in range(3):
        part = render_gpu(gpu_device, w, h, draws, shard=(rank, 3, sharding.BANDS), tile_mode=api.TILE_AUTO)
        own = owner == rank
        assert (part[0][own] == full[0][own]).all() and (part[1].view(np.uint32)[own] == full[1].view(np.uint32)[own]).all(), rank


def test_heavily_clipped_strips_are_right_or_fail_loudly(gpu_device):
    """62 strip positions own 124 record slots.  Strips in which every triangle is guard-clipped into a fan stress that
    bound (it takes more than two fan triangles per position, over a whole chunk, to exceed it: the hardest scene found by
    search reserves 132): each scene must either equal the oracle bit for bit or fail with MTR_E_OVERFLOW -- never a frame
    with triangles missing"""
    from mt_renderer_amd import api
    outcomes = []
    for seed in (167, 3, 41, 90, 125):
        rng = np.random.default_rng(seed)
        verts = []
        for i in range(40):
            a = rng.uniform(0, 2 * np.pi)
            verts += [(0.5 * np.cos(0.9 * i), 0.5 * np.sin(0.9 * i), 1.0), (float(np.cos(a)), float(np.sin(a)), 1e-3 if seed != 125 else 1e-9)]
        if seed == 125:  # a strip around a circle 2.5 million pixels out: every triangle holds most of the guard-band square
            verts = [(40000.0 * np.cos(np.radians(125.0 * i)), 40000.0 * np.sin(np.radians(125.0 * i)), 1.0) for i in range(80)]
        md = pixel_model([dict(verts=verts, indices=list(range(len(verts))), topology=scene.TOPO_STRIP, debug_id=2)])
        md.prim_states = np.array([(0, 1, 1, 1)], dtype=np.uint8)  # two-sided: nothing is culled
        draws = [dict(md=md, M=M_W)]
        ref = render_oracle(128, 96, draws)
        assert ref[2]["tris_setup"] > 60
        try:
            assert_same(render_gpu(gpu_device, 128, 96, draws), ref, f"heavy clipping {seed}")
            outcomes.append("exact")
        except api.MtrError as e:
            assert e.code == api.MTR_E_OVERFLOW and "124 records" in str(e), e
            outcomes.append("overflow")
    assert "exact" in outcomes
    # the device is fine afterwards
    ok = [dict(md=_tris([[(-0.5, -0.5, 1.0), (0.3, 0.4, 1.0), (0.5, -0.5, 1.0)]]), M=M_W)]
    assert_same(render_gpu(gpu_device, 128, 96, ok), render_oracle(128, 96, ok), "after the heavy scenes")
