"""BC textures kept block-compressed in HBM and decoded per fetch (mtr_device_set_texture_residency(MTR_TEXRES_BLOCKS),
csrc/bc_sample.h): the same pixels as the default decode-at-upload path and as the oracle, for every BC7 mode /
partition / rotation a few thousand random blocks reach, BC1 with and without its punch-through alpha, mip chains,
magnified (linear, four texels from up to four blocks) and minified (nearest) samples, both tile kernels."""
import numpy as np
import pytest

from mt_renderer_amd import scene
from tests.helpers import assert_same, render_gpu, render_oracle
from tests.pixel_scenes import pixel_model, pixel_to_ndc_matrix

pytestmark = pytest.mark.gpu


def _both_residencies(dev, w, h, draws, label):
    from mt_renderer_amd import api
    ref = render_oracle(w, h, draws)
    try:
        dev.set_texture_residency(api.TEXRES_BLOCKS)
        blocks = render_gpu(dev, w, h, draws)
    finally:
        dev.set_texture_residency(api.TEXRES_DECODED)
    decoded = render_gpu(dev, w, h, draws)
    assert_same(blocks, ref, label + " (blocks resident)")
    assert_same(decoded, ref, label + " (decoded at upload)")


def _texel_quad(tex, w, h, z=0.5, uv=(0.0, 0.0, 1.0, 1.0)):
    u0, v0, u1, v1 = uv
    verts = [(0, 0, z, u0, v0), (0, h, z, u0, v1), (w, 0, z, u1, v0), (w, h, z, u1, v1)]
    return pixel_model([dict(verts=verts, indices=[0, 1, 2, 3], topology=scene.TOPO_STRIP, texture=0)], textures=[tex])


@pytest.mark.parametrize("kind", ["bc7", "bc7_opaque", "bc1"])
def test_every_texel_one_to_one(gpu_device, kind):
    """texture w x h on a w x h pixel quad: every texel of every block is fetched once"""
    w, h = 208, 132  # 52 x 33 blocks = 1716 random blocks
    tex = (scene.random_bc1_texture(w, h, seed=11) if kind == "bc1"
           else scene.random_bc7_texture(w, h, seed=12, opaque_modes_only=kind == "bc7_opaque"))
    _both_residencies(gpu_device, w, h, [dict(md=_texel_quad(tex, w, h), M=pixel_to_ndc_matrix(w, h))], kind + " 1:1")


@pytest.mark.parametrize("kind", ["bc7", "bc1"])
def test_magnified_and_ragged_sizes(gpu_device, kind):
    """texture sizes that are not multiples of 4 (partial edge blocks), magnified 3.3x: bilinear footprints straddle blocks"""
    tw, th = 37, 22
    tex = scene.random_bc1_texture(tw, th, seed=5) if kind == "bc1" else scene.random_bc7_texture(tw, th, seed=6)
    w, h = 122, 73
    draws = [dict(md=_texel_quad(tex, w, h, uv=(-0.1, -0.05, 1.1, 1.07)), M=pixel_to_ndc_matrix(w, h))]  # clamp-to-edge on all sides
    _both_residencies(gpu_device, w, h, draws, kind + " magnified")


@pytest.mark.parametrize("kind", ["bc7", "bc1"])
def test_mip_chain_from_blocks(gpu_device, kind):
    """minified skinned meshes: nearest texel of the selected level, level offsets counted in blocks"""
    rng_seed = 3
    tw, th, levels = 256, 128, 8
    data, lw, lh = b"", tw, th
    for l in range(levels):
        t = (scene.random_bc1_texture(lw, lh, seed=rng_seed * 31 + l) if kind == "bc1"
             else scene.random_bc7_texture(lw, lh, seed=rng_seed * 31 + l, opaque_modes_only=True))
        data += t.data
        lw, lh = max(1, lw >> 1), max(1, lh >> 1)
    tex = scene.TextureData(tw, th, scene.TEX_BC1 if kind == "bc1" else scene.TEX_BC7, data, levels=levels)
    w, h = 320, 200
    md = scene.mesh50k(textured=True, textures=[tex], rows=30, cols=48)
    draws = []
    for dist, sx in ((2.3, 1.6), (6.0, 1.6), (12.0, 0.4)):
        M = scene.to_f32_colmajor(scene.reference_view_proj(w, h) @ scene.mat_translate(-5.0, 0.0, 1.0 - dist) @ scene.mat_scale(sx, 0.9, 1.0))
        draws.append(dict(md=md, M=M, palette=scene.bone_palette()))
    _both_residencies(gpu_device, w, h, draws, kind + " mips")


def test_instanced_bc7_translucent_and_opaque(gpu_device):
    """C5's shape at a size the oracle finishes in seconds, blocks resident"""
    w, h = 960, 540
    for opaque in (True, False):
        texs = [scene.random_bc7_texture(128, 128, seed=300 + i, opaque_modes_only=opaque) for i in range(4)]
        md = scene.mesh50k(textured=True, textures=texs, rows=25, cols=40)
        mats, pals = scene.instance_lattice(4, 4)
        vp = scene.to_f32_colmajor(scene.reference_view_proj(w, h))
        draws = [dict(md=md, vp=vp, model_mats=mats, palettes=pals, tex_override=[i // 4 for i in range(16)])]
        _both_residencies(gpu_device, w, h, draws, f"instanced opaque={opaque}")


def test_blocks_resident_textures_cannot_be_read_back(gpu_device):
    from mt_renderer_amd import api
    tex = scene.random_bc7_texture(16, 16, seed=1)
    try:
        gpu_device.set_texture_residency(api.TEXRES_BLOCKS)
        t = api.Texture.new(gpu_device, tex)
    finally:
        gpu_device.set_texture_residency(api.TEXRES_DECODED)
    with pytest.raises(api.MtrError, match="resident as BC blocks"):
        t.read_rgba8()
    t.close()
    with pytest.raises(api.MtrError):
        gpu_device.set_texture_residency(7)
