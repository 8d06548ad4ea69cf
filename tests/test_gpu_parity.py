"""-m gpu: the HIP draw path (through the C ABI) against the CPU oracle, bit for bit."""
import numpy as np
import pytest

from mt_renderer_amd import scene
from tests.helpers import assert_same, render_gpu, render_oracle

pytestmark = pytest.mark.gpu


def test_cube_c1(gpu_device):
    """BASELINE config C1: the reference cube (src/debug_overlay.rs:10-35), 256x256."""
    md = scene.cube_model(3)
    M = scene.to_f32_colmajor(scene.cube_transform(256, 256))
    draws = [dict(md=md, M=M)]
    assert_same(render_gpu(gpu_device, 256, 256, draws), render_oracle(256, 256, draws), "cube")


@pytest.mark.parametrize("size", [(64, 48), (333, 171), (640, 360)])
def test_small_skinned_mesh(gpu_device, size):
    w, h = size
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=20, cols=31)
    M = scene.to_f32_colmajor(scene.headline_transform(w, h))
    draws = [dict(md=md, M=M, palette=scene.bone_palette())]
    assert_same(render_gpu(gpu_device, w, h, draws), render_oracle(w, h, draws), f"mesh {size}")


def test_mesh50k_c2(gpu_device):
    """BASELINE config C2: one mesh50k, 64 bones, 1920x1080."""
    md = scene.mesh50k()
    M = scene.to_f32_colmajor(scene.headline_transform(1920, 1080))
    draws = [dict(md=md, M=M, palette=scene.bone_palette())]
    assert_same(render_gpu(gpu_device, 1920, 1080, draws), render_oracle(1920, 1080, draws), "C2")


def test_headline_1m(gpu_device):
    """The headline scene: 1 000 000 triangles, 64 bones, 1920x1080."""
    md = scene.headline_model()
    M = scene.to_f32_colmajor(scene.headline_transform(1920, 1080))
    draws = [dict(md=md, M=M, palette=scene.bone_palette())]
    assert_same(render_gpu(gpu_device, 1920, 1080, draws), render_oracle(1920, 1080, draws, nthreads=8), "headline")
