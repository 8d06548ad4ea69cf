"""Committed golden vectors (tests/golden, made by tools/make_goldens.py): the oracle must reproduce
them on CPU, the HIP path must reproduce them on the GPU -- including by hash at BASELINE.json's full
sizes (C2 and the 1M-triangle headline scene)."""
import hashlib
import json
import os

import numpy as np
import pytest

from mt_renderer_amd import scene
from tests.golden_scenes import SCENES
from tests.helpers import render_gpu, render_oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
INDEX = json.load(open(os.path.join(GOLDEN, "index.json")))


def _check(name, col, dep):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert (col == g["color"]).all(), name
    assert (dep.view(np.uint32) == g["depth"].view(np.uint32)).all(), name
    assert hashlib.sha256(col.tobytes()).hexdigest() == INDEX[name]["color_sha256"]


@pytest.mark.parametrize("name", sorted(SCENES))
def test_oracle_reproduces_goldens(name):
    w, h, draws = SCENES[name]()
    col, dep, st = render_oracle(w, h, draws)
    _check(name, col, dep)
    assert st == INDEX[name]["stats"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SCENES))
def test_gpu_reproduces_goldens(gpu_device, name):
    w, h, draws = SCENES[name]()
    col, dep, st = render_gpu(gpu_device, w, h, draws)
    _check(name, col, dep)
    assert st["tris_setup"] == INDEX[name]["stats"]["tris_setup"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2_mesh50k_1080p", "headline_1m_1080p"])
def test_gpu_full_size_hashes(gpu_device, name):
    md = scene.mesh50k() if name.startswith("c2") else scene.headline_model()
    M = scene.to_f32_colmajor(scene.headline_transform(1920, 1080))
    col, dep, st = render_gpu(gpu_device, 1920, 1080, [dict(md=md, M=M, palette=scene.bone_palette())])
    assert hashlib.sha256(col.tobytes()).hexdigest() == INDEX[name]["color_sha256"]
    assert hashlib.sha256(dep.tobytes()).hexdigest() == INDEX[name]["depth_sha256"]
    assert st["tris_in"] == INDEX[name]["stats"]["tris_in"] and st["tris_setup"] == INDEX[name]["stats"]["tris_setup"]
