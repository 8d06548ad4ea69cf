"""The small scenes whose oracle output is committed under tests/golden/*.npz (tools/make_goldens.py)."""
import numpy as np

from mt_renderer_amd import scene
from tests.pixel_scenes import pixel_model, pixel_to_ndc_matrix


def _cube():
    w = h = 64
    return w, h, [dict(md=scene.cube_model(3), M=scene.to_f32_colmajor(scene.cube_transform(w, h)))]


def _skinned():
    w, h = 96, 64
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=12, cols=16)
    return w, h, [dict(md=md, M=scene.to_f32_colmajor(scene.headline_transform(w, h)), palette=scene.bone_palette())]


def _textured():
    w, h = 64, 64
    texs = [scene.random_bc7_texture(16, 16, 3), scene.random_bc1_texture(8, 8, 4),
            scene.checker_rgba8_texture(4, 4, cell=1, alpha=(90, 200))]

    def quad(x0, y0, x1, y1, z, tex):
        v = [(x0, y0, z, 0, 0), (x0, y1, z, 0, 1), (x1, y1, z, 1, 1), (x1, y0, z, 1, 0)]
        return dict(verts=v, indices=[0, 1, 2, 0, 2, 3], texture=tex)
    md = pixel_model([quad(2, 2, 40, 40, .6, 0), quad(20, 10, 62, 50, .5, 2), quad(30, 30, 38, 38, .4, 1),
                      quad(0, 44, 64, 64, .3, 0)], texs)
    return w, h, [dict(md=md, M=pixel_to_ndc_matrix(w, h))]


def _near_clip():
    w, h = 80, 48
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=10, cols=14)
    M = scene.reference_view_proj(w, h) @ scene.mat_translate(-5.0, 0.05, 1.0 - 0.2) @ scene.mat_rot_x(0.7)
    return w, h, [dict(md=md, M=scene.to_f32_colmajor(M), palette=scene.bone_palette())]


def _instances_overlay():
    w, h = 96, 54
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=8, cols=10)
    mats, pals = scene.instance_lattice(3, 2)
    vp = scene.to_f32_colmajor(scene.reference_view_proj(w, h))
    inst = np.stack([scene.to_f32_colmajor(scene.mat_translate(-5 + 0.5 * i - 0.5, 0.1 * i, -1.0) @ scene.mat_scale(.15, .15, .15))
                     for i in range(3)])
    return w, h, [dict(md=md, vp=vp, model_mats=mats, palettes=pals), dict(md=md, vp=vp, overlay=inst)]


SCENES = {"cube_64": _cube, "skinned_96x64": _skinned, "textured_64": _textured, "near_clip_80x48": _near_clip,
          "instances_overlay_96x54": _instances_overlay}
