"""-m gpu: BASELINE configs C4 and C5 at FULL size through the multi-GPU path that ships -- bands of bin rows (equal and
balanced by a calibration frame, as bench.py cuts them), per-rank culling (k_cull_instances, k_cull_chunks), the culled
geometry launch k_geom<., true> and its remainder kernel k_geom_rest, single-pass binning, the visibility kernel, pack ->
gather -> unpack.  One GPU stands in for every rank, one after the other.

C4 = 128 instanced mesh50k (6.4 M triangles) at 3840x2160, "sharded across 2/4/8 GPUs" (BASELINE.json): EVERY rank at
N = 2, 4, 8.  C5 = 1024 instances (51.2 M triangles) + 64 BC7 1024x1024 textures at 3840x2160 on 8 GPUs: ranks 0, 3, 7 of
8, plus one run with MTR_GEOM_SLOTS forcing the kept instances through k_geom_rest.  Full size is beyond what the scalar
oracle renders in seconds, so the checks are exact-equality properties: a rank's own pixels are those of the unsharded
frame (which tests/test_gpu_parity.py pins to the oracle at C4 size and at 64 instances of C5), culling drops no
triangle that reaches the rank (same set-up and queue counts with culling off), and the frame rebuilt from the packed
shards is the unsharded frame.  C5 at 64 instances is compared with the ORACLE itself under 8-way bands."""
import numpy as np
import pytest

from mt_renderer_amd import scene, sharding
from tests.helpers import render_oracle

pytestmark = pytest.mark.gpu

W4K, H4K = 3840, 2160


class Resident:
    """a batch scene kept in HBM across many frames (the 64 textures of C5 are uploaded once)"""

    def __init__(self, dev, md, nx, ny, tex_override=None):
        from mt_renderer_amd import api
        self.dev = dev
        self.mats, self.pals = scene.instance_lattice(nx, ny)
        self.tex_override = tex_override
        self.md = md
        self.model = api.Model.new(dev, md)
        self.batch = api.Batch(dev, self.model, self.mats, self.pals, tex_override)
        self.vp = scene.to_f32_colmajor(scene.reference_view_proj(W4K, H4K))

    def frame(self, shard=None):
        from mt_renderer_amd import api
        fr = api.Frame(self.dev, W4K, H4K)
        if shard:
            fr.set_shard(*shard)
        fr.draw_batch(self.batch, self.vp)
        fr.end()
        return fr

    def render(self, shard=None):
        fr = self.frame(shard)
        out = fr.color(), fr.depth().view(np.uint32), fr.stats()
        fr.close()
        return out

    def balanced_bands(self, world):
        fr = self.frame()
        entries, _ = fr.bin_counts()
        fr.close()
        nbx, nby, _ = sharding.grid(W4K, H4K)
        return sharding.balanced_bands(entries.reshape(nby, nbx).sum(axis=1).astype(np.float64) + 8.0 * nbx, world)  # bench.py's cut

    def close(self):
        self.batch.close()
        self.model.close()


def _check_rank(scn, full, bands, rank, world, culled_expected=True):
    """own pixels == unsharded frame; culling on / off: same pixels, same set-up and queue counts"""
    owner_rows = np.repeat(np.searchsorted(np.asarray(bands[1:]), np.arange((H4K + 15) // 16), side="right"), 16)[:H4K]
    own = owner_rows == rank  # bands: ownership is a property of the pixel row
    shard = (rank, world, sharding.BANDS, 0, bands)
    res = {}
    for cull in (True, False):
        scn.dev.set_culling(cull)
        try:
            res[cull] = scn.render(shard)
        finally:
            scn.dev.set_culling(True)
        c, d, st = res[cull]
        assert (c[own] == full[0][own]).all() and (d[own] == full[1][own]).all(), (world, rank, cull)
        assert st["binning"] == 1 and st["shard_map"] == sharding.BANDS
    for key in ("tris_setup", "bin_entries"):
        assert res[True][2][key] == res[False][2][key], (world, rank, key, "culling dropped (or added) a triangle of the rank")
    assert res[False][2]["chunks_culled"] == 0
    if culled_expected and world >= 4:
        assert res[True][2]["chunks_culled"] > res[True][2]["chunks"] // 2, (world, rank, res[True][2])
    return res[True]


def _gather_and_compare(scn, full, bands, world):
    """pack every rank's bins, concatenate as an all-gather would, unpack: the unsharded frame"""
    import torch
    from mt_renderer_amd import api
    nbytes = api.shard_bytes_map(W4K, H4K, world, sharding.BANDS, 0, bands)
    shards, last = [], None
    for rank in range(world):
        fr = scn.frame((rank, world, sharding.BANDS, 0, bands))
        assert fr.shard_bytes() == nbytes
        buf = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        fr.pack_color_shard(buf.data_ptr(), nbytes)
        scn.dev.synchronize()
        shards.append(buf)
        if last is not None:
            last.close()
        last = fr
    gathered = torch.cat(shards)
    out = torch.zeros(W4K * H4K * 4, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    last.unpack_color_shards(gathered.data_ptr(), out.data_ptr())
    scn.dev.synchronize()
    torch.cuda.synchronize()
    last.close()
    assert (out.cpu().numpy().reshape(H4K, W4K, 4) == full[0]).all(), world


def test_c4_full_size_every_rank_bands_and_culling(gpu_device):
    scn = Resident(gpu_device, scene.mesh50k(), 16, 8)
    try:
        full = scn.render()
        assert full[2]["tris_in"] == 128 * 50000 and full[2]["tile_kernel"] == 2
        for world in (2, 4, 8):
            for kind in ("equal", "balanced"):
                bands = sharding.equal_bands(H4K, world) if kind == "equal" else scn.balanced_bands(world)
                assert len(bands) == world + 1 and bands[0] == 0 and bands[-1] == (H4K + 15) // 16
                setup = 0
                for rank in range(world):
                    setup += _check_rank(scn, full, bands, rank, world)[2]["tris_setup"]
                assert setup >= full[2]["tris_setup"]  # every triangle is set up by at least one rank (border ones by two)
                _gather_and_compare(scn, full, bands, world)
    finally:
        scn.close()


def _c5_scene(dev, nx, ny, tex_size=1024, opaque=True):
    n = nx * ny
    texs = [scene.random_bc7_texture(tex_size, tex_size, seed=200 + i, opaque_modes_only=opaque) for i in range(max(1, n // 16))]
    md = scene.mesh50k(textured=True, textures=texs)
    return Resident(dev, md, nx, ny, tex_override=[i // 16 for i in range(n)])


def test_c5_full_size_ranks_of_8_bands_and_culling(gpu_device):
    scn = _c5_scene(gpu_device, 32, 32)
    try:
        full = scn.render()
        assert full[2]["tris_in"] == 1024 * 50000 and full[2]["tile_kernel"] == 2
        assert full[0][..., :3].reshape(-1, 3).std(axis=0).min() > 1.0  # an image, not a constant
        for kind in ("equal", "balanced"):
            bands = sharding.equal_bands(H4K, 8) if kind == "equal" else scn.balanced_bands(8)
            for rank in (0, 3, 7):
                part = _check_rank(scn, full, bands, rank, 8)
                # a rank of 8 keeps about an eighth of the 832 512 chunks (plus the straddlers of its borders)
                kept = part[2]["chunks"] - part[2]["chunks_culled"]
                assert part[2]["chunks"] == 1024 * 813 and kept < part[2]["chunks"] // 4, (kind, rank, kept)
        _gather_and_compare(scn, full, scn.balanced_bands(8), 8)
    finally:
        scn.close()


def test_c5_full_size_remainder_kernel(monkeypatch):
    """MTR_GEOM_SLOTS (read at device creation) bounds the instance slots of the full-rate culled launch: with 5, nearly
    every instance a rank of 8 keeps (~130-180 of 1024) goes through k_geom_rest.  Same pixels, same counts."""
    from mt_renderer_amd import api
    dev0 = api.Device(0)
    monkeypatch.setenv("MTR_GEOM_SLOTS", "5")
    dev1 = api.Device(0)
    monkeypatch.delenv("MTR_GEOM_SLOTS")
    a = b = None
    try:
        a, b = _c5_scene(dev0, 32, 32), _c5_scene(dev1, 32, 32)
        # the first frame of this scene on a fresh device overflows the default per-bin queue bound: it is re-run through
        # the exact two-pass queues and the bound doubles for the frames that follow (tests/test_gpu_overflow.py)
        for s in (a, b):
            for _ in range(3):
                first = s.render()
        assert first[2]["binning"] == 1
        full = a.render()
        bands = a.balanced_bands(8)
        for rank in (3, 6):
            ref = a.render((rank, 8, sharding.BANDS, 0, bands))
            got = _check_rank(b, full, bands, rank, 8)
            assert (got[0] == ref[0]).all() and (got[1] == ref[1]).all()
            for key in ("tris_setup", "bin_entries", "chunks_culled"):
                assert got[2][key] == ref[2][key], (rank, key)
    finally:
        for s in (a, b):
            if s is not None:
                s.close()
        dev1.close()
        dev0.close()


@pytest.mark.parametrize("opaque", [True, False], ids=["opaque", "translucent"])
def test_c5_64_instances_vs_oracle_under_8_way_bands(gpu_device, opaque):
    """C5 at the size the oracle renders in seconds (64 instances, 4 BC7 textures, 3840x2160): the frame assembled from
    the own pixels of the 8 ranks (balanced bands, culling on) is the ORACLE's frame, bit for bit.  Translucent set: the
    ranks resolve alpha blending through the visibility kernel's order lists (flagged bins: the ordered kernel)."""
    scn = _c5_scene(gpu_device, 8, 8, tex_size=256, opaque=opaque)
    try:
        draws = [dict(md=scn.md, vp=scn.vp, model_mats=scn.mats, palettes=scn.pals, tex_override=scn.tex_override)]
        oc, od, ost = render_oracle(W4K, H4K, draws, nthreads=16)
        bands = scn.balanced_bands(8)
        color = np.zeros_like(oc)
        depth = np.zeros((H4K, W4K), dtype=np.uint32)
        rows = np.repeat(np.searchsorted(np.asarray(bands[1:]), np.arange((H4K + 15) // 16), side="right"), 16)[:H4K]
        for rank in range(8):
            c, d, st = scn.render((rank, 8, sharding.BANDS, 0, bands))
            own = rows == rank
            color[own], depth[own] = c[own], d[own]
            assert st["tris_in"] == ost["tris_in"]
        assert (color == oc).all() and (depth == od.view(np.uint32)).all()
    finally:
        scn.close()
