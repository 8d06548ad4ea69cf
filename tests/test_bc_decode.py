"""CPU: the oracle's BC1 / BC7 decoders.  BC7 is bit-exact by specification, so it is cross-checked
against an independent decoder (Pillow's DDS reader) on random blocks of every mode; BC1 has
implementation-defined interpolation rounding (SPEC.md "BC1"), so Pillow must agree within 1 LSB and the
chosen rule is pinned by hand-computed vectors."""
import numpy as np
import pytest

from mt_renderer_amd import scene
from oracle import oracle as orc
from tools.bc7_probe_pillow import dds_bc, decode_bc7_blocks as pillow_bc7

PIL = pytest.importorskip("PIL")


def _random_bc7(n, mode, seed):
    r = scene.splitmix64(np.uint64(seed) + np.arange(2 * n, dtype=np.uint64))
    b = r.view(np.uint8).reshape(n, 16).copy()
    low = (1 << (mode + 1)) - 1
    b[:, 0] = (b[:, 0] & ~np.uint8(low)) | np.uint8(1 << mode)
    return b


@pytest.mark.parametrize("mode", range(8))
def test_bc7_matches_independent_decoder(mode):
    blocks = _random_bc7(512, mode, 1000 + mode)
    assert (orc.decode_bc7_blocks(blocks) == pillow_bc7(blocks)).all()


def test_bc7_reserved_mode_is_transparent_black():
    blk = np.zeros((1, 16), dtype=np.uint8)
    blk[0, 1:] = 0xFF
    assert (orc.decode_bc7_blocks(blk) == 0).all()


def test_bc7_tables_match_spec_masks():
    """Independent recollection of the 64 two-subset partition masks (bit i = pixel i in subset 1) against
    the generated table: protects oracle/bc7_tables.h against a regression of the generator."""
    masks = [0xCCCC, 0x8888, 0xEEEE, 0xECC8, 0xC880, 0xFEEC, 0xFEC8, 0xEC80, 0xC800, 0xFFEC, 0xFE80, 0xE800, 0xFFE8,
             0xFF00, 0xFFF0, 0xF000, 0xF710, 0x008E, 0x7100, 0x08CE, 0x008C, 0x7310, 0x3100, 0x8CCE, 0x088C, 0x3110,
             0x6666, 0x366C, 0x17E8, 0x0FF0, 0x718E, 0x399C, 0xAAAA, 0xF0F0, 0x5A5A, 0x33CC, 0x3C3C, 0x55AA, 0x9696,
             0xA55A, 0x73CE, 0x13C8, 0x324C, 0x3BDC, 0x6996, 0xC33C, 0x9966, 0x0660, 0x0272, 0x04E4, 0x4E40, 0x2720,
             0xC936, 0x936C, 0x39C6, 0x639C, 0x9336, 0x9CC6, 0x817E, 0xE718, 0xCCF0, 0x0FCC, 0x7744, 0xEE22]
    src = open(__file__.replace("tests/test_bc_decode.py", "oracle/bc7_tables.h")).read()
    body = src[src.index("BC7_PART2[64][16] = {") + 21:]
    vals = [int(x) for x in body[:body.index("}")].split(",")]
    assert len(vals) == 1024
    for p in range(64):
        m = sum(vals[p * 16 + i] << i for i in range(16))
        assert m == masks[p], p


def test_bc1_hand_vectors():
    def blk(c0, c1, idx):
        return np.frombuffer(int(c0).to_bytes(2, "little") + int(c1).to_bytes(2, "little") + int(idx).to_bytes(4, "little"),
                             dtype=np.uint8)[None]
    # c0 > c1: four-colour mode; 0xF800 = pure red (255,0,0), 0x001F = pure blue (0,0,255)
    out = orc.decode_bc1_blocks(blk(0xF800, 0x001F, 0b11100100))[0]
    assert tuple(out[0]) == (255, 0, 0, 255) and tuple(out[1]) == (0, 0, 255, 255)
    assert tuple(out[2]) == ((2 * 255 + 0 + 1) // 3, 0, (0 + 255 + 1) // 3, 255)      # (2*c0 + c1 + 1)/3
    assert tuple(out[3]) == ((255 + 0 + 1) // 3, 0, (0 + 2 * 255 + 1) // 3, 255)      # (c0 + 2*c1 + 1)/3
    # c0 <= c1: three colours + transparent black
    out = orc.decode_bc1_blocks(blk(0x001F, 0xF800, 0b11100100))[0]
    assert tuple(out[2]) == (128, 0, 128, 255) and tuple(out[3]) == (0, 0, 0, 0)
    # 565 -> 888 by bit replication: 0x8410 = (16,32,16) -> (132,130,132)
    out = orc.decode_bc1_blocks(blk(0x8410, 0x8410, 0))[0]
    assert tuple(out[0]) == (132, 130, 132, 255)


def test_bc1_close_to_independent_decoder():
    import io
    from PIL import Image
    n = 256
    blocks = scene.splitmix64(np.uint64(77) + np.arange(n, dtype=np.uint64)).view(np.uint8).reshape(n, 8)
    img = Image.open(io.BytesIO(dds_bc(blocks.tobytes(), 4 * n, 4, 71)))  # DXGI_FORMAT_BC1_UNORM
    ref = np.asarray(img.convert("RGBA")).reshape(4, n, 4, 4).transpose(1, 0, 2, 3).reshape(n, 16, 4)
    mine = orc.decode_bc1_blocks(blocks)
    assert np.abs(mine.astype(int) - ref.astype(int)).max() <= 1
    assert (mine[..., 3] == ref[..., 3]).all()


def test_texture_decode_layout_and_errors():
    t = scene.random_bc7_texture(12, 8, seed=5)
    full = orc.decode_texture(t.fmt, 12, 8, t.data)
    blocks = np.frombuffer(t.data, dtype=np.uint8).reshape(-1, 16)
    dec = orc.decode_bc7_blocks(blocks).reshape(2, 3, 4, 4, 4)  # by, bx, y, x, c
    assert (full == dec.transpose(0, 2, 1, 3, 4).reshape(8, 12, 4)).all()
    # non-multiple-of-4 sizes keep the top-left part of the edge blocks
    part = orc.decode_texture(t.fmt, 10, 7, t.data)
    assert (part == full[:7, :10]).all()
    with pytest.raises(orc.OracleError):
        orc.decode_texture(t.fmt, 12, 8, t.data[:-1])
    assert (orc.decode_texture(scene.TEX_BC7_ALT, 12, 8, t.data) == full).all()  # ids 42 and 54, src/rtexture.rs:156-158
