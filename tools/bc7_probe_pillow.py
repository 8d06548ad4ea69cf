"""Derive the BC7 partition / anchor tables by probing an INDEPENDENT decoder (Pillow's DDS
reader), and emit them as a C header fragment.  Pillow is not part of the reference; it is used
here only as an instrument so the tables committed in oracle/bc_tables.h and in the HIP decoder
are not typed from memory.  Run: python tools/bc7_probe_pillow.py > /tmp/bc7_tables.txt
"""
import io, struct, sys
import numpy as np
from PIL import Image


def dds_bc(blocks: bytes, w: int, h: int, dxgi: int) -> bytes:
    pf = struct.pack("<II4sIIIII", 32, 0x4, b"DX10", 0, 0, 0, 0, 0)
    hdr = struct.pack("<IIIIIII44s", 124, 0x1 | 0x2 | 0x4 | 0x1000 | 0x80000, h, w, len(blocks), 0, 0, b"\0" * 44)
    hdr += pf + struct.pack("<IIIII", 0x1000, 0, 0, 0, 0)
    assert len(hdr) == 124
    dx10 = struct.pack("<IIIII", dxgi, 3, 0, 1, 0)
    return b"DDS " + hdr + dx10 + blocks


def decode_bc7_blocks(blocks: np.ndarray) -> np.ndarray:
    """blocks: (n,16) uint8 -> (n,16,4) uint8 RGBA, pixel index = y*4+x"""
    n = blocks.shape[0]
    img = Image.open(io.BytesIO(dds_bc(blocks.tobytes(), 4 * n, 4, 98)))
    a = np.asarray(img.convert("RGBA"))  # (4, 4n, 4)
    return a.reshape(4, n, 4, 4).transpose(1, 0, 2, 3).reshape(n, 16, 4)


class BitWriter:
    def __init__(self):
        self.v = 0
        self.n = 0

    def put(self, val, bits):
        self.v |= (val & ((1 << bits) - 1)) << self.n
        self.n += bits

    def bytes(self):
        assert self.n <= 128
        return np.frombuffer(self.v.to_bytes(16, "little"), dtype=np.uint8)


def block_mode1(partition, ep, idxbits=0):
    """mode 1: 2 subsets, 6-bit endpoints (r,g,b) x 4 endpoints, shared pbit per subset, 3-bit idx.
    ep: 4 endpoints (s0e0,s0e1,s1e0,s1e1) each (r,g,b) 6-bit."""
    bw = BitWriter()
    bw.put(1 << 1, 2)
    bw.put(partition, 6)
    for c in range(3):
        for e in range(4):
            bw.put(ep[e][c], 6)
    bw.put(0, 2)  # pbits
    bw.put(idxbits, 46)
    return bw.bytes()


def block_mode2(partition, ep, idxbits=0):
    """mode 2: 3 subsets, 5-bit endpoints x 6, no pbits, 2-bit idx (29 bits of indices)."""
    bw = BitWriter()
    bw.put(1 << 2, 3)
    bw.put(partition, 6)
    for c in range(3):
        for e in range(6):
            bw.put(ep[e][c], 5)
    bw.put(idxbits, 29)
    return bw.bytes()


def probe():
    # ---- partition maps: constant colour per subset, all indices zero
    p2 = np.zeros((64, 16), dtype=np.uint8)
    blocks = np.stack([block_mode1(p, [(0, 0, 0), (0, 0, 0), (63, 0, 0), (63, 0, 0)]) for p in range(64)])
    out = decode_bc7_blocks(blocks)
    p2[:] = (out[:, :, 0] > 128).astype(np.uint8)
    p3 = np.zeros((64, 16), dtype=np.uint8)
    blocks = np.stack([block_mode2(p, [(0, 0, 0), (0, 0, 0), (31, 0, 0), (31, 0, 0), (0, 31, 0), (0, 31, 0)]) for p in range(64)])
    out = decode_bc7_blocks(blocks)
    p3[:] = np.where(out[:, :, 0] > 128, 1, np.where(out[:, :, 1] > 128, 2, 0)).astype(np.uint8)

    # ---- anchors: endpoints e0=0,e1=max per subset; set one index bit at a time and see which
    # pixel moves and how far.  A pixel owning only (IB-1) bits is an anchor.
    def anchors(nsub, mk, nbits_total, ib, part_tab):
        res = np.zeros((64, nsub), dtype=np.int32)
        for p in range(64):
            blocks = np.stack([mk(p, 1 << b) for b in range(nbits_total)])
            out = decode_bc7_blocks(blocks)  # (nbits,16,4)
            nb = np.zeros(16, dtype=np.int32)
            for b in range(nbits_total):
                moved = np.nonzero(out[b, :, 0].astype(int) + out[b, :, 1].astype(int) + out[b, :, 2].astype(int) > 0)[0]
                assert len(moved) == 1, (p, b, moved)
                nb[moved[0]] += 1
            anc = np.nonzero(nb == ib - 1)[0]
            assert len(anc) == nsub, (p, anc)
            for a in anc:
                res[p, part_tab[p, a]] = a
            assert (np.sort(part_tab[p, anc]) == np.arange(nsub)).all()
        return res

    a2 = anchors(2, lambda p, bits: block_mode1(p, [(0, 0, 0), (63, 63, 63)] * 2, bits), 46, 3, p2)
    a3 = anchors(3, lambda p, bits: block_mode2(p, [(0, 0, 0), (31, 31, 31)] * 3, bits), 29, 2, p3)
    return p2, p3, a2, a3


if __name__ == "__main__":
    p2, p3, a2, a3 = probe()
    assert (a2[:, 0] == 0).all() and (a3[:, 0] == 0).all()

    def arr(name, a, ctype="uint8_t"):
        flat = ", ".join(str(int(x)) for x in a.reshape(-1))
        dims = "".join(f"[{d}]" for d in a.shape)
        print(f"static const {ctype} {name}{dims} = {{ {flat} }};")

    arr("BC7_PART2", p2)
    arr("BC7_PART3", p3)
    arr("BC7_ANCHOR2_1", a2[:, 1])
    arr("BC7_ANCHOR3_1", a3[:, 1])
    arr("BC7_ANCHOR3_2", a3[:, 2])
