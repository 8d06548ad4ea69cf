#!/usr/bin/env python3
"""CPU-side statistics of the headline scene's (triangle, bin) entries: entries per 16x16 bin, bbox pixels per
entry, and the lane efficiency of a lane = triangle walk.  Analysis only (uses the oracle's vertex stage)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mt_renderer_amd import scene
from oracle import oracle as orc

W, H, BIN = 1920, 1080, 16
md = scene.headline_model()
om = orc.OracleModel(md)
M = scene.to_f32_colmajor(scene.headline_transform(W, H))
pal = scene.bone_palette()
ents = []  # (bin, npx)
for p in range(md.nprims):
    f = scene.unpack_primitive(md.prims[p])
    clip, _ = om.vertex_stage(p, M, pal)
    w = clip[:, 3]
    X = np.rint((clip[:, 0] / w * 0.5 + 0.5) * W * 256).astype(np.int64)
    Y = np.rint((0.5 - clip[:, 1] / w * 0.5) * H * 256).astype(np.int64)
    idx = md.index_buf[f["index_ofs"]: f["index_ofs"] + f["index_num"]].astype(np.int64)
    i0, i1, i2 = idx[:-2], idx[1:-1], idx[2:]
    ok = (i0 != 0xFFFF) & (i1 != 0xFFFF) & (i2 != 0xFFFF)
    par = np.zeros(len(i0), dtype=bool)
    q = 0
    for k, v in enumerate(idx):  # parity within strips
        if v == 0xFFFF: q = 0; continue
        q += 1
        if q >= 3: par[k - 2] = ((q - 3) & 1) != 0
    a, b, c = i0, np.where(par, i2, i1), np.where(par, i1, i2)
    a, b, c = a[ok], b[ok], c[ok]
    A2 = (X[c] - X[a]) * (Y[b] - Y[a]) - (X[b] - X[a]) * (Y[c] - Y[a])
    fr = A2 > 0
    a, b, c = a[fr], b[fr], c[fr]
    xs = np.stack([X[a], X[b], X[c]]); ys = np.stack([Y[a], Y[b], Y[c]])
    px0 = np.maximum((xs.min(0) + 127) >> 8, 0); px1 = np.minimum((xs.max(0) - 128) >> 8, W - 1)
    py0 = np.maximum((ys.min(0) + 127) >> 8, 0); py1 = np.minimum((ys.max(0) - 128) >> 8, H - 1)
    keep = (px0 <= px1) & (py0 <= py1)
    px0, px1, py0, py1 = px0[keep], px1[keep], py0[keep], py1[keep]
    for t in range(len(px0)):
        for by in range(py0[t] // BIN, py1[t] // BIN + 1):
            for bx in range(px0[t] // BIN, px1[t] // BIN + 1):
                w_ = min(px1[t], bx * BIN + 15) - max(px0[t], bx * BIN) + 1
                h_ = min(py1[t], by * BIN + 15) - max(py0[t], by * BIN) + 1
                ents.append((by * 120 + bx, w_ * h_, w_, h_))
e = np.array(ents)
print("entries", len(e), "sum npx", e[:, 1].sum(), "mean npx %.2f" % e[:, 1].mean())
print("npx percentiles 50/90/99/max", np.percentile(e[:, 1], [50, 90, 99]), e[:, 1].max())
print("width pct", np.percentile(e[:, 2], [50, 90, 99]), "height pct", np.percentile(e[:, 3], [50, 90, 99]))
for c in (1, 2, 4, 6, 8, 12, 16, 32):
    print("  npx<=%d: %.1f%%" % (c, 100 * (e[:, 1] <= c).mean()))
nb = np.bincount(e[:, 0], minlength=8160)
print("bins nonempty", (nb > 0).sum(), "N percentiles 50/90/99/max", np.percentile(nb[nb > 0], [50, 90, 99]), nb.max())
# lane = triangle walk: iterations per pass = max npx among its 64 entries (queue order ~ submission order)
order = np.argsort(e[:, 0], kind="stable")
es = e[order]
it = 0; useful = 0
start = np.concatenate([[0], np.cumsum(nb)])
for b in range(8160):
    q = es[start[b]:start[b + 1], 1]
    for p in range(0, len(q), 64):
        it += q[p:p + 64].max(); useful += q[p:p + 64].sum()
print("lane=triangle: wave iterations", it, "lane efficiency %.1f%%" % (100 * useful / (64 * it)))

# ---- cost simulation (instruction units per wave) of small-triangle walks, per pass of 64 queue entries ----
def coop_cost(npx):
    return 80 + np.ceil(npx / 64) * 40

def sim(mode, it_cost, thr_set):
    total = 0.0
    for b in range(8160):
        q = es[start[b]:start[b + 1]]
        for p in range(0, len(q), 64):
            npx, w_, h_ = q[p:p + 64, 1], q[p:p + 64, 2], q[p:p + 64, 3]
            best = None
            for c in thr_set:
                small = npx <= c
                cost = coop_cost(npx[~small]).sum()
                if small.any():
                    if mode == "lane":
                        cost += npx[small].max() * it_cost
                    else:
                        G = mode
                        per = np.where(small, np.ceil(w_ / G) * h_, 0)
                        T = 64 // G
                        for g in range(0, len(per), T):
                            m = per[g:g + T].max()
                            if m: cost += m * it_cost + 40
                best = cost if best is None else min(best, cost)
            total += best + 250  # setup etc.
    return total

base = sim("lane", 45, (16, 32, 64, 128, 256))
print("lane=triangle (now)        : %.1f M units" % (base / 1e6))
print("lane=triangle trimmed loop : %.1f M" % (sim("lane", 30, (16, 32, 64, 128, 256)) / 1e6))
for G in (2, 4, 8):
    print("G=%d column walk            : %.1f M" % (G, sim(G, 32, (8, 16, 32, 64, 128, 256)) / 1e6))

# ---- per-bin totals: what one workgroup of the visibility kernel has to walk ----
pairs = np.bincount(e[:, 0], weights=e[:, 1], minlength=8160)
nz = pairs[pairs > 0]
print("pairs per non-empty bin: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f  (sum %.0f)" % (nz.mean(), *np.percentile(nz, [50, 90, 99]), nz.max(), nz.sum()))
print("bins with more than 4x the mean: %d; their share of all pairs: %.1f%%" % ((nz > 4 * nz.mean()).sum(), 100 * nz[nz > 4 * nz.mean()].sum() / nz.sum()))
