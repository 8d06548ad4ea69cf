#!/usr/bin/env python3
"""Writes tools/abl/k_geom_abl.hip = csrc/k_geom.hip + #ifdef ABL_* hooks (timing ablations, results are WRONG pixels).
ABL_NOSKIN ABL_EARLY ABL_NOREC ABL_NOBIN ABL_NOCLIP ABL_NOSTAT ABL_NOMAT ABL_LDS=<bytes>; see run_geom.sh."""
import os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(R, "mt_renderer_amd/csrc/k_geom.hip")).read()


def sub(old, new):
    global s
    assert s.count(old) == 1, old
    s = s.replace(old, new)


sub("    const VOut v = shade_vertex_mfma(P.vbuf, pr, vid, vvalid, M, s_pal, P.npal, skinned);",
    "#ifdef ABL_NOSKIN\n    const VOut v = shade_vertex_mfma(P.vbuf, pr, vid, vvalid, M, s_pal, P.npal, false);\n#else\n"
    "    const VOut v = shade_vertex_mfma(P.vbuf, pr, vid, vvalid, M, s_pal, P.npal, skinned);\n#endif")
sub("    // ---- strip assembly: which lanes complete",
    "#ifdef ABL_EARLY\n    if (me.X == 0x7fffffff) P.fb.chunk_info[gid].n = me.Y;\n    return;\n#endif\n"
    "    // ---- strip assembly: which lanes complete")
sub("        P.fb.rec_a[base + rank] = r0.a;\n        P.fb.rec_hdr[base + rank] = r0.h;",
    "#ifndef ABL_NOREC\n        P.fb.rec_a[base + rank] = r0.a;\n        P.fb.rec_hdr[base + rank] = r0.h;\n#endif")
sub("    for (uint32_t round = 0; round * 64 < total; ++round) {\n        const uint32_t j = round * 64 + lane;",
    "#ifdef ABL_NOBIN\n    if (s_hdr[wave][lane].bx0 == 0x7fff) P.fb.chunk_info[gid].n = 1;\n    return;\n#endif\n"
    "    for (uint32_t round = 0; round * 64 < total; ++round) {\n        const uint32_t j = round * 64 + lane;")
sub("        if (!((f_or >> 2) & OC_ZN)) {\n            if (setup_tri(ta, tb, tc, W, H, mat, r0)) n_out = 1;\n        } else {",
    "        if (!((f_or >> 2) & OC_ZN)) {\n            if (setup_tri(ta, tb, tc, W, H, mat, r0)) n_out = 1;\n        }\n"
    "#ifndef ABL_NOCLIP\n        else {")
sub("                n_out = (s0 ? 1u : 0u) + (s1 ? 1u : 0u);\n            }\n        }\n    }\n",
    "                n_out = (s0 ? 1u : 0u) + (s1 ? 1u : 0u);\n            }\n        }\n#endif\n    }\n")
sub("    size_t lds = (size_t)p.npal * 64;", "#ifndef ABL_LDS\n#define ABL_LDS 0\n#endif\n    size_t lds = (size_t)p.npal * 64 + ABL_LDS;")
old = "        if (total) atomicAdd(&P.fb.counters[MTR_CTR(CTR_REC, gid)], total);  // statistics only"
sub(old, "#ifndef ABL_NOSTAT\n" + old + "\n#endif")
old = "    const DMat dmat = P.mats[mat];  // wave-uniform"
sub(old, "#ifdef ABL_NOMAT\n    DMat dmat{}; dmat.rgba8 = mat; dmat.shader = MTR_SH_DEBUG;\n#else\n" + old + "\n#endif")
open(os.path.join(R, "tools/abl/k_geom_abl.hip"), "w").write(s)
