#!/usr/bin/env python3
"""Writes tools/abl/k_tile_vis_abl.hip = csrc/k_tile_vis.hip + #ifdef ABL_T_* hooks (timing ablations, WRONG pixels)."""
import os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(R, "mt_renderer_amd/csrc/k_tile_vis.hip")).read()


def sub(old, new):
    global s
    assert s.count(old) == 1, old
    s = s.replace(old, new)


sub("        if (valid) setup_tri(a_cur, ord_cur, binx0, biny0, vw, vh, s);",
    "#ifdef ABL_T_NOSETUP\n        if (valid && a_cur.X0 == 0x7ffffff0) s_key[0] = 1;\n#else\n"
    "        if (valid) setup_tri(a_cur, ord_cur, binx0, biny0, vw, vh, s);\n#endif")
sub("        for (uint64_t todo = zlim_ok ? __ballot(npx != 0 && !large) : 0ull; todo;) {",
    "#ifdef ABL_T_NOFLAT\n        for (uint64_t todo = 0; todo;) {\n#else\n"
    "        for (uint64_t todo = zlim_ok ? __ballot(npx != 0 && !large) : 0ull; todo;) {\n#endif")
sub("        for (uint64_t mb = __ballot(npx != 0 && large); mb; mb &= mb - 1) {",
    "#ifdef ABL_T_NOCOOP\n        for (uint64_t mb = 0; mb; mb &= mb - 1) {\n#else\n"
    "        for (uint64_t mb = __ballot(npx != 0 && large); mb; mb &= mb - 1) {\n#endif")
sub("            const uint4 tail = reinterpret_cast<const uint4*>(&P.fb.rec_a[r])[2];",
    "#ifdef ABL_T_NOWINNER\n            const uint4 tail = {r, r, r, 0u};\n#else\n"
    "            const uint4 tail = reinterpret_cast<const uint4*>(&P.fb.rec_a[r])[2];\n#endif")
open(os.path.join(R, "tools/abl/k_tile_vis_abl.hip"), "w").write(s)
