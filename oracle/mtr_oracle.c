/* mtr_oracle.c -- CPU ORACLE for the rModel draw path.  TEST INFRASTRUCTURE ONLY (see
 * mtr_oracle.h).  Scalar C, no fast-math; compile with -ffp-contract=off: every fused
 * multiply-add is an explicit fmaf(), every other float op is a single IEEE-754 binary32
 * operation with round-to-nearest-even, denormals kept.  "parity unpinned" vs the reference's real
 * pixels (the reference has no raster tests); normative for this build (SPEC.md).
 */
#include "mtr_oracle.h"
#include "bc7_tables.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * crc32, MT flavour: table poly 0xEDB88320, no final xor, stops at the first NUL byte.
 * follows src/util/crc.rs:36-50 (table generated here instead of spelled out).
 * ---------------------------------------------------------------------------------------- */
static uint32_t g_crc_table[256];
static int g_crc_init = 0;
static void crc_init(void) {
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        g_crc_table[i] = c;
    }
    g_crc_init = 1;
}
uint32_t orc_crc32(const uint8_t *bytes, size_t len, uint32_t init) {
    if (!g_crc_init) crc_init();
    uint32_t val = init;
    for (size_t i = 0; i < len; i++) {
        uint8_t b = bytes[i];
        if (b == 0) break; /* src/util/crc.rs:39-42 */
        val = g_crc_table[(b ^ val) & 0xff] ^ (val >> 8);
    }
    return val;
}

/* ------------------------------------------------------------------------------------------
 * PrimitiveInfo accessors -- src/rmodel.rs:173-225 (struct layout :135-171, 14 LE u32 words)
 * ---------------------------------------------------------------------------------------- */
static uint32_t rd32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
uint32_t orc_prim_vertex_num(const uint8_t *p) { return (rd32(p + 0) >> 16) & 0xffff; }
uint32_t orc_prim_parts_no(const uint8_t *p) { return rd32(p + 4) & 0xfff; }
uint32_t orc_prim_material_no(const uint8_t *p) { return (rd32(p + 4) >> 12) & 0xfff; }
uint32_t orc_prim_weight_num(const uint8_t *p) { return (rd32(p + 8) >> 3) & 0x1f; }
uint32_t orc_prim_vertex_stride(const uint8_t *p) { return (rd32(p + 8) >> 16) & 0xff; }
uint32_t orc_prim_topology(const uint8_t *p) { return (rd32(p + 8) >> 24) & 0x3f; }
uint32_t orc_prim_vertex_base(const uint8_t *p) { return rd32(p + 16); }
uint32_t orc_prim_inputlayout(const uint8_t *p) { return rd32(p + 20); }
uint32_t orc_prim_index_ofs(const uint8_t *p) { return rd32(p + 24); }
uint32_t orc_prim_index_num(const uint8_t *p) { return rd32(p + 28); }
uint32_t orc_prim_index_base(const uint8_t *p) { return rd32(p + 32); }
uint32_t orc_prim_boundary_num(const uint8_t *p) { return (rd32(p + 36) >> 8) & 0xff; }

/* ------------------------------------------------------------------------------------------
 * BC1 (Bc1RgbaUnorm) -- format id 19, src/rtexture.rs:155.  SPEC.md "BC1".
 * ---------------------------------------------------------------------------------------- */
void orc_bc1_decode_block(const uint8_t blk[8], uint8_t out[64]) {
    uint32_t c0 = (uint32_t)blk[0] | ((uint32_t)blk[1] << 8);
    uint32_t c1 = (uint32_t)blk[2] | ((uint32_t)blk[3] << 8);
    uint32_t bits = rd32(blk + 4);
    uint32_t col[4][4];
    uint32_t c[2] = {c0, c1};
    for (int i = 0; i < 2; i++) {
        uint32_t r5 = (c[i] >> 11) & 31, g6 = (c[i] >> 5) & 63, b5 = c[i] & 31;
        col[i][0] = (r5 << 3) | (r5 >> 2);
        col[i][1] = (g6 << 2) | (g6 >> 4);
        col[i][2] = (b5 << 3) | (b5 >> 2);
        col[i][3] = 255;
    }
    if (c0 > c1) {
        for (int k = 0; k < 3; k++) {
            col[2][k] = (2 * col[0][k] + col[1][k] + 1) / 3;
            col[3][k] = (col[0][k] + 2 * col[1][k] + 1) / 3;
        }
        col[2][3] = 255;
        col[3][3] = 255;
    } else {
        for (int k = 0; k < 3; k++) {
            col[2][k] = (col[0][k] + col[1][k] + 1) / 2;
            col[3][k] = 0;
        }
        col[2][3] = 255;
        col[3][3] = 0;
    }
    for (int i = 0; i < 16; i++) {
        uint32_t s = (bits >> (2 * i)) & 3;
        for (int k = 0; k < 4; k++) out[i * 4 + k] = (uint8_t)col[s][k];
    }
}

/* ------------------------------------------------------------------------------------------
 * BC7 (Bc7RgbaUnorm) -- format ids 42 and 54, src/rtexture.rs:156-158.
 * Khronos Data Format Specification, BPTC: bit-exact by spec.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t ns, pb, rb, isb, cb, ab, epb, spb, ib, ib2;
} bc7_mode;
static const bc7_mode BC7_MODES[8] = {
    {3, 4, 0, 0, 4, 0, 1, 0, 3, 0}, {2, 6, 0, 0, 6, 0, 0, 1, 3, 0}, {3, 6, 0, 0, 5, 0, 0, 0, 2, 0},
    {2, 6, 0, 0, 7, 0, 1, 0, 2, 0}, {1, 0, 2, 1, 5, 6, 0, 0, 2, 3}, {1, 0, 2, 0, 7, 8, 0, 0, 2, 2},
    {1, 0, 0, 0, 7, 7, 1, 0, 4, 0}, {2, 6, 0, 0, 5, 5, 1, 0, 2, 0},
};
static const uint8_t BC7_W2[4] = {0, 21, 43, 64};
static const uint8_t BC7_W3[8] = {0, 9, 18, 27, 37, 46, 55, 64};
static const uint8_t BC7_W4[16] = {0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64};

typedef struct {
    const uint8_t *d;
    uint32_t pos;
} bitrd;
static uint32_t getbits(bitrd *b, uint32_t n) {
    uint32_t v = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t p = b->pos + i;
        v |= (uint32_t)((b->d[p >> 3] >> (p & 7)) & 1) << i;
    }
    b->pos += n;
    return v;
}
static uint32_t bc7_interp(uint32_t e0, uint32_t e1, uint32_t idx, uint32_t bits) {
    uint32_t w = bits == 2 ? BC7_W2[idx] : bits == 3 ? BC7_W3[idx] : BC7_W4[idx];
    return ((64 - w) * e0 + w * e1 + 32) >> 6;
}
void orc_bc7_decode_block(const uint8_t blk[16], uint8_t out[64]) {
    uint32_t mode = 0;
    while (mode < 8 && !((blk[0] >> mode) & 1)) mode++;
    if (mode >= 8) { /* reserved: all zero */
        memset(out, 0, 64);
        return;
    }
    const bc7_mode *m = &BC7_MODES[mode];
    bitrd br = {blk, mode + 1};
    uint32_t part = getbits(&br, m->pb);
    uint32_t rot = getbits(&br, m->rb);
    uint32_t isel = getbits(&br, m->isb);
    uint32_t ne = m->ns * 2u;
    uint32_t ep[6][4];
    for (int c = 0; c < 3; c++)
        for (uint32_t e = 0; e < ne; e++) ep[e][c] = getbits(&br, m->cb);
    for (uint32_t e = 0; e < ne; e++) ep[e][3] = m->ab ? getbits(&br, m->ab) : 255u;
    uint32_t cprec = m->cb, aprec = m->ab;
    if (m->epb) {
        for (uint32_t e = 0; e < ne; e++) {
            uint32_t p = getbits(&br, 1);
            for (int c = 0; c < 3; c++) ep[e][c] = (ep[e][c] << 1) | p;
            if (m->ab) ep[e][3] = (ep[e][3] << 1) | p;
        }
        cprec++;
        if (m->ab) aprec++;
    } else if (m->spb) {
        for (uint32_t s = 0; s < m->ns; s++) {
            uint32_t p = getbits(&br, 1);
            for (uint32_t e = 2 * s; e < 2 * s + 2; e++) {
                for (int c = 0; c < 3; c++) ep[e][c] = (ep[e][c] << 1) | p;
                if (m->ab) ep[e][3] = (ep[e][3] << 1) | p;
            }
        }
        cprec++;
        if (m->ab) aprec++;
    }
    for (uint32_t e = 0; e < ne; e++) {
        for (int c = 0; c < 3; c++) {
            uint32_t v = ep[e][c] << (8 - cprec);
            ep[e][c] = v | (v >> cprec);
        }
        if (m->ab) {
            uint32_t v = ep[e][3] << (8 - aprec);
            ep[e][3] = v | (v >> aprec);
        }
    }
    const uint8_t *ptab = m->ns == 2 ? BC7_PART2[part] : BC7_PART3[part];
    uint32_t anchor[3] = {0, 0, 0};
    if (m->ns == 2) anchor[1] = BC7_ANCHOR2_1[part];
    if (m->ns == 3) {
        anchor[1] = BC7_ANCHOR3_1[part];
        anchor[2] = BC7_ANCHOR3_2[part];
    }
    uint32_t idx1[16], idx2[16];
    for (uint32_t i = 0; i < 16; i++) {
        uint32_t s = m->ns == 1 ? 0 : ptab[i];
        uint32_t nb = m->ib - (i == anchor[s] ? 1u : 0u);
        idx1[i] = getbits(&br, nb);
    }
    for (uint32_t i = 0; i < 16; i++) {
        idx2[i] = 0;
        if (m->ib2) idx2[i] = getbits(&br, m->ib2 - (i == 0 ? 1u : 0u));
    }
    for (uint32_t i = 0; i < 16; i++) {
        uint32_t s = m->ns == 1 ? 0 : ptab[i];
        const uint32_t *e0 = ep[2 * s], *e1 = ep[2 * s + 1];
        uint32_t ci = idx1[i], cbits = m->ib, ai = idx1[i], abits = m->ib;
        if (m->ib2) {
            if (isel) {
                ci = idx2[i];
                cbits = m->ib2;
            } else {
                ai = idx2[i];
                abits = m->ib2;
            }
        }
        uint32_t px[4];
        for (int c = 0; c < 3; c++) px[c] = bc7_interp(e0[c], e1[c], ci, cbits);
        px[3] = m->ab ? bc7_interp(e0[3], e1[3], ai, abits) : 255u;
        if (rot) {
            uint32_t t = px[3];
            px[3] = px[rot - 1];
            px[rot - 1] = t;
        }
        for (int c = 0; c < 4; c++) out[i * 4 + c] = (uint8_t)px[c];
    }
}

int orc_texture_decode(uint32_t fmt, uint32_t w, uint32_t h, const uint8_t *data, size_t len,
                       uint8_t *out) {
    if (w == 0 || h == 0) return ORC_E_INVALID;
    if (fmt == ORC_TEX_RGBA8) {
        if (len < (size_t)w * h * 4) return ORC_E_INVALID;
        memcpy(out, data, (size_t)w * h * 4);
        return ORC_OK;
    }
    if (fmt != ORC_TEX_BC1 && fmt != ORC_TEX_BC7 && fmt != ORC_TEX_BC7_ALT)
        return ORC_E_UNSUPPORTED; /* src/rtexture.rs:159 todo!() */
    uint32_t bw = (w + 3) / 4, bh = (h + 3) / 4;
    size_t bsz = fmt == ORC_TEX_BC1 ? 8 : 16;
    if (len < (size_t)bw * bh * bsz) return ORC_E_INVALID;
    for (uint32_t by = 0; by < bh; by++)
        for (uint32_t bx = 0; bx < bw; bx++) {
            uint8_t px[64];
            const uint8_t *blk = data + ((size_t)by * bw + bx) * bsz;
            if (fmt == ORC_TEX_BC1)
                orc_bc1_decode_block(blk, px);
            else
                orc_bc7_decode_block(blk, px);
            for (uint32_t y = 0; y < 4; y++)
                for (uint32_t x = 0; x < 4; x++) {
                    uint32_t X = bx * 4 + x, Y = by * 4 + y;
                    if (X < w && Y < h) memcpy(out + ((size_t)Y * w + X) * 4, px + (y * 4 + x) * 4, 4);
                }
        }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * vertex fetch -- address rule src/model.rs:337-342,357-361; format table src/rshader2.rs:516-564
 * ---------------------------------------------------------------------------------------- */
static float half_to_float(uint16_t h) {
    uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31, f = h & 1023, bits;
    if (e == 0) {
        if (f == 0)
            bits = s;
        else { /* subnormal: value = f * 2^-24, exact in binary32 */
            float v = (float)f * 5.9604644775390625e-08f;
            memcpy(&bits, &v, 4);
            bits |= s;
        }
    } else if (e == 31)
        bits = s | 0x7f800000u | (f << 13);
    else
        bits = s | ((e + 112) << 23) | (f << 13);
    float r;
    memcpy(&r, &bits, 4);
    return r;
}
static float snorm16f(int16_t v) {
    float f = (float)v / 32767.0f;
    return f < -1.0f ? -1.0f : f;
}
static float snorm8f(int8_t v) {
    float f = (float)v / 127.0f;
    return f < -1.0f ? -1.0f : f;
}
static float unorm8f(uint8_t v) { return (float)v / 255.0f; }

/* Decodes a float-class element to (x,y,z,w) with WebGPU's (0,0,0,1) fill.  Returns the number of
 * bytes the wgpu format reads, or 0 if the (format,count) pair is not in the reference's table
 * (todo!() there) or is an integer format that cannot feed a vec4f/vec2f shader input. */
static uint32_t elem_bytes(uint8_t fmt, uint8_t cnt) {
    switch (fmt) {
    case ORC_IEF_U8N: return cnt == 1 ? 2 : cnt == 4 ? 4 : 0;      /* Unorm8x2 / Unorm8x4 */
    case ORC_IEF_S8N: return cnt == 1 ? 2 : (cnt == 3 || cnt == 4) ? 4 : 0; /* Snorm8x2/x4 */
    case ORC_IEF_S16N: return cnt == 1 ? 4 : cnt == 3 ? 8 : 0;     /* Snorm16x2 / Snorm16x4 */
    case ORC_IEF_F16: return cnt == 2 ? 4 : 0;                     /* Float16x2 */
    case ORC_IEF_F32: return cnt == 3 ? 12 : 0;                    /* Float32x3 */
    case ORC_IEF_U8NL: return cnt == 3 ? 4 : 0;                    /* Unorm8x4 */
    case ORC_IEF_SCMP3N: return 4;                                 /* 10:10:10 signed normalised (build extension, SPEC 2) */
    default: return 0;
    }
}
static float snorm10f(uint32_t bits10) {
    int32_t v = (int32_t)(bits10 << 22) >> 22;
    float f = (float)v / 511.0f;
    return f < -1.0f ? -1.0f : f;
}
static void elem_decode(uint8_t fmt, uint8_t cnt, const uint8_t *p, float out[4]) {
    out[0] = out[1] = out[2] = 0.0f;
    out[3] = 1.0f;
    uint32_t nb = elem_bytes(fmt, cnt);
    switch (fmt) {
    case ORC_IEF_U8N:
    case ORC_IEF_U8NL:
        for (uint32_t i = 0; i < nb; i++) out[i] = unorm8f(p[i]);
        break;
    case ORC_IEF_S8N:
        for (uint32_t i = 0; i < nb; i++) out[i] = snorm8f((int8_t)p[i]);
        break;
    case ORC_IEF_S16N:
        for (uint32_t i = 0; i < nb / 2; i++)
            out[i] = snorm16f((int16_t)((uint16_t)p[2 * i] | ((uint16_t)p[2 * i + 1] << 8)));
        break;
    case ORC_IEF_F16:
        for (uint32_t i = 0; i < 2; i++)
            out[i] = half_to_float((uint16_t)((uint16_t)p[2 * i] | ((uint16_t)p[2 * i + 1] << 8)));
        break;
    case ORC_IEF_F32:
        for (uint32_t i = 0; i < 3; i++) memcpy(&out[i], p + 4 * i, 4);
        break;
    case ORC_IEF_SCMP3N: {
        uint32_t w = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        out[0] = snorm10f(w & 0x3ff);
        out[1] = snorm10f((w >> 10) & 0x3ff);
        out[2] = snorm10f((w >> 20) & 0x3ff);
        break;
    }
    default: break;
    }
}

typedef struct {
    int has_pos, has_uv, has_joint, has_weight;
    orc_element pos, uv, joint, weight;
} layout_info;

static int layout_resolve(const orc_layout *l, layout_info *li) {
    memset(li, 0, sizeof *li);
    if (l->num > 8) return ORC_E_INVALID;
    for (uint32_t i = 0; i < l->num; i++) {
        const orc_element *e = &l->el[i];
        /* src/rshader2.rs:509-512 skips these; pad0 bit 0 (ORC_ELEM_DECODE_SCMP3N) opts into this build's decode */
        if (e->format == ORC_IEF_SCMP3N && !(e->pad0 & 1)) continue;
        switch (e->semantic) {
        case ORC_SEM_POSITION:
        case ORC_SEM_TEXCOORD:
            if (elem_bytes(e->format, e->count) == 0) return ORC_E_UNSUPPORTED;
            if (e->semantic == ORC_SEM_POSITION) {
                li->has_pos = 1;
                li->pos = *e;
            } else {
                li->has_uv = 1;
                li->uv = *e;
            }
            break;
        case ORC_SEM_JOINT:
            if (e->format != ORC_IEF_U8 || e->count != 4) return ORC_E_UNSUPPORTED;
            li->has_joint = 1;
            li->joint = *e;
            break;
        case ORC_SEM_WEIGHT:
            if (e->format != ORC_IEF_U8N || e->count != 4) return ORC_E_UNSUPPORTED;
            li->has_weight = 1;
            li->weight = *e;
            break;
        default: break; /* src/rshader2.rs:506 `_ => continue` */
        }
    }
    if (!li->has_pos) return ORC_E_UNSUPPORTED; /* both WGSL vertex shaders need location 0 */
    return ORC_OK;
}

typedef struct {
    float x, y, z, w, u, v;
} cvert;

typedef struct {
    const orc_model *m;
    const uint8_t *prim;
    layout_info li;
    uint32_t stride, vbase, vnum;
    float M[16];
    const float *palette;
    size_t npal;
    int skinned;
} vs_ctx;

/* The vertex shader: LBS (build extension, SPEC.md "LBS") then clip = M * (q,1)
 * (src/shaders/textured.wgsl:15, src/shaders/debug_ids.wgsl:13). */
static cvert shade_vertex(const vs_ctx *c, uint32_t vidx) {
    const uint8_t *vp = c->m->vertex_buf + c->vbase + (size_t)vidx * c->stride;
    float p[4], t[4] = {0, 0, 0, 1};
    elem_decode(c->li.pos.format, c->li.pos.count, vp + c->li.pos.offset, p);
    if (c->li.has_uv) elem_decode(c->li.uv.format, c->li.uv.count, vp + c->li.uv.offset, t);
    float q[4] = {p[0], p[1], p[2], 1.0f};
    if (c->skinned) {
        const uint8_t *jp = vp + c->li.joint.offset, *wp = vp + c->li.weight.offset;
        float acc[3] = {0.0f, 0.0f, 0.0f};
        float pin[4] = {p[0], p[1], p[2], 1.0f};
        for (int k = 0; k < 4; k++) {
            uint32_t j = jp[k];
            if (j >= c->npal) j = (uint32_t)c->npal - 1;
            const float *P = c->palette + (size_t)j * 16;
            float wk = unorm8f(wp[k]);
            for (int col = 0; col < 4; col++) {
                float s = wk * pin[col];
                for (int i = 0; i < 3; i++) acc[i] = fmaf(P[col * 4 + i], s, acc[i]);
            }
        }
        q[0] = acc[0];
        q[1] = acc[1];
        q[2] = acc[2];
    }
    float clip[4];
    for (int i = 0; i < 4; i++) {
        float a = 0.0f;
        for (int col = 0; col < 4; col++) a = fmaf(c->M[col * 4 + i], q[col], a);
        clip[i] = a;
    }
    cvert r = {clip[0], clip[1], clip[2], clip[3], t[0], t[1]};
    return r;
}

void orc_mat4_mul(const float A[16], const float B[16], float out[16]) {
    float r[16];
    for (int c = 0; c < 4; c++)
        for (int i = 0; i < 4; i++) {
            float a = 0.0f;
            for (int k = 0; k < 4; k++) a = fmaf(A[k * 4 + i], B[c * 4 + k], a);
            r[c * 4 + i] = a;
        }
    memcpy(out, r, sizeof r);
}

static int vs_ctx_init(vs_ctx *c, const orc_model *m, size_t prim, const float M[16],
                       const float *palette, size_t npal) {
    if (prim >= m->nprims) return ORC_E_INVALID;
    c->m = m;
    c->prim = m->prims + prim * 0x38;
    int rc = layout_resolve(&m->layouts[prim], &c->li);
    if (rc) return rc;
    c->stride = orc_prim_vertex_stride(c->prim);
    c->vbase = orc_prim_vertex_base(c->prim);
    c->vnum = orc_prim_vertex_num(c->prim);
    memcpy(c->M, M, sizeof c->M);
    c->palette = palette;
    c->npal = npal;
    c->skinned = c->li.has_joint && c->li.has_weight && palette && npal > 0;
    /* every element must lie inside the stride and the bound slice inside the buffer */
    const orc_element *es[4] = {&c->li.pos, c->li.has_uv ? &c->li.uv : NULL,
                                c->skinned ? &c->li.joint : NULL, c->skinned ? &c->li.weight : NULL};
    for (int i = 0; i < 4; i++) {
        if (!es[i]) continue;
        uint32_t nb = (es[i]->semantic <= ORC_SEM_TEXCOORD) ? elem_bytes(es[i]->format, es[i]->count) : 4;
        if ((uint32_t)es[i]->offset + nb > c->stride) return ORC_E_INVALID;
    }
    if ((size_t)c->vbase + (size_t)c->vnum * c->stride > m->vertex_len) return ORC_E_INVALID;
    return ORC_OK;
}

int orc_vertex_stage(const orc_model *m, size_t prim, const float M[16], const float *palette,
                     size_t npal, float *out_clip, float *out_uv) {
    vs_ctx c;
    int rc = vs_ctx_init(&c, m, prim, M, palette, npal);
    if (rc) return rc;
    for (uint32_t v = 0; v < c.vnum; v++) {
        cvert r = shade_vertex(&c, v);
        out_clip[4 * v + 0] = r.x;
        out_clip[4 * v + 1] = r.y;
        out_clip[4 * v + 2] = r.z;
        out_clip[4 * v + 3] = r.w;
        out_uv[2 * v + 0] = r.u;
        out_uv[2 * v + 1] = r.v;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * frame
 * ---------------------------------------------------------------------------------------- */
struct orc_frame {
    uint32_t w, h;
    uint8_t *color;
    float *depth;
    uint64_t tris_in, tris_setup, frags;
};

static uint8_t quant8(float x) {
    if (!(x > 0.0f)) x = 0.0f; /* also NaN -> 0 */
    if (x > 1.0f) x = 1.0f;
    return (uint8_t)rintf(x * 255.0f);
}

orc_frame *orc_frame_create(uint32_t w, uint32_t h, const float clear[4], float clear_depth) {
    if (w == 0 || h == 0 || w > 16384 || h > 16384) return NULL;
    orc_frame *f = (orc_frame *)calloc(1, sizeof *f);
    f->w = w;
    f->h = h;
    f->color = (uint8_t *)malloc((size_t)w * h * 4);
    f->depth = (float *)malloc((size_t)w * h * sizeof(float));
    uint8_t c[4] = {quant8(clear[0]), quant8(clear[1]), quant8(clear[2]), quant8(clear[3])};
    for (size_t i = 0; i < (size_t)w * h; i++) {
        memcpy(f->color + 4 * i, c, 4);
        f->depth[i] = clear_depth;
    }
    return f;
}
void orc_frame_destroy(orc_frame *f) {
    if (!f) return;
    free(f->color);
    free(f->depth);
    free(f);
}
const uint8_t *orc_frame_color(const orc_frame *f) { return f->color; }
const float *orc_frame_depth(const orc_frame *f) { return f->depth; }
uint64_t orc_frame_tris_in(const orc_frame *f) { return f->tris_in; }
uint64_t orc_frame_tris_setup(const orc_frame *f) { return f->tris_setup; }
uint64_t orc_frame_frags(const orc_frame *f) { return f->frags; }

/* ------------------------------------------------------------------------------------------
 * clip, project, setup  (fixed-function state: src/model.rs:249-262)
 * ---------------------------------------------------------------------------------------- */
#define ORC_GUARD_BAND 1048576.0f /* 2^20 px */

enum { SH_DEBUG = 0, SH_TEXTURED = 1, SH_CONST = 2 };

typedef struct {
    int32_t X[3], Y[3];
    int64_t A2;
    float z0, dz1, dz2;
    float iw0, diw1, diw2, up0, dup1, dup2, vp0, dvp1, dvp2;
    float rcpA;
    int valid;
} tri_setup;

typedef struct {
    int ok;
    int32_t X, Y;
    float z, iw, up, vp;
} pvert;

static pvert project(cvert c, uint32_t W, uint32_t H) {
    pvert p;
    memset(&p, 0, sizeof p);
    if (!(c.w > 0.0f)) return p;
    float iw = 1.0f / c.w;
    float xn = c.x * iw, yn = c.y * iw, zn = c.z * iw;
    float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
    float xf = fmaf(xn, hw, hw);
    float yf = fmaf(-yn, hh, hh);
    if (!(fabsf(xf) <= ORC_GUARD_BAND && fabsf(yf) <= ORC_GUARD_BAND)) return p;
    p.ok = 1;
    p.X = (int32_t)rintf(xf * 256.0f);
    p.Y = (int32_t)rintf(yf * 256.0f);
    p.z = zn;
    p.iw = iw;
    p.up = c.u * iw;
    p.vp = c.v * iw;
    return p;
}

static cvert clip_lerp(cvert in, cvert out) { /* in.z >= 0 > out.z */
    float t = in.z / (in.z - out.z);
    cvert r;
    r.x = fmaf(t, out.x - in.x, in.x);
    r.y = fmaf(t, out.y - in.y, in.y);
    r.z = 0.0f;
    r.w = fmaf(t, out.w - in.w, in.w);
    r.u = fmaf(t, out.u - in.u, in.u);
    r.v = fmaf(t, out.v - in.v, in.v);
    return r;
}

enum { CULL_BACK = 0, CULL_NONE = 1, CULL_FRONT = 2 };

static int setup_tri(cvert a, cvert b, cvert c, uint32_t W, uint32_t H, int cull, tri_setup *s) {
    pvert p[3] = {project(a, W, H), project(b, W, H), project(c, W, H)};
    s->valid = 0;
    if (!(p[0].ok && p[1].ok && p[2].ok)) return 0;
    int64_t A2 = (int64_t)(p[2].X - p[0].X) * (int64_t)(p[1].Y - p[0].Y) -
                 (int64_t)(p[1].X - p[0].X) * (int64_t)(p[2].Y - p[0].Y);
    /* cull_mode Back, front_face Ccw: src/model.rs:252.  Material state (SPEC 10): a back face that is kept is set
     * up with its second and third vertex exchanged, i.e. as the front face it is when seen from behind. */
    if (cull == CULL_BACK) {
        if (A2 <= 0) return 0;
    } else {
        if (A2 == 0 || (cull == CULL_FRONT && A2 > 0)) return 0;
        if (A2 < 0) {
            pvert t = p[1];
            p[1] = p[2];
            p[2] = t;
            A2 = -A2;
        }
    }
    for (int i = 0; i < 3; i++) {
        s->X[i] = p[i].X;
        s->Y[i] = p[i].Y;
    }
    s->A2 = A2;
    s->rcpA = 1.0f / (float)A2;
    s->z0 = p[0].z;
    s->dz1 = p[1].z - p[0].z;
    s->dz2 = p[2].z - p[0].z;
    s->iw0 = p[0].iw;
    s->diw1 = p[1].iw - p[0].iw;
    s->diw2 = p[2].iw - p[0].iw;
    s->up0 = p[0].up;
    s->dup1 = p[1].up - p[0].up;
    s->dup2 = p[2].up - p[0].up;
    s->vp0 = p[0].vp;
    s->dvp1 = p[1].vp - p[0].vp;
    s->dvp2 = p[2].vp - p[0].vp;
    /* bbox of pixel centres; reject if it holds none inside the target */
    int32_t xmin = s->X[0], xmax = s->X[0], ymin = s->Y[0], ymax = s->Y[0];
    for (int i = 1; i < 3; i++) {
        if (s->X[i] < xmin) xmin = s->X[i];
        if (s->X[i] > xmax) xmax = s->X[i];
        if (s->Y[i] < ymin) ymin = s->Y[i];
        if (s->Y[i] > ymax) ymax = s->Y[i];
    }
    int32_t px0 = (xmin + 127) >> 8, px1 = (xmax - 128) >> 8;
    int32_t py0 = (ymin + 127) >> 8, py1 = (ymax - 128) >> 8;
    if (px0 < 0) px0 = 0;
    if (py0 < 0) py0 = 0;
    if (px1 > (int32_t)W - 1) px1 = (int32_t)W - 1;
    if (py1 > (int32_t)H - 1) py1 = (int32_t)H - 1;
    if (px0 > px1 || py0 > py1) return 0;
    s->valid = 1;
    return 1;
}

/* Guard band (SPEC 5.3): the rasteriser works in 24.8 fixed point inside +-2^20 px.  A polygon with a vertex it cannot
 * project (w <= 0, or beyond the band) is clipped against the four planes |x| <= 64 w, |y| <= 64 w -- 64 half-targets from
 * the centre, inside the band for every target up to 16384 px -- in that order: +x, -x, +y, -y.  Same Sutherland-Hodgman
 * step as the near plane: a is emitted if inside, then the intersection, always computed from the inside vertex towards
 * the outside one with t = d_I / (d_I - d_O), d = the plane's signed distance, every attribute fma(t, O - I, I). */
#define ORC_GUARD_K 64.0f
#define ORC_MAX_POLY 8

static float guard_dist(const cvert *v, int plane) {
    switch (plane) {
    case 0: return fmaf(ORC_GUARD_K, v->w, -v->x);  /* x <= 64 w */
    case 1: return fmaf(ORC_GUARD_K, v->w, v->x);   /* x >= -64 w */
    case 2: return fmaf(ORC_GUARD_K, v->w, -v->y);
    default: return fmaf(ORC_GUARD_K, v->w, v->y);
    }
}

static cvert plane_lerp(cvert in, cvert out, float din, float dout) { /* din >= 0 > dout */
    float t = din / (din - dout);
    cvert r;
    r.x = fmaf(t, out.x - in.x, in.x);
    r.y = fmaf(t, out.y - in.y, in.y);
    r.z = fmaf(t, out.z - in.z, in.z);
    r.w = fmaf(t, out.w - in.w, in.w);
    r.u = fmaf(t, out.u - in.u, in.u);
    r.v = fmaf(t, out.v - in.v, in.v);
    return r;
}

static int clip_poly_plane(const cvert *in, int n, int plane, cvert *out) {
    int m = 0;
    for (int i = 0; i < n; i++) {
        cvert p = in[i], q = in[(i + 1) % n];
        float dp = guard_dist(&p, plane), dq = guard_dist(&q, plane);
        int pin = !(dp < 0.0f), qin = !(dq < 0.0f);
        if (pin && m < ORC_MAX_POLY) out[m++] = p;
        if (pin != qin && m < ORC_MAX_POLY) out[m++] = pin ? plane_lerp(p, q, dp, dq) : plane_lerp(q, p, dq, dp);
    }
    return m;
}

static int projectable(cvert c, uint32_t W, uint32_t H) { return project(c, W, H).ok; }

/* assembles the screen triangles of one clip-space triangle (at most ORC_MAX_POLY - 2); returns how many slots of out[]
 * were used (a slot may hold an invalid set-up: culled or empty) */
static int clip_and_setup(cvert a, cvert b, cvert c, uint32_t W, uint32_t H, int cull, tri_setup out[ORC_MAX_POLY - 2]) {
    for (int i = 0; i < ORC_MAX_POLY - 2; i++) out[i].valid = 0;
    /* trivial frustum reject; the WebGPU clip volume is -w<=x<=w, -w<=y<=w, 0<=z<=w */
    if (a.x < -a.w && b.x < -b.w && c.x < -c.w) return 0;
    if (a.x > a.w && b.x > b.w && c.x > c.w) return 0;
    if (a.y < -a.w && b.y < -b.w && c.y < -c.w) return 0;
    if (a.y > a.w && b.y > b.w && c.y > c.w) return 0;
    if (a.z < 0.0f && b.z < 0.0f && c.z < 0.0f) return 0;
    if (a.z > a.w && b.z > b.w && c.z > c.w) return 0;
    cvert poly[ORC_MAX_POLY], tmp[ORC_MAX_POLY];
    int n = 3;
    poly[0] = a; poly[1] = b; poly[2] = c;
    if (a.z < 0.0f || b.z < 0.0f || c.z < 0.0f) {
        /* near-plane clip (z >= 0), Sutherland-Hodgman over the cycle a,b,c; intersections are always
         * computed from the inside vertex towards the outside one so shared edges stay watertight */
        cvert v[3] = {a, b, c};
        n = 0;
        for (int i = 0; i < 3; i++) {
            cvert p = v[i], q = v[(i + 1) % 3];
            int pin = !(p.z < 0.0f), qin = !(q.z < 0.0f);
            if (pin) poly[n++] = p;
            if (pin != qin) poly[n++] = pin ? clip_lerp(p, q) : clip_lerp(q, p);
        }
        if (n < 3) return 0;
    }
    int all_ok = 1;
    for (int i = 0; i < n; i++) all_ok = all_ok && projectable(poly[i], W, H);
    if (!all_ok) { /* guard-band clip, only for polygons the rasteriser could not take as they are */
        for (int plane = 0; plane < 4; plane++) {
            n = clip_poly_plane(poly, n, plane, tmp);
            memcpy(poly, tmp, sizeof(cvert) * (size_t)n);
            if (n < 3) return 0;
        }
    }
    for (int i = 0; i + 2 < n; i++) setup_tri(poly[0], poly[i + 1], poly[i + 2], W, H, cull, &out[i]);
    return n - 2;
}

/* ------------------------------------------------------------------------------------------
 * fragment stage
 * ---------------------------------------------------------------------------------------- */
static const uint8_t DEBUG_PALETTE[20][3] = {
    /* src/shaders/debug_ids.wgsl:23-44 */
    {215, 62, 103}, {95, 190, 80},  {133, 95, 213},  {180, 184, 53}, {213, 87, 180},
    {72, 138, 55},  {145, 79, 158}, {91, 196, 153},  {206, 78, 55},  {74, 174, 209},
    {225, 133, 58}, {92, 122, 198}, {207, 162, 81},  {188, 144, 216}, {152, 173, 92},
    {161, 71, 103}, {53, 133, 98},  {225, 131, 152}, {111, 111, 40}, {162, 99, 55},
};

enum { BLEND_ALPHA = 0, BLEND_OFF = 1, BLEND_ADD = 2 };

typedef struct {
    int shader;
    int blend;       /* BLEND_*: ALPHA is the reference's pipeline (src/model.rs:243-246) */
    int depth_write; /* src/model.rs:257 */
    int depth_test;  /* LessEqual (src/model.rs:258) or always */
    int cull;        /* CULL_* */
    float const_rgba[4];
    const orc_texture *tex;
} fs_state;

static inline int64_t edge_fn(int32_t Xa, int32_t Ya, int32_t Xb, int32_t Yb, int64_t Px, int64_t Py) {
    return (int64_t)(Yb - Ya) * (Px - Xa) - (int64_t)(Xb - Xa) * (Py - Ya);
}
static inline int edge_topleft(int32_t Xa, int32_t Ya, int32_t Xb, int32_t Yb) {
    int32_t dy = Yb - Ya, dx = Xb - Xa;
    return dy > 0 || (dy == 0 && dx < 0);
}

static void uv_at(const tri_setup *s, int32_t px, int32_t py, float *u, float *v) {
    int64_t Px = (int64_t)px * 256 + 128, Py = (int64_t)py * 256 + 128;
    int64_t E1 = edge_fn(s->X[2], s->Y[2], s->X[0], s->Y[0], Px, Py);
    int64_t E2 = edge_fn(s->X[0], s->Y[0], s->X[1], s->Y[1], Px, Py);
    float b1 = (float)E1 * s->rcpA, b2 = (float)E2 * s->rcpA;
    float iw = fmaf(b2, s->diw2, fmaf(b1, s->diw1, s->iw0));
    float up = fmaf(b2, s->dup2, fmaf(b1, s->dup1, s->up0));
    float vp = fmaf(b2, s->dvp2, fmaf(b1, s->dvp1, s->vp0));
    *u = up / iw;
    *v = vp / iw;
}

static int32_t clamp_texel(float f, uint32_t n) {
    if (!(f >= 0.0f)) f = 0.0f;
    if (f > (float)(n - 1)) f = (float)(n - 1);
    return (int32_t)f;
}
static void texel_f(const orc_texture *t, int32_t x, int32_t y, float out[4]) {
    const uint8_t *p = t->rgba + ((size_t)y * t->w + (size_t)x) * 4;
    for (int c = 0; c < 4; c++) out[c] = unorm8f(p[c]);
}
/* level l of a mip chain: max(1, w >> l) x max(1, h >> l) texels, stored right after level l - 1 */
static void texel_level_f(const orc_texture *t, uint32_t level, int32_t x, int32_t y, float out[4]) {
    size_t off = 0;
    uint32_t lw = t->w, lh = t->h;
    for (uint32_t l = 0; l < level; l++) {
        off += (size_t)lw * lh * 4;
        lw = lw > 1 ? lw >> 1 : 1;
        lh = lh > 1 ? lh >> 1 : 1;
    }
    const uint8_t *p = t->rgba + off + ((size_t)y * lw + (size_t)x) * 4;
    for (int c = 0; c < 4; c++) out[c] = unorm8f(p[c]);
}

/* textureSample with the reference's sampler (src/texture.rs:33-42): clamp-to-edge, mag linear,
 * min nearest, one mip level (src/texture.rs:21).  SPEC.md "sampling". */
static void sample_texture(const orc_texture *t, const tri_setup *s, int32_t px, int32_t py, float out[4]) {
    float u, v;
    uv_at(s, px, py, &u, &v);
    int32_t qx = px & ~1, qy = py & ~1;
    float ua, va, ub, vb;
    uv_at(s, qx, py, &ua, &va);
    uv_at(s, qx + 1, py, &ub, &vb);
    float dudx = ub - ua, dvdx = vb - va;
    uv_at(s, px, qy, &ua, &va);
    uv_at(s, px, qy + 1, &ub, &vb);
    float dudy = ub - ua, dvdy = vb - va;
    float fw = (float)t->w, fh = (float)t->h;
    int linear = (fabsf(dudx) * fw <= 1.0f) && (fabsf(dvdx) * fh <= 1.0f) &&
                 (fabsf(dudy) * fw <= 1.0f) && (fabsf(dvdy) * fh <= 1.0f);
    if (!linear) {
        /* minification: nearest texel of the nearest mip level (src/texture.rs:39-40).  With m the largest of the four
         * products above, level l is taken while m > 2^(l - 1/2), i.e. m*m > 2^(2l - 1); one level (the reference,
         * src/texture.rs:21) never leaves level 0.  NaN compares false: level 0. */
        uint32_t level = 0, nlev = t->levels ? t->levels : 1;
        if (nlev > 1) {
            float m = fmaxf(fmaxf(fabsf(dudx) * fw, fabsf(dvdx) * fh), fmaxf(fabsf(dudy) * fw, fabsf(dvdy) * fh));
            float m2 = m * m, thr = 2.0f;
            while (level + 1 < nlev && m2 > thr) {
                level++;
                thr *= 4.0f;
            }
        }
        uint32_t lw = t->w, lh = t->h;
        for (uint32_t l = 0; l < level; l++) {
            lw = lw > 1 ? lw >> 1 : 1;
            lh = lh > 1 ? lh >> 1 : 1;
        }
        int32_t tx = clamp_texel(floorf(u * (float)lw), lw), ty = clamp_texel(floorf(v * (float)lh), lh);
        texel_level_f(t, level, tx, ty, out);
        return;
    }
    float x = u * fw - 0.5f, y = v * fh - 0.5f;
    float x0 = floorf(x), y0 = floorf(y);
    float fx = x - x0, fy = y - y0;
    int32_t ix0 = clamp_texel(x0, t->w), ix1 = clamp_texel(x0 + 1.0f, t->w);
    int32_t iy0 = clamp_texel(y0, t->h), iy1 = clamp_texel(y0 + 1.0f, t->h);
    float c00[4], c10[4], c01[4], c11[4];
    texel_f(t, ix0, iy0, c00);
    texel_f(t, ix1, iy0, c10);
    texel_f(t, ix0, iy1, c01);
    texel_f(t, ix1, iy1, c11);
    for (int c = 0; c < 4; c++) {
        float top = fmaf(fx, c10[c] - c00[c], c00[c]);
        float bot = fmaf(fx, c11[c] - c01[c], c01[c]);
        out[c] = fmaf(fy, bot - top, top);
    }
}

/* rasterises rows [y_lo, y_hi) of one set-up triangle; returns depth-passing fragments */
static uint64_t raster_tri(orc_frame *f, const tri_setup *s, const fs_state *fs, int32_t y_lo, int32_t y_hi) {
    int32_t xmin = s->X[0], xmax = s->X[0], ymin = s->Y[0], ymax = s->Y[0];
    for (int i = 1; i < 3; i++) {
        if (s->X[i] < xmin) xmin = s->X[i];
        if (s->X[i] > xmax) xmax = s->X[i];
        if (s->Y[i] < ymin) ymin = s->Y[i];
        if (s->Y[i] > ymax) ymax = s->Y[i];
    }
    int32_t px0 = (xmin + 127) >> 8, px1 = (xmax - 128) >> 8;
    int32_t py0 = (ymin + 127) >> 8, py1 = (ymax - 128) >> 8;
    if (px0 < 0) px0 = 0;
    if (px1 > (int32_t)f->w - 1) px1 = (int32_t)f->w - 1;
    if (py0 < y_lo) py0 = y_lo;
    if (py1 > y_hi - 1) py1 = y_hi - 1;
    int tl0 = edge_topleft(s->X[1], s->Y[1], s->X[2], s->Y[2]);
    int tl1 = edge_topleft(s->X[2], s->Y[2], s->X[0], s->Y[0]);
    int tl2 = edge_topleft(s->X[0], s->Y[0], s->X[1], s->Y[1]);
    uint64_t frags = 0;
    for (int32_t py = py0; py <= py1; py++)
        for (int32_t px = px0; px <= px1; px++) {
            int64_t Px = (int64_t)px * 256 + 128, Py = (int64_t)py * 256 + 128;
            int64_t E0 = edge_fn(s->X[1], s->Y[1], s->X[2], s->Y[2], Px, Py);
            int64_t E1 = edge_fn(s->X[2], s->Y[2], s->X[0], s->Y[0], Px, Py);
            int64_t E2 = edge_fn(s->X[0], s->Y[0], s->X[1], s->Y[1], Px, Py);
            if (!((E0 > 0 || (E0 == 0 && tl0)) && (E1 > 0 || (E1 == 0 && tl1)) &&
                  (E2 > 0 || (E2 == 0 && tl2))))
                continue;
            float b1 = (float)E1 * s->rcpA, b2 = (float)E2 * s->rcpA;
            float z = fmaf(b2, s->dz2, fmaf(b1, s->dz1, s->z0));
            if (!(z >= 0.0f && z <= 1.0f)) continue; /* near/far clip, unclipped_depth off */
            size_t pi = (size_t)py * f->w + (size_t)px;
            if (fs->depth_test && !(z <= f->depth[pi])) continue; /* LessEqual, src/model.rs:258 */
            if (fs->depth_write) f->depth[pi] = z;                /* depth_write_enabled, src/model.rs:257 */
            frags++;
            float src[4];
            if (fs->shader == SH_TEXTURED)
                sample_texture(fs->tex, s, px, py, src);
            else
                memcpy(src, fs->const_rgba, sizeof src);
            uint8_t *dst = f->color + pi * 4;
            if (fs->blend == BLEND_ALPHA) {
                /* SrcAlpha / OneMinusSrcAlpha colour, One / Zero alpha: src/model.rs:243-246 */
                float a = src[3], ia = 1.0f - a;
                for (int c = 0; c < 3; c++) {
                    float d = unorm8f(dst[c]);
                    float t = d * ia;
                    dst[c] = quant8(fmaf(src[c], a, t));
                }
                dst[3] = quant8(src[3]);
            } else if (fs->blend == BLEND_ADD) {
                /* additive (material state, SPEC 10): SrcAlpha / One colour, One / Zero alpha */
                float a = src[3];
                for (int c = 0; c < 3; c++) dst[c] = quant8(fmaf(src[c], a, unorm8f(dst[c])));
                dst[3] = quant8(src[3]);
            } else {
                for (int c = 0; c < 4; c++) dst[c] = quant8(src[c]);
            }
        }
    return frags;
}

/* ------------------------------------------------------------------------------------------
 * primitive assembly + the draw loop (src/model.rs:317-362)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int64_t i0, i1, i2; /* index-buffer positions, already in winding order; -1 = no triangle */
} tri_pos;

static int draw_primitive(orc_frame *f, const vs_ctx *vc, const fs_state *fs, uint32_t topology,
                          const uint16_t *index_buf, size_t index_total, int strip_restart,
                          int nthreads) {
    uint32_t index_ofs = orc_prim_index_ofs(vc->prim), index_num = orc_prim_index_num(vc->prim);
    uint32_t index_base = orc_prim_index_base(vc->prim);
    if ((size_t)index_ofs + index_num > index_total) return ORC_E_INVALID;
    if (index_num == 0) return ORC_OK;
    const uint16_t *ib = index_buf + index_ofs;
    /* one slot per index position that completes a triangle, two set-ups per slot */
    tri_pos *tp = (tri_pos *)malloc(sizeof(tri_pos) * index_num);
    uint64_t ntri = 0;
    if (topology == 4) { /* TriangleStrip = 4, src/rmodel.rs:121-123; restart 0xFFFF src/model.rs:251 */
        uint32_t q = 0;  /* indices since the last restart */
        for (uint32_t i = 0; i < index_num; i++) {
            tp[i].i0 = -1;
            if (strip_restart && ib[i] == 0xFFFF) {
                q = 0;
                continue;
            }
            q++;
            if (q >= 3) {
                uint32_t t = q - 3; /* triangle number within the strip */
                tp[i].i0 = i - 2;
                tp[i].i1 = (t & 1) ? i : i - 1;
                tp[i].i2 = (t & 1) ? i - 1 : i;
                ntri++;
            }
        }
    } else if (topology == 3) { /* TriangleList (build extension; the overlay cube) */
        for (uint32_t i = 0; i < index_num; i++) {
            tp[i].i0 = -1;
            if (i % 3 == 2) {
                tp[i].i0 = i - 2;
                tp[i].i1 = i - 1;
                tp[i].i2 = i;
                ntri++;
            }
        }
    } else {
        free(tp);
        return ORC_E_UNSUPPORTED; /* src/rmodel.rs:215 from_repr().unwrap() */
    }
    f->tris_in += ntri;
    /* two set-up slots per index position (near clipping yields at most two triangles); the rare position whose
     * polygon went through the guard-band clip and fanned into more parks the rest in `ovf`, and only then is the array
     * rebuilt with ORC_MAX_POLY - 2 slots per position */
    size_t SPT = 2;
    tri_setup *su = (tri_setup *)malloc(sizeof(tri_setup) * SPT * (size_t)index_num);
    typedef struct { int64_t pos; int k; tri_setup s; } ovf_t;
    ovf_t *ovf = NULL;
    size_t novf = 0, capovf = 0;
    uint64_t nsetup = 0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) reduction(+ : nsetup) schedule(static)
#endif
    for (int64_t i = 0; i < (int64_t)index_num; i++) {
        su[2 * i].valid = su[2 * i + 1].valid = 0;
        if (tp[i].i0 < 0) continue;
        uint32_t vi[3] = {(uint32_t)ib[tp[i].i0] + index_base, (uint32_t)ib[tp[i].i1] + index_base,
                          (uint32_t)ib[tp[i].i2] + index_base};
        /* a vertex outside the bound slice (src/model.rs:337-342): triangle dropped (SPEC.md) */
        if (vi[0] >= vc->vnum || vi[1] >= vc->vnum || vi[2] >= vc->vnum) continue;
        cvert a = shade_vertex(vc, vi[0]), b = shade_vertex(vc, vi[1]), c = shade_vertex(vc, vi[2]);
        tri_setup loc[ORC_MAX_POLY - 2];
        clip_and_setup(a, b, c, f->w, f->h, fs->cull, loc);
        su[2 * i] = loc[0];
        su[2 * i + 1] = loc[1];
        for (int k = 0; k < ORC_MAX_POLY - 2; k++) nsetup += (uint64_t)loc[k].valid;
        for (int k = 2; k < ORC_MAX_POLY - 2; k++)
            if (loc[k].valid) {
#ifdef _OPENMP
#pragma omp critical(orc_ovf)
#endif
                {
                    if (novf == capovf) {
                        capovf = capovf ? capovf * 2 : 64;
                        ovf = (ovf_t *)realloc(ovf, capovf * sizeof(ovf_t));
                    }
                    ovf[novf].pos = i;
                    ovf[novf].k = k;
                    ovf[novf].s = loc[k];
                    novf++;
                }
            }
    }
    if (novf) {
        const size_t S6 = ORC_MAX_POLY - 2;
        tri_setup *su6 = (tri_setup *)calloc(S6 * (size_t)index_num, sizeof(tri_setup));
        for (size_t i = 0; i < (size_t)index_num; i++) {
            su6[S6 * i] = su[2 * i];
            su6[S6 * i + 1] = su[2 * i + 1];
        }
        for (size_t j = 0; j < novf; j++) su6[S6 * (size_t)ovf[j].pos + (size_t)ovf[j].k] = ovf[j].s;
        free(su);
        su = su6;
        SPT = S6;
    }
    free(ovf);
    f->tris_setup += nsetup;
    uint64_t frags = 0;
    if (nthreads <= 1) {
        for (size_t i = 0; i < SPT * (size_t)index_num; i++)
            if (su[i].valid) frags += raster_tri(f, &su[i], fs, 0, (int32_t)f->h);
    } else {
#ifdef _OPENMP
        /* 8-row bands: every pixel row is owned by exactly one band, each band walks its own list of
         * set-ups in submission order (lists built serially by a counting sort), so results equal
         * the scalar path whatever the thread count */
        const int32_t BH = 8, nband = ((int32_t)f->h + BH - 1) / BH;
        uint32_t *bstart = (uint32_t *)calloc((size_t)nband + 1, sizeof(uint32_t));
        for (size_t i = 0; i < SPT * (size_t)index_num; i++) {
            if (!su[i].valid) continue;
            int32_t ymin = su[i].Y[0], ymax = su[i].Y[0];
            for (int k = 1; k < 3; k++) {
                if (su[i].Y[k] < ymin) ymin = su[i].Y[k];
                if (su[i].Y[k] > ymax) ymax = su[i].Y[k];
            }
            int32_t b0 = ((ymin + 127) >> 8), b1 = ((ymax - 128) >> 8);
            if (b0 < 0) b0 = 0;
            if (b1 > (int32_t)f->h - 1) b1 = (int32_t)f->h - 1;
            for (int32_t b = b0 / BH; b <= b1 / BH; b++) bstart[b + 1]++;
        }
        for (int32_t b = 0; b < nband; b++) bstart[b + 1] += bstart[b];
        uint32_t *blist = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)bstart[nband] + 1));
        uint32_t *bfill = (uint32_t *)calloc((size_t)nband, sizeof(uint32_t));
        for (size_t i = 0; i < SPT * (size_t)index_num; i++) {
            if (!su[i].valid) continue;
            int32_t ymin = su[i].Y[0], ymax = su[i].Y[0];
            for (int k = 1; k < 3; k++) {
                if (su[i].Y[k] < ymin) ymin = su[i].Y[k];
                if (su[i].Y[k] > ymax) ymax = su[i].Y[k];
            }
            int32_t b0 = ((ymin + 127) >> 8), b1 = ((ymax - 128) >> 8);
            if (b0 < 0) b0 = 0;
            if (b1 > (int32_t)f->h - 1) b1 = (int32_t)f->h - 1;
            for (int32_t b = b0 / BH; b <= b1 / BH; b++) blist[bstart[b] + bfill[b]++] = (uint32_t)i;
        }
#pragma omp parallel for num_threads(nthreads) reduction(+ : frags) schedule(dynamic, 1)
        for (int32_t b = 0; b < nband; b++) {
            int32_t lo = b * BH, hi = lo + BH > (int32_t)f->h ? (int32_t)f->h : lo + BH;
            for (uint32_t k = bstart[b]; k < bstart[b + 1]; k++) frags += raster_tri(f, &su[blist[k]], fs, lo, hi);
        }
        free(bfill);
        free(blist);
        free(bstart);
#else
        for (size_t i = 0; i < SPT * (size_t)index_num; i++)
            if (su[i].valid) frags += raster_tri(f, &su[i], fs, 0, (int32_t)f->h);
#endif
    }
    f->frags += frags;
    free(su);
    free(tp);
    return ORC_OK;
}

int orc_draw(orc_frame *f, const orc_model *m, const float M[16], const float *palette, size_t npal,
             int32_t tex_override, int nthreads) {
    if (!f || !m) return ORC_E_INVALID;
    /* validate everything first so a failing draw leaves the frame untouched */
    for (size_t p = 0; p < m->nprims; p++) {
        const uint8_t *pr = m->prims + p * 0x38;
        if (orc_prim_parts_no(pr) >= m->nparts) return ORC_E_INVALID; /* src/model.rs:318 would panic */
        vs_ctx vc;
        int rc = vs_ctx_init(&vc, m, p, M, palette, npal);
        if (rc) return rc;
        uint32_t topo = orc_prim_topology(pr);
        if (topo != 4 && topo != 3) return ORC_E_UNSUPPORTED;
        int32_t tex = m->prim_to_texture ? m->prim_to_texture[p] : -1;
        if (tex >= 0 && tex_override >= 0) tex = tex_override;
        if (tex >= (int32_t)m->ntextures) return ORC_E_INVALID;
        if ((size_t)orc_prim_index_ofs(pr) + orc_prim_index_num(pr) > m->index_num) return ORC_E_INVALID;
    }
    for (size_t p = 0; p < m->nprims; p++) {
        const uint8_t *pr = m->prims + p * 0x38;
        if (!m->parts_disp[orc_prim_parts_no(pr)]) continue; /* src/model.rs:318-320 */
        vs_ctx vc;
        vs_ctx_init(&vc, m, p, M, palette, npal);
        int32_t tex = m->prim_to_texture ? m->prim_to_texture[p] : -1;
        if (tex >= 0 && tex_override >= 0) tex = tex_override;
        fs_state fs;
        memset(&fs, 0, sizeof fs);
        fs.blend = BLEND_ALPHA; /* src/model.rs:243-246 */
        fs.depth_write = fs.depth_test = 1;
        fs.cull = CULL_BACK;
        if (m->prim_state) { /* material state per primitive (SPEC 10): blend, depth write, depth test, cull */
            const uint8_t *st = m->prim_state + 4 * p;
            if (st[0] > 2 || st[3] > 2) return ORC_E_INVALID;
            fs.blend = st[0];
            fs.depth_write = st[1] != 0;
            fs.depth_test = st[2] != 0;
            fs.cull = st[3];
        }
        /* pipeline choice, src/model.rs:212-216: textured && attributes.len() != 1 */
        if (tex >= 0 && vc.li.has_uv) {
            fs.shader = SH_TEXTURED;
            fs.tex = &m->textures[tex];
        } else {
            fs.shader = SH_DEBUG; /* src/shaders/debug_ids.wgsl:46 */
            uint32_t id = m->prim_debug_id ? m->prim_debug_id[p] : 0;
            for (int c = 0; c < 3; c++) fs.const_rgba[c] = (float)DEBUG_PALETTE[id % 20][c] / 255.0f;
            fs.const_rgba[3] = 1.0f;
        }
        int rc = draw_primitive(f, &vc, &fs, orc_prim_topology(pr), m->index_buf, m->index_num, 1, nthreads);
        if (rc) return rc;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * debug overlay cubes -- src/debug_overlay.rs:10-35 (geometry), :119-188 (state), :202-221 (draw)
 * ---------------------------------------------------------------------------------------- */
int orc_draw_overlay_cubes(orc_frame *f, const float cam[16], const float *inst, size_t n) {
    static const float verts[24] = {1, 1, -1, 1, -1, -1, 1, 1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1};
    static const uint16_t idx[36] = {4, 2, 0, 2, 7, 3, 6, 5, 7, 1, 7, 5, 0, 3, 1, 4, 1, 5,
                                     4, 6, 2, 2, 6, 7, 6, 4, 5, 1, 3, 7, 0, 2, 3, 4, 0, 1};
    uint8_t prim[0x38];
    memset(prim, 0, sizeof prim);
    uint32_t w0 = 8u << 16, w2 = (12u << 16) | (3u << 24), inum = 36;
    memcpy(prim + 0, &w0, 4);
    memcpy(prim + 8, &w2, 4);
    memcpy(prim + 28, &inum, 4);
    orc_layout lay;
    memset(&lay, 0, sizeof lay);
    lay.num = 1;
    lay.el[0].semantic = ORC_SEM_POSITION;
    lay.el[0].format = ORC_IEF_F32;
    lay.el[0].count = 3;
    orc_model m;
    memset(&m, 0, sizeof m);
    m.vertex_buf = (const uint8_t *)verts;
    m.vertex_len = sizeof verts;
    m.index_buf = idx;
    m.index_num = 36;
    m.prims = prim;
    m.nprims = 1;
    m.layouts = &lay;
    fs_state fs;
    memset(&fs, 0, sizeof fs);
    fs.shader = SH_CONST;
    fs.blend = BLEND_OFF; /* blend: None, src/debug_overlay.rs:174 */
    fs.depth_write = fs.depth_test = 1;
    fs.cull = CULL_BACK;
    fs.const_rgba[0] = 0.1f;
    fs.const_rgba[1] = 0.2f;
    fs.const_rgba[2] = 0.3f;
    fs.const_rgba[3] = 1.0f; /* src/shaders/debug_overlay.wgsl:30 */
    for (size_t i = 0; i < n; i++) {
        float M[16];
        orc_mat4_mul(cam, inst + 16 * i, M); /* camera_transform * position_matrix, debug_overlay.wgsl:24-26 */
        vs_ctx vc;
        int rc = vs_ctx_init(&vc, &m, 0, M, NULL, 0);
        if (rc) return rc;
        rc = draw_primitive(f, &vc, &fs, 3, idx, 36, 0, 1);
        if (rc) return rc;
    }
    return ORC_OK;
}
