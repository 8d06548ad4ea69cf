/* mtr_oracle.h -- CPU ORACLE for the rModel draw path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the normative CPU restatement of the draw path that the reference (ReplayCoding/
 * mt-renderer) configures but never computes itself: the reference records wgpu commands
 * (src/model.rs:299-363) and a GPU driver does all transform / raster / shade arithmetic.
 * Nothing under oracle/ may be imported, linked or executed by the product (libmtr.so,
 * mt_renderer_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * PARITY STATUS: "parity unpinned" against the reference's real pixels -- the reference holds no
 * image / depth / vertex golden of any kind.  What IS pinned by reference data:
 * packed-struct sizes and bit-field accessors (src/rmodel.rs:135-225,486-494), the crc32 KATs
 * (src/util/crc.rs:52-63, src/dti.txt), the cube geometry (src/debug_overlay.rs:10-35), the
 * 20-colour palette (src/shaders/debug_ids.wgsl:23-44) and the pipeline state
 * (src/model.rs:232-264).  BC7 decoding is additionally cross-checked against an independent
 * decoder (Pillow) in tests/.
 *
 * The numeric rules are written out in SPEC.md; every function below cites the reference
 * file:line that selects the behaviour it restates.
 */
#ifndef MTR_ORACLE_H
#define MTR_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* semantics the draw path binds (src/rshader2.rs:503-507) + the build's LBS extension */
enum { ORC_SEM_POSITION = 0, ORC_SEM_TEXCOORD = 1, ORC_SEM_JOINT = 2, ORC_SEM_WEIGHT = 3 };

/* InputElementFormat, numeric values of src/rshader2.rs:73-90 */
enum {
    ORC_IEF_F32 = 1, ORC_IEF_F16 = 2, ORC_IEF_S16 = 3, ORC_IEF_U16 = 4, ORC_IEF_S16N = 5,
    ORC_IEF_U16N = 6, ORC_IEF_S8 = 7, ORC_IEF_U8 = 8, ORC_IEF_S8N = 9, ORC_IEF_U8N = 10,
    ORC_IEF_SCMP3N = 11, ORC_IEF_UCMP3N = 12, ORC_IEF_U8NL = 13, ORC_IEF_COLOR4N = 14
};

enum { ORC_OK = 0, ORC_E_INVALID = 1, ORC_E_UNSUPPORTED = 2 };

/* rTexture format ids, src/rtexture.rs:152-161 */
enum { ORC_TEX_RGBA8 = 7, ORC_TEX_BC1 = 19, ORC_TEX_BC7 = 42, ORC_TEX_BC7_ALT = 54 };

typedef struct {
    uint8_t semantic, format, count, pad0;
    uint16_t offset, pad1;
} orc_element;

typedef struct {
    uint32_t num;
    orc_element el[8];
} orc_layout;

typedef struct {
    uint32_t w, h;
    const uint8_t *rgba; /* decoded RGBA8, row-major, w*h*4 bytes; further mip levels follow, each max(1, w>>l) x max(1, h>>l) */
    uint32_t levels;     /* mip levels present (0 and 1 both mean level 0 only: the reference, src/texture.rs:21) */
} orc_texture;

typedef struct {
    const uint8_t *vertex_buf;
    size_t vertex_len;
    const uint16_t *index_buf;
    size_t index_num;
    const uint8_t *prims; /* nprims * 0x38 bytes, PrimitiveInfo verbatim (src/rmodel.rs:135-171) */
    size_t nprims;
    const orc_layout *layouts;       /* one per primitive */
    const int32_t *prim_to_texture;  /* per primitive: index into textures[] or -1 */
    const uint32_t *prim_debug_id;   /* per primitive (src/model.rs:140-141) */
    const uint8_t *parts_disp;       /* src/model.rs:295,318 */
    size_t nparts;
    const orc_texture *textures;
    size_t ntextures;
    /* material state per primitive, 4 bytes each {blend 0 alpha / 1 off / 2 additive, depth write, depth test, cull 0 back /
     * 1 none / 2 front}, or NULL: the reference's pipeline state (src/model.rs:240-262).  Build extension, SPEC 10. */
    const uint8_t *prim_state;
} orc_model;

typedef struct orc_frame orc_frame;

/* ---- utilities pinned by reference KATs ---- */
uint32_t orc_crc32(const uint8_t *bytes, size_t len, uint32_t init); /* src/util/crc.rs:36-50 */

/* PrimitiveInfo accessors, src/rmodel.rs:173-225.  p points at 0x38 bytes. */
uint32_t orc_prim_vertex_stride(const uint8_t *p);
uint32_t orc_prim_parts_no(const uint8_t *p);
uint32_t orc_prim_material_no(const uint8_t *p);
uint32_t orc_prim_weight_num(const uint8_t *p);
uint32_t orc_prim_inputlayout(const uint8_t *p);
uint32_t orc_prim_vertex_base(const uint8_t *p);
uint32_t orc_prim_index_ofs(const uint8_t *p);
uint32_t orc_prim_index_base(const uint8_t *p);
uint32_t orc_prim_index_num(const uint8_t *p);
uint32_t orc_prim_topology(const uint8_t *p);
uint32_t orc_prim_vertex_num(const uint8_t *p);
uint32_t orc_prim_boundary_num(const uint8_t *p);

/* ---- texture decode (src/rtexture.rs:152-161 selects the formats; hardware decodes them) ---- */
void orc_bc1_decode_block(const uint8_t blk[8], uint8_t out_rgba[64]);
void orc_bc7_decode_block(const uint8_t blk[16], uint8_t out_rgba[64]);
/* decodes level 0 of a w*h texture to RGBA8; returns ORC_E_UNSUPPORTED for other format ids
 * (the reference todo!()s, src/rtexture.rs:159) and ORC_E_INVALID if data is too short. */
int orc_texture_decode(uint32_t fmt, uint32_t w, uint32_t h, const uint8_t *data, size_t len,
                       uint8_t *out_rgba);

/* ---- vertex stage only (for unit parity of the HIP vertex kernel) ----
 * For primitive `prim` computes, for every vertex v in [0, vertex_num): clip = M * (skin(pos),1)
 * and the decoded texcoord.  out_clip: vertex_num*4 floats, out_uv: vertex_num*2 floats. */
int orc_vertex_stage(const orc_model *m, size_t prim, const float M[16], const float *palette,
                     size_t npal, float *out_clip, float *out_uv);

/* M = A * B for column-major 4x4 with the normative fma chain (SPEC.md "matrix compose") */
void orc_mat4_mul(const float A[16], const float B[16], float out[16]);

/* ---- frame ---- */
orc_frame *orc_frame_create(uint32_t w, uint32_t h, const float clear_rgba[4], float clear_depth);
void orc_frame_destroy(orc_frame *f);
const uint8_t *orc_frame_color(const orc_frame *f); /* RGBA8, w*h*4 */
const float *orc_frame_depth(const orc_frame *f);   /* f32, w*h */
uint64_t orc_frame_tris_in(const orc_frame *f);     /* input triangles counted per SURVEY 8(d) */
uint64_t orc_frame_tris_setup(const orc_frame *f);  /* triangles that survived clip/cull/bbox */
uint64_t orc_frame_frags(const orc_frame *f);       /* fragments that passed the depth test */

/* Draws every visible primitive of the model with transform M (= view_proj, or view_proj*model
 * composed by the caller with orc_mat4_mul).  palette: npal column-major 4x4 matrices or NULL.
 * tex_override: texture index used instead of prim_to_texture for textured primitives, or -1.
 * nthreads: 1 = the normative scalar path; >1 = row-band parallel variant of the same
 * arithmetic (used only as the timed CPU baseline; results are identical by construction). */
int orc_draw(orc_frame *f, const orc_model *m, const float M[16], const float *palette, size_t npal,
             int32_t tex_override, int nthreads);

/* instanced cubes of the debug overlay (src/debug_overlay.rs:119-221): TriangleList, no blend,
 * constant colour (0.1,0.2,0.3,1) (src/shaders/debug_overlay.wgsl:29-31) */
int orc_draw_overlay_cubes(orc_frame *f, const float cam[16], const float *inst_mats, size_t n);

#ifdef __cplusplus
}
#endif
#endif
