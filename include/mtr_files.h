/* mtr_files.h -- MT Framework resource-file readers behind the C ABI.
 *
 * The reference parses these files in Rust (and stays free to: mtr.h takes already-parsed buffers).  These entry
 * points let a C / C++ / Python host feed real assets to the HIP draw path without the Rust side, with the same
 * field meaning, the same struct sizes and the same acceptance rules; where the reference panics / unwraps on a
 * malformed file these return MTR_E_INVALID and a message (mtr_files_last_error).  Host-only code: no GPU needed,
 * except mtr_model_create_from_files which ends in mtr_model_create.
 *
 *   rModel    (.mod)  src/rmodel.rs:84-171 (structs), :307-455 (ModelFile::new), sizes tested :487-494
 *   rTexture  (.tex)  src/rtexture.rs:24-78 (header bit fields), :89-138 (TextureFile::new)
 *   rShader2  (.mfx)  src/rshader2.rs:14-66 (structs), :298-486 (Shader2File::new), sizes :573-582
 *   rMaterial (.mrl)  src/rmaterial.rs:12-116 (structs), :179-298 (MaterialFile::new), sizes :317-322
 *   rScheduler(.sdl)  src/rscheduler.rs:36-84 (structs), :88-216 (SchedulerFile::new), size :222
 *   rArchive  (.arc)  src/rarchive.rs:24-43 (structs), :73-176 (ArchiveFile::new / get_resource), sizes :163-169
 *
 * All multi-byte fields are little-endian and unaligned ("repr(C, packed)" in the reference).  "View" structs
 * point INTO the caller's buffer, which must outlive them; their typed pointers (primitives, lmats, imats,
 * index_buf) sit wherever the file put the data and may be UNALIGNED: copy with memcpy before dereferencing on a
 * strict-alignment target (the library itself only ever memcpy's them).  Handles own their memory.
 */
#ifndef MTR_FILES_H
#define MTR_FILES_H

#include "mtr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* message of the last failed mtr_r*_parse / mtr_model_create_from_files call on this thread; never NULL */
const char *mtr_files_last_error(void);

/* on-disk struct sizes, the reference's own size tests as data:
 * kind 0 ModelHdr 0xa0, 1 PrimitiveInfo 0x38, 2 PartsInfo 0x20, 3 BoundaryInfo 0x90, 4 JointInfo 24, 5 MtMatrix 64,
 * 6 TextureHeader 0x10, 7 Shader2Header 0x20, 8 RawShader2Object 0x28, 9 RawShader2InputElement 0x10,
 * 10 RawShader2InputLayout 16, 11 RawShader2Struct 16, 12 RawShader2Variable 0x30, 13 RawShader2CBuffer 24,
 * 14 MaterialHeader 0x28, 15 RawTextureInfo 0x98, 16 RawMaterialInfo 0x48, 17 RawMaterialState 0x18,
 * 18 SchedulerTrack 0x30, 19 SchedulerHeader 0x20, 20 ArchiveHeader 8, 21 RawResourceInfo 0x90.   Unknown kind: 0. */
size_t mtr_file_struct_size(uint32_t kind);

/* ---------------------------------------------------------------- rModel ---- */
typedef struct mtr_rmodel_view {
    uint32_t magic;
    uint16_t version, jnt_num, primitive_num, material_num;
    uint32_t vertex_num, index_num, polygon_num, vertexbuf_size, texture_num, parts_num, boundary_num;
    float bounding_sphere[4];        /* x y z r */
    float bounding_box_min[4], bounding_box_max[4];
    const uint8_t *material_names;   /* material_num x 128 bytes, NUL-terminated (src/rmodel.rs:315-328) */
    const mtr_primitive *primitives; /* primitive_num x 0x38 (unaligned: copy before use on strict targets) */
    const uint8_t *boundary_infos;   /* boundary_num x 0x90, right after the primitive array (src/rmodel.rs:358-359) */
    const uint8_t *joint_infos;      /* jnt_num x 24, or NULL */
    const float *lmats, *imats;      /* jnt_num x 16 f32 each, or NULL (src/rmodel.rs:392-395) */
    const uint8_t *joint_table;      /* 256 bytes, or NULL (the reference substitutes 255s) */
    const uint8_t *parts;            /* parts_num x 0x20 */
    const uint8_t *vertex_buf;       /* vertexbuf_size bytes */
    const uint16_t *index_buf;       /* index_num u16 */
} mtr_rmodel_view;

int32_t mtr_rmodel_parse(const void *data, size_t len, mtr_rmodel_view *out);
/* PrimitiveInfo bit fields (src/rmodel.rs:173-225) */
enum {
    MTR_PRIM_VERTEX_NUM = 0, MTR_PRIM_PARTS_NO = 1, MTR_PRIM_MATERIAL_NO = 2, MTR_PRIM_WEIGHT_NUM = 3,
    MTR_PRIM_VERTEX_STRIDE = 4, MTR_PRIM_TOPOLOGY = 5, MTR_PRIM_VERTEX_OFS = 6, MTR_PRIM_VERTEX_BASE = 7,
    MTR_PRIM_INPUTLAYOUT = 8, MTR_PRIM_INDEX_OFS = 9, MTR_PRIM_INDEX_NUM = 10, MTR_PRIM_INDEX_BASE = 11,
    MTR_PRIM_BOUNDARY_NUM = 12
};
uint32_t mtr_primitive_field(const mtr_primitive *prim, uint32_t field);
/* BoundaryInfo::joint of boundary i (src/rmodel.rs:245-249): the debug id source of Model::new */
int32_t mtr_rmodel_boundary_joint(const mtr_rmodel_view *m, uint32_t i, uint32_t *out);
/* JointInfo of joint i: no / parent / symmetry bytes and the offset vector (src/rmodel.rs:253-275) */
int32_t mtr_rmodel_joint(const mtr_rmodel_view *m, uint32_t i, uint32_t *no, uint32_t *parent, uint32_t *symmetry,
                         float offset[3]);

/* Skin palette of a model's skeleton (row f-3; the reference parses lmats / imats / joint_table and never combines them,
 * src/rmodel.rs:392-413, so this formation rule is the build's -- the usual one for MT Framework data):
 *     world_j = world_parent(j) * local_j      (parent 255, or a joint that is its own parent: world_j = local_j)
 *     palette[j] = world_j * imat_j            (imat_j = inverse of the bind-pose world matrix)
 * j is the joint's INDEX in the file's joint array, which is what the vertex `Joint` bytes hold; joint_table maps a joint
 * NUMBER (JointInfo::no) to that index (mtr_rmodel_joint_index).  local_mats: jnt_num matrices, or NULL for the file's
 * lmats (the bind pose, which must give the identity wherever the file is consistent).  Matrices are the file's 16 floats
 * read as column-major M * v (MtMatrix rows = glam columns); every product is the fma chain of SPEC.md section 4.
 * out_palette: jnt_num * 16 floats for mtr_model_set_palette.  MTR_E_INVALID: no joints, a parent outside the array, a cycle. */
int32_t mtr_rmodel_palette(const mtr_rmodel_view *m, const float *local_mats, float *out_palette, size_t cap_mats);
/* index of the joint numbered `no` (joint_table[no]), or -1 */
int32_t mtr_rmodel_joint_index(const mtr_rmodel_view *m, uint32_t no);

/* -------------------------------------------------------------- rTexture ---- */
typedef struct mtr_rtexture_view {
    uint32_t version, prebias, type, level_count, array_count, format; /* format: MTR_TEX_* when supported */
    uint32_t width, height;      /* already shifted left by prebias (src/rtexture.rs:57-62) */
    uint64_t level0_offset;      /* first entry of the offset table (src/rtexture.rs:126) */
    const uint8_t *data;         /* from level0_offset to the end of the file (src/rtexture.rs:129-130) */
    size_t data_len;
} mtr_rtexture_view;
/* MTR_E_INVALID: magic != "TEX\0" or not a 2-D texture (asserts at src/rtexture.rs:105-106), truncated file */
int32_t mtr_rtexture_parse(const void *data, size_t len, mtr_rtexture_view *out);
/* TextureFile -> Texture::new -> mtr_texture_create in one call; MTR_E_UNSUPPORTED for a format
 * TextureFile::format_wgpu todo!()s */
int32_t mtr_texture_create_from_file(mtr_device *dev, const void *data, size_t len, mtr_texture **out);
/* the same with the file's mip chain (row f-4): the reference keeps every byte from the level-0 offset to the end of the
 * file (src/rtexture.rs:129-130) and uploads level 0 only (src/texture.rs:21); here the offsets table (one u64 per level,
 * src/rtexture.rs:111-126) locates up to max_levels levels of array slice 0, each checked to lie inside the file, and
 * mtr_texture_create_mips samples them (SPEC.md section 7).  max_levels = 1 is mtr_texture_create_from_file. */
int32_t mtr_texture_create_from_file_mips(mtr_device *dev, const void *data, size_t len, uint32_t max_levels, mtr_texture **out);

/* -------------------------------------------------------------- rShader2 ---- */
typedef struct mtr_rshader2 mtr_rshader2;
int32_t mtr_rshader2_parse(const void *data, size_t len, mtr_rshader2 **out);
void mtr_rshader2_destroy(mtr_rshader2 *sh);
uint32_t mtr_rshader2_num_objects(const mtr_rshader2 *sh); /* header.num_objects - 1 (src/rshader2.rs:316-318) */
/* object i: name, object type (ObjectType, src/rshader2.rs:129-152), name hash = crc32(name) & 0xfffff */
int32_t mtr_rshader2_object(const mtr_rshader2 *sh, uint32_t i, const char **name, uint32_t *obj_type,
                            uint32_t *name_hash);
/* Shader2File::get_object_by_handle (src/rshader2.rs:487-492): index of the object whose name hash is
 * (handle >> 12) & 0xfffff, or -1 */
int32_t mtr_rshader2_find(const mtr_rshader2 *sh, uint32_t handle);
/* input layout of object i (must be OT_INPUTLAYOUT = 9): stride, every raw element, and the mtr_layout the draw
 * path consumes.  Element selection follows create_vertex_buffer_elements (src/rshader2.rs:496-571): "Position"
 * and "TexCoord" are bound, SCMP3N elements are skipped; "Joint" / "Weight" map to this build's skinning
 * extension; everything else is ignored.  A bound element whose (format, count) the reference todo!()s makes
 * mtr_model_create fail with MTR_E_UNSUPPORTED later, exactly like a hand-made layout. */
typedef struct mtr_raw_element {
    const char *name;
    uint32_t sindex, format, count, start, offset, instance; /* bit fields at src/rshader2.rs:419-442 */
} mtr_raw_element;
int32_t mtr_rshader2_input_layout(const mtr_rshader2 *sh, uint32_t i, uint32_t *stride, mtr_layout *layout,
                                  mtr_raw_element *raw, uint32_t raw_cap, uint32_t *raw_num);

/* ------------------------------------------------------------- rMaterial ---- */
typedef struct mtr_rmaterial mtr_rmaterial;
/* needs the shader package: state objects are resolved by handle (src/rmaterial.rs:211-298) */
int32_t mtr_rmaterial_parse(const void *data, size_t len, const mtr_rshader2 *sh, mtr_rmaterial **out);
void mtr_rmaterial_destroy(mtr_rmaterial *m);
uint32_t mtr_rmaterial_num_textures(const mtr_rmaterial *m);
const char *mtr_rmaterial_texture_path(const mtr_rmaterial *m, uint32_t i); /* NULL if out of range */
uint32_t mtr_rmaterial_num_materials(const mtr_rmaterial *m);
typedef struct mtr_material_info {
    uint32_t name_hash;        /* crc32(material name), full 32 bits (src/rmaterial.rs:304-311) */
    uint32_t dti_hash;         /* material class */
    int32_t albedo_texture;    /* index into the texture list bound to "tAlbedoMap", or -1 (src/rmaterial.rs:276-279) */
    uint32_t bsstate, dsstate, rsstate; /* blend / depth-stencil / rasterizer state object handles (logged only) */
    uint32_t state_num;
    float blend_factor[4];
} mtr_material_info;
int32_t mtr_rmaterial_info(const mtr_rmaterial *m, uint32_t i, mtr_material_info *out);
/* MaterialFile::material_by_name (src/rmaterial.rs:304-311): index or -1 */
int32_t mtr_rmaterial_find(const mtr_rmaterial *m, const char *name);

/* Material state by name (row f-4).  The reference resolves a material's three state handles to shader-package objects
 * and logs their NAMES (src/rmaterial.rs:211-230); it neither parses the objects' contents (OT_BLEND / OT_DEPTHSTENCIL
 * / OT_RASTERIZER fall through `_ => None`, src/rshader2.rs:451) nor applies them.  MT Framework names its state objects
 * by what they do, and this build maps the names it knows (a convention, not something the reference pins):
 *   blend          "BSSolid" and names without "Blend"/"Add"/"Alpha" -> MTR_BLEND_OFF; names containing "Add" -> MTR_BLEND_ADD;
 *                  other names containing "Blend" or "Alpha" -> MTR_BLEND_ALPHA
 *   depth-stencil  depth_test = the name contains "ZTest"; depth_write = it contains "ZTestWrite" or "ZWrite"
 *   rasterizer     names ending in "CN" or containing "CullNone" / "TwoSide" -> MTR_CULL_NONE; ending in "CF" or containing
 *                  "CullFront" -> MTR_CULL_FRONT; otherwise MTR_CULL_BACK
 * A NULL or empty name leaves the reference's state for that part.  Returns how many of the three names were recognised
 * by an explicit rule (0..3) -- an unknown name maps to the default and is not counted. */
int32_t mtr_state_from_names(const char *blend_name, const char *depth_stencil_name, const char *rasterizer_name, mtr_prim_state *out);
/* per primitive of `model`: material name -> rMaterial entry -> its three state objects' names -> mtr_state_from_names.
 * states: primitive_num entries (for mtr_model_set_prim_states); a primitive whose material is not in `mat` keeps the
 * reference state. */
int32_t mtr_model_states_from_files(const mtr_rmodel_view *model, const mtr_rshader2 *sh, const mtr_rmaterial *mat,
                                    mtr_prim_state *states, size_t nstates);

/* ------------------------------------------------------------ rScheduler ---- */
typedef struct mtr_rscheduler mtr_rscheduler;
int32_t mtr_rscheduler_parse(const void *data, size_t len, mtr_rscheduler **out);
void mtr_rscheduler_destroy(mtr_rscheduler *s);
uint32_t mtr_rscheduler_num_tracks(const mtr_rscheduler *s);
typedef struct mtr_track_info {
    uint32_t track_type; /* SchedulerTrackType, src/rscheduler.rs:15-33 */
    uint32_t prop_type;  /* dti::PropType, src/dti.rs:6-70 */
    uint32_t key_num;
    uint32_t parent;     /* field_4 */
    uint32_t dti_or_prop;/* field_10: class hash for UNIT / SYSTEM tracks */
    const char *name;    /* track / property name */
} mtr_track_info;
int32_t mtr_rscheduler_track(const mtr_rscheduler *s, uint32_t i, mtr_track_info *out);
/* key k of track i: frame number (24 bits), mode (8 bits), and the value as the reference decodes it
 * (src/rscheduler.rs:146-205): BOOL -> u8, INT -> u32, FLOAT -> f32 (bit pattern in value_bits),
 * RESOURCE -> class hash in value_bits and the path in *resource (NULL for a null reference).
 * VECTOR / MATRIX keys: mtr_rscheduler_key_floats.  Other key types (INT64, STRING, ...: todo!() in the reference) return
 * MTR_E_UNSUPPORTED. */
int32_t mtr_rscheduler_key(const mtr_rscheduler *s, uint32_t track, uint32_t k, uint32_t *frame, uint32_t *mode,
                           uint64_t *value_bits, const char **resource);
/* value of a BOOL / INT / FLOAT track at `frame`: the key with the greatest frame number <= frame (step hold; the
 * reference never evaluates tracks, so this rule is this build's and is documented as such).  MTR_E_INVALID if the
 * track has no key at or before `frame`. */
int32_t mtr_rscheduler_eval(const mtr_rscheduler *s, uint32_t track, uint32_t frame, uint64_t *value_bits);
/* VECTOR (4 f32 per key) and MATRIX (16 f32 per key) tracks -- `todo!()` in the reference (src/rscheduler.rs:207), decoded
 * here as MtVector4 / MtMatrix arrays.  key_floats: the floats of key k (out: 4 or 16, *n says which); eval_floats: the
 * key with the greatest frame number <= frame (step hold), as mtr_rscheduler_eval.  FLOAT tracks answer with n = 1. */
int32_t mtr_rscheduler_key_floats(const mtr_rscheduler *s, uint32_t track, uint32_t k, float out[16], uint32_t *n);
int32_t mtr_rscheduler_eval_floats(const mtr_rscheduler *s, uint32_t track, uint32_t frame, float out[16], uint32_t *n);
/* first track with this name, or -1 */
int32_t mtr_rscheduler_find_track(const mtr_rscheduler *s, const char *name);
/* Binding tracks to what the draw path animates (row f-2).  The reference never evaluates a track, so WHICH track drives
 * WHAT is the host's statement: one binding = (track, target, index).  mtr_rscheduler_apply evaluates every binding at
 * `frame` (step hold) into the caller's arrays, ready for mtr_model_set_parts_disp and mtr_batch_create / _update:
 *   MTR_SDL_PARTS_DISP          BOOL / INT track   -> parts_disp[index] = value != 0
 *   MTR_SDL_INSTANCE_MATRIX     MATRIX track       -> model_mats[index] = the 16 floats
 *   MTR_SDL_INSTANCE_TRANSLATION VECTOR track      -> floats 12..14 of model_mats[index] = x, y, z
 *   MTR_SDL_INSTANCE_TRANSLATE_X / _Y / _Z  FLOAT track -> float 12 / 13 / 14 of model_mats[index]
 * A binding whose track has no key at or before `frame` leaves its target untouched.  MTR_E_INVALID: track / index out of
 * range, or a track of the wrong type for its target. */
enum { MTR_SDL_PARTS_DISP = 0, MTR_SDL_INSTANCE_MATRIX = 1, MTR_SDL_INSTANCE_TRANSLATION = 2, MTR_SDL_INSTANCE_TRANSLATE_X = 3,
       MTR_SDL_INSTANCE_TRANSLATE_Y = 4, MTR_SDL_INSTANCE_TRANSLATE_Z = 5 };
typedef struct mtr_sdl_binding {
    uint32_t track, target, index;
} mtr_sdl_binding;
int32_t mtr_rscheduler_apply(const mtr_rscheduler *s, uint32_t frame, const mtr_sdl_binding *bindings, size_t nbindings,
                             uint8_t *parts_disp, size_t nparts, float *model_mats, size_t ninstances);

/* -------------------------------------------------------------- rArchive ---- */
/* ArchiveFile over a caller-owned file image (zlib-compressed resources behind a table of 0x90-byte entries).
 * MTR_E_INVALID: magic != "ARC\0" or version != 7 (asserts at src/rarchive.rs:79-80), truncated table, path not
 * terminated.  The view points INTO the caller's buffer. */
typedef struct mtr_rarchive_view {
    uint32_t num_resources;
    const uint8_t *table;   /* num_resources x 0x90: path[128], dti_type, size_compressed, orgsize:29|quality:3, offset */
    const uint8_t *file;    /* the archive image */
    size_t file_len;
} mtr_rarchive_view;
typedef struct mtr_resource_info {
    const char *path;       /* backslash-separated, no extension (src/rarchive.rs:138-141) */
    uint32_t dti_hash;      /* resource class (crc32(name) & 0x7fffffff, src/dti.rs:174) */
    uint32_t size_compressed, size_uncompressed, quality, offset;
} mtr_resource_info;
int32_t mtr_rarchive_parse(const void *data, size_t len, mtr_rarchive_view *out);
int32_t mtr_rarchive_info(const mtr_rarchive_view *a, uint32_t i, mtr_resource_info *out);
/* ArchiveFile::get_resource (src/rarchive.rs:143-176): index of the resource with this path ('/' accepted for '\\')
 * and class hash, or -1 */
int32_t mtr_rarchive_find(const mtr_rarchive_view *a, const char *path, uint32_t dti_hash);
/* inflate resource i into out (capacity cap >= size_uncompressed).  MTR_E_INVALID: compressed bytes outside the file,
 * corrupt stream, or a length other than size_uncompressed (assert at src/rarchive.rs:173) */
int32_t mtr_rarchive_extract(const mtr_rarchive_view *a, uint32_t i, void *out, size_t cap, size_t *out_len);

/* ------------------------------------------- Model::new from parsed files ---- */
/* src/model.rs:36-293 over real files: per primitive the input layout is looked up in the shader package by
 * handle (missing -> MTR_E_INVALID, the reference panics), the texture through material name -> rMaterial ->
 * albedo texture index -> textures[] (the caller loads rmaterial texture i into textures[i]; NULL = could not be
 * loaded, an error only if a primitive needs it: "no texture found!", src/model.rs:167), the debug id from the
 * primitive's boundary joint.  `mat` may be NULL (every primitive untextured).  The joints' offsets become the model's
 * joint positions (src/model.rs:283-291) for mtr_frame_draw_model_joints. */
int32_t mtr_model_create_from_files(mtr_device *dev, const mtr_rmodel_view *model, const mtr_rshader2 *sh,
                                    const mtr_rmaterial *mat, mtr_texture *const *textures, size_t ntextures,
                                    mtr_model **out);

#ifdef __cplusplus
}
#endif
#endif
