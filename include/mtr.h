/* mtr.h -- C ABI of libmtr.so: the MI355X-native (HIP, gfx950) rModel draw path.
 *
 * The reference (ReplayCoding/mt-renderer) has no FFI: its seam is the Rust API typed over wgpu
 *     Model::new / Model::set_parts_disp / Model::render      (src/model.rs:36-45, :295, :299-305)
 *     Texture::new                                            (src/texture.rs:11)
 *     RendererApp::{setup, render, post_render}               (src/renderer_app_manager.rs:14-32)
 * and everything below `wgpu::RenderPass::draw_indexed` happens inside a GPU driver.  This header
 * is what a Rust `mtr-sys` crate would bind: plain pointers and sizes, an
 * int32 status return, out-parameters last, no unwinding, opaque handles freed by the matching
 * *_destroy.  The caller owns every host buffer passed in; the library copies at *_create.
 *
 * Matrix convention: 16 f32, column-major, M*v (glam::Mat4 bytes, src/bin/modelviewer.rs:217-221).
 * Threading: one host thread per mtr_device at a time (externally synchronised), as the reference's
 * single winit thread.  There is no CPU fallback: every entry point needs a HIP device.
 */
#ifndef MTR_H
#define MTR_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTR_ABI_VERSION 2

enum {
    MTR_OK = 0,
    MTR_E_INVALID = 1,     /* bad argument / out-of-range offset (reference: panic) */
    MTR_E_UNSUPPORTED = 2, /* vertex/texture format the reference todo!()s: src/rshader2.rs:519-563, src/rtexture.rs:159 */
    MTR_E_NOMEM = 3,
    MTR_E_HIP = 4,         /* a HIP runtime call failed; see mtr_last_error */
    MTR_E_OVERFLOW = 5     /* internal queue capacity exceeded and could not be grown */
};

/* vertex element semantics: Position / TexCoord are the only names the reference binds
 * (src/rshader2.rs:503-507); Joint / Weight are this build's linear-blend-skinning extension */
enum { MTR_SEM_POSITION = 0, MTR_SEM_TEXCOORD = 1, MTR_SEM_JOINT = 2, MTR_SEM_WEIGHT = 3 };

/* InputElementFormat, numeric values of src/rshader2.rs:73-90 */
enum {
    MTR_IEF_F32 = 1, MTR_IEF_F16 = 2, MTR_IEF_S16 = 3, MTR_IEF_U16 = 4, MTR_IEF_S16N = 5,
    MTR_IEF_U16N = 6, MTR_IEF_S8 = 7, MTR_IEF_U8 = 8, MTR_IEF_S8N = 9, MTR_IEF_U8N = 10,
    MTR_IEF_SCMP3N = 11, MTR_IEF_UCMP3N = 12, MTR_IEF_U8NL = 13, MTR_IEF_COLOR4N = 14
};

/* rTexture format ids accepted by TextureFile::format_wgpu (src/rtexture.rs:152-161) */
enum { MTR_TEX_RGBA8 = 7, MTR_TEX_BC1 = 19, MTR_TEX_BC7 = 42, MTR_TEX_BC7_ALT = 54 };

/* PrimitiveInfo, verbatim 0x38 bytes (src/rmodel.rs:135-171; size test :489) */
typedef struct mtr_primitive {
    uint32_t w[14];
} mtr_primitive;

/* one decoded input-layout element (src/rshader2.rs:425-442): replaces the wgpu::VertexAttribute
 * list that Shader2File::create_vertex_buffer_elements builds (src/rshader2.rs:496-571) */
/* mtr_element.flags: the reference skips IEF_SCMP3N elements (src/rshader2.rs:509-512); with this bit a Position /
 * TexCoord element of that format is decoded as three signed 10-bit fields, x = bits 0..9, max(v / 511, -1)
 * (build extension, row f-4; SPEC.md section 2) */
#define MTR_ELEM_DECODE_SCMP3N 1u
typedef struct mtr_element {
    uint8_t semantic; /* MTR_SEM_* */
    uint8_t format;   /* MTR_IEF_* */
    uint8_t count;
    uint8_t flags;    /* MTR_ELEM_* (0 = the reference's behaviour) */
    uint16_t offset;  /* byte offset inside the vertex (9 bits in the file) */
    uint16_t pad1;
} mtr_element;

typedef struct mtr_layout {
    uint32_t num_elements; /* <= 8 */
    mtr_element elements[8];
} mtr_layout;

typedef struct mtr_device mtr_device;
typedef struct mtr_texture mtr_texture;
typedef struct mtr_model mtr_model;
typedef struct mtr_batch mtr_batch;
typedef struct mtr_frame mtr_frame;

/* per-frame counters (device side, read back by mtr_frame_get_stats) */
typedef struct mtr_frame_stats {
    uint64_t tris_in;     /* input triangles, SURVEY 8(d) definition */
    uint64_t tris_setup;  /* triangles that survived clip / cull / empty-bbox (sharded frame: and touch a bin of this rank); one that
                             covers no pixel centre is counted here and queued nowhere: bin_entries counts what is queued */
    uint64_t bin_entries; /* (triangle, 16x16 bin) pairs */
    uint64_t segments;    /* per-bin ordered runs */
    uint32_t width, height, nbins, ndraws;
    uint32_t tile_kernel; /* MTR_TILE_ORDERED, MTR_TILE_VISIBILITY or MTR_TILE_MIXED: which tile kernel(s) rendered the frame */
    uint32_t binning;     /* 1 = single-pass bounded queues, 2 = exact two-pass (count, scan, fill) queues */
    uint64_t chunks;        /* geometry waves of the frame (62 strip positions each, per instance) */
    uint64_t chunks_culled; /* of those, skipped by this rank before any vertex work: bounds outside its bins (sharded frames;
                             * whole instances of a batch and single chunks alike) */
    uint32_t shard_map;     /* MTR_OWN_* */
    uint32_t shard_bins;    /* bins this rank rendered */
} mtr_frame_stats;

/* stage timings of the last submitted frame, milliseconds, from hipEvents recorded on the
 * device's stream (enabled by mtr_device_set_profiling) */
enum { MTR_STAGE_GEOM = 0, MTR_STAGE_SCAN = 1, MTR_STAGE_FILL = 2, MTR_STAGE_TILE = 3, MTR_STAGE_COUNT = 4 };

/* ---- device (replaces wgpu::Device + wgpu::Queue, src/renderer_app_manager.rs:103-115) ---- */
int32_t mtr_device_create(int32_t hip_device, mtr_device **out);
/* same, but all work is issued on a caller-owned hipStream_t (e.g. torch's current stream) */
int32_t mtr_device_create_on_stream(int32_t hip_device, void *hip_stream, mtr_device **out);
void mtr_device_destroy(mtr_device *dev);
const char *mtr_last_error(const mtr_device *dev); /* never NULL; dev may be NULL for create errors */
int32_t mtr_device_set_profiling(mtr_device *dev, int32_t enable);
/* wgpu::Device::poll(Maintain::Wait): returns once every frame submitted so far (and every exchange handed over) has
 * left the GPU, with the error of any frame that was released without mtr_frame_wait and turned out to have overflowed
 * its bin queues (MTR_E_OVERFLOW: that frame is missing triangles; later frames get larger queues). */
int32_t mtr_device_synchronize(mtr_device *dev);
/* tile-kernel choice.  AUTO: the visibility-key kernel when every material of the frame is opaque (debug-id /
 * overlay colours, textures whose alpha is 255 everywhere -- the blend is then a replace), else the ordered
 * kernel.  ORDERED forces the ordered kernel (tests compare both); VISIBILITY is honoured only when eligible. */
enum { MTR_TILE_AUTO = 0, MTR_TILE_ORDERED = 1, MTR_TILE_VISIBILITY = 2,
       MTR_TILE_MIXED = 3 /* reported only: visibility kernel on the bins that hold no translucent triangle, ordered kernel on the rest */ };
int32_t mtr_device_set_tile_mode(mtr_device *dev, int32_t mode);
/* triangle -> bin queues.  single_pass != 0 (default): k_geom writes straight into bounded per-bin queues of
 * queue_capacity entries (0 keeps the current bound); a frame that overflows a queue is transparently re-run with the
 * exact two-pass queues (count, scan, fill) and the bound doubles for later frames.  single_pass == 0: always two-pass. */
int32_t mtr_device_set_binning(mtr_device *dev, int32_t single_pass, uint32_t queue_capacity);
int32_t mtr_abi_version(void);

/* ---- Texture::new (src/texture.rs:11-30): level 0 only, 2-D, decoded on upload ---- */
int32_t mtr_texture_create(mtr_device *dev, uint32_t width, uint32_t height, uint32_t format,
                           const void *data, size_t len, mtr_texture **out);
/* the same with a mip chain (row f-4): `levels` levels in `data`, level l = max(1, width >> l) x max(1, height >> l),
 * level 0 first, each in `format`.  The reference uploads level 0 only (src/texture.rs:21) although rTexture files
 * carry the chain (src/rtexture.rs:111-130); with more levels a minified sample (SPEC.md section 7) takes the nearest
 * texel of the nearest level: level l while max |d(uv)/d(xy)| * size > 2^(l - 1/2).  levels = 1 is mtr_texture_create. */
int32_t mtr_texture_create_mips(mtr_device *dev, uint32_t width, uint32_t height, uint32_t format, uint32_t levels,
                                const void *data, size_t len, mtr_texture **out);
void mtr_texture_destroy(mtr_texture *tex);
/* decoded RGBA8 texels of level 0 (row-major, width*height*4 bytes): what the sampler reads */
int32_t mtr_texture_read_rgba8(mtr_texture *tex, void *out, size_t len);
/* What a BC1 / BC7 texture created AFTER this call keeps in HBM.  The reference hands the blocks to the GPU's texture unit
 * (device feature TEXTURE_COMPRESSION_BC, src/renderer_app_manager.rs:107); here either
 *   MTR_TEXRES_DECODED  (default) the image is decoded to RGBA8 once at creation and the sampler reads plain texels, or
 *   MTR_TEXRES_BLOCKS   the blocks stay as uploaded (1/4 of the bytes for BC7, 1/8 for BC1) and every fetch decodes its
 *                       texel from its block.  Same pixels; mtr_texture_read_rgba8 returns MTR_E_UNSUPPORTED.
 * DESIGN.md section 3 has the measured trade (C5: 64 textures of 1024 x 1024). */
enum { MTR_TEXRES_DECODED = 0, MTR_TEXRES_BLOCKS = 1 };
int32_t mtr_device_set_texture_residency(mtr_device *dev, uint32_t mode);

/* ---- Model::new (src/model.rs:36-293) ----
 * vertex_buf/index_buf: ModelFile::vertex_buf()/index_buf() (src/rmodel.rs:457-463).
 * prims: ModelFile::primitives(), one layout / texture slot / debug id per primitive:
 *   prim_to_texture[p] = index into textures[] or -1 (mat_to_tex, src/model.rs:60-75,167,324)
 *   prim_debug_id[p]   = boundary_infos[prim.boundary_num()].joint() (src/model.rs:140-141)
 * parts_disp defaults to all-true with len = nprims (src/model.rs:270). */
int32_t mtr_model_create(mtr_device *dev, const void *vertex_buf, size_t vertex_len,
                         const uint16_t *index_buf, size_t index_num, const mtr_primitive *prims,
                         size_t nprims, const mtr_layout *layouts, const int32_t *prim_to_texture,
                         mtr_texture *const *textures, size_t ntextures,
                         const uint32_t *prim_debug_id, mtr_model **out);
void mtr_model_destroy(mtr_model *model);
/* Material state per primitive (row f-4).  The reference builds every pipeline with one fixed state (alpha blend,
 * depth LessEqual + write, cull back: src/model.rs:240-262) and only LOGS the blend / depth-stencil / rasterizer
 * state objects a material names (src/rmaterial.rs:104-106, :211-230); applying them is this build's extension:
 *   blend       MTR_BLEND_ALPHA  rgb = src * a + dst * (1 - a), alpha = src (the reference)
 *               MTR_BLEND_OFF    replace
 *               MTR_BLEND_ADD    rgb = src * a + dst, alpha = src
 *   depth_write 1 (the reference) / 0: fragments that pass leave the depth buffer alone
 *   depth_test  1: LessEqual (the reference) / 0: always pass (near / far clipping still applies)
 *   cull        MTR_CULL_BACK (the reference) / MTR_CULL_NONE / MTR_CULL_FRONT; a back face that is kept is
 *               rasterised with its second and third vertex exchanged
 * states: nprims entries, or NULL to go back to the reference state.  Takes effect for frames drawn afterwards.
 * mtr_files.h maps MT Framework state-object names to these (mtr_state_from_names). */
enum { MTR_BLEND_ALPHA = 0, MTR_BLEND_OFF = 1, MTR_BLEND_ADD = 2 };
enum { MTR_CULL_BACK = 0, MTR_CULL_NONE = 1, MTR_CULL_FRONT = 2 };
typedef struct mtr_prim_state {
    uint8_t blend, depth_write, depth_test, cull;
} mtr_prim_state;
int32_t mtr_model_set_prim_states(mtr_model *model, const mtr_prim_state *states, size_t nprims);
/* Model::set_parts_disp (src/model.rs:295-297) */
int32_t mtr_model_set_parts_disp(mtr_model *model, const uint8_t *parts_disp, size_t n);
/* bone palette for linear-blend skinning (build extension; n x 16 f32 column-major, n <= 256) */
int32_t mtr_model_set_palette(mtr_model *model, const float *mats, size_t n);

/* ---- instance batch ("scheduler" submission): n instances of one model, each
 * with its own model matrix and (optionally) palette and albedo override, resident in HBM ---- */
int32_t mtr_batch_create(mtr_device *dev, mtr_model *model, size_t n, const float *model_mats,
                         const float *palettes /* n * npal * 16 or NULL */, size_t npal,
                         const int32_t *texture_override /* n entries or NULL */, mtr_batch **out);
void mtr_batch_destroy(mtr_batch *batch);

/* ---- frame = one render pass (src/bin/modelviewer.rs:190-210: clear colour / clear depth) ---- */
int32_t mtr_frame_begin(mtr_device *dev, uint32_t width, uint32_t height, const float clear_rgba[4],
                        float clear_depth, mtr_frame **out);
/* Multi-GPU, one process per GPU: the frame renders only the 16x16-pixel bins that `rank` of `world` owns; colour / depth
 * of the other bins are untouched.  The sharded colour is exchanged by the caller (RCCL all-gather, see bench.py).
 * Which bins a rank owns is the host's choice, per frame (every rank of a frame must make the same choice):
 *   MTR_OWN_INTERLEAVED  bin b (row-major) -> rank b % world.  Finest balance; every object touches every rank.
 *   MTR_OWN_BANDS        rank r owns the bin rows [band_rows[r], band_rows[r+1]) (band_rows[0] = 0, band_rows[world] =
 *                        ceil(height / 16), non-decreasing; NULL = equal bands).  An object touches the few ranks
 *                        whose bands it crosses, so the geometry a rank must process shrinks with the world.
 *   MTR_OWN_SUPERTILES   squares of (1 << param) x (1 << param) bins (param <= 6) dealt round-robin, row-major.
 * A sharded rank does not process geometry that cannot reach its bins: every geometry wave first tests the object-space
 * bounds of its 62 strip positions (one box per joint that carries weight, transformed by that joint's palette matrix
 * and the view-projection; computed at mtr_model_create) against the rank's bins, and a batch draw first compacts its
 * instance list the same way.  The test is conservative: the pixels of a rank's bins are exactly those of the unsharded
 * frame.  mtr_device_set_culling(dev, 0) turns it off (tests compare both).  mtr_frame_set_shard = INTERLEAVED. */
enum { MTR_OWN_INTERLEAVED = 0, MTR_OWN_BANDS = 1, MTR_OWN_SUPERTILES = 2 };
int32_t mtr_frame_set_shard(mtr_frame *frame, uint32_t rank, uint32_t world);
int32_t mtr_frame_set_shard_map(mtr_frame *frame, uint32_t rank, uint32_t world, uint32_t map, uint32_t param,
                                const uint32_t *band_rows /* world + 1 entries, or NULL */);
/* mode: MTR_GEOM_CULL_OFF; MTR_GEOM_CULL_SHARDED (default: sharded frames with bands / super-tiles); MTR_GEOM_CULL_ALL_FRAMES:
 * unsharded frames too -- the same test against the whole target, i.e. instances and chunks that are off the target (or
 * behind the near plane as a whole) are skipped before any vertex work.  Costs two small kernels per draw (about 30 us
 * on 1024 instances), so it pays when a sizeable part of a batch is out of view; never changes a pixel. */
enum { MTR_GEOM_CULL_OFF = 0, MTR_GEOM_CULL_SHARDED = 1, MTR_GEOM_CULL_ALL_FRAMES = 2 };
int32_t mtr_device_set_culling(mtr_device *dev, int32_t mode);
/* Model::render (src/model.rs:299-363) with transform = view_proj (src/bin/modelviewer.rs:217-221) */
int32_t mtr_frame_draw_model(mtr_frame *frame, mtr_model *model, const float view_proj[16]);
int32_t mtr_frame_draw_batch(mtr_frame *frame, mtr_batch *batch, const float view_proj[16]);
/* convenience: host arrays, builds a temporary batch (npal palettes per instance, may be 0) */
int32_t mtr_frame_draw_instances(mtr_frame *frame, mtr_model *model, const float *model_mats,
                                 const float *palettes, size_t npal, size_t n,
                                 const float view_proj[16]);
/* The per-joint cubes Model::render adds to the debug overlay every frame (src/model.rs:309-315): one cube per joint at
 * joint_position * 0.01 with scale 0.005 (from_scale_rotation_translation, src/debug_overlay.rs:227-231), drawn like
 * mtr_frame_draw_overlay_cubes.  The positions are JointInfo::offset of every joint (src/model.rs:283-291), given with
 * mtr_model_set_joint_positions (mtr_model_create_from_files sets them).  A model without joints draws nothing. */
int32_t mtr_model_set_joint_positions(mtr_model *model, const float *xyz, size_t njoints);
int32_t mtr_frame_draw_model_joints(mtr_frame *frame, mtr_model *model, const float camera[16]);
/* DebugOverlay::render (src/debug_overlay.rs:202-221): n instanced cubes, no blend, constant colour */
int32_t mtr_frame_draw_overlay_cubes(mtr_frame *frame, const float camera[16], const float *inst_mats,
                                     size_t n);
/* enqueue all kernels; returns without waiting.  A frame that is submitted and then destroyed (or consumed through its
 * device pointers) without mtr_frame_wait cannot be re-run if a bounded bin queue overflows: its tile kernels then
 * leave every bin at the clear colour instead of rendering from incomplete queues, and the NEXT mtr_frame_begin /
 * mtr_frame_submit / mtr_device_synchronize returns MTR_E_OVERFLOW once (the bound has been raised by then). */
int32_t mtr_frame_submit(mtr_frame *frame);
/* framebuffer complete in HBM; a frame whose bin queues overflowed is transparently re-run first (exact two-pass queues) */
int32_t mtr_frame_wait(mtr_frame *frame);
int32_t mtr_frame_end(mtr_frame *frame);    /* submit + wait */
int32_t mtr_frame_read_color(mtr_frame *frame, void *rgba8, size_t len);  /* width*height*4 */
int32_t mtr_frame_read_depth(mtr_frame *frame, float *depth, size_t count); /* width*height */
void *mtr_frame_color_devptr(mtr_frame *frame); /* RGBA8 in HBM, row-major */
void *mtr_frame_depth_devptr(mtr_frame *frame); /* f32 in HBM, row-major */
/* multi-GPU exchange helpers (device pointers, enqueued on the device's stream):
 * pack: this frame's own bins (bin % world == rank), bin-major, 16x16 RGBA8 each, into dst
 *       (mtr_shard_bytes(width,height,world) bytes) = the all-gather send buffer;
 * unpack: gathered = world such blocks in rank order -> linear RGBA8 width*height at dst. */
size_t mtr_shard_bytes(uint32_t width, uint32_t height, uint32_t world); /* INTERLEAVED */
/* all-gather send size of any map = 1 KiB x the largest share of bins (0: bad arguments); pack zero-fills past a rank's own share */
size_t mtr_shard_bytes_map(uint32_t width, uint32_t height, uint32_t world, uint32_t map, uint32_t param, const uint32_t *band_rows);
size_t mtr_frame_shard_bytes(mtr_frame *frame); /* of this frame's map */
int32_t mtr_frame_pack_color_shard(mtr_frame *frame, void *dst_dev, size_t dst_bytes);
/* gathered (world blocks of mtr_frame_shard_bytes each, rank order) -> linear RGBA8 at dst_dev, by this frame's map;
 * hip_stream NULL = the device's public stream */
int32_t mtr_frame_unpack_color_shards_on_stream(mtr_frame *frame, const void *gathered_dev, void *dst_dev, void *hip_stream);
int32_t mtr_device_unpack_color_shards(mtr_device *dev, const void *gathered_dev, uint32_t world,
                                       uint32_t width, uint32_t height, void *dst_dev);
/* the same two steps on a caller-chosen hipStream_t instead of the device's public stream (the pack waits, on that
 * stream, for the frame's completion).  One in-order stream makes the exchange chain of frame k+1 (pack -> all-gather
 * -> unpack, with a cross-stream hand-over before and after the collective) wait for that of frame k; a host that
 * rotates a few exchange streams, each with its own send / receive buffers, overlaps them (bench.py). */
int32_t mtr_frame_pack_color_shard_on_stream(mtr_frame *frame, void *dst_dev, size_t dst_bytes, void *hip_stream);
int32_t mtr_device_unpack_color_shards_on_stream(mtr_device *dev, const void *gathered_dev, uint32_t world,
                                                 uint32_t width, uint32_t height, void *dst_dev, void *hip_stream);
/* The exchange of every sharded frame issued by a second host thread, owned by the device.  all-gather is the host's:
 * `fn` has ncclAllGather's signature (sendbuff, recvbuff, sendcount, datatype, comm, stream) and is called with
 * (send_dev, gathered_dev, send_bytes, dtype_u8, comm, hip_stream), so an RCCL host passes &ncclAllGather, its
 * communicator and ncclUint8; the library links no collective library itself.  Per frame the thread runs, on hip_stream:
 * pack (after the frame completes) -> fn -> unpack into dst_dev, then destroys the frame.
 *   (a frame whose bin queues overflowed is re-run by the thread before it is packed: the gathered frame is never missing
 *   triangles; the all-gather count is mtr_shard_bytes(frame width, height, world), which send_bytes must cover)
 *   mtr_frame_submit_exchange: submits the frame if it was not yet, hands it to the thread and CONSUMES the handle
 *       (blocks while 8 frames are waiting; on an error return the handle is NOT consumed and stays the caller's);   mtr_device_exchange_drain: returns once every handed-over frame has been
 *       issued (not: finished on the GPU -- synchronise hip_stream for that) with the first error of the thread, if any;
 *   mtr_device_exchange_stop: drain + join (also done by mtr_device_destroy).
 *   mtr_device_exchange_add_lane (optional, while the thread is idle, up to 4 lanes): one more (communicator, stream, send /
 *       gathered / destination buffers) set; frames are dealt to the lanes in turn (the i-th frame handed over goes to
 *       lane i % lanes, lane 0 being the one given to _start).  A lane is an in-order stream and completes one pack +
 *       all-gather + unpack latency per frame; two lanes keep two collectives in flight.  Every rank must use the same
 *       number of lanes, and each lane needs a communicator of its own.
 * Frames of one device are still begun / drawn / submitted by ONE thread; only these calls cross threads. */
typedef int (*mtr_allgather_fn)(const void *send, void *recv, size_t count, int datatype, void *comm, void *stream);
int32_t mtr_device_exchange_start(mtr_device *dev, mtr_allgather_fn fn, void *comm, int dtype_u8, void *send_dev,
                                  size_t send_bytes, void *gathered_dev, void *dst_dev, uint32_t world, void *hip_stream);
int32_t mtr_device_exchange_add_lane(mtr_device *dev, void *comm, void *send_dev, void *gathered_dev, void *dst_dev,
                                     void *hip_stream);
int32_t mtr_frame_submit_exchange(mtr_frame *frame);
int32_t mtr_device_exchange_drain(mtr_device *dev);
int32_t mtr_device_exchange_stop(mtr_device *dev);
/* ---- one host thread, N devices ("mtr_group_create(const int* devices, int n, mtr_group**) + same frame calls") ----
 * The reference has one event loop and one wgpu::Device (src/renderer_app_manager.rs:202-272).  A host that keeps that shape
 * and owns several GPUs of one node drives them through a group: rank r of the group is an mtr_device on hip_devices[r]
 * (an index may repeat: several ranks on one card), a group frame is one sharded frame per rank under one ownership map
 * (mtr_frame_set_shard_map's arguments), and ending it gathers the colour of every part into one linear RGBA8 image on
 * the device of rank 0: pack on each device, one peer copy per rank (xGMI where the devices are peers), unpack.
 *   resources are per device: create the model / textures / batches on mtr_group_device(group, r) for every r;
 *   drawing is "the same frame calls": mtr_frame_draw_* into mtr_group_frame_part(gf, r) with rank r's objects
 *       (the parts are owned by the group frame: never submit, wait for or destroy them yourself);
 *   mtr_group_frame_end submits every part, waits for all of them (a part whose bin queues overflowed is re-run first)
 *       and returns with the gathered image complete; it stays readable until the next group frame of the group ends;
 *   mtr_group_destroy destroys the group's devices: destroy their models, textures, batches and group frames first.
 * A group is driven by one thread and ends one frame at a time; the throughput path for N GPUs is one process (or host
 * thread) per device with the exchange thread above.  Errors: mtr_group_last_error(group)
 * (NULL: the last mtr_group_create failure). */
typedef struct mtr_group mtr_group;
typedef struct mtr_group_frame mtr_group_frame;
int32_t mtr_group_create(const int32_t *hip_devices, int32_t n, mtr_group **out); /* 1 <= n <= 64 */
void mtr_group_destroy(mtr_group *group);
int32_t mtr_group_size(const mtr_group *group);
mtr_device *mtr_group_device(mtr_group *group, int32_t rank);
const char *mtr_group_last_error(const mtr_group *group);
int32_t mtr_group_frame_begin(mtr_group *group, uint32_t width, uint32_t height, const float clear_rgba[4], float clear_depth,
                              uint32_t map, uint32_t param, const uint32_t *band_rows, mtr_group_frame **out);
mtr_frame *mtr_group_frame_part(mtr_group_frame *gframe, int32_t rank);
int32_t mtr_group_frame_end(mtr_group_frame *gframe);
int32_t mtr_group_frame_read_color(mtr_group_frame *gframe, void *rgba8, size_t len); /* width*height*4 */
void *mtr_group_frame_color_devptr(mtr_group_frame *gframe); /* on rank 0's device; NULL once a later group frame has ended */
void mtr_group_frame_destroy(mtr_group_frame *gframe);
int32_t mtr_frame_get_stats(mtr_frame *frame, mtr_frame_stats *out);
int32_t mtr_frame_get_timings(mtr_frame *frame, float ms[MTR_STAGE_COUNT]);
/* tuning / test hook: per-bin queue sizes of the frame just rendered (valid until the next frame is submitted on
 * this device): entries[b] = (triangle, bin) pairs of 16x16 bin b, segments[b] = ordered runs; nbins each.  A sharded
 * frame reports its own bins; the bins of the other ranks read 0. */
int32_t mtr_frame_read_bin_counts(mtr_frame *frame, uint32_t *entries, uint32_t *segments, size_t nbins);
void mtr_frame_destroy(mtr_frame *frame);

/* ---- unit-test hooks: one stage at a time through the same kernels ---- */
/* vertex stage of primitive `prim`: clip = M * (skin(position),1) and decoded texcoord for every
 * vertex in [0, vertex_num); out_clip vertex_num*4 f32, out_uv vertex_num*2 f32 (host memory) */
int32_t mtr_model_vertex_stage(mtr_model *model, size_t prim, const float M[16], float *out_clip,
                               float *out_uv);
/* MT crc32 (src/util/crc.rs:36-50), host side: material-name -> material lookups */
uint32_t mtr_crc32(const uint8_t *bytes, size_t len, uint32_t init);

#ifdef __cplusplus
}
#endif
#endif
