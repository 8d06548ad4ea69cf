//! Raw bindings of `include/mtr.h`.  One declaration per exported symbol, same order as the header.
//! NOTE: written without a Rust toolchain available (the build container has none); kept
//! declaration-only so that it is checkable by eye against the header.
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_void};

pub const MTR_OK: i32 = 0;
pub const MTR_E_INVALID: i32 = 1;
pub const MTR_E_UNSUPPORTED: i32 = 2;
pub const MTR_E_NOMEM: i32 = 3;
pub const MTR_E_HIP: i32 = 4;
pub const MTR_E_OVERFLOW: i32 = 5;

pub const MTR_SEM_POSITION: u8 = 0;
pub const MTR_SEM_TEXCOORD: u8 = 1;
pub const MTR_SEM_JOINT: u8 = 2;
pub const MTR_SEM_WEIGHT: u8 = 3;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct mtr_primitive {
    pub w: [u32; 14], // rmodel::PrimitiveInfo, 0x38 bytes verbatim
}
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mtr_element {
    pub semantic: u8,
    pub format: u8, // rshader2::InputElementFormat as u32 -> u8
    pub count: u8,
    pub pad0: u8,
    pub offset: u16,
    pub pad1: u16,
}
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mtr_layout {
    pub num_elements: u32,
    pub elements: [mtr_element; 8],
}
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mtr_frame_stats {
    pub tris_in: u64,
    pub tris_setup: u64,
    pub bin_entries: u64,
    pub segments: u64,
    pub width: u32,
    pub height: u32,
    pub nbins: u32,
    pub ndraws: u32,
    pub tile_kernel: u32,
    pub binning: u32,
    pub chunks: u64,
    pub chunks_culled: u64,
    pub shard_map: u32,  // MTR_OWN_*
    pub shard_bins: u32,
}
/// per-primitive material state (mtr.h): blend MTR_BLEND_*, depth write, depth test, cull MTR_CULL_*
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mtr_prim_state {
    pub blend: u8,
    pub depth_write: u8,
    pub depth_test: u8,
    pub cull: u8,
}
pub const MTR_OWN_INTERLEAVED: u32 = 0;
pub const MTR_OWN_BANDS: u32 = 1;
pub const MTR_OWN_SUPERTILES: u32 = 2;
pub const MTR_TEXRES_DECODED: u32 = 0;
pub const MTR_TEXRES_BLOCKS: u32 = 1;
macro_rules! opaque { ($($n:ident),*) => { $( #[repr(C)] pub struct $n { _p: [u8; 0] } )* } }
opaque!(mtr_device, mtr_texture, mtr_model, mtr_batch, mtr_frame);

pub type mtr_allgather_fn = Option<unsafe extern "C" fn(send: *const c_void, recv: *mut c_void, count: usize, datatype: i32,
                                                         comm: *mut c_void, stream: *mut c_void) -> i32>;

extern "C" {
    pub fn mtr_abi_version() -> i32;
    pub fn mtr_device_create(hip_device: i32, out: *mut *mut mtr_device) -> i32;
    pub fn mtr_device_create_on_stream(hip_device: i32, hip_stream: *mut c_void, out: *mut *mut mtr_device) -> i32;
    pub fn mtr_device_destroy(dev: *mut mtr_device);
    pub fn mtr_last_error(dev: *const mtr_device) -> *const c_char;
    pub fn mtr_device_set_profiling(dev: *mut mtr_device, enable: i32) -> i32;
    pub fn mtr_device_set_tile_mode(dev: *mut mtr_device, mode: i32) -> i32;
    pub fn mtr_device_set_binning(dev: *mut mtr_device, single_pass: i32, queue_capacity: u32) -> i32;
    pub fn mtr_frame_read_bin_counts(frame: *mut mtr_frame, entries: *mut u32, segments: *mut u32, nbins: usize) -> i32;
    pub fn mtr_texture_create(dev: *mut mtr_device, width: u32, height: u32, format: u32, data: *const c_void, len: usize,
                              out: *mut *mut mtr_texture) -> i32;
    pub fn mtr_texture_destroy(tex: *mut mtr_texture);
    pub fn mtr_texture_read_rgba8(tex: *mut mtr_texture, out: *mut c_void, len: usize) -> i32;
    pub fn mtr_model_create(dev: *mut mtr_device, vertex_buf: *const c_void, vertex_len: usize, index_buf: *const u16,
                            index_num: usize, prims: *const mtr_primitive, nprims: usize, layouts: *const mtr_layout,
                            prim_to_texture: *const i32, textures: *const *mut mtr_texture, ntextures: usize,
                            prim_debug_id: *const u32, out: *mut *mut mtr_model) -> i32;
    pub fn mtr_model_destroy(model: *mut mtr_model);
    pub fn mtr_model_set_parts_disp(model: *mut mtr_model, parts_disp: *const u8, n: usize) -> i32;
    pub fn mtr_model_set_palette(model: *mut mtr_model, mats: *const f32, n: usize) -> i32;
    pub fn mtr_batch_create(dev: *mut mtr_device, model: *mut mtr_model, n: usize, model_mats: *const f32,
                            palettes: *const f32, npal: usize, texture_override: *const i32, out: *mut *mut mtr_batch) -> i32;
    pub fn mtr_batch_destroy(batch: *mut mtr_batch);
    pub fn mtr_frame_begin(dev: *mut mtr_device, width: u32, height: u32, clear_rgba: *const f32, clear_depth: f32,
                           out: *mut *mut mtr_frame) -> i32;
    pub fn mtr_frame_set_shard(frame: *mut mtr_frame, rank: u32, world: u32) -> i32;
    pub fn mtr_frame_draw_model(frame: *mut mtr_frame, model: *mut mtr_model, view_proj: *const f32) -> i32;
    pub fn mtr_frame_draw_batch(frame: *mut mtr_frame, batch: *mut mtr_batch, view_proj: *const f32) -> i32;
    pub fn mtr_frame_draw_instances(frame: *mut mtr_frame, model: *mut mtr_model, model_mats: *const f32,
                                    palettes: *const f32, npal: usize, n: usize, view_proj: *const f32) -> i32;
    pub fn mtr_frame_draw_overlay_cubes(frame: *mut mtr_frame, camera: *const f32, inst_mats: *const f32, n: usize) -> i32;
    pub fn mtr_frame_submit(frame: *mut mtr_frame) -> i32;
    pub fn mtr_frame_wait(frame: *mut mtr_frame) -> i32;
    pub fn mtr_frame_end(frame: *mut mtr_frame) -> i32;
    pub fn mtr_frame_read_color(frame: *mut mtr_frame, rgba8: *mut c_void, len: usize) -> i32;
    pub fn mtr_frame_read_depth(frame: *mut mtr_frame, depth: *mut f32, count: usize) -> i32;
    pub fn mtr_frame_color_devptr(frame: *mut mtr_frame) -> *mut c_void;
    pub fn mtr_frame_depth_devptr(frame: *mut mtr_frame) -> *mut c_void;
    pub fn mtr_shard_bytes(width: u32, height: u32, world: u32) -> usize;
    pub fn mtr_frame_pack_color_shard(frame: *mut mtr_frame, dst_dev: *mut c_void, dst_bytes: usize) -> i32;
    pub fn mtr_device_unpack_color_shards(dev: *mut mtr_device, gathered_dev: *const c_void, world: u32, width: u32,
                                          height: u32, dst_dev: *mut c_void) -> i32;
    pub fn mtr_frame_pack_color_shard_on_stream(frame: *mut mtr_frame, dst_dev: *mut c_void, dst_bytes: usize, hip_stream: *mut c_void) -> i32;
    pub fn mtr_device_unpack_color_shards_on_stream(dev: *mut mtr_device, gathered_dev: *const c_void, world: u32, width: u32,
                                                    height: u32, dst_dev: *mut c_void, hip_stream: *mut c_void) -> i32;
    /// exchange thread (mtr.h): `fn_` has ncclAllGather's signature; `mtr_frame_submit_exchange` consumes the frame
    pub fn mtr_device_exchange_start(dev: *mut mtr_device, fn_: mtr_allgather_fn, comm: *mut c_void, dtype_u8: i32,
                                     send_dev: *mut c_void, send_bytes: usize, gathered_dev: *mut c_void, dst_dev: *mut c_void,
                                     world: u32, hip_stream: *mut c_void) -> i32;
    pub fn mtr_device_exchange_add_lane(dev: *mut mtr_device, comm: *mut c_void, send_dev: *mut c_void, gathered_dev: *mut c_void,
                                        dst_dev: *mut c_void, hip_stream: *mut c_void) -> i32;
    pub fn mtr_frame_submit_exchange(frame: *mut mtr_frame) -> i32;
    pub fn mtr_device_exchange_drain(dev: *mut mtr_device) -> i32;
    pub fn mtr_device_exchange_stop(dev: *mut mtr_device) -> i32;
    pub fn mtr_frame_get_stats(frame: *mut mtr_frame, out: *mut mtr_frame_stats) -> i32;
    pub fn mtr_frame_get_timings(frame: *mut mtr_frame, ms: *mut f32) -> i32;
    pub fn mtr_frame_destroy(frame: *mut mtr_frame);
    pub fn mtr_model_vertex_stage(model: *mut mtr_model, prim: usize, m: *const f32, out_clip: *mut f32, out_uv: *mut f32) -> i32;
    pub fn mtr_crc32(bytes: *const u8, len: usize, init: u32) -> u32;
    // round 2
    pub fn mtr_device_synchronize(dev: *mut mtr_device) -> i32;
    pub fn mtr_device_set_culling(dev: *mut mtr_device, mode: i32) -> i32; // MTR_GEOM_CULL_OFF / _SHARDED / _ALL_FRAMES = 0 / 1 / 2
    pub fn mtr_device_set_texture_residency(dev: *mut mtr_device, mode: u32) -> i32;
    pub fn mtr_texture_create_mips(dev: *mut mtr_device, width: u32, height: u32, format: u32, levels: u32, data: *const c_void,
                                   len: usize, out: *mut *mut mtr_texture) -> i32;
    pub fn mtr_model_set_prim_states(model: *mut mtr_model, states: *const mtr_prim_state, nprims: usize) -> i32;
    pub fn mtr_model_set_joint_positions(model: *mut mtr_model, xyz: *const f32, njoints: usize) -> i32;
    pub fn mtr_frame_draw_model_joints(frame: *mut mtr_frame, model: *mut mtr_model, camera: *const f32) -> i32;
    pub fn mtr_frame_set_shard_map(frame: *mut mtr_frame, rank: u32, world: u32, map: u32, param: u32, band_rows: *const u32) -> i32;
    pub fn mtr_shard_bytes_map(width: u32, height: u32, world: u32, map: u32, param: u32, band_rows: *const u32) -> usize;
    pub fn mtr_frame_shard_bytes(frame: *mut mtr_frame) -> usize;
    pub fn mtr_frame_unpack_color_shards_on_stream(frame: *mut mtr_frame, gathered_dev: *const c_void, dst_dev: *mut c_void,
                                                   hip_stream: *mut c_void) -> i32;
}
pub mod files;
