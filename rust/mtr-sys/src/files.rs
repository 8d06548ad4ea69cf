//! Raw bindings of include/mtr_files.h (resource-file readers).  Uncompiled here: no Rust toolchain in the build image.
//! A maintainer keeps the reference's own readers (src/rmodel.rs, src/rshader2.rs, ...) and does not need these; they
//! exist for hosts that want the library to parse the files.
use super::*;
use std::os::raw::c_char;

#[repr(C)]
pub struct mtr_rmodel_view {
    pub magic: u32,
    pub version: u16,
    pub jnt_num: u16,
    pub primitive_num: u16,
    pub material_num: u16,
    pub vertex_num: u32,
    pub index_num: u32,
    pub polygon_num: u32,
    pub vertexbuf_size: u32,
    pub texture_num: u32,
    pub parts_num: u32,
    pub boundary_num: u32,
    pub bounding_sphere: [f32; 4],
    pub bounding_box_min: [f32; 4],
    pub bounding_box_max: [f32; 4],
    pub material_names: *const u8,
    pub primitives: *const mtr_primitive,
    pub boundary_infos: *const u8,
    pub joint_infos: *const u8,
    pub lmats: *const f32,
    pub imats: *const f32,
    pub joint_table: *const u8,
    pub parts: *const u8,
    pub vertex_buf: *const u8,
    pub index_buf: *const u16,
}

#[repr(C)]
pub struct mtr_rtexture_view {
    pub version: u32,
    pub prebias: u32,
    pub type_: u32,
    pub level_count: u32,
    pub array_count: u32,
    pub format: u32,
    pub width: u32,
    pub height: u32,
    pub level0_offset: u64,
    pub data: *const u8,
    pub data_len: usize,
}

#[repr(C)]
pub struct mtr_raw_element {
    pub name: *const c_char,
    pub sindex: u32,
    pub format: u32,
    pub count: u32,
    pub start: u32,
    pub offset: u32,
    pub instance: u32,
}

#[repr(C)]
pub struct mtr_material_info {
    pub name_hash: u32,
    pub dti_hash: u32,
    pub albedo_texture: i32,
    pub bsstate: u32,
    pub dsstate: u32,
    pub rsstate: u32,
    pub state_num: u32,
    pub blend_factor: [f32; 4],
}

#[repr(C)]
pub struct mtr_track_info {
    pub track_type: u32,
    pub prop_type: u32,
    pub key_num: u32,
    pub parent: u32,
    pub dti_or_prop: u32,
    pub name: *const c_char,
}

#[repr(C)]
pub struct mtr_rarchive_view {
    pub num_resources: u32,
    pub table: *const u8,
    pub file: *const u8,
    pub file_len: usize,
}

#[repr(C)]
pub struct mtr_resource_info {
    pub path: *const c_char,
    pub dti_hash: u32,
    pub size_compressed: u32,
    pub size_uncompressed: u32,
    pub quality: u32,
    pub offset: u32,
}

pub enum mtr_rshader2 {}
pub enum mtr_rmaterial {}
pub enum mtr_rscheduler {}

extern "C" {
    pub fn mtr_files_last_error() -> *const c_char;
    pub fn mtr_file_struct_size(kind: u32) -> usize;
    pub fn mtr_rmodel_parse(data: *const c_void, len: usize, out: *mut mtr_rmodel_view) -> i32;
    pub fn mtr_primitive_field(prim: *const mtr_primitive, field: u32) -> u32;
    pub fn mtr_rmodel_boundary_joint(m: *const mtr_rmodel_view, i: u32, out: *mut u32) -> i32;
    pub fn mtr_rmodel_joint(m: *const mtr_rmodel_view, i: u32, no: *mut u32, parent: *mut u32, symmetry: *mut u32, offset: *mut f32) -> i32;
    pub fn mtr_rtexture_parse(data: *const c_void, len: usize, out: *mut mtr_rtexture_view) -> i32;
    pub fn mtr_texture_create_from_file(dev: *mut mtr_device, data: *const c_void, len: usize, out: *mut *mut mtr_texture) -> i32;
    pub fn mtr_rshader2_parse(data: *const c_void, len: usize, out: *mut *mut mtr_rshader2) -> i32;
    pub fn mtr_rshader2_destroy(sh: *mut mtr_rshader2);
    pub fn mtr_rshader2_num_objects(sh: *const mtr_rshader2) -> u32;
    pub fn mtr_rshader2_object(sh: *const mtr_rshader2, i: u32, name: *mut *const c_char, obj_type: *mut u32, name_hash: *mut u32) -> i32;
    pub fn mtr_rshader2_find(sh: *const mtr_rshader2, handle: u32) -> i32;
    pub fn mtr_rshader2_input_layout(sh: *const mtr_rshader2, i: u32, stride: *mut u32, layout: *mut mtr_layout,
                                     raw: *mut mtr_raw_element, raw_cap: u32, raw_num: *mut u32) -> i32;
    pub fn mtr_rmaterial_parse(data: *const c_void, len: usize, sh: *const mtr_rshader2, out: *mut *mut mtr_rmaterial) -> i32;
    pub fn mtr_rmaterial_destroy(m: *mut mtr_rmaterial);
    pub fn mtr_rmaterial_num_textures(m: *const mtr_rmaterial) -> u32;
    pub fn mtr_rmaterial_texture_path(m: *const mtr_rmaterial, i: u32) -> *const c_char;
    pub fn mtr_rmaterial_num_materials(m: *const mtr_rmaterial) -> u32;
    pub fn mtr_rmaterial_info(m: *const mtr_rmaterial, i: u32, out: *mut mtr_material_info) -> i32;
    pub fn mtr_rmaterial_find(m: *const mtr_rmaterial, name: *const c_char) -> i32;
    pub fn mtr_rscheduler_parse(data: *const c_void, len: usize, out: *mut *mut mtr_rscheduler) -> i32;
    pub fn mtr_rscheduler_destroy(s: *mut mtr_rscheduler);
    pub fn mtr_rscheduler_num_tracks(s: *const mtr_rscheduler) -> u32;
    pub fn mtr_rscheduler_track(s: *const mtr_rscheduler, i: u32, out: *mut mtr_track_info) -> i32;
    pub fn mtr_rscheduler_key(s: *const mtr_rscheduler, track: u32, k: u32, frame: *mut u32, mode: *mut u32, value_bits: *mut u64,
                              resource: *mut *const c_char) -> i32;
    pub fn mtr_rscheduler_eval(s: *const mtr_rscheduler, track: u32, frame: u32, value_bits: *mut u64) -> i32;
    pub fn mtr_rarchive_parse(data: *const c_void, len: usize, out: *mut mtr_rarchive_view) -> i32;
    pub fn mtr_rarchive_info(a: *const mtr_rarchive_view, i: u32, out: *mut mtr_resource_info) -> i32;
    pub fn mtr_rarchive_find(a: *const mtr_rarchive_view, path: *const c_char, dti_hash: u32) -> i32;
    pub fn mtr_rarchive_extract(a: *const mtr_rarchive_view, i: u32, out: *mut c_void, cap: usize, out_len: *mut usize) -> i32;
    pub fn mtr_model_create_from_files(dev: *mut mtr_device, model: *const mtr_rmodel_view, sh: *const mtr_rshader2, mat: *const mtr_rmaterial,
                                       textures: *const *mut mtr_texture, ntextures: usize, out: *mut *mut mtr_model) -> i32;
    // round 2 (mtr_files.h): skeleton -> skin palette, state objects by name, mip chains, VECTOR / MATRIX keys, track bindings
    pub fn mtr_rmodel_palette(m: *const mtr_rmodel_view, local_mats: *const f32, out_palette: *mut f32, cap_mats: usize) -> i32;
    pub fn mtr_rmodel_joint_index(m: *const mtr_rmodel_view, no: u32) -> i32;
    pub fn mtr_texture_create_from_file_mips(dev: *mut mtr_device, data: *const c_void, len: usize, max_levels: u32, out: *mut *mut mtr_texture) -> i32;
    pub fn mtr_state_from_names(blend_name: *const c_char, depth_stencil_name: *const c_char, rasterizer_name: *const c_char,
                                out: *mut super::mtr_prim_state) -> i32;
    pub fn mtr_model_states_from_files(model: *const mtr_rmodel_view, sh: *const mtr_rshader2, mat: *const mtr_rmaterial,
                                       states: *mut super::mtr_prim_state, nstates: usize) -> i32;
    pub fn mtr_rscheduler_key_floats(s: *const mtr_rscheduler, track: u32, k: u32, out: *mut f32, n: *mut u32) -> i32;
    pub fn mtr_rscheduler_eval_floats(s: *const mtr_rscheduler, track: u32, frame: u32, out: *mut f32, n: *mut u32) -> i32;
    pub fn mtr_rscheduler_find_track(s: *const mtr_rscheduler, name: *const c_char) -> i32;
    pub fn mtr_rscheduler_apply(s: *const mtr_rscheduler, frame: u32, bindings: *const mtr_sdl_binding, nbindings: usize,
                                parts_disp: *mut u8, nparts: usize, model_mats: *mut f32, ninstances: usize) -> i32;
}
/// one binding of a track to what the draw path animates (mtr_files.h: MTR_SDL_*)
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mtr_sdl_binding {
    pub track: u32,
    pub target: u32,
    pub index: u32,
}
