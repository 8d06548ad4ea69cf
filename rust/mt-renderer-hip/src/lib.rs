//! Drop-in shapes for mt-renderer's GPU-object layer on top of libmtr.so.
//!
//! `Model::new / set_parts_disp / render` and `Texture::new` keep the reference's names and
//! argument roles (reference: src/model.rs:36-45, :295, :299-305; src/texture.rs:11), with
//! `wgpu::Device + wgpu::Queue` replaced by [`Device`] and `wgpu::RenderPass` by [`Frame`].
//! The caller passes what the parsed files contribute to the draw path: `ModelFile::vertex_buf()`,
//! `index_buf()`, `primitives()` (0x38-byte records, transmuted), the decoded input layouts
//! (`Shader2ObjectInputLayoutInfo::elements`), `mat_to_tex` and the debug ids.
//! UNCOMPILED in the build container (no cargo/rustc); see INTEGRATION.md.
use anyhow::{anyhow, Result};
use mtr_sys as sys;
use std::ffi::CStr;
use std::ptr;

pub struct Device(*mut sys::mtr_device);
pub struct Texture { h: *mut sys::mtr_texture }
pub struct Model { h: *mut sys::mtr_model, _textures: Vec<Texture> }
pub struct Frame<'d> { h: *mut sys::mtr_frame, dev: &'d Device }

fn check(dev: *const sys::mtr_device, rc: i32) -> Result<()> {
    if rc == sys::MTR_OK { return Ok(()); }
    let msg = unsafe { CStr::from_ptr(sys::mtr_last_error(dev)) }.to_string_lossy().into_owned();
    Err(anyhow!("mtr error {rc}: {msg}"))
}

impl Device {
    pub fn new(hip_device: i32) -> Result<Self> {
        let mut h = ptr::null_mut();
        check(ptr::null(), unsafe { sys::mtr_device_create(hip_device, &mut h) })?;
        Ok(Device(h))
    }
    /// MTR_TILE_AUTO (0) / ORDERED (1) / VISIBILITY (2)
    pub fn set_tile_mode(&self, mode: i32) -> Result<()> { check(self.0, unsafe { sys::mtr_device_set_tile_mode(self.0, mode) }) }
    pub fn set_binning(&self, single_pass: bool, queue_capacity: u32) -> Result<()> {
        check(self.0, unsafe { sys::mtr_device_set_binning(self.0, single_pass as i32, queue_capacity) })
    }
}
impl Drop for Device { fn drop(&mut self) { unsafe { sys::mtr_device_destroy(self.0) } } }

impl Texture {
    /// `Texture::new(device, queue, resource: TextureFile)` -- src/texture.rs:11
    pub fn new(device: &Device, width: u32, height: u32, format: u32, data: &[u8]) -> Result<Self> {
        let mut h = ptr::null_mut();
        check(device.0, unsafe { sys::mtr_texture_create(device.0, width, height, format, data.as_ptr().cast(), data.len(), &mut h) })?;
        Ok(Texture { h })
    }
}
impl Drop for Texture { fn drop(&mut self) { unsafe { sys::mtr_texture_destroy(self.h) } } }

pub struct ModelInputs<'a> {
    pub vertex_buf: &'a [u8],                  // ModelFile::vertex_buf()
    pub index_buf: &'a [u16],                  // ModelFile::index_buf()
    pub primitives: &'a [sys::mtr_primitive],  // ModelFile::primitives(), 0x38 bytes each
    pub layouts: &'a [sys::mtr_layout],        // per primitive: shader2.get_object_by_handle(prim.inputlayout())
    pub prim_to_texture: &'a [i32],            // mat_to_tex[prim.material_no()] or -1
    pub prim_debug_id: &'a [u32],              // boundary_infos[prim.boundary_num()].joint()
}

impl Model {
    /// `Model::new(model_file, material_file, shader2, resource_manager, device, queue, ..)` -- src/model.rs:36-45
    pub fn new(device: &Device, inputs: &ModelInputs, textures: Vec<Texture>) -> Result<Self> {
        let handles: Vec<*mut sys::mtr_texture> = textures.iter().map(|t| t.h).collect();
        let mut h = ptr::null_mut();
        check(device.0, unsafe {
            sys::mtr_model_create(device.0, inputs.vertex_buf.as_ptr().cast(), inputs.vertex_buf.len(), inputs.index_buf.as_ptr(),
                                  inputs.index_buf.len(), inputs.primitives.as_ptr(), inputs.primitives.len(), inputs.layouts.as_ptr(),
                                  inputs.prim_to_texture.as_ptr(), handles.as_ptr(), handles.len(), inputs.prim_debug_id.as_ptr(), &mut h)
        })?;
        Ok(Model { h, _textures: textures })
    }
    /// `Model::set_parts_disp(&mut self, parts_disp: &[bool])` -- src/model.rs:295
    pub fn set_parts_disp(&mut self, parts_disp: &[bool]) -> Result<()> {
        let v: Vec<u8> = parts_disp.iter().map(|b| *b as u8).collect();
        check(ptr::null(), unsafe { sys::mtr_model_set_parts_disp(self.h, v.as_ptr(), v.len()) })
    }
    pub fn set_palette(&mut self, mats: &[[f32; 16]]) -> Result<()> {
        check(ptr::null(), unsafe { sys::mtr_model_set_palette(self.h, mats.as_ptr().cast(), mats.len()) })
    }
    /// `Model::render(&self, rpass, queue, transform_bind_group, debug_overlay)` -- src/model.rs:299-305;
    /// the 64-byte transform uniform (src/bin/modelviewer.rs:217-221) is passed directly.
    pub fn render(&self, frame: &mut Frame, view_proj: &[f32; 16]) -> Result<()> {
        check(frame.dev.0, unsafe { sys::mtr_frame_draw_model(frame.h, self.h, view_proj.as_ptr()) })
    }
}
impl Drop for Model { fn drop(&mut self) { unsafe { sys::mtr_model_destroy(self.h) } } }

impl<'d> Frame<'d> {
    /// begin_render_pass with LoadOp::Clear(colour) / Clear(depth) -- src/bin/modelviewer.rs:190-210
    pub fn begin(dev: &'d Device, width: u32, height: u32, clear: [f32; 4], clear_depth: f32) -> Result<Self> {
        let mut h = ptr::null_mut();
        check(dev.0, unsafe { sys::mtr_frame_begin(dev.0, width, height, clear.as_ptr(), clear_depth, &mut h) })?;
        Ok(Frame { h, dev })
    }
    /// queue.submit + wait -- src/renderer_app_manager.rs:185
    pub fn end(&mut self) -> Result<()> { check(self.dev.0, unsafe { sys::mtr_frame_end(self.h) }) }
    /// `end` in two halves: keep several frames in flight (the library overlaps them on its own streams)
    pub fn submit(&mut self) -> Result<()> { check(self.dev.0, unsafe { sys::mtr_frame_submit(self.h) }) }
    pub fn wait(&mut self) -> Result<()> { check(self.dev.0, unsafe { sys::mtr_frame_wait(self.h) }) }
    /// multi-GPU: render only the bins with `bin % world == rank`
    pub fn set_shard(&mut self, rank: u32, world: u32) -> Result<()> { check(self.dev.0, unsafe { sys::mtr_frame_set_shard(self.h, rank, world) }) }
    pub fn read_color(&mut self, out: &mut [u8]) -> Result<()> {
        check(self.dev.0, unsafe { sys::mtr_frame_read_color(self.h, out.as_mut_ptr().cast(), out.len()) })
    }
}
impl<'d> Drop for Frame<'d> { fn drop(&mut self) { unsafe { sys::mtr_frame_destroy(self.h) } } }
