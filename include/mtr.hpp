// mtr.hpp -- C++ host mirror of the reference's GPU-object layer over the C ABI (mtr.h).
// Same names and argument roles as mt-renderer's Rust API: Texture::new (src/texture.rs:11),
// Model::new / set_parts_disp / render (src/model.rs:36-45, :295, :299-305); wgpu::Device+Queue -> mtr::Device,
// wgpu::RenderPass -> mtr::Frame.  Errors (the reference's anyhow::Result / panics) become mtr::Error.
#pragma once
#include "mtr.h"

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace mtr {

struct Error : std::runtime_error {
    int32_t code;
    Error(int32_t c, const std::string& m) : std::runtime_error("mtr error " + std::to_string(c) + ": " + m), code(c) {}
};

class Device {
  public:
    explicit Device(int hip_device = 0) {
        int32_t rc = mtr_device_create(hip_device, &h_);
        if (rc) throw Error(rc, mtr_last_error(nullptr));
    }
    // rank r of a Group: the group owns the device
    struct Borrowed {};
    Device(mtr_device* of_group, Borrowed) : h_(of_group), owned_(false) {}
    Device(const Device&) = delete;
    Device& operator=(const Device&) = delete;
    ~Device() {
        if (owned_) mtr_device_destroy(h_);
    }
    mtr_device* handle() const { return h_; }
    void set_tile_mode(int32_t mode) const { check(mtr_device_set_tile_mode(h_, mode)); }  // MTR_TILE_AUTO / ORDERED / VISIBILITY
    void set_binning(bool single_pass, uint32_t queue_capacity = 0) const { check(mtr_device_set_binning(h_, single_pass ? 1 : 0, queue_capacity)); }
    void unpack_color_shards(const void* gathered_dev, uint32_t world, uint32_t w, uint32_t h, void* dst_dev) const {
        check(mtr_device_unpack_color_shards(h_, gathered_dev, world, w, h, dst_dev));
    }
    // the exchange of sharded frames on a second host thread (mtr.h): fn = &ncclAllGather for an RCCL host
    void exchange_start(mtr_allgather_fn fn, void* comm, int dtype_u8, void* send_dev, size_t send_bytes, void* gathered_dev,
                        void* dst_dev, uint32_t world, void* hip_stream) const {
        check(mtr_device_exchange_start(h_, fn, comm, dtype_u8, send_dev, send_bytes, gathered_dev, dst_dev, world, hip_stream));
    }
    void exchange_add_lane(void* comm, void* send_dev, void* gathered_dev, void* dst_dev, void* hip_stream) const {
        check(mtr_device_exchange_add_lane(h_, comm, send_dev, gathered_dev, dst_dev, hip_stream));
    }
    // waits for every frame in flight; reports (once) an overflow latched by a frame nobody waited for
    void synchronize() const { check(mtr_device_synchronize(h_)); }
    void set_culling(int32_t mode) const { check(mtr_device_set_culling(h_, mode)); }  // MTR_GEOM_CULL_OFF / _SHARDED (default) / _ALL_FRAMES
    void set_texture_residency(uint32_t mode) const { check(mtr_device_set_texture_residency(h_, mode)); }  // MTR_TEXRES_*
    void exchange_drain() const { check(mtr_device_exchange_drain(h_)); }
    void exchange_stop() const { check(mtr_device_exchange_stop(h_)); }
    void check(int32_t rc) const {
        if (rc) throw Error(rc, mtr_last_error(h_));
    }

  private:
    mtr_device* h_ = nullptr;
    bool owned_ = true;
};

class Texture {
  public:
    // Texture::new(device, queue, resource) -- src/texture.rs:11
    Texture(const Device& dev, uint32_t width, uint32_t height, uint32_t format, const void* data, size_t len) {
        dev.check(mtr_texture_create(dev.handle(), width, height, format, data, len, &h_));
    }
    // the same with the file's mip chain (level 0 first); the reference uploads level 0 only (src/texture.rs:21)
    Texture(const Device& dev, uint32_t width, uint32_t height, uint32_t format, uint32_t levels, const void* data, size_t len) {
        dev.check(mtr_texture_create_mips(dev.handle(), width, height, format, levels, data, len, &h_));
    }
    Texture(Texture&& o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    Texture(const Texture&) = delete;
    ~Texture() { mtr_texture_destroy(h_); }
    mtr_texture* handle() const { return h_; }

  private:
    mtr_texture* h_ = nullptr;
};

class Frame;

class Model {
  public:
    // Model::new(model_file, material_file, shader2, resource_manager, device, queue, ..) -- src/model.rs:36-45
    Model(const Device& dev, const void* vertex_buf, size_t vertex_len, const uint16_t* index_buf, size_t index_num,
          const std::vector<mtr_primitive>& prims, const std::vector<mtr_layout>& layouts,
          const std::vector<int32_t>& prim_to_texture, const std::vector<const Texture*>& textures,
          const std::vector<uint32_t>& prim_debug_id)
        : dev_(dev) {
        if (layouts.size() != prims.size() || prim_to_texture.size() != prims.size() || prim_debug_id.size() != prims.size())
            throw Error(MTR_E_INVALID, "per-primitive arrays must have one entry per primitive");
        std::vector<mtr_texture*> th;
        for (const Texture* t : textures) th.push_back(t->handle());
        dev.check(mtr_model_create(dev.handle(), vertex_buf, vertex_len, index_buf, index_num, prims.data(), prims.size(),
                                   layouts.data(), prim_to_texture.data(), th.data(), th.size(), prim_debug_id.data(), &h_));
    }
    Model(const Model&) = delete;
    ~Model() { mtr_model_destroy(h_); }
    // Model::set_parts_disp(&mut self, parts_disp: &[bool]) -- src/model.rs:295
    void set_parts_disp(const std::vector<uint8_t>& parts_disp) {
        dev_.check(mtr_model_set_parts_disp(h_, parts_disp.data(), parts_disp.size()));
    }
    void set_palette(const float* mats, size_t n) { dev_.check(mtr_model_set_palette(h_, mats, n)); }
    // the state objects a material names (src/rmaterial.rs:211-230), applied per primitive; default = src/model.rs:240-262
    void set_prim_states(const std::vector<mtr_prim_state>& states) { dev_.check(mtr_model_set_prim_states(h_, states.data(), states.size())); }
    // joint positions for the per-joint debug cubes of Model::render (src/model.rs:309-315)
    void set_joint_positions(const float* xyz, size_t njoints) { dev_.check(mtr_model_set_joint_positions(h_, xyz, njoints)); }
    void render_with_joints(Frame& frame, const float view_proj[16]) const;
    // Model::render(&self, rpass, queue, transform_bind_group, debug_overlay) -- src/model.rs:299-305
    void render(Frame& frame, const float view_proj[16]) const;
    mtr_model* handle() const { return h_; }

  private:
    const Device& dev_;
    mtr_model* h_ = nullptr;
};

class Frame {
  public:
    // begin_render_pass, LoadOp::Clear(WHITE) / Clear(1.0) -- src/bin/modelviewer.rs:190-210
    Frame(const Device& dev, uint32_t width, uint32_t height, const float clear_rgba[4], float clear_depth) : dev_(dev) {
        dev.check(mtr_frame_begin(dev.handle(), width, height, clear_rgba, clear_depth, &h_));
    }
    // part r of a GroupFrame: the group frame owns (submits, waits for, destroys) it; draw into it, nothing else
    Frame(const Device& dev, mtr_frame* part_of_group_frame) : dev_(dev), h_(part_of_group_frame), owned_(false) {}
    Frame(const Frame&) = delete;
    ~Frame() {
        if (owned_) mtr_frame_destroy(h_);
    }
    void end() { dev_.check(mtr_frame_end(h_)); }  // queue.submit, src/renderer_app_manager.rs:185
    // end() in two halves, so a host can keep several frames in flight (the library overlaps them on its own streams)
    void submit() { dev_.check(mtr_frame_submit(h_)); }
    void wait() { dev_.check(mtr_frame_wait(h_)); }
    // multi-GPU: render only the bins with (bin % world) == rank, then pack them for the all-gather
    void set_shard(uint32_t rank, uint32_t world) { dev_.check(mtr_frame_set_shard(h_, rank, world)); }
    // ownership map of a sharded frame: MTR_OWN_BANDS (band_rows: world + 1 bin rows or nullptr), MTR_OWN_SUPERTILES (param), ...
    void set_shard_map(uint32_t rank, uint32_t world, uint32_t map, uint32_t param = 0, const uint32_t* band_rows = nullptr) {
        dev_.check(mtr_frame_set_shard_map(h_, rank, world, map, param, band_rows));
    }
    size_t shard_bytes() const { return mtr_frame_shard_bytes(h_); }
    void unpack_color_shards_on_stream(const void* gathered_dev, void* dst_dev, void* hip_stream) {
        dev_.check(mtr_frame_unpack_color_shards_on_stream(h_, gathered_dev, dst_dev, hip_stream));
    }
    void pack_color_shard(void* dst_dev, size_t dst_bytes) { dev_.check(mtr_frame_pack_color_shard(h_, dst_dev, dst_bytes)); }
    // hands the frame to the device's exchange thread, which owns (and destroys) it from here on
    void submit_exchange() {
        dev_.check(mtr_frame_submit_exchange(h_));
        h_ = nullptr;
    }
    void* color_devptr() const { return mtr_frame_color_devptr(h_); }
    void* depth_devptr() const { return mtr_frame_depth_devptr(h_); }
    void read_color(void* rgba8, size_t len) { dev_.check(mtr_frame_read_color(h_, rgba8, len)); }
    void read_depth(float* d, size_t count) { dev_.check(mtr_frame_read_depth(h_, d, count)); }
    mtr_frame_stats stats() {
        mtr_frame_stats s{};
        dev_.check(mtr_frame_get_stats(h_, &s));
        return s;
    }
    mtr_frame* handle() const { return h_; }
    const Device& device() const { return dev_; }

  private:
    const Device& dev_;
    mtr_frame* h_ = nullptr;
    bool owned_ = true;
};

inline void Model::render(Frame& frame, const float view_proj[16]) const {
    dev_.check(mtr_frame_draw_model(frame.handle(), h_, view_proj));
}

inline void Model::render_with_joints(Frame& frame, const float view_proj[16]) const {
    dev_.check(mtr_frame_draw_model_joints(frame.handle(), h_, view_proj));
}

// One host thread, N devices (mtr.h: mtr_group_*).  device(r) creates rank r's models / textures; a GroupFrame is one sharded
// Frame per rank -- draw into part(r) with rank r's objects -- and end() leaves the gathered RGBA8 image on rank 0's device.
class Group {
  public:
    explicit Group(const std::vector<int32_t>& hip_devices) {
        int32_t rc = mtr_group_create(hip_devices.data(), (int32_t)hip_devices.size(), &h_);
        if (rc) throw Error(rc, mtr_group_last_error(nullptr));
        for (int32_t r = 0; r < mtr_group_size(h_); r++) devs_.emplace_back(new Device(mtr_group_device(h_, r), Device::Borrowed{}));
    }
    Group(const Group&) = delete;
    Group& operator=(const Group&) = delete;
    ~Group() {
        devs_.clear();
        mtr_group_destroy(h_);
    }
    size_t size() const { return devs_.size(); }
    const Device& device(size_t rank) const { return *devs_.at(rank); }
    mtr_group* handle() const { return h_; }
    void check(int32_t rc) const {
        if (rc) throw Error(rc, mtr_group_last_error(h_));
    }

  private:
    mtr_group* h_ = nullptr;
    std::vector<std::unique_ptr<Device>> devs_;
};

class GroupFrame {
  public:
    GroupFrame(const Group& group, uint32_t width, uint32_t height, const float clear_rgba[4], float clear_depth,
               uint32_t map = MTR_OWN_BANDS, uint32_t param = 0, const uint32_t* band_rows = nullptr) : group_(group) {
        group.check(mtr_group_frame_begin(group.handle(), width, height, clear_rgba, clear_depth, map, param, band_rows, &h_));
        for (size_t r = 0; r < group.size(); r++) parts_.emplace_back(new Frame(group.device(r), mtr_group_frame_part(h_, (int32_t)r)));
    }
    GroupFrame(const GroupFrame&) = delete;
    ~GroupFrame() {
        parts_.clear();
        mtr_group_frame_destroy(h_);
    }
    Frame& part(size_t rank) { return *parts_.at(rank); }
    void end() { group_.check(mtr_group_frame_end(h_)); }
    void read_color(void* rgba8, size_t len) { group_.check(mtr_group_frame_read_color(h_, rgba8, len)); }
    void* color_devptr() const { return mtr_group_frame_color_devptr(h_); }

  private:
    const Group& group_;
    mtr_group_frame* h_ = nullptr;
    std::vector<std::unique_ptr<Frame>> parts_;
};

// The reference's app seam (src/renderer_app_manager.rs:14-32), headless: the "frame_view + encoder" pair is the Frame.
struct RendererApp {
    virtual ~RendererApp() = default;
    virtual void render(Frame& frame) = 0;
    virtual void post_render() {}
};

}  // namespace mtr
