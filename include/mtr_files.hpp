// mtr_files.hpp -- C++ mirror of the reference's resource readers over include/mtr_files.h, with the reference's type
// names: ModelFile (src/rmodel.rs:295), TextureFile (src/rtexture.rs:80), Shader2File (src/rshader2.rs:246),
// MaterialFile (src/rmaterial.rs:172), SchedulerFile (src/rscheduler.rs:84), ArchiveFile (src/rarchive.rs:66);
// model_from_files = Model::new over them
// (src/model.rs:36-293).  Each reader owns a copy of the file bytes; errors (the reference's panics) throw mtr::Error.
#pragma once
#include "mtr.hpp"
#include "mtr_files.h"

#include <fstream>
#include <iterator>
#include <memory>

namespace mtr {

inline std::vector<uint8_t> read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error(MTR_E_INVALID, "cannot open " + path);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

inline void files_check(int32_t rc) {
    if (rc) throw Error(rc, mtr_files_last_error());
}

class ModelFile {
  public:
    explicit ModelFile(std::vector<uint8_t> bytes) : bytes_(std::move(bytes)) {
        files_check(mtr_rmodel_parse(bytes_.data(), bytes_.size(), &v_));
    }
    const mtr_rmodel_view& view() const { return v_; }
    std::string material_name(uint32_t i) const { return reinterpret_cast<const char*>(v_.material_names + (size_t)i * 128); }
    uint32_t primitive_field(uint32_t p, uint32_t field) const { return mtr_primitive_field(v_.primitives + p, field); }
    uint32_t boundary_joint(uint32_t i) const {
        uint32_t j = 0;
        files_check(mtr_rmodel_boundary_joint(&v_, i, &j));
        return j;
    }

  private:
    std::vector<uint8_t> bytes_;
    mtr_rmodel_view v_{};
};

class TextureFile {
  public:
    explicit TextureFile(std::vector<uint8_t> bytes) : bytes_(std::move(bytes)) {
        files_check(mtr_rtexture_parse(bytes_.data(), bytes_.size(), &v_));
    }
    uint32_t width() const { return v_.width; }
    uint32_t height() const { return v_.height; }
    uint32_t format() const { return v_.format; }
    const mtr_rtexture_view& view() const { return v_; }
    const std::vector<uint8_t>& bytes() const { return bytes_; }

  private:
    std::vector<uint8_t> bytes_;
    mtr_rtexture_view v_{};
};

class Shader2File {
  public:
    explicit Shader2File(const std::vector<uint8_t>& bytes) {
        mtr_rshader2* h = nullptr;
        files_check(mtr_rshader2_parse(bytes.data(), bytes.size(), &h));
        h_.reset(h);
    }
    mtr_rshader2* handle() const { return h_.get(); }
    uint32_t num_objects() const { return mtr_rshader2_num_objects(h_.get()); }
    std::string object_name(uint32_t i) const {
        const char* n = nullptr;
        files_check(mtr_rshader2_object(h_.get(), i, &n, nullptr, nullptr));
        return n;
    }
    int32_t get_object_by_handle(uint32_t handle) const { return mtr_rshader2_find(h_.get(), handle); }  // -1: none
    mtr_layout input_layout(uint32_t i, uint32_t* stride = nullptr) const {
        mtr_layout l{};
        files_check(mtr_rshader2_input_layout(h_.get(), i, stride, &l, nullptr, 0, nullptr));
        return l;
    }

  private:
    struct Del { void operator()(mtr_rshader2* p) const { mtr_rshader2_destroy(p); } };
    std::unique_ptr<mtr_rshader2, Del> h_;
};

class MaterialFile {
  public:
    MaterialFile(const std::vector<uint8_t>& bytes, const Shader2File& shader2) {
        mtr_rmaterial* h = nullptr;
        files_check(mtr_rmaterial_parse(bytes.data(), bytes.size(), shader2.handle(), &h));
        h_.reset(h);
    }
    mtr_rmaterial* handle() const { return h_.get(); }
    std::vector<std::string> textures() const {
        std::vector<std::string> out;
        for (uint32_t i = 0; i < mtr_rmaterial_num_textures(h_.get()); i++) out.emplace_back(mtr_rmaterial_texture_path(h_.get(), i));
        return out;
    }
    uint32_t num_materials() const { return mtr_rmaterial_num_materials(h_.get()); }
    mtr_material_info material(uint32_t i) const {
        mtr_material_info m{};
        files_check(mtr_rmaterial_info(h_.get(), i, &m));
        return m;
    }
    int32_t material_by_name(const std::string& name) const { return mtr_rmaterial_find(h_.get(), name.c_str()); }  // -1: none

  private:
    struct Del { void operator()(mtr_rmaterial* p) const { mtr_rmaterial_destroy(p); } };
    std::unique_ptr<mtr_rmaterial, Del> h_;
};

class SchedulerFile {
  public:
    explicit SchedulerFile(const std::vector<uint8_t>& bytes) {
        mtr_rscheduler* h = nullptr;
        files_check(mtr_rscheduler_parse(bytes.data(), bytes.size(), &h));
        h_.reset(h);
    }
    uint32_t num_tracks() const { return mtr_rscheduler_num_tracks(h_.get()); }
    mtr_track_info track(uint32_t i) const {
        mtr_track_info t{};
        files_check(mtr_rscheduler_track(h_.get(), i, &t));
        return t;
    }
    uint64_t eval(uint32_t track, uint32_t frame) const {
        uint64_t v = 0;
        files_check(mtr_rscheduler_eval(h_.get(), track, frame, &v));
        return v;
    }

  private:
    struct Del { void operator()(mtr_rscheduler* p) const { mtr_rscheduler_destroy(p); } };
    std::unique_ptr<mtr_rscheduler, Del> h_;
};

class ArchiveFile {
  public:
    explicit ArchiveFile(std::vector<uint8_t> bytes) : bytes_(std::move(bytes)) {
        files_check(mtr_rarchive_parse(bytes_.data(), bytes_.size(), &v_));
    }
    uint32_t num_resources() const { return v_.num_resources; }
    mtr_resource_info info(uint32_t i) const {
        mtr_resource_info r{};
        files_check(mtr_rarchive_info(&v_, i, &r));
        return r;
    }
    // ArchiveFile::get_resource_with_path (src/rarchive.rs:130-141): empty optional-like result = {false, {}}
    bool get_resource(const std::string& path, uint32_t dti_hash, std::vector<uint8_t>& out) const {
        const int32_t i = mtr_rarchive_find(&v_, path.c_str(), dti_hash);
        if (i < 0) return false;
        out.resize(info((uint32_t)i).size_uncompressed);
        size_t n = 0;
        files_check(mtr_rarchive_extract(&v_, (uint32_t)i, out.data(), out.size(), &n));
        return true;
    }

  private:
    std::vector<uint8_t> bytes_;
    mtr_rarchive_view v_{};
};

// Texture::new(device, queue, TextureFile) -- src/texture.rs:11
inline mtr_texture* texture_from_file(const Device& dev, const TextureFile& tf) {
    mtr_texture* t = nullptr;
    files_check(mtr_texture_create_from_file(dev.handle(), tf.bytes().data(), tf.bytes().size(), &t));
    return t;
}

// Model::new(model_file, material_file, shader2, ..) -- src/model.rs:36-45; textures[i] = rMaterial texture i or nullptr
inline mtr_model* model_from_files(const Device& dev, const ModelFile& model, const Shader2File& shader2, const MaterialFile* material,
                                   const std::vector<mtr_texture*>& textures) {
    mtr_model* m = nullptr;
    files_check(mtr_model_create_from_files(dev.handle(), &model.view(), shader2.handle(), material ? material->handle() : nullptr,
                                            textures.data(), textures.size(), &m));
    return m;
}

}  // namespace mtr
