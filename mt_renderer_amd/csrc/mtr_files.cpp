// mtr_files.cpp -- host-side readers of the MT Framework resource files that feed the draw path
// (include/mtr_files.h).  Every struct below restates an on-disk layout the
// reference declares with repr(C, packed); the static_asserts are the reference's own size tests
// (src/rmodel.rs:487-494, src/rtexture.rs:168-172, src/rshader2.rs:573-582, src/rmaterial.rs:317-322,
// src/rscheduler.rs:221-223).  Nothing here trusts an offset or a count read from the file: every access
// goes through Span::at(), which fails the parse instead of reading out of bounds (the reference panics).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mtr_files.h"

#include <zlib.h>

namespace {

thread_local char g_err[256] = "";

int32_t ferr(int32_t code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#pragma pack(push, 1)
struct ModelHdr {  // src/rmodel.rs:94-117
    uint32_t magic;
    uint16_t version, jnt_num, primitive_num, material_num;
    uint32_t vertex_num, index_num, polygon_num, vertexbuf_size, texture_num, parts_num, padding1;
    uint64_t joint_info, parts_info, material_info, primitive_info, vertex_data, index_data, rcn_data;
    float sphere[4];
    float box_min[4], box_max[4];
    int32_t middist, lowdist;
    uint32_t light_group;
    uint16_t memory, reserved;
};
struct JointInfo {  // src/rmodel.rs:253-258
    uint32_t bits;
    float radius, length, offset[3];
};
struct TextureHeader {  // src/rtexture.rs:43-48
    uint32_t magic, b4, b8, bc;
};
struct Shader2Header {  // src/rshader2.rs:16-28
    uint32_t magic;
    uint16_t major_version, minor_version;
    uint32_t shader_version, num_objects;
    uint64_t stringtable_offs, pobjects;
};
struct RawShader2Object {  // src/rshader2.rs:32-42
    uint64_t name_offs, sname_offs;
    uint32_t b10, b14, hash, padding1;
    uint64_t annotations;
};
struct RawShader2InputElement {  // src/rshader2.rs:64-68
    uint64_t name;
    uint32_t bits, padding1;
};
struct RawShader2InputLayout {  // src/rshader2.rs:181-186
    uint32_t b0, padding1;
    uint64_t pdefaultvalues;
};
struct RawShader2Struct {
    uint32_t b0, padding1;
    uint64_t members;
};
struct RawShader2Variable {
    uint64_t name;
    uint32_t b8, field_4;
    uint64_t sname;
    uint32_t b18, padding1;
    uint64_t annotations, pinitvalues;
};
struct RawShader2CBuffer {
    uint32_t b0, crc;
    uint64_t variables, pinitvalues;
};
struct MaterialHeader {  // src/rmaterial.rs:14-24
    uint32_t magic, version, material_num, texture_num, shader_version, padding1;
    uint64_t textures, materials;
};
struct RawTextureInfo {  // src/rmaterial.rs:28-37
    uint32_t dti_hash, padding;
    uint64_t ptex, plut;
    char path[128];
};
struct RawMaterialState {  // src/rmaterial.rs:68-76
    uint32_t b0, padding;
    uint64_t sh_value;
    uint32_t sh_crc, padding1;
};
struct RawMaterialInfo {  // src/rmaterial.rs:98-116
    uint32_t dti_hash, padding, name_hash, state_bufsize, bsstate, dsstate, rsstate, b1c, b20;
    float blend_factor[4];
    uint32_t animation_bufsize;
    uint64_t states, animation_list;
};
struct SchedulerTrack {  // src/rscheduler.rs:37-49
    uint32_t b0, field_4;
    uint64_t track_prop_name;
    uint32_t field_10, pad_14;
    uint64_t unit_group, key_frame, key_value;
};
struct ArchiveHeader {  // src/rarchive.rs:24-30
    uint32_t magic;
    uint16_t version, num_resources;
};
struct RawResourceInfo {  // src/rarchive.rs:32-43
    char path[128];
    uint32_t dti_type, size_compressed, orgsize_quality, offset;
};
struct SchedulerHeader {  // src/rscheduler.rs:67-79
    uint32_t magic;
    uint16_t version, track_num;
    uint32_t crc, bitfield_c, base_track, pad_14;
    uint64_t metadata;
};
#pragma pack(pop)

static_assert(sizeof(ModelHdr) == 0xa0, "ModelHdr");
static_assert(sizeof(mtr_primitive) == 0x38, "PrimitiveInfo");
static_assert(sizeof(JointInfo) == 24, "JointInfo");
static_assert(sizeof(TextureHeader) == 0x10, "TextureHeader");
static_assert(sizeof(Shader2Header) == 0x20, "Shader2Header");
static_assert(sizeof(RawShader2Object) == 0x28, "RawShader2Object");
static_assert(sizeof(RawShader2InputElement) == 0x10, "RawShader2InputElement");
static_assert(sizeof(RawShader2InputLayout) == 16, "RawShader2InputLayout");
static_assert(sizeof(RawShader2Struct) == 16, "RawShader2Struct");
static_assert(sizeof(RawShader2Variable) == 0x30, "RawShader2Variable");
static_assert(sizeof(RawShader2CBuffer) == 24, "RawShader2CBuffer");
static_assert(sizeof(MaterialHeader) == 0x28, "MaterialHeader");
static_assert(sizeof(RawTextureInfo) == 0x98, "RawTextureInfo");
static_assert(sizeof(RawMaterialInfo) == 0x48, "RawMaterialInfo");
static_assert(sizeof(RawMaterialState) == 0x18, "RawMaterialState");
static_assert(sizeof(SchedulerTrack) == 0x30, "SchedulerTrack");
static_assert(sizeof(SchedulerHeader) == 0x20, "SchedulerHeader");
static_assert(sizeof(ArchiveHeader) == 8, "ArchiveHeader");
static_assert(sizeof(RawResourceInfo) == 0x90, "RawResourceInfo");

constexpr size_t kPartsInfo = 0x20, kBoundaryInfo = 0x90, kMtMatrix = 64;

struct Span {
    const uint8_t* p;
    size_t n;
    // pointer to `count` items of `size` bytes at byte offset `off`, or nullptr when that leaves the file
    const uint8_t* at(uint64_t off, uint64_t count, uint64_t size) const {
        if (size != 0 && count > (UINT64_MAX / size)) return nullptr;
        const uint64_t bytes = count * size;
        if (off > n || bytes > n - off) return nullptr;
        return p + off;
    }
    // NUL-terminated string starting at `off`, or nullptr (CStr::from_bytes_until_nul failing in the reference)
    const char* cstr(uint64_t off) const {
        if (off >= n) return nullptr;
        return memchr(p + off, 0, n - off) ? reinterpret_cast<const char*>(p + off) : nullptr;
    }
};

template <class T>
bool rd(const Span& s, uint64_t off, T& out) {
    const uint8_t* q = s.at(off, 1, sizeof(T));
    if (!q) return false;
    memcpy(&out, q, sizeof(T));
    return true;
}

uint32_t crc_str(const char* s) { return mtr_crc32(reinterpret_cast<const uint8_t*>(s), strlen(s), 0xffffffffu); }

}  // namespace

struct mtr_rshader2 {
    struct Element {
        std::string name;
        uint32_t sindex, format, count, start, offset, instance;
    };
    struct Object {
        std::string name;
        uint32_t obj_type, name_hash, sindex, index;
        bool is_layout = false;
        uint32_t stride = 0;
        std::vector<Element> elements;
    };
    std::vector<Object> objects;
    std::unordered_map<uint32_t, uint32_t> by_hash;
};

struct mtr_rmaterial {
    std::vector<std::string> textures;
    std::vector<mtr_material_info> materials;
};

struct mtr_rscheduler {
    struct Track {
        uint32_t track_type, prop_type, key_num, parent, dti_or_prop;
        std::string name;
        std::vector<uint32_t> frames;     // raw key words: frame | mode << 24
        std::vector<uint64_t> values;     // decoded values (see mtr_rscheduler_key)
        std::vector<std::string> res;     // RESOURCE paths ("" + has_res false for null)
        std::vector<uint8_t> has_res;
        std::vector<float> fvalues;       // VECTOR (4 per key) / MATRIX (16 per key) tracks
        uint32_t fwidth = 0;
        bool decodable = false;
    };
    std::vector<Track> tracks;
};

extern "C" {

const char* mtr_files_last_error(void) { return g_err; }

size_t mtr_file_struct_size(uint32_t kind) {
    static const size_t sizes[] = {sizeof(ModelHdr), sizeof(mtr_primitive), kPartsInfo, kBoundaryInfo, sizeof(JointInfo),
                                   kMtMatrix, sizeof(TextureHeader), sizeof(Shader2Header), sizeof(RawShader2Object),
                                   sizeof(RawShader2InputElement), sizeof(RawShader2InputLayout), sizeof(RawShader2Struct),
                                   sizeof(RawShader2Variable), sizeof(RawShader2CBuffer), sizeof(MaterialHeader),
                                   sizeof(RawTextureInfo), sizeof(RawMaterialInfo), sizeof(RawMaterialState),
                                   sizeof(SchedulerTrack), sizeof(SchedulerHeader), sizeof(ArchiveHeader), sizeof(RawResourceInfo)};
    return kind < sizeof sizes / sizeof sizes[0] ? sizes[kind] : 0;
}

// ---------------------------------------------------------------------------------------------- rModel
int32_t mtr_rmodel_parse(const void* data, size_t len, mtr_rmodel_view* out) {
    if (!data || !out) return ferr(MTR_E_INVALID, "rModel: null argument");
    const Span s{static_cast<const uint8_t*>(data), len};
    ModelHdr h;
    uint32_t boundary_num;
    // the boundary count is the u32 that follows the header (src/rmodel.rs:308-310)
    if (!rd(s, 0, h) || !rd(s, sizeof(ModelHdr), boundary_num)) return ferr(MTR_E_INVALID, "rModel: truncated header");
    mtr_rmodel_view v{};
    v.magic = h.magic; v.version = h.version; v.jnt_num = h.jnt_num; v.primitive_num = h.primitive_num;
    v.material_num = h.material_num; v.vertex_num = h.vertex_num; v.index_num = h.index_num; v.polygon_num = h.polygon_num;
    v.vertexbuf_size = h.vertexbuf_size; v.texture_num = h.texture_num; v.parts_num = h.parts_num; v.boundary_num = boundary_num;
    memcpy(v.bounding_sphere, h.sphere, 16); memcpy(v.bounding_box_min, h.box_min, 16); memcpy(v.bounding_box_max, h.box_max, 16);

    if (!(v.material_names = s.at(h.material_info, h.material_num, 128))) return ferr(MTR_E_INVALID, "rModel: material names out of range");
    for (uint32_t i = 0; i < h.material_num; i++)
        if (!memchr(v.material_names + (size_t)i * 128, 0, 128)) return ferr(MTR_E_INVALID, "rModel: material name %u is not terminated", i);
    const uint8_t* prims = s.at(h.primitive_info, h.primitive_num, sizeof(mtr_primitive));
    if (!prims) return ferr(MTR_E_INVALID, "rModel: primitive array out of range");
    v.primitives = reinterpret_cast<const mtr_primitive*>(prims);
    // the boundary infos are read from where the primitive array ended (src/rmodel.rs:358-359)
    const uint64_t after_prims = h.primitive_info + (uint64_t)h.primitive_num * sizeof(mtr_primitive);
    if (!(v.boundary_infos = s.at(after_prims, boundary_num, kBoundaryInfo))) return ferr(MTR_E_INVALID, "rModel: boundary infos out of range");
    if (h.jnt_num) {  // joint infos, then lmats, imats, the 256-byte joint table, back to back (src/rmodel.rs:372-406)
        uint64_t o = h.joint_info;
        if (!(v.joint_infos = s.at(o, h.jnt_num, sizeof(JointInfo)))) return ferr(MTR_E_INVALID, "rModel: joint infos out of range");
        o += (uint64_t)h.jnt_num * sizeof(JointInfo);
        const uint8_t* lm = s.at(o, h.jnt_num, kMtMatrix);
        o += (uint64_t)h.jnt_num * kMtMatrix;
        const uint8_t* im = s.at(o, h.jnt_num, kMtMatrix);
        o += (uint64_t)h.jnt_num * kMtMatrix;
        if (!lm || !im || !(v.joint_table = s.at(o, 256, 1))) return ferr(MTR_E_INVALID, "rModel: joint matrices / table out of range");
        v.lmats = reinterpret_cast<const float*>(lm);
        v.imats = reinterpret_cast<const float*>(im);
    }
    if (!(v.parts = s.at(h.parts_info, h.parts_num, kPartsInfo))) return ferr(MTR_E_INVALID, "rModel: parts out of range");
    if (!(v.vertex_buf = s.at(h.vertex_data, h.vertexbuf_size, 1))) return ferr(MTR_E_INVALID, "rModel: vertex data out of range");
    const uint8_t* ib = s.at(h.index_data, h.index_num, 2);
    if (!ib) return ferr(MTR_E_INVALID, "rModel: index data out of range");
    v.index_buf = reinterpret_cast<const uint16_t*>(ib);
    // what Model::new indexes with file-provided numbers (it would panic): material_no, boundary_num
    for (uint32_t i = 0; i < h.primitive_num; i++) {
        mtr_primitive p;
        memcpy(&p, prims + (size_t)i * sizeof p, sizeof p);
        if (mtr_primitive_field(&p, MTR_PRIM_MATERIAL_NO) >= h.material_num)
            return ferr(MTR_E_INVALID, "rModel: primitive %u names material %u of %u", i, mtr_primitive_field(&p, MTR_PRIM_MATERIAL_NO), h.material_num);
        if (mtr_primitive_field(&p, MTR_PRIM_BOUNDARY_NUM) >= boundary_num)
            return ferr(MTR_E_INVALID, "rModel: primitive %u names boundary %u of %u", i, mtr_primitive_field(&p, MTR_PRIM_BOUNDARY_NUM), boundary_num);
    }
    *out = v;
    return MTR_OK;
}

uint32_t mtr_primitive_field(const mtr_primitive* prim, uint32_t field) {
    if (!prim) return 0;
    uint32_t w[14];
    memcpy(w, prim, sizeof w);  // views may be unaligned
    switch (field) {            // src/rmodel.rs:173-225
        case MTR_PRIM_VERTEX_NUM: return (w[0] >> 16) & 0xffff;
        case MTR_PRIM_PARTS_NO: return w[1] & 0xfff;
        case MTR_PRIM_MATERIAL_NO: return (w[1] >> 12) & 0xfff;
        case MTR_PRIM_WEIGHT_NUM: return (w[2] >> 3) & 0x1f;
        case MTR_PRIM_VERTEX_STRIDE: return (w[2] >> 16) & 0xff;
        case MTR_PRIM_TOPOLOGY: return (w[2] >> 24) & 0x3f;
        case MTR_PRIM_VERTEX_OFS: return w[3];
        case MTR_PRIM_VERTEX_BASE: return w[4];
        case MTR_PRIM_INPUTLAYOUT: return w[5];
        case MTR_PRIM_INDEX_OFS: return w[6];
        case MTR_PRIM_INDEX_NUM: return w[7];
        case MTR_PRIM_INDEX_BASE: return w[8];
        case MTR_PRIM_BOUNDARY_NUM: return (w[9] >> 8) & 0xff;
        default: return 0;
    }
}

int32_t mtr_rmodel_boundary_joint(const mtr_rmodel_view* m, uint32_t i, uint32_t* out) {
    if (!m || !out || i >= m->boundary_num) return ferr(MTR_E_INVALID, "rModel: boundary %u out of range", i);
    memcpy(out, m->boundary_infos + (size_t)i * kBoundaryInfo, 4);
    return MTR_OK;
}

int32_t mtr_rmodel_joint(const mtr_rmodel_view* m, uint32_t i, uint32_t* no, uint32_t* parent, uint32_t* symmetry, float offset[3]) {
    if (!m || i >= m->jnt_num || !m->joint_infos) return ferr(MTR_E_INVALID, "rModel: joint %u out of range", i);
    JointInfo j;
    memcpy(&j, m->joint_infos + (size_t)i * sizeof j, sizeof j);
    if (no) *no = j.bits & 0xff;
    if (parent) *parent = (j.bits >> 8) & 0xff;
    if (symmetry) *symmetry = (j.bits >> 16) & 0xff;
    if (offset) memcpy(offset, j.offset, 12);
    return MTR_OK;
}

// -------------------------------------------------------------------------------------------- rTexture
int32_t mtr_rtexture_parse(const void* data, size_t len, mtr_rtexture_view* out) {
    if (!data || !out) return ferr(MTR_E_INVALID, "rTexture: null argument");
    const Span s{static_cast<const uint8_t*>(data), len};
    TextureHeader h;
    if (!rd(s, 0, h)) return ferr(MTR_E_INVALID, "rTexture: truncated header");
    if (memcmp(&h.magic, "TEX\0", 4) != 0) return ferr(MTR_E_INVALID, "rTexture: bad magic %08x", h.magic);
    mtr_rtexture_view v{};
    v.version = h.b4 & 0xffff;
    v.prebias = (h.b4 >> 24) & 0xf;
    v.type = (h.b4 >> 28) & 0xf;
    v.level_count = h.b8 & 0x3f;
    v.width = ((h.b8 >> 6) & 0x1fff) << v.prebias;
    v.height = ((h.b8 >> 19) & 0x1fff) << v.prebias;
    v.array_count = h.bc & 0xff;
    v.format = (h.bc >> 8) & 0xff;
    if (v.type > 9) return ferr(MTR_E_INVALID, "rTexture: unknown texture type %u", v.type);
    if (v.type != 2) return ferr(MTR_E_INVALID, "rTexture: type %u is not TT_2D", v.type);
    const uint32_t num_images = v.array_count * v.level_count;  // src/rtexture.rs:111
    if (num_images == 0) return ferr(MTR_E_INVALID, "rTexture: no images (the reference indexes offsets[0])");
    if (!s.at(sizeof h, num_images, 8) || !rd(s, sizeof h, v.level0_offset)) return ferr(MTR_E_INVALID, "rTexture: truncated offset table");
    if (v.level0_offset > len) return ferr(MTR_E_INVALID, "rTexture: level 0 offset beyond the file");
    v.data = s.p + v.level0_offset;
    v.data_len = len - (size_t)v.level0_offset;
    *out = v;
    return MTR_OK;
}

int32_t mtr_texture_create_from_file_mips(mtr_device* dev, const void* data, size_t len, uint32_t max_levels, mtr_texture** out) {
    mtr_rtexture_view v;
    int32_t rc = mtr_rtexture_parse(data, len, &v);
    if (rc) return rc;
    if (v.format != MTR_TEX_RGBA8 && v.format != MTR_TEX_BC1 && v.format != MTR_TEX_BC7 && v.format != MTR_TEX_BC7_ALT)
        return ferr(MTR_E_UNSUPPORTED, "rTexture: unhandled texture format %u", v.format);
    uint32_t levels = std::min(std::max(max_levels, 1u), std::min(v.level_count, 15u));
    while (levels > 1 && (v.width >> (levels - 1)) == 0 && (v.height >> (levels - 1)) == 0) levels--;
    if (levels <= 1) {
        rc = mtr_texture_create(dev, v.width, v.height, v.format, v.data, v.data_len, out);
        if (rc) ferr(rc, "rTexture: %s", mtr_last_error(dev));
        return rc;
    }
    // gather the levels of array slice 0 through the offsets table (src/rtexture.rs:111-126): one u64 per image
    const Span s{static_cast<const uint8_t*>(data), len};
    std::vector<uint8_t> chain;
    for (uint32_t l = 0; l < levels; l++) {
        uint64_t off = 0;
        if (!rd(s, sizeof(TextureHeader) + (uint64_t)l * 8, off)) return ferr(MTR_E_INVALID, "rTexture: truncated offset table");
        const size_t lw = std::max(1u, v.width >> l), lh = std::max(1u, v.height >> l);
        const size_t nb = v.format == MTR_TEX_RGBA8 ? lw * lh * 4 : ((lw + 3) / 4) * ((lh + 3) / 4) * (v.format == MTR_TEX_BC1 ? 8 : 16);
        if (!s.at(off, nb, 1)) return ferr(MTR_E_INVALID, "rTexture: mip level %u lies outside the file", l);
        chain.insert(chain.end(), s.p + off, s.p + off + nb);
    }
    rc = mtr_texture_create_mips(dev, v.width, v.height, v.format, levels, chain.data(), chain.size(), out);
    if (rc) ferr(rc, "rTexture: %s", mtr_last_error(dev));
    return rc;
}

int32_t mtr_texture_create_from_file(mtr_device* dev, const void* data, size_t len, mtr_texture** out) {
    return mtr_texture_create_from_file_mips(dev, data, len, 1, out);
}

// -------------------------------------------------------------------------------------------- rShader2
int32_t mtr_rshader2_parse(const void* data, size_t len, mtr_rshader2** out) {
    if (!data || !out) return ferr(MTR_E_INVALID, "rShader2: null argument");
    const Span s{static_cast<const uint8_t*>(data), len};
    Shader2Header h;
    if (!rd(s, 0, h)) return ferr(MTR_E_INVALID, "rShader2: truncated header");
    if (h.magic != 0x58464d) return ferr(MTR_E_INVALID, "rShader2 magic incorrect: %08x", h.magic);
    if (h.num_objects == 0 || h.stringtable_offs > len) return ferr(MTR_E_INVALID, "rShader2: bad object count / string table");
    const Span strs{s.p + h.stringtable_offs, len - (size_t)h.stringtable_offs};
    // the object pointer array starts right after the header; index 0 is unused, so num_objects - 1 entries
    const uint32_t n = h.num_objects - 1;
    if (!s.at(sizeof h, n, 8)) return ferr(MTR_E_INVALID, "rShader2: object pointer array out of range");
    mtr_rshader2* sh = new (std::nothrow) mtr_rshader2;
    if (!sh) return ferr(MTR_E_NOMEM, "rShader2: out of memory");
    sh->objects.reserve(n);
    for (uint32_t i = 0; i < n; i++) {
        uint64_t ptr;
        RawShader2Object o;
        rd(s, sizeof h + (uint64_t)i * 8, ptr);
        if (!rd(s, ptr, o)) { delete sh; return ferr(MTR_E_INVALID, "rShader2: object %u out of range", i); }
        const char* name = o.name_offs ? strs.cstr(o.name_offs) : nullptr;
        if (!name) { delete sh; return ferr(MTR_E_INVALID, "rShader2: object %u has no name", i); }
        mtr_rshader2::Object ob;
        ob.name = name;
        ob.obj_type = o.b10 & 0x3f;
        if (ob.obj_type > 17) { delete sh; return ferr(MTR_E_INVALID, "rShader2: object %u has unknown type %u", i, ob.obj_type); }
        ob.sindex = o.b14 & 0xffff;
        ob.index = (o.b14 >> 16) & 0xffff;
        ob.name_hash = crc_str(name) & 0xfffff;  // src/rshader2.rs:343
        if (ob.obj_type == 9) {                  // OT_INPUTLAYOUT, src/rshader2.rs:403-450
            RawShader2InputLayout il;
            const uint64_t spec = ptr + sizeof o;
            if (!rd(s, spec, il)) { delete sh; return ferr(MTR_E_INVALID, "rShader2: input layout %u out of range", i); }
            const uint32_t count = il.b0 & 0xffff;
            ob.stride = (il.b0 >> 16) & 0xffff;
            ob.is_layout = true;
            if (!s.at(spec + sizeof il, count, sizeof(RawShader2InputElement))) { delete sh; return ferr(MTR_E_INVALID, "rShader2: input elements of object %u out of range", i); }
            for (uint32_t e = 0; e < count; e++) {
                RawShader2InputElement re;
                rd(s, spec + sizeof il + (uint64_t)e * sizeof re, re);
                const char* en = strs.cstr(re.name);
                const uint32_t fmt = (re.bits >> 6) & 0x1f;
                if (!en || fmt > 15) { delete sh; return ferr(MTR_E_INVALID, "rShader2: input element %u of object %u is malformed", e, i); }
                ob.elements.push_back({en, re.bits & 0x3f, fmt, (re.bits >> 11) & 0x7f, (re.bits >> 18) & 0xf, (re.bits >> 22) & 0x1ff, (re.bits >> 31) & 1});
            }
        }
        if (sh->by_hash.count(ob.name_hash)) {  // the reference asserts on a collision (src/rshader2.rs:467-475)
            const std::string other = sh->objects[sh->by_hash[ob.name_hash]].name;
            delete sh;
            return ferr(MTR_E_INVALID, "Shader Object name hash collision: %s and %s", name, other.c_str());
        }
        sh->by_hash[ob.name_hash] = i;
        sh->objects.push_back(std::move(ob));
    }
    *out = sh;
    return MTR_OK;
}

void mtr_rshader2_destroy(mtr_rshader2* sh) { delete sh; }
uint32_t mtr_rshader2_num_objects(const mtr_rshader2* sh) { return sh ? (uint32_t)sh->objects.size() : 0; }

int32_t mtr_rshader2_object(const mtr_rshader2* sh, uint32_t i, const char** name, uint32_t* obj_type, uint32_t* name_hash) {
    if (!sh || i >= sh->objects.size()) return ferr(MTR_E_INVALID, "rShader2: object %u out of range", i);
    if (name) *name = sh->objects[i].name.c_str();
    if (obj_type) *obj_type = sh->objects[i].obj_type;
    if (name_hash) *name_hash = sh->objects[i].name_hash;
    return MTR_OK;
}

int32_t mtr_rshader2_find(const mtr_rshader2* sh, uint32_t handle) {
    if (!sh) return -1;
    auto it = sh->by_hash.find((handle & 0xfffff000u) >> 12);
    return it == sh->by_hash.end() ? -1 : (int32_t)it->second;
}

int32_t mtr_rshader2_input_layout(const mtr_rshader2* sh, uint32_t i, uint32_t* stride, mtr_layout* layout, mtr_raw_element* raw,
                                  uint32_t raw_cap, uint32_t* raw_num) {
    if (!sh || i >= sh->objects.size()) return ferr(MTR_E_INVALID, "rShader2: object %u out of range", i);
    const mtr_rshader2::Object& o = sh->objects[i];
    if (!o.is_layout) return ferr(MTR_E_INVALID, "rShader2: object %s isn't an inputlayout", o.name.c_str());
    if (stride) *stride = o.stride;
    if (raw_num) *raw_num = (uint32_t)o.elements.size();
    for (uint32_t e = 0; raw && e < o.elements.size() && e < raw_cap; e++) {
        const auto& el = o.elements[e];
        raw[e] = {el.name.c_str(), el.sindex, el.format, el.count, el.start, el.offset, el.instance};
    }
    if (layout) {
        mtr_layout L{};
        for (const auto& el : o.elements) {
            int sem;
            if (el.name == "Position") sem = MTR_SEM_POSITION;       // src/rshader2.rs:503-507
            else if (el.name == "TexCoord") sem = MTR_SEM_TEXCOORD;
            else if (el.name == "Joint") sem = MTR_SEM_JOINT;        // skinning extension (SPEC.md)
            else if (el.name == "Weight") sem = MTR_SEM_WEIGHT;
            else continue;
            if (el.format == MTR_IEF_SCMP3N) continue;               // src/rshader2.rs:509-512
            if (L.num_elements == 8) return ferr(MTR_E_UNSUPPORTED, "rShader2: more than 8 bound elements in %s", o.name.c_str());
            mtr_element& d = L.elements[L.num_elements++];
            d.semantic = (uint8_t)sem; d.format = (uint8_t)el.format; d.count = (uint8_t)el.count; d.offset = (uint16_t)el.offset;
        }
        *layout = L;
    }
    return MTR_OK;
}

// ------------------------------------------------------------------------------------------- rMaterial
int32_t mtr_rmaterial_parse(const void* data, size_t len, const mtr_rshader2* sh, mtr_rmaterial** out) {
    if (!data || !out || !sh) return ferr(MTR_E_INVALID, "rMaterial: null argument");
    const Span s{static_cast<const uint8_t*>(data), len};
    MaterialHeader h;
    if (!rd(s, 0, h)) return ferr(MTR_E_INVALID, "rMaterial: truncated header");
    if (!s.at(h.textures, h.texture_num, sizeof(RawTextureInfo))) return ferr(MTR_E_INVALID, "rMaterial: texture infos out of range");
    if (!s.at(h.materials, h.material_num, sizeof(RawMaterialInfo))) return ferr(MTR_E_INVALID, "rMaterial: material infos out of range");
    mtr_rmaterial* m = new (std::nothrow) mtr_rmaterial;
    if (!m) return ferr(MTR_E_NOMEM, "rMaterial: out of memory");
    const uint32_t rtexture_hash = crc_str("rTexture") & 0x7fffffffu;  // DTI hash rule, src/dti.rs:174
    for (uint32_t i = 0; i < h.texture_num; i++) {
        RawTextureInfo ti;
        rd(s, h.textures + (uint64_t)i * sizeof ti, ti);
        if (!memchr(ti.path, 0, sizeof ti.path)) { delete m; return ferr(MTR_E_INVALID, "rMaterial: texture %u path is not terminated", i); }
        // the reference asserts the class is rTexture (src/rmaterial.rs:194)
        if (ti.dti_hash != rtexture_hash) { delete m; return ferr(MTR_E_INVALID, "rMaterial: texture %u has class %08x, not rTexture", i, ti.dti_hash); }
        m->textures.emplace_back(ti.path);
    }
    for (uint32_t i = 0; i < h.material_num; i++) {
        RawMaterialInfo mi;
        rd(s, h.materials + (uint64_t)i * sizeof mi, mi);
        mtr_material_info info{};
        info.name_hash = mi.name_hash; info.dti_hash = mi.dti_hash; info.albedo_texture = -1;
        info.bsstate = mi.bsstate; info.dsstate = mi.dsstate; info.rsstate = mi.rsstate;
        info.state_num = mi.b1c & 0xfff;
        memcpy(info.blend_factor, mi.blend_factor, 16);
        // the reference resolves (and unwraps) the three state handles for its log line (src/rmaterial.rs:218-229)
        if (mtr_rshader2_find(sh, mi.bsstate) < 0 || mtr_rshader2_find(sh, mi.dsstate) < 0 || mtr_rshader2_find(sh, mi.rsstate) < 0) {
            delete m;
            return ferr(MTR_E_INVALID, "rMaterial: material %u names a state object the shader package lacks", i);
        }
        if (!s.at(mi.states, info.state_num, sizeof(RawMaterialState))) { delete m; return ferr(MTR_E_INVALID, "rMaterial: states of material %u out of range", i); }
        for (uint32_t k = 0; k < info.state_num; k++) {
            RawMaterialState st;
            rd(s, mi.states + (uint64_t)k * sizeof st, st);
            const uint32_t type = st.b0 & 0xf;
            const int32_t obj = mtr_rshader2_find(sh, st.sh_crc);
            if (type > 4 || obj < 0) { delete m; return ferr(MTR_E_INVALID, "rMaterial: state %u of material %u is malformed", k, i); }
            if (type == 0 || type == 2) {  // FUNCTION / SAMPLER: the value is an object handle too (src/rmaterial.rs:250-263)
                if (st.sh_value > 0xffffffffull || mtr_rshader2_find(sh, (uint32_t)st.sh_value) < 0) {
                    delete m;
                    return ferr(MTR_E_INVALID, "rMaterial: state %u of material %u names an unknown object", k, i);
                }
            } else if (type == 3 && st.sh_value != 0) {  // TEXTURE: 1-based index into the texture list
                if (st.sh_value - 1 >= m->textures.size()) {
                    const size_t have = m->textures.size();
                    delete m;
                    return ferr(MTR_E_INVALID, "rMaterial: state %u of material %u names texture %llu of %zu", k, i, (unsigned long long)st.sh_value, have);
                }
                if (sh->objects[(size_t)obj].name == "tAlbedoMap") info.albedo_texture = (int32_t)(st.sh_value - 1);
            }
        }
        m->materials.push_back(info);
    }
    *out = m;
    return MTR_OK;
}

void mtr_rmaterial_destroy(mtr_rmaterial* m) { delete m; }
uint32_t mtr_rmaterial_num_textures(const mtr_rmaterial* m) { return m ? (uint32_t)m->textures.size() : 0; }
const char* mtr_rmaterial_texture_path(const mtr_rmaterial* m, uint32_t i) { return m && i < m->textures.size() ? m->textures[i].c_str() : nullptr; }
uint32_t mtr_rmaterial_num_materials(const mtr_rmaterial* m) { return m ? (uint32_t)m->materials.size() : 0; }

int32_t mtr_rmaterial_info(const mtr_rmaterial* m, uint32_t i, mtr_material_info* out) {
    if (!m || !out || i >= m->materials.size()) return ferr(MTR_E_INVALID, "rMaterial: material %u out of range", i);
    *out = m->materials[i];
    return MTR_OK;
}

int32_t mtr_rmaterial_find(const mtr_rmaterial* m, const char* name) {
    if (!m || !name) return -1;
    const uint32_t hsh = crc_str(name);
    for (size_t i = 0; i < m->materials.size(); i++)
        if (m->materials[i].name_hash == hsh) return (int32_t)i;
    return -1;
}

// ------------------------------------------------------------------------------------------ material state by name
int32_t mtr_state_from_names(const char* bs, const char* ds, const char* rs, mtr_prim_state* out) {
    if (!out) return 0;
    mtr_prim_state st = {MTR_BLEND_ALPHA, 1, 1, MTR_CULL_BACK};  // the reference's pipeline, src/model.rs:240-262
    int32_t known = 0;
    auto has = [](const std::string& n, const char* sub) { return n.find(sub) != std::string::npos; };
    auto ends = [](const std::string& n, const char* suf) { const size_t k = strlen(suf); return n.size() >= k && n.compare(n.size() - k, k, suf) == 0; };
    if (bs && *bs) {
        const std::string n(bs);
        if (has(n, "Add")) { st.blend = MTR_BLEND_ADD; known++; }
        else if (has(n, "Blend") || has(n, "Alpha")) { st.blend = MTR_BLEND_ALPHA; known++; }
        else if (n.compare(0, 2, "BS") == 0) { st.blend = MTR_BLEND_OFF; known++; }  // "BSSolid" and the other opaque states
    }
    if (ds && *ds) {
        const std::string n(ds);
        if (n.compare(0, 2, "DS") == 0) {
            st.depth_test = has(n, "ZTest") ? 1 : 0;
            st.depth_write = (has(n, "ZTestWrite") || has(n, "ZWrite")) ? 1 : 0;
            known++;
        }
    }
    if (rs && *rs) {
        const std::string n(rs);
        if (ends(n, "CN") || has(n, "CullNone") || has(n, "TwoSide")) { st.cull = MTR_CULL_NONE; known++; }
        else if (ends(n, "CF") || has(n, "CullFront")) { st.cull = MTR_CULL_FRONT; known++; }
        else if (n.compare(0, 2, "RS") == 0) { st.cull = MTR_CULL_BACK; known++; }
    }
    *out = st;
    return known;
}

int32_t mtr_model_states_from_files(const mtr_rmodel_view* model, const mtr_rshader2* sh, const mtr_rmaterial* mat, mtr_prim_state* states,
                                    size_t nstates) {
    if (!model || !sh || !mat || !states) return ferr(MTR_E_INVALID, "states from files: null argument");
    if (nstates != model->primitive_num) return ferr(MTR_E_INVALID, "states from files: one state per primitive");
    for (size_t p = 0; p < nstates; p++) {
        mtr_primitive pr;
        memcpy(&pr, reinterpret_cast<const uint8_t*>(model->primitives) + p * sizeof pr, sizeof pr);
        const uint32_t mno = mtr_primitive_field(&pr, MTR_PRIM_MATERIAL_NO);
        mtr_state_from_names(nullptr, nullptr, nullptr, &states[p]);
        if (mno >= model->material_num) return ferr(MTR_E_INVALID, "primitive %zu: material_no %u outside the name table", p, mno);
        const int32_t mi = mtr_rmaterial_find(mat, reinterpret_cast<const char*>(model->material_names + (size_t)mno * 128));
        if (mi < 0) continue;
        const mtr_material_info& info = mat->materials[(size_t)mi];
        const char* names[3] = {nullptr, nullptr, nullptr};
        const uint32_t handles[3] = {info.bsstate, info.dsstate, info.rsstate};
        for (int k = 0; k < 3; k++) {
            const int32_t oi = mtr_rshader2_find(sh, handles[k]);
            if (oi >= 0) names[k] = sh->objects[(size_t)oi].name.c_str();
        }
        mtr_state_from_names(names[0], names[1], names[2], &states[p]);
    }
    return MTR_OK;
}

// ------------------------------------------------------------------------------------------ rScheduler
int32_t mtr_rscheduler_parse(const void* data, size_t len, mtr_rscheduler** out) {
    if (!data || !out) return ferr(MTR_E_INVALID, "rScheduler: null argument");
    const Span s{static_cast<const uint8_t*>(data), len};
    SchedulerHeader h;
    if (!rd(s, 0, h)) return ferr(MTR_E_INVALID, "rScheduler: truncated header");
    if (memcmp(&h.magic, "SDL\0", 4) != 0) return ferr(MTR_E_INVALID, "rScheduler: bad magic %08x", h.magic);
    if (h.version != 0x16) return ferr(MTR_E_INVALID, "rScheduler: version %x, expected 16", h.version);
    if (!s.at(sizeof h, h.track_num, sizeof(SchedulerTrack))) return ferr(MTR_E_INVALID, "rScheduler: track array out of range");
    mtr_rscheduler* sc = new (std::nothrow) mtr_rscheduler;
    if (!sc) return ferr(MTR_E_NOMEM, "rScheduler: out of memory");
    auto fail = [&](const char* what, uint32_t i) { delete sc; return ferr(MTR_E_INVALID, "rScheduler: track %u: %s", i, what); };
    for (uint32_t i = 0; i < h.track_num; i++) {
        SchedulerTrack t;
        rd(s, sizeof h + (uint64_t)i * sizeof t, t);
        mtr_rscheduler::Track tr;
        tr.track_type = t.b0 & 0xff; tr.prop_type = (t.b0 >> 8) & 0xff; tr.key_num = (t.b0 >> 16) & 0xffff;
        tr.parent = t.field_4; tr.dti_or_prop = t.field_10;
        if (tr.track_type == 0 || tr.track_type == 4) return fail("track type the reference todo!()s", i);
        if (tr.track_type > 16) return fail("unknown track type", i);
        const char* name = (h.metadata <= len && t.track_prop_name <= len - h.metadata) ? s.cstr(h.metadata + t.track_prop_name) : nullptr;
        tr.name = name ? name : "";  // the reference only logs the Result
        if (tr.track_type >= 6) {    // keyed tracks (src/rscheduler.rs:128-205)
            if (!s.at(t.key_frame, tr.key_num, 4)) return fail("key frames out of range", i);
            tr.frames.resize(tr.key_num);
            if (tr.key_num) memcpy(tr.frames.data(), s.p + t.key_frame, (size_t)tr.key_num * 4);
            size_t vsz = 0;
            switch (tr.track_type) {
                case 11: vsz = 1; break;  // BOOL
                case 6: vsz = 4; break;   // INT
                case 9: vsz = 4; break;   // FLOAT
                case 13: vsz = 8; break;  // RESOURCE
                default: vsz = 0;         // todo!() in the reference once a key exists
            }
            if (tr.track_type == 8 || tr.track_type == 16) {  // VECTOR = MtVector4, MATRIX = MtMatrix (todo!() in the reference)
                tr.fwidth = tr.track_type == 8 ? 4 : 16;
                if (!s.at(t.key_value, tr.key_num, (uint64_t)tr.fwidth * 4)) return fail("key values out of range", i);
                tr.fvalues.resize((size_t)tr.key_num * tr.fwidth);
                if (tr.key_num) memcpy(tr.fvalues.data(), s.p + t.key_value, tr.fvalues.size() * 4);
            }
            if (vsz) {
                if (tr.track_type == 11 && tr.prop_type != 3) return fail("BOOL track whose property is not bool", i);   // asserts,
                if (tr.track_type == 9 && tr.prop_type != 12) return fail("FLOAT track whose property is not f32", i);   // :147,:165
                if (!s.at(t.key_value, tr.key_num, vsz)) return fail("key values out of range", i);
                tr.decodable = true;
                tr.values.resize(tr.key_num);
                tr.res.resize(tr.key_num);
                tr.has_res.assign(tr.key_num, 0);
                for (uint32_t k = 0; k < tr.key_num; k++) {
                    uint64_t v = 0;
                    memcpy(&v, s.p + t.key_value + (size_t)k * vsz, vsz);
                    if (tr.track_type == 13 && v != 0) {  // pointer into the metadata block: u32 class hash + path
                        uint32_t dti;
                        const uint64_t o = h.metadata + v;
                        const char* path = (v <= len && h.metadata <= len) ? s.cstr(o + 4) : nullptr;
                        if (!rd(s, o, dti) || !path) return fail("resource key out of range", i);
                        tr.res[k] = path;
                        tr.has_res[k] = 1;
                        v = dti;
                    }
                    tr.values[k] = v;
                }
            }
        }
        sc->tracks.push_back(std::move(tr));
    }
    *out = sc;
    return MTR_OK;
}

void mtr_rscheduler_destroy(mtr_rscheduler* s) { delete s; }
uint32_t mtr_rscheduler_num_tracks(const mtr_rscheduler* s) { return s ? (uint32_t)s->tracks.size() : 0; }

int32_t mtr_rscheduler_track(const mtr_rscheduler* s, uint32_t i, mtr_track_info* out) {
    if (!s || !out || i >= s->tracks.size()) return ferr(MTR_E_INVALID, "rScheduler: track %u out of range", i);
    const auto& t = s->tracks[i];
    *out = {t.track_type, t.prop_type, t.key_num, t.parent, t.dti_or_prop, t.name.c_str()};
    return MTR_OK;
}

int32_t mtr_rscheduler_key(const mtr_rscheduler* s, uint32_t track, uint32_t k, uint32_t* frame, uint32_t* mode, uint64_t* value_bits,
                           const char** resource) {
    if (!s || track >= s->tracks.size()) return ferr(MTR_E_INVALID, "rScheduler: track %u out of range", track);
    const auto& t = s->tracks[track];
    if (k >= t.frames.size()) return ferr(MTR_E_INVALID, "rScheduler: key %u of track %u out of range", k, track);
    if (frame) *frame = t.frames[k] & 0xffffff;
    if (mode) *mode = (t.frames[k] >> 24) & 0xff;
    if (!t.decodable) return ferr(MTR_E_UNSUPPORTED, "rScheduler: key values of track type %u are not decoded", t.track_type);
    if (value_bits) *value_bits = t.values[k];
    if (resource) *resource = t.has_res[k] ? t.res[k].c_str() : nullptr;
    return MTR_OK;
}

int32_t mtr_rscheduler_eval(const mtr_rscheduler* s, uint32_t track, uint32_t frame, uint64_t* value_bits) {
    if (!s || !value_bits || track >= s->tracks.size()) return ferr(MTR_E_INVALID, "rScheduler: track %u out of range", track);
    const auto& t = s->tracks[track];
    if (!t.decodable || t.track_type == 13) return ferr(MTR_E_UNSUPPORTED, "rScheduler: track type %u cannot be evaluated", t.track_type);
    int64_t best = -1;
    uint32_t best_frame = 0;
    for (size_t k = 0; k < t.frames.size(); k++) {
        const uint32_t f = t.frames[k] & 0xffffff;
        if (f <= frame && (best < 0 || f >= best_frame)) { best = (int64_t)k; best_frame = f; }
    }
    if (best < 0) return ferr(MTR_E_INVALID, "rScheduler: track %u has no key at or before frame %u", track, frame);
    *value_bits = t.values[(size_t)best];
    return MTR_OK;
}

static int64_t held_key(const mtr_rscheduler::Track& t, uint32_t frame) {
    int64_t best = -1;
    uint32_t best_frame = 0;
    for (size_t k = 0; k < t.frames.size(); k++) {
        const uint32_t f = t.frames[k] & 0xffffff;
        if (f <= frame && (best < 0 || f >= best_frame)) { best = (int64_t)k; best_frame = f; }
    }
    return best;
}

int32_t mtr_rscheduler_key_floats(const mtr_rscheduler* s, uint32_t track, uint32_t k, float out[16], uint32_t* n) {
    if (!s || !out || !n || track >= s->tracks.size()) return ferr(MTR_E_INVALID, "rScheduler: track %u out of range", track);
    const auto& t = s->tracks[track];
    if (k >= t.frames.size()) return ferr(MTR_E_INVALID, "rScheduler: key %u of track %u out of range", k, track);
    if (t.track_type == 9 && t.decodable) {
        const uint32_t bits = (uint32_t)t.values[k];
        memcpy(out, &bits, 4);
        *n = 1;
        return MTR_OK;
    }
    if (!t.fwidth) return ferr(MTR_E_UNSUPPORTED, "rScheduler: track type %u has no float keys", t.track_type);
    memcpy(out, &t.fvalues[(size_t)k * t.fwidth], (size_t)t.fwidth * 4);
    *n = t.fwidth;
    return MTR_OK;
}

int32_t mtr_rscheduler_eval_floats(const mtr_rscheduler* s, uint32_t track, uint32_t frame, float out[16], uint32_t* n) {
    if (!s || !out || !n || track >= s->tracks.size()) return ferr(MTR_E_INVALID, "rScheduler: track %u out of range", track);
    const int64_t k = held_key(s->tracks[track], frame);
    if (k < 0) return ferr(MTR_E_INVALID, "rScheduler: track %u has no key at or before frame %u", track, frame);
    return mtr_rscheduler_key_floats(s, track, (uint32_t)k, out, n);
}

int32_t mtr_rscheduler_find_track(const mtr_rscheduler* s, const char* name) {
    if (!s || !name) return -1;
    for (size_t i = 0; i < s->tracks.size(); i++)
        if (s->tracks[i].name == name) return (int32_t)i;
    return -1;
}

int32_t mtr_rscheduler_apply(const mtr_rscheduler* s, uint32_t frame, const mtr_sdl_binding* b, size_t nb, uint8_t* parts_disp, size_t nparts,
                             float* model_mats, size_t ninst) {
    if (!s || (!b && nb)) return ferr(MTR_E_INVALID, "rScheduler: null argument");
    for (size_t i = 0; i < nb; i++) {
        if (b[i].track >= s->tracks.size()) return ferr(MTR_E_INVALID, "binding %zu: track %u out of range", i, b[i].track);
        const auto& t = s->tracks[b[i].track];
        const int64_t k = held_key(t, frame);
        if (b[i].target == MTR_SDL_PARTS_DISP) {
            if ((t.track_type != 11 && t.track_type != 6) || !t.decodable) return ferr(MTR_E_INVALID, "binding %zu: parts_disp needs a BOOL / INT track", i);
            if (!parts_disp || b[i].index >= nparts) return ferr(MTR_E_INVALID, "binding %zu: parts_disp index %u out of range", i, b[i].index);
            if (k >= 0) parts_disp[b[i].index] = t.values[(size_t)k] != 0;
            continue;
        }
        if (!model_mats || b[i].index >= ninst) return ferr(MTR_E_INVALID, "binding %zu: instance %u out of range", i, b[i].index);
        float* M = model_mats + (size_t)b[i].index * 16;
        const uint32_t need = b[i].target == MTR_SDL_INSTANCE_MATRIX ? 16u : (b[i].target == MTR_SDL_INSTANCE_TRANSLATION ? 4u : 1u);
        if (b[i].target > MTR_SDL_INSTANCE_TRANSLATE_Z) return ferr(MTR_E_INVALID, "binding %zu: unknown target %u", i, b[i].target);
        const uint32_t have = t.track_type == 9 && t.decodable ? 1u : t.fwidth;
        if (have != need) return ferr(MTR_E_INVALID, "binding %zu: track type %u does not fit target %u", i, t.track_type, b[i].target);
        if (k < 0) continue;
        float v[16];
        uint32_t n = 0;
        const int32_t rc = mtr_rscheduler_key_floats(s, b[i].track, (uint32_t)k, v, &n);
        if (rc) return rc;
        if (need == 16) memcpy(M, v, 64);
        else if (need == 4) memcpy(M + 12, v, 12);
        else M[12 + (b[i].target - MTR_SDL_INSTANCE_TRANSLATE_X)] = v[0];
    }
    return MTR_OK;
}

// -------------------------------------------------------------------------------------------- skin palette
int32_t mtr_rmodel_joint_index(const mtr_rmodel_view* m, uint32_t no) {
    if (!m || !m->joint_table || no > 255) return -1;
    const uint8_t idx = m->joint_table[no];
    return (idx == 255 || idx >= m->jnt_num) ? -1 : (int32_t)idx;
}

static void mat4_mul_fma(const float* A, const float* B, float* out) {  // SPEC.md section 4: k-ordered fma chain from 0
    float r[16];
    for (int c = 0; c < 4; c++)
        for (int i = 0; i < 4; i++) {
            float a = 0.0f;
            for (int k = 0; k < 4; k++) a = std::fmaf(A[k * 4 + i], B[c * 4 + k], a);
            r[c * 4 + i] = a;
        }
    memcpy(out, r, sizeof r);
}

int32_t mtr_rmodel_palette(const mtr_rmodel_view* m, const float* local_mats, float* out, size_t cap) {
    if (!m || !out) return ferr(MTR_E_INVALID, "palette: null argument");
    const uint32_t n = m->jnt_num;
    if (n == 0 || !m->joint_infos || !m->lmats || !m->imats) return ferr(MTR_E_INVALID, "palette: the model has no joints");
    if (cap < n) return ferr(MTR_E_INVALID, "palette: room for %zu matrices, the model has %u joints", cap, n);
    std::vector<float> locals((size_t)n * 16), imats((size_t)n * 16), world((size_t)n * 16);
    memcpy(locals.data(), local_mats ? local_mats : m->lmats, locals.size() * 4);  // the file's arrays may be unaligned
    memcpy(imats.data(), m->imats, imats.size() * 4);
    std::vector<uint8_t> state(n, 0);  // 0 new, 1 on the stack, 2 done
    for (uint32_t j0 = 0; j0 < n; j0++) {
        // walk up to the first finished ancestor, then come back down
        std::vector<uint32_t> chain;
        uint32_t j = j0;
        while (state[j] != 2) {
            if (state[j] == 1) return ferr(MTR_E_INVALID, "palette: joint %u is its own ancestor", j);
            state[j] = 1;
            chain.push_back(j);
            uint32_t no, parent, sym;
            float off[3];
            mtr_rmodel_joint(m, j, &no, &parent, &sym, off);
            if (parent == 255 || parent == j) break;
            if (parent >= n) return ferr(MTR_E_INVALID, "palette: joint %u has parent %u of %u joints", j, parent, n);
            j = parent;
        }
        for (size_t k = chain.size(); k-- > 0;) {
            const uint32_t c = chain[k];
            uint32_t no, parent, sym;
            float off[3];
            mtr_rmodel_joint(m, c, &no, &parent, &sym, off);
            if (parent == 255 || parent == c) memcpy(&world[(size_t)c * 16], &locals[(size_t)c * 16], 64);
            else mat4_mul_fma(&world[(size_t)parent * 16], &locals[(size_t)c * 16], &world[(size_t)c * 16]);
            state[c] = 2;
        }
    }
    for (uint32_t j = 0; j < n; j++) mat4_mul_fma(&world[(size_t)j * 16], &imats[(size_t)j * 16], out + (size_t)j * 16);
    return MTR_OK;
}

// -------------------------------------------------------------------------------------------- rArchive
int32_t mtr_rarchive_parse(const void* data, size_t len, mtr_rarchive_view* out) {
    if (!data || !out) return ferr(MTR_E_INVALID, "rArchive: null argument");
    const Span s{static_cast<const uint8_t*>(data), len};
    ArchiveHeader h;
    if (!rd(s, 0, h)) return ferr(MTR_E_INVALID, "rArchive: truncated header");
    if (memcmp(&h.magic, "ARC\0", 4) != 0) return ferr(MTR_E_INVALID, "rArchive: bad magic %08x", h.magic);
    if (h.version != 7) return ferr(MTR_E_INVALID, "rArchive: version %u, expected 7", h.version);
    const uint8_t* table = s.at(sizeof h, h.num_resources, sizeof(RawResourceInfo));
    if (!table) return ferr(MTR_E_INVALID, "rArchive: resource table out of range");
    for (uint32_t i = 0; i < h.num_resources; i++)
        if (!memchr(table + (size_t)i * sizeof(RawResourceInfo), 0, 128)) return ferr(MTR_E_INVALID, "rArchive: path of resource %u is not terminated", i);
    *out = mtr_rarchive_view{h.num_resources, table, s.p, len};
    return MTR_OK;
}

int32_t mtr_rarchive_info(const mtr_rarchive_view* a, uint32_t i, mtr_resource_info* out) {
    if (!a || !out || i >= a->num_resources) return ferr(MTR_E_INVALID, "rArchive: resource %u out of range", i);
    const uint8_t* e = a->table + (size_t)i * sizeof(RawResourceInfo);
    RawResourceInfo r;
    memcpy(&r, e, sizeof r);
    out->path = reinterpret_cast<const char*>(e);
    out->dti_hash = r.dti_type;
    out->size_compressed = r.size_compressed;
    out->size_uncompressed = r.orgsize_quality & ((1u << 29) - 1u);  // ORGSIZE_MASK, src/rarchive.rs:19
    out->quality = (r.orgsize_quality >> 29) & 7u;
    out->offset = r.offset;
    return MTR_OK;
}

int32_t mtr_rarchive_find(const mtr_rarchive_view* a, const char* path, uint32_t dti_hash) {
    if (!a || !path) return -1;
    std::string want(path);
    for (char& c : want)
        if (c == '/') c = '\\';  // get_resource_with_path, src/rarchive.rs:138-141
    for (uint32_t i = 0; i < a->num_resources; i++) {
        mtr_resource_info ri;
        mtr_rarchive_info(a, i, &ri);
        if (ri.dti_hash == dti_hash && want == ri.path) return (int32_t)i;
    }
    return -1;
}

int32_t mtr_rarchive_extract(const mtr_rarchive_view* a, uint32_t i, void* out, size_t cap, size_t* out_len) {
    mtr_resource_info ri;
    int32_t rc = mtr_rarchive_info(a, i, &ri);
    if (rc) return rc;
    if (!out && ri.size_uncompressed) return ferr(MTR_E_INVALID, "rArchive: null output");
    if (cap < ri.size_uncompressed) return ferr(MTR_E_INVALID, "rArchive: output holds %zu bytes, resource %u needs %u", cap, i, ri.size_uncompressed);
    const Span s{a->file, a->file_len};
    const uint8_t* src = s.at(ri.offset, ri.size_compressed, 1);
    if (!src) return ferr(MTR_E_INVALID, "rArchive: compressed bytes of resource %u lie outside the file", i);
    uLongf n = (uLongf)cap;
    static uint8_t dummy;
    const int z = uncompress(static_cast<Bytef*>(out ? out : &dummy), &n, src, ri.size_compressed);  // zlib stream (flate2 ZlibDecoder)
    if (z != Z_OK) return ferr(MTR_E_INVALID, "rArchive: resource %u does not inflate (zlib %d)", i, z);
    if (n != ri.size_uncompressed) return ferr(MTR_E_INVALID, "rArchive: resource %u inflated to %lu bytes, table says %u", i, (unsigned long)n, ri.size_uncompressed);
    if (out_len) *out_len = n;
    return MTR_OK;
}

// ------------------------------------------------------------------------- Model::new from parsed files
int32_t mtr_model_create_from_files(mtr_device* dev, const mtr_rmodel_view* model, const mtr_rshader2* sh, const mtr_rmaterial* mat,
                                    mtr_texture* const* textures, size_t ntextures, mtr_model** out) {
    if (!dev || !model || !sh || !out) return ferr(MTR_E_INVALID, "Model::new: null argument");
    const size_t np = model->primitive_num;
    // mat_to_tex (src/model.rs:60-75): material name -> rMaterial entry -> albedo texture index
    std::vector<int32_t> mat_to_tex(model->material_num, -1);
    for (uint32_t i = 0; mat && i < model->material_num; i++) {
        const int32_t mi = mtr_rmaterial_find(mat, reinterpret_cast<const char*>(model->material_names + (size_t)i * 128));
        if (mi >= 0) mat_to_tex[i] = mat->materials[(size_t)mi].albedo_texture;
    }
    std::vector<mtr_primitive> prims(np);
    std::vector<mtr_layout> layouts(np);
    std::vector<int32_t> p2t(np);
    std::vector<uint32_t> dids(np);
    if (np) memcpy(prims.data(), model->primitives, np * sizeof(mtr_primitive));
    // textures the caller could not load stay NULL (Vec<Option<Texture>>, src/model.rs:46-58); only a primitive that
    // needs one of them is an error, so the model is created over the loaded ones
    std::vector<mtr_texture*> loaded;
    std::vector<int32_t> remap(ntextures, -1);
    for (size_t t = 0; textures && t < ntextures; t++)
        if (textures[t]) { remap[t] = (int32_t)loaded.size(); loaded.push_back(textures[t]); }
    for (size_t p = 0; p < np; p++) {
        const uint32_t handle = mtr_primitive_field(&prims[p], MTR_PRIM_INPUTLAYOUT);
        const int32_t oi = mtr_rshader2_find(sh, handle);
        if (oi < 0) return ferr(MTR_E_INVALID, "invalid inputlayout %08x", handle);  // panic at src/model.rs:182-183
        int32_t rc = mtr_rshader2_input_layout(sh, (uint32_t)oi, nullptr, &layouts[p], nullptr, 0, nullptr);
        if (rc) return rc;
        const int32_t t = mat_to_tex[mtr_primitive_field(&prims[p], MTR_PRIM_MATERIAL_NO)];
        if (t >= 0 && ((size_t)t >= ntextures || remap[(size_t)t] < 0))
            return ferr(MTR_E_INVALID, "no texture found! (primitive %zu wants texture %d)", p, t);  // expect() at src/model.rs:167
        p2t[p] = t >= 0 ? remap[(size_t)t] : -1;
        if ((rc = mtr_rmodel_boundary_joint(model, mtr_primitive_field(&prims[p], MTR_PRIM_BOUNDARY_NUM), &dids[p]))) return rc;
    }
    const int32_t rc = mtr_model_create(dev, model->vertex_buf, model->vertexbuf_size, model->index_buf, model->index_num, prims.data(), np,
                                        layouts.data(), p2t.data(), loaded.data(), loaded.size(), dids.data(), out);
    if (rc) return ferr(rc, "Model::new: %s", mtr_last_error(dev));
    if (model->jnt_num && model->joint_infos) {  // joint_positions, src/model.rs:283-291
        std::vector<float> pos((size_t)model->jnt_num * 3);
        for (uint32_t j = 0; j < model->jnt_num; j++) {
            uint32_t no, parent, sym;
            mtr_rmodel_joint(model, j, &no, &parent, &sym, &pos[(size_t)j * 3]);
        }
        mtr_model_set_joint_positions(*out, pos.data(), model->jnt_num);
    }
    return rc;
}

}  // extern "C"
