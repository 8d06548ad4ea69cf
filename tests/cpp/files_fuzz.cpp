// AddressSanitizer / UBSan run of the resource-file parsers (host build only: GPU sanitizers are not available).
// Compiles csrc/mtr_files.cpp directly, with stand-ins for the few entry points of mtr.h it calls, and feeds it
// deterministic mutations of the valid files named on the command line.
// usage: files_fuzz <iterations> model.mod shader.mfx material.mrl texture.tex schedule.sdl archive.arc
#include "../../mt_renderer_amd/csrc/mtr_files.cpp"

#include <cstdlib>
#include <fstream>
#include <iterator>

extern "C" {
uint32_t mtr_crc32(const uint8_t* bytes, size_t len, uint32_t init) {  // src/util/crc.rs:36-50, bitwise form
    uint32_t v = init;
    for (size_t i = 0; i < len && bytes[i]; i++) {
        v ^= bytes[i];
        for (int k = 0; k < 8; k++) v = (v >> 1) ^ ((v & 1u) ? 0xEDB88320u : 0u);
    }
    return v;
}
const char* mtr_last_error(const mtr_device*) { return "stub"; }
int32_t mtr_model_set_joint_positions(mtr_model*, const float*, size_t) { return MTR_OK; }
int32_t mtr_texture_create(mtr_device*, uint32_t, uint32_t, uint32_t, const void*, size_t, mtr_texture**) { return MTR_E_HIP; }
// reads every byte it is handed, so a mip chain gathered from offsets outside the file would trip ASan
int32_t mtr_texture_create_mips(mtr_device*, uint32_t, uint32_t, uint32_t, uint32_t, const void* data, size_t len, mtr_texture**) {
    volatile uint8_t sink = 0;
    for (size_t i = 0; i < len; i++) sink ^= static_cast<const uint8_t*>(data)[i];
    return MTR_E_HIP;
}
int32_t mtr_model_create(mtr_device*, const void*, size_t, const uint16_t*, size_t, const mtr_primitive*, size_t, const mtr_layout*,
                         const int32_t*, mtr_texture* const*, size_t, const uint32_t*, mtr_model**) { return MTR_E_HIP; }
}

static std::vector<uint8_t> slurp(const char* p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {  // splitmix64
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static std::vector<uint8_t> mutate(std::vector<uint8_t> a) {
    if (a.empty()) return a;
    switch (rnd() % 4) {
        case 0:
            for (int i = 0, n = 1 + (int)(rnd() % 5); i < n; i++) a[rnd() % a.size()] = (uint8_t)rnd();
            break;
        case 1: {
            static const uint64_t big[] = {0xFFFFFFFFFFFFFFFFull, 0x7FFFFFFFull, 0x80000000ull, 0x100000000ull, 0ull};
            uint64_t v = big[rnd() % 5];
            if (rnd() % 3 == 0) v = a.size() - (rnd() % 3);
            const size_t off = (rnd() % (a.size() < 0x108 ? a.size() : 0x108)) & ~(size_t)3;
            for (size_t i = 0; i < 8 && off + i < a.size(); i++) a[off + i] = (uint8_t)(v >> (8 * i));
            break;
        }
        case 2: a.resize(rnd() % a.size()); break;
        default: {
            const size_t off = rnd() % a.size();
            for (size_t i = 0; i < 4 && off + i < a.size(); i++) a[off + i] = 0;
        }
    }
    return a;
}

int main(int argc, char** argv) {
    if (argc != 8) return 64;
    const long iters = strtol(argv[1], nullptr, 10);
    const std::vector<uint8_t> good[6] = {slurp(argv[2]), slurp(argv[3]), slurp(argv[4]), slurp(argv[5]), slurp(argv[6]), slurp(argv[7])};
    mtr_rshader2* sh_ok = nullptr;
    if (mtr_rshader2_parse(good[1].data(), good[1].size(), &sh_ok)) return 3;
    long ok = 0, err = 0;
    for (long it = 0; it < iters; it++) {
        const int which = (int)(rnd() % 6);
        const std::vector<uint8_t> b = it < 6 ? good[it] : mutate(good[which]);
        const int w = it < 6 ? (int)it : which;
        int32_t rc = 0;
        if (w == 0) {
            mtr_rmodel_view v;
            rc = mtr_rmodel_parse(b.data(), b.size(), &v);
            if (!rc) {
                for (uint32_t p = 0; p < v.primitive_num; p++) {
                    uint32_t j;
                    mtr_rmodel_boundary_joint(&v, mtr_primitive_field(v.primitives + p, MTR_PRIM_BOUNDARY_NUM), &j);
                }
                for (uint32_t j = 0; j < v.jnt_num; j++) mtr_rmodel_joint(&v, j, nullptr, nullptr, nullptr, nullptr);
                volatile uint8_t sink = 0;
                for (uint32_t i = 0; i < v.vertexbuf_size; i += 97) sink ^= v.vertex_buf[i];
                for (uint32_t i = 0; i < v.index_num; i += 31) {  // typed views may be unaligned: read them with memcpy
                    uint16_t ix;
                    memcpy(&ix, reinterpret_cast<const uint8_t*>(v.index_buf) + 2 * (size_t)i, 2);
                    sink ^= (uint8_t)ix;
                }
                mtr_model* m = nullptr;  // Model::new glue up to the (stubbed) device call
                static char fake_dev[8];
                mtr_model_create_from_files(reinterpret_cast<mtr_device*>(fake_dev), &v, sh_ok, nullptr, nullptr, 0, &m);
            }
        } else if (w == 1) {
            mtr_rshader2* sh = nullptr;
            rc = mtr_rshader2_parse(b.data(), b.size(), &sh);
            if (!rc) {
                for (uint32_t i = 0; i < mtr_rshader2_num_objects(sh); i++) {
                    mtr_layout l;
                    mtr_raw_element raw[8];
                    uint32_t n, stride;
                    mtr_rshader2_input_layout(sh, i, &stride, &l, raw, 8, &n);
                }
                mtr_rshader2_destroy(sh);
            }
        } else if (w == 2) {
            mtr_rmaterial* m = nullptr;
            rc = mtr_rmaterial_parse(b.data(), b.size(), sh_ok, &m);
            if (!rc) {
                for (uint32_t i = 0; i < mtr_rmaterial_num_materials(m); i++) {
                    mtr_material_info mi;
                    mtr_rmaterial_info(m, i, &mi);
                }
                mtr_rmaterial_destroy(m);
            }
        } else if (w == 3) {
            mtr_rtexture_view v;
            rc = mtr_rtexture_parse(b.data(), b.size(), &v);
            if (!rc && v.data_len) { volatile uint8_t s = v.data[0] ^ v.data[v.data_len - 1]; (void)s; }
            { mtr_texture* t = nullptr; (void)mtr_texture_create_from_file_mips(nullptr, b.data(), b.size(), 1 + (uint32_t)(it % 9), &t); }  // offsets table walk
        } else if (w == 5) {
            mtr_rarchive_view v;
            rc = mtr_rarchive_parse(b.data(), b.size(), &v);
            if (!rc) {
                std::vector<uint8_t> out;
                for (uint32_t i = 0; i < v.num_resources; i++) {
                    mtr_resource_info ri;
                    mtr_rarchive_info(&v, i, &ri);
                    mtr_rarchive_find(&v, ri.path, ri.dti_hash);
                    if (ri.size_uncompressed <= (1u << 22)) {
                        out.resize(ri.size_uncompressed ? ri.size_uncompressed : 1);
                        size_t n;
                        mtr_rarchive_extract(&v, i, out.data(), ri.size_uncompressed, &n);
                    }
                }
            }
        } else {
            mtr_rscheduler* sc = nullptr;
            rc = mtr_rscheduler_parse(b.data(), b.size(), &sc);
            if (!rc) {
                for (uint32_t t = 0; t < mtr_rscheduler_num_tracks(sc); t++) {
                    mtr_track_info ti;
                    mtr_rscheduler_track(sc, t, &ti);
                    for (uint32_t k = 0; k < ti.key_num; k++) {
                        uint32_t f, m;
                        uint64_t v;
                        const char* r;
                        mtr_rscheduler_key(sc, t, k, &f, &m, &v, &r);
                    }
                    uint64_t v;
                    mtr_rscheduler_eval(sc, t, 5, &v);
                }
                mtr_rscheduler_destroy(sc);
            }
        }
        if (it < 6 && rc) { std::fprintf(stderr, "valid file %ld rejected: %s\n", it, mtr_files_last_error()); return 4; }
        (rc ? err : ok)++;
    }
    mtr_rshader2_destroy(sh_ok);
    std::printf("ok=%ld err=%ld\n", ok, err);
    return 0;
}
