// AddressSanitizer / UBSan run of mtr_group_* (csrc/mtr_group.cpp) over the stand-in HIP runtime of tests/cpp/hip_stub:
// groups of 1..5 ranks (every rank on the stub's one device), frames of odd sizes under every ownership map.  The stub
// pack writes a byte derived from the BIN ID into each 1 KiB block of a rank's own bins and the stub unpack spreads the
// byte it finds at src_of_bin[b] over bin b, so the gathered image is right exactly when every rank packed its own bins,
// the group copied each shard to its rank's offset, and the tables agree -- and every copy is bounds-checked by ASan.
// usage: group_asan <rounds>
#include "../../mt_renderer_amd/csrc/mtr_api.cpp"
#include "../../mt_renderer_amd/csrc/mtr_group.cpp"

void mtr_launch_geom(const GeomParams&, hipStream_t) {}
void mtr_launch_scan(const FrameBuffers&, hipStream_t) {}
void mtr_launch_fill(const FrameBuffers&, uint32_t, hipStream_t) {}
static long g_tiles = 0, g_reruns = 0;
static void stub_status(const TileParams& p) {
    uint32_t flags = 0;
    if (p.fb.direct && g_tiles++ % 11 == 3) flags = 4u;  // a bounded bin queue "filled up": mtr_frame_wait re-runs the part
    if (!p.fb.direct) g_reruns++;
    if (p.host_status) __atomic_store_n(p.host_status, 0x80000000u | flags, __ATOMIC_RELEASE);
}
void mtr_launch_tile(const TileParams& p, bool, hipStream_t) { stub_status(p); }
void mtr_launch_tile_vis(const TileParams& p, bool, hipStream_t) { stub_status(p); }
void mtr_launch_alpha_min(const uint8_t*, size_t, uint32_t* out_min, hipStream_t) { *out_min = 255; }
void mtr_launch_vertex_stage(const GeomParams&, uint32_t, float*, float*, hipStream_t) {}
void mtr_launch_bc1_decode(const uint8_t*, uint8_t*, uint32_t, uint32_t, hipStream_t) {}
void mtr_launch_bc7_decode(const uint8_t*, uint8_t*, uint32_t, uint32_t, hipStream_t) {}
static uint8_t bin_byte(uint32_t b) { return (uint8_t)(b * 7u + 1u); }
void mtr_launch_pack_shard(const uint8_t*, uint8_t* dst, uint32_t, uint32_t, const uint32_t* own_list, uint32_t n, uint32_t stride_bins, hipStream_t) {
    for (uint32_t k = 0; k < stride_bins; k++) memset(dst + (size_t)k * 1024, k < n ? bin_byte(own_list[k]) : 0, 1024);
}
void mtr_launch_unpack_shards(const uint8_t* g, uint8_t* dst, uint32_t W, uint32_t H, const uint32_t* src_of_bin, hipStream_t) {
    const uint32_t nbx = (W + 15) / 16, nby = (H + 15) / 16;
    for (uint32_t b = 0; b < nbx * nby; b++) {
        const uint8_t v = g[(size_t)src_of_bin[b] * 1024];
        for (uint32_t y = b / nbx * 16; y < std::min(H, b / nbx * 16 + 16); y++)
            for (uint32_t x = b % nbx * 16; x < std::min(W, b % nbx * 16 + 16); x++) memset(dst + ((size_t)y * W + x) * 4, v, 4);
    }
}
void mtr_launch_cull_instances(const CullParams& p, hipStream_t) { for (uint32_t i = 0; i < p.ninst; i++) { p.strad[p.count[1]++] = *p.count; p.list[(*p.count)++] = i; } }
void mtr_launch_cull_chunks(const ChunkCullParams& p, hipStream_t) { if (p.nchunks && p.ninst) p.work_mask[(size_t)((p.nchunks + 15) / 16) * p.ninst - 1] = 0xFFFF; }

#define REQ(x) do { if ((x) != MTR_OK) { fprintf(stderr, "%s failed: %s\n", #x, mtr_group_last_error(g)); return 4; } } while (0)
#define MUSTFAIL(x) do { if ((x) == MTR_OK) { fprintf(stderr, "%s should have failed\n", #x); return 5; } } while (0)

int main(int argc, char** argv) {
    const long rounds = argc > 1 ? strtol(argv[1], nullptr, 10) : 40;
    const float verts[9] = {-0.5f, -0.5f, 0.5f, 0.5f, -0.5f, 0.5f, 0.0f, 0.5f, 0.5f};
    const uint16_t idx[3] = {0, 1, 2};
    mtr_primitive pr;
    memset(&pr, 0, sizeof pr);
    pr.w[0] = 3u << 16; pr.w[2] = 1 | (12u << 16) | (3u << 24); pr.w[7] = 3;
    mtr_layout l;
    memset(&l, 0, sizeof l);
    l.elements[l.num_elements++] = mtr_element{MTR_SEM_POSITION, MTR_IEF_F32, 3, 0, 0, 0};
    const float clear[4] = {1, 1, 1, 1}, M[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    mtr_group* g = nullptr;
    const int32_t none[1] = {0}, bad[2] = {0, 3};
    MUSTFAIL(mtr_group_create(none, 0, &g));
    MUSTFAIL(mtr_group_create(bad, 2, &g));  // the stub has one device
    MUSTFAIL(mtr_group_create(nullptr, 2, &g));
    long frames = 0, pixels = 0;
    for (long round = 0; round < rounds; round++) {
        const int32_t world = 1 + (int32_t)(round % 5);
        const int32_t devs[5] = {0, 0, 0, 0, 0};
        REQ(mtr_group_create(devs, world, &g));
        if (mtr_group_size(g) != world || mtr_group_device(g, world) || mtr_group_device(g, -1) || !mtr_group_device(g, world - 1)) return 6;
        std::vector<mtr_model*> models((size_t)world, nullptr);
        for (int32_t r = 0; r < world; r++)
            if (mtr_model_create(mtr_group_device(g, r), verts, sizeof verts, idx, 3, &pr, 1, &l, nullptr, nullptr, 0, nullptr, &models[(size_t)r])) return 7;
        mtr_group_frame* prev = nullptr;
        for (int k = 0; k < 6; k++) {
            // sizes shrink and grow: the group's send / gathered / image buffers are reallocated while a frame is alive
            const uint32_t W = 17 + (uint32_t)((round * 37 + k * 101) % 300), H = 16 + (uint32_t)((round * 53 + k * 67) % 200);
            const uint32_t nby = (H + 15) / 16;
            const uint32_t map = (uint32_t)k % 3;
            std::vector<uint32_t> bands;
            const uint32_t* bp = nullptr;
            if (map == MTR_OWN_BANDS && k >= 3 && nby >= (uint32_t)world) {  // explicit uneven bands, some of them empty
                bands.assign((size_t)world + 1, nby);
                bands[0] = 0;
                for (int32_t r = 1; r < world; r++) bands[(size_t)r] = std::min(nby, (uint32_t)r * (nby / (uint32_t)world) / 2);
                bp = bands.data();
            }
            mtr_group_frame* gf = nullptr;
            MUSTFAIL(mtr_group_frame_begin(g, W, H, clear, 1.0f, 9, 0, nullptr, &gf));  // no such map
            MUSTFAIL(mtr_group_frame_begin(g, 0, H, clear, 1.0f, map, 1, bp, &gf));
            REQ(mtr_group_frame_begin(g, W, H, clear, 1.0f, map, 1, bp, &gf));
            if (mtr_group_frame_part(gf, world) || mtr_group_frame_part(gf, -1)) return 8;
            std::vector<uint8_t> img((size_t)W * H * 4);
            MUSTFAIL(mtr_group_frame_read_color(gf, img.data(), img.size()));  // not ended
            for (int32_t r = 0; r < world; r++)
                if (mtr_frame_draw_model(mtr_group_frame_part(gf, r), models[(size_t)r], M)) return 9;
            REQ(mtr_group_frame_end(gf));
            MUSTFAIL(mtr_group_frame_end(gf));
            MUSTFAIL(mtr_group_frame_read_color(gf, img.data(), img.size() - 1));
            REQ(mtr_group_frame_read_color(gf, img.data(), img.size()));
            if (!mtr_group_frame_color_devptr(gf)) return 10;
            const uint32_t nbx = (W + 15) / 16;
            for (uint32_t y = 0; y < H; y++)
                for (uint32_t x = 0; x < W; x++)
                    if (img[((size_t)y * W + x) * 4] != bin_byte(y / 16 * nbx + x / 16)) {
                        fprintf(stderr, "world %d map %u %ux%u: pixel (%u, %u) came from the wrong shard\n", world, map, W, H, x, y);
                        return 11;
                    }
            pixels += (long)W * H;
            if (prev) {
                MUSTFAIL(mtr_group_frame_read_color(prev, img.data(), img.size()));  // a later group frame has ended
                if (mtr_group_frame_color_devptr(prev)) return 12;
                mtr_group_frame_destroy(prev);
            }
            prev = gf;
            frames++;
        }
        mtr_group_frame_destroy(prev);
        mtr_group_frame_destroy(nullptr);
        for (mtr_model* m : models) mtr_model_destroy(m);
        mtr_group_destroy(g);
        g = nullptr;
    }
    mtr_group_destroy(nullptr);
    printf("frames=%ld pixels=%ld reruns=%ld\n", frames, pixels, g_reruns);
    return 0;
}
