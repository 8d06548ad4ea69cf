// AddressSanitizer / UBSan run of the HOST side of the library (csrc/mtr_api.cpp + csrc/mtr_files.cpp) over a stand-in
// HIP runtime (tests/cpp/hip_stub): model / texture / batch creation with random, mostly malformed arguments, frames
// drawn, submitted, waited and read back.  The kernels are no-ops here; what is exercised is everything the host does
// with caller-provided sizes, offsets and indices before a kernel may trust them.   usage: host_fuzz <iterations>
#include "../../mt_renderer_amd/csrc/mtr_api.cpp"
#include "../../mt_renderer_amd/csrc/mtr_files.cpp"

// kernel launchers: inert, except the ones whose output the host reads back
void mtr_launch_geom(const GeomParams&, hipStream_t) {}
void mtr_launch_scan(const FrameBuffers&, hipStream_t) {}
void mtr_launch_fill(const FrameBuffers&, uint32_t, hipStream_t) {}
// what tile_prologue does on the device: publish the (clean) overflow flags of the frame
static void stub_status(const TileParams& p) { if (p.host_status) __atomic_store_n(p.host_status, 0x80000000u, __ATOMIC_RELEASE); }
void mtr_launch_tile(const TileParams& p, bool, hipStream_t) { stub_status(p); }
void mtr_launch_tile_vis(const TileParams& p, bool, hipStream_t) { stub_status(p); }
void mtr_launch_alpha_min(const uint8_t* rgba, size_t npixels, uint32_t* out_min, hipStream_t) {
    uint32_t m = 255;
    for (size_t i = 0; i < npixels; i++) m = std::min<uint32_t>(m, rgba[4 * i + 3]);  // reads every texel: ASan checks the size
    *out_min = m;
}
void mtr_launch_vertex_stage(const GeomParams&, uint32_t, float*, float*, hipStream_t) {}
void mtr_launch_bc1_decode(const uint8_t* b, uint8_t* rgba, uint32_t w, uint32_t h, hipStream_t) {
    memset(rgba, b[(size_t)((w + 3) / 4) * ((h + 3) / 4) * 8 - 1], (size_t)w * h * 4);  // touches the last block byte
}
void mtr_launch_bc7_decode(const uint8_t* b, uint8_t* rgba, uint32_t w, uint32_t h, hipStream_t) {
    memset(rgba, b[(size_t)((w + 3) / 4) * ((h + 3) / 4) * 16 - 1], (size_t)w * h * 4);
}
void mtr_launch_pack_shard(const uint8_t*, uint8_t*, uint32_t, uint32_t, const uint32_t*, uint32_t, uint32_t, hipStream_t) {}
void mtr_launch_unpack_shards(const uint8_t*, uint8_t*, uint32_t, uint32_t, const uint32_t*, hipStream_t) {}
void mtr_launch_cull_instances(const CullParams& p, hipStream_t) { for (uint32_t i = 0; i < p.ninst; i++) { p.strad[p.count[1]++] = *p.count; p.list[(*p.count)++] = i; } }
void mtr_launch_cull_chunks(const ChunkCullParams& p, hipStream_t) { if (p.nchunks && p.ninst) p.work_mask[(size_t)((p.nchunks + 15) / 16) * p.ninst - 1] = 0xFFFF; }  // the last mask of the draw

static uint64_t rs = 0x243F6A8885A308D3ull;
static uint64_t rnd() {
    uint64_t z = (rs += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint32_t pick(uint32_t lo, uint32_t hi) { return lo + (uint32_t)(rnd() % (hi - lo + 1)); }
static uint32_t weird() {  // mostly sane, sometimes hostile
    switch (rnd() % 8) {
        case 0: return 0xFFFFFFFFu;
        case 1: return 0x7FFFFFFFu;
        case 2: return (uint32_t)rnd();
        default: return (uint32_t)(rnd() % 64);
    }
}

int main(int argc, char** argv) {
    const long iters = argc > 1 ? strtol(argv[1], nullptr, 10) : 2000;
    mtr_device* dev = nullptr;
    if (mtr_device_create(0, &dev)) return 3;
    long created = 0, rejected = 0;
    for (long it = 0; it < iters; it++) {
        // ---- textures ----
        std::vector<mtr_texture*> texs;
        for (int t = 0, nt = (int)(rnd() % 3); t < nt; t++) {
            const uint32_t w = rnd() % 5 ? pick(1, 40) : weird(), h = rnd() % 5 ? pick(1, 40) : weird();
            static const uint32_t fmts[] = {MTR_TEX_RGBA8, MTR_TEX_BC1, MTR_TEX_BC7, MTR_TEX_BC7_ALT, 3, 255};
            const uint32_t fmt = fmts[rnd() % 6];
            const size_t need = fmt == MTR_TEX_RGBA8 ? (size_t)w * h * 4 : (size_t)((w + 3) / 4) * ((h + 3) / 4) * (fmt == MTR_TEX_BC1 ? 8 : 16);
            size_t len = need < (1u << 20) ? need : 64;
            if (rnd() % 4 == 0) len = len ? len - 1 - (size_t)(rnd() % len) : 0;  // short buffer
            std::vector<uint8_t> data(len ? len : 1, (uint8_t)rnd());
            mtr_texture* tx = nullptr;
            if (mtr_texture_create(dev, w, h, fmt, data.data(), len, &tx) == MTR_OK) texs.push_back(tx);
        }
        // ---- model ----
        const uint32_t np = pick(0, 4);
        const size_t vlen = rnd() % 6 ? pick(0, 4000) : 0, inum = rnd() % 6 ? pick(0, 600) : 0;
        std::vector<uint8_t> vb(vlen ? vlen : 1);
        std::vector<uint16_t> ib(inum ? inum : 1);
        for (auto& x : vb) x = (uint8_t)rnd();
        for (auto& x : ib) x = rnd() % 7 ? (uint16_t)(rnd() % 300) : (uint16_t)0xFFFF;
        std::vector<mtr_primitive> prims(np ? np : 1);
        std::vector<mtr_layout> lays(np ? np : 1);
        std::vector<int32_t> p2t(np ? np : 1);
        std::vector<uint32_t> dids(np ? np : 1);
        for (uint32_t p = 0; p < np; p++) {
            // a VALID primitive first ...
            mtr_primitive& pr = prims[p];
            const bool uv = rnd() % 2, skin = rnd() % 2;
            const uint32_t stride = 12 + (uv ? 4 : 0) + (skin ? 8 : 0);
            const uint32_t vbase = vlen >= 4 ? (pick(0, (uint32_t)vlen / 4) & ~3u) : 0;
            const uint32_t vnum = (uint32_t)std::min<size_t>((vlen - vbase) / stride, 200);
            const uint32_t iofs = inum ? pick(0, (uint32_t)inum - 1) : 0, icnt = inum ? pick(0, (uint32_t)inum - iofs) : 0;
            memset(&pr, 0, sizeof pr);
            pr.w[0] = vnum << 16;
            pr.w[1] = p | (p << 12);
            pr.w[2] = 1 | (stride << 16) | ((rnd() % 2 ? 4u : 3u) << 24);
            pr.w[4] = vbase; pr.w[6] = iofs; pr.w[7] = icnt;
            mtr_layout& l = lays[p];
            memset(&l, 0, sizeof l);
            uint32_t off = 12;
            l.elements[l.num_elements++] = mtr_element{MTR_SEM_POSITION, MTR_IEF_F32, 3, 0, 0, 0};
            if (uv) { l.elements[l.num_elements++] = mtr_element{MTR_SEM_TEXCOORD, MTR_IEF_F16, 2, 0, (uint16_t)off, 0}; off += 4; }
            if (skin) {
                l.elements[l.num_elements++] = mtr_element{MTR_SEM_JOINT, MTR_IEF_U8, 4, 0, (uint16_t)off, 0};
                l.elements[l.num_elements++] = mtr_element{MTR_SEM_WEIGHT, MTR_IEF_U8N, 4, 0, (uint16_t)(off + 4), 0};
            }
            p2t[p] = (!texs.empty() && rnd() % 2) ? (int32_t)(rnd() % texs.size()) : -1;
            dids[p] = (uint32_t)rnd();
            // ... then, half of the time, one hostile field
            if (rnd() % 2) {
                switch (rnd() % 12) {
                    case 0: pr.w[0] = weird() << 16; break;                                  // vertex_num
                    case 1: pr.w[1] = weird(); break;                                        // parts_no / material_no
                    case 2: pr.w[2] = weird(); break;                                        // stride / topology
                    case 3: pr.w[4] = weird(); break;                                        // vertex_base
                    case 4: pr.w[6] = weird(); break;                                        // index_ofs
                    case 5: pr.w[7] = weird(); break;                                        // index_num
                    case 6: pr.w[8] = weird(); break;                                        // index_base
                    case 7: l.num_elements = weird(); break;
                    case 8: l.elements[rnd() % 8].offset = (uint16_t)weird(); break;
                    case 9: { mtr_element& e = l.elements[rnd() % 8]; e.format = (uint8_t)(rnd() % 20); e.count = (uint8_t)(rnd() % 9); break; }
                    case 10: l.elements[rnd() % 8].semantic = (uint8_t)(rnd() % 6); break;
                    default: p2t[p] = (int32_t)weird();
                }
            }
        }
        mtr_model* model = nullptr;
        const int32_t rc = mtr_model_create(dev, vb.data(), vlen, ib.data(), inum, prims.data(), np, lays.data(), rnd() % 5 ? p2t.data() : nullptr,
                                            texs.empty() ? nullptr : texs.data(), texs.size(), rnd() % 5 ? dids.data() : nullptr, &model);
        if (rc != MTR_OK) {
            rejected++;
        } else {
            created++;
            std::vector<uint8_t> disp(pick(0, 6), (uint8_t)(rnd() & 1));
            mtr_model_set_parts_disp(model, disp.data(), disp.size());
            std::vector<float> pal((size_t)pick(0, 70) * 16, 0.5f);
            mtr_model_set_palette(model, pal.empty() ? nullptr : pal.data(), pal.size() / 16);
            const uint32_t W = rnd() % 6 ? pick(1, 300) : weird(), H = rnd() % 6 ? pick(1, 200) : weird();
            const float clear[4] = {1, 1, 1, 1}, M[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
            mtr_frame* fr = nullptr;
            if (mtr_frame_begin(dev, W, H, clear, 1.0f, &fr) == MTR_OK) {
                if (rnd() % 3 == 0) mtr_frame_set_shard(fr, weird() % 9, weird() % 9);
                if (rnd() % 3 == 0) {  // ownership maps, band tables that are mostly malformed
                    uint32_t bands[10];
                    for (auto& b : bands) b = weird() % 40;
                    if (rnd() % 2) bands[0] = 0;
                    mtr_frame_set_shard_map(fr, weird() % 9, weird() % 9, weird() % 4, weird() % 9, rnd() % 2 ? bands : nullptr);
                }
                mtr_frame_draw_model(fr, model, M);
                const uint32_t ninst = pick(0, 5);
                std::vector<float> mm((size_t)(ninst ? ninst : 1) * 16, 1.0f), pp((size_t)(ninst ? ninst : 1) * 2 * 16, 0.25f);
                std::vector<int32_t> tov(ninst ? ninst : 1, rnd() % 2 ? -1 : (int32_t)weird());
                { const bool wp = rnd() % 2; mtr_frame_draw_instances(fr, model, mm.data(), wp ? pp.data() : nullptr, wp ? 2 : 0, ninst, M); }
                mtr_batch* batch = nullptr;
                if (mtr_batch_create(dev, model, ninst, mm.data(), rnd() % 2 ? pp.data() : nullptr, rnd() % 2 ? 2 : 0, rnd() % 2 ? tov.data() : nullptr, &batch) == MTR_OK) {
                    mtr_frame_draw_batch(fr, batch, M);
                }
                mtr_frame_draw_overlay_cubes(fr, M, mm.data(), ninst);
                if (mtr_frame_end(fr) == MTR_OK && (uint64_t)W * H < (1u << 22)) {
                    std::vector<uint8_t> px((size_t)W * H * 4);
                    std::vector<float> dp((size_t)W * H);
                    mtr_frame_read_color(fr, px.data(), px.size() - (rnd() % 4 == 0 ? 1 : 0));
                    mtr_frame_read_depth(fr, dp.data(), dp.size());
                    mtr_frame_stats st;
                    mtr_frame_get_stats(fr, &st);
                }
                mtr_frame_destroy(fr);
                if (batch) mtr_batch_destroy(batch);
            }
            mtr_model_destroy(model);
        }
        for (mtr_texture* t : texs) mtr_texture_destroy(t);
    }
    mtr_device_destroy(dev);
    std::printf("created=%ld rejected=%ld\n", created, rejected);
    return 0;
}
