// ThreadSanitizer run of the device's exchange thread (csrc/mtr_api.cpp: mtr_device_exchange_start and friends) over the
// stand-in HIP runtime of tests/cpp/hip_stub: the render thread begins, draws and hands over thousands of sharded frames
// while the exchange thread packs, "gathers", unpacks and destroys them; frames that never reach the exchange thread are
// begun and destroyed in between, so the framebuffer pool is touched from both sides.  Any unsynchronised access to
// shared host state is reported by TSan (exit code 66).          usage: exchange_tsan <frames>
#include "../../mt_renderer_amd/csrc/mtr_api.cpp"

#include <atomic>

void mtr_launch_geom(const GeomParams&, hipStream_t) {}
void mtr_launch_scan(const FrameBuffers&, hipStream_t) {}
void mtr_launch_fill(const FrameBuffers&, uint32_t, hipStream_t) {}
// what tile_prologue does on the device: publish the (clean) overflow flags of the frame
static void stub_status(const TileParams& p) { if (p.host_status) __atomic_store_n(p.host_status, 0x80000000u, __ATOMIC_RELEASE); }
void mtr_launch_tile(const TileParams& p, bool, hipStream_t) { stub_status(p); }
void mtr_launch_tile_vis(const TileParams& p, bool, hipStream_t) { stub_status(p); }
void mtr_launch_alpha_min(const uint8_t*, size_t, uint32_t* out_min, hipStream_t) { *out_min = 255; }
void mtr_launch_vertex_stage(const GeomParams&, uint32_t, float*, float*, hipStream_t) {}
void mtr_launch_bc1_decode(const uint8_t*, uint8_t*, uint32_t, uint32_t, hipStream_t) {}
void mtr_launch_bc7_decode(const uint8_t*, uint8_t*, uint32_t, uint32_t, hipStream_t) {}
// the copies the real kernels make, so that the send / gathered / destination buffers are really written by this thread
void mtr_launch_pack_shard(const uint8_t* color, uint8_t* dst, uint32_t, uint32_t, const uint32_t* own_list, uint32_t n, uint32_t, hipStream_t) { dst[0] = color[0] + (n ? (uint8_t)own_list[0] : 0); }
void mtr_launch_unpack_shards(const uint8_t* g, uint8_t* dst, uint32_t, uint32_t, const uint32_t* src_of_bin, hipStream_t) { dst[0] = g[0] + (uint8_t)src_of_bin[0]; }
void mtr_launch_cull_instances(const CullParams& p, hipStream_t) { for (uint32_t i = 0; i < p.ninst; i++) { p.strad[p.count[1]++] = *p.count; p.list[(*p.count)++] = i; } }
void mtr_launch_cull_chunks(const ChunkCullParams& p, hipStream_t) { if (p.nchunks && p.ninst) p.work_mask[(size_t)((p.nchunks + 15) / 16) * p.ninst - 1] = 0xFFFF; }  // the last mask of the draw

static std::atomic<long> g_calls{0};
static int fake_allgather(const void* send, void* recv, size_t count, int, void*, void*) {
    if (count) static_cast<uint8_t*>(recv)[0] = static_cast<const uint8_t*>(send)[0];
    g_calls.fetch_add(1, std::memory_order_relaxed);
    return 0;
}

#define REQ(x) do { if ((x) != MTR_OK) { fprintf(stderr, "%s failed: %s\n", #x, mtr_last_error(dev)); return 4; } } while (0)

int main(int argc, char** argv) {
    const long frames = argc > 1 ? strtol(argv[1], nullptr, 10) : 3000;
    mtr_device* dev = nullptr;
    if (mtr_device_create(0, &dev)) return 3;
    // one triangle, float3 positions
    const float verts[9] = {-0.5f, -0.5f, 0.5f, 0.5f, -0.5f, 0.5f, 0.0f, 0.5f, 0.5f};
    const uint16_t idx[3] = {0, 1, 2};
    mtr_primitive pr;
    memset(&pr, 0, sizeof pr);
    pr.w[0] = 3u << 16; pr.w[2] = 1 | (12u << 16) | (3u << 24); pr.w[7] = 3;
    mtr_layout l;
    memset(&l, 0, sizeof l);
    l.elements[l.num_elements++] = mtr_element{MTR_SEM_POSITION, MTR_IEF_F32, 3, 0, 0, 0};
    mtr_model* model = nullptr;
    REQ(mtr_model_create(dev, verts, sizeof verts, idx, 3, &pr, 1, &l, nullptr, nullptr, 0, nullptr, &model));
    const uint32_t W = 64, H = 48, world = 2;
    const size_t nbytes = std::max({mtr_shard_bytes(W, H, world), mtr_shard_bytes_map(W, H, world, MTR_OWN_BANDS, 0, nullptr),
                                    mtr_shard_bytes_map(W, H, world, MTR_OWN_SUPERTILES, 1, nullptr)});
    std::vector<uint8_t> send(nbytes), gathered(nbytes * world), final_((size_t)W * H * 4);
    int fake_stream = 0;
    REQ(mtr_device_exchange_start(dev, fake_allgather, nullptr, 1, send.data(), nbytes, gathered.data(), final_.data(), world, &fake_stream));
    // a second lane: alternate frames use these buffers
    std::vector<uint8_t> send2(nbytes), gathered2(nbytes * world), final2((size_t)W * H * 4);
    int fake_stream2 = 0;
    REQ(mtr_device_exchange_add_lane(dev, nullptr, send2.data(), gathered2.data(), final2.data(), &fake_stream2));
    const float clear[4] = {1, 1, 1, 1}, M[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    long handed = 0;
    for (long i = 0; i < frames; i++) {
        mtr_frame* f = nullptr;
        REQ(mtr_frame_begin(dev, W, H, clear, 1.0f, &f));
        if (i % 3 == 0) REQ(mtr_frame_set_shard(f, (uint32_t)(i & 1), world));
        else REQ(mtr_frame_set_shard_map(f, (uint32_t)(i & 1), world, i % 3 == 1 ? MTR_OWN_BANDS : MTR_OWN_SUPERTILES, 1, nullptr));
        if (i % 5 == 2) {
            // a frame that OWNS a temporary batch: the exchange thread destroys it (mtr_batch_destroy: garbage list,
            // frame index) while this thread keeps submitting
            const float inst[32] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0.1f, 0, 0, 1};
            REQ(mtr_frame_draw_instances(f, model, inst, nullptr, 0, 2, M));
        } else if (i % 5 == 4) {
            REQ(mtr_frame_draw_overlay_cubes(f, M, M, 1));
        } else {
            REQ(mtr_frame_draw_model(f, model, M));
        }
        if (i % 7 == 3) {  // a frame that is submitted, waited and destroyed here: the pool is used from this thread too
            REQ(mtr_frame_submit(f));
            REQ(mtr_frame_wait(f));
            mtr_frame_destroy(f);
        } else if (i % 11 == 5) {  // begun, never submitted
            mtr_frame_destroy(f);
        } else {
            REQ(mtr_frame_submit_exchange(f));
            handed++;
        }
        if (i % 257 == 0) REQ(mtr_device_exchange_drain(dev));
    }
    REQ(mtr_device_exchange_drain(dev));
    if (g_calls.load() != handed) { fprintf(stderr, "all-gather calls %ld != frames handed over %ld\n", g_calls.load(), handed); return 5; }
    REQ(mtr_device_exchange_stop(dev));
    mtr_model_destroy(model);
    mtr_device_destroy(dev);
    printf("handed=%ld\n", handed);
    return 0;
}
