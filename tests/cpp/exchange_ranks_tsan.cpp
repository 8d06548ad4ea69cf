// Two "ranks" (two devices with an exchange thread each) in one process over the stand-in HIP runtime, joined by an
// all-gather that behaves like a real collective: call k returns only once BOTH ranks have issued call k (20 s timeout =
// a hang).  Rank 1's pack fails at one frame.  What must hold:
//   * no rank skips a collective: both ranks issue exactly as many all-gathers as frames were handed over, nothing hangs;
//   * the failing rank's mtr_device_exchange_drain returns the error, the healthy rank's returns MTR_OK, and the MIN of
//     the two (what bench.py all-reduces) tells every rank;
//   * frames whose bin queues "overflow" (the stub tile launch publishes flag 4 now and then) are re-run by the exchange
//     thread while the render thread flips parts_disp and the palette of the same model: ThreadSanitizer sees no race.
// usage: exchange_ranks_tsan <frames>
#include "../../mt_renderer_amd/csrc/mtr_api.cpp"

#include <atomic>
#include <chrono>

static std::atomic<long> g_tile_launches{0};
static std::atomic<long> g_reruns{0};
void mtr_launch_geom(const GeomParams&, hipStream_t) {}
void mtr_launch_scan(const FrameBuffers&, hipStream_t) {}
void mtr_launch_fill(const FrameBuffers&, uint32_t, hipStream_t) {}
static void stub_status(const TileParams& p) {
    uint32_t flags = 0;
    if (p.fb.direct && g_tile_launches.fetch_add(1, std::memory_order_relaxed) % 13 == 5) flags = 4u;  // a bounded bin queue "filled up"
    if (!p.fb.direct) g_reruns.fetch_add(1, std::memory_order_relaxed);
    if (p.host_status) __atomic_store_n(p.host_status, 0x80000000u | flags, __ATOMIC_RELEASE);
}
void mtr_launch_tile(const TileParams& p, bool, hipStream_t) { stub_status(p); }
void mtr_launch_tile_vis(const TileParams& p, bool, hipStream_t) { stub_status(p); }
void mtr_launch_alpha_min(const uint8_t*, size_t, uint32_t* out_min, hipStream_t) { *out_min = 255; }
void mtr_launch_vertex_stage(const GeomParams&, uint32_t, float*, float*, hipStream_t) {}
void mtr_launch_bc1_decode(const uint8_t*, uint8_t*, uint32_t, uint32_t, hipStream_t) {}
void mtr_launch_bc7_decode(const uint8_t*, uint8_t*, uint32_t, uint32_t, hipStream_t) {}
static uint8_t* g_fail_send = nullptr;          // rank 1's send buffer
static std::atomic<long> g_packs_rank1{0};
static long g_fail_at = 37;
void mtr_launch_pack_shard(const uint8_t* color, uint8_t* dst, uint32_t, uint32_t, const uint32_t* own_list, uint32_t n, uint32_t, hipStream_t) {
    dst[0] = color[0] + (n ? (uint8_t)own_list[0] : 0);
    if (dst == g_fail_send && g_packs_rank1.fetch_add(1) == g_fail_at) hipStubInjectedError() = 719;  // "the launch failed" on this thread
}
void mtr_launch_unpack_shards(const uint8_t* g, uint8_t* dst, uint32_t, uint32_t, const uint32_t* src_of_bin, hipStream_t) { dst[0] = g[0] + (uint8_t)src_of_bin[0]; }
void mtr_launch_cull_instances(const CullParams& p, hipStream_t) { for (uint32_t i = 0; i < p.ninst; i++) { p.strad[p.count[1]++] = *p.count; p.list[(*p.count)++] = i; } }
void mtr_launch_cull_chunks(const ChunkCullParams& p, hipStream_t) { if (p.nchunks && p.ninst) p.work_mask[(size_t)((p.nchunks + 15) / 16) * p.ninst - 1] = 0xFFFF; }

struct Collective {
    std::mutex m;
    std::condition_variable cv;
    long issued[2] = {0, 0};
    bool hung = false;
};
static Collective g_coll;
struct Comm { int rank; };
static int barrier_allgather(const void* send, void* recv, size_t count, int, void* comm, void*) {
    const int r = static_cast<Comm*>(comm)->rank;
    if (count) static_cast<uint8_t*>(recv)[0] = static_cast<const uint8_t*>(send)[0];
    std::unique_lock<std::mutex> lk(g_coll.m);
    const long mine = ++g_coll.issued[r];
    g_coll.cv.notify_all();
    // system_clock: pthread_cond_timedwait, which this libtsan intercepts (wait_for's pthread_cond_clockwait it does not)
    if (!g_coll.cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::seconds(20), [&] { return g_coll.issued[1 - r] >= mine || g_coll.hung; }))
        g_coll.hung = true;
    if (g_coll.hung) { g_coll.cv.notify_all(); return 1; }
    return 0;
}

struct Rank {
    int rank = 0;
    mtr_device* dev = nullptr;
    mtr_model* model = nullptr;
    Comm comm;
    std::vector<uint8_t> send, gathered, final_;
    int fake_stream = 0;
    long handed = 0;
    int32_t drain_rc = MTR_OK;
    int fatal = 0;
};

static const uint32_t W = 64, H = 48, WORLD = 2;

static void render_loop(Rank* R, long frames) {
    mtr_device* dev = R->dev;
    const float clear[4] = {1, 1, 1, 1}, M[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float pal[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
#define RREQ(x) do { if ((x) != MTR_OK) { fprintf(stderr, "[rank %d] %s failed: %s\n", R->rank, #x, mtr_last_error(dev)); R->fatal = 4; return; } } while (0)
    for (long i = 0; i < frames; i++) {
        // the host animates the model between frames: parts_disp (its length changes, so the chunk table is rebuilt every
        // time) and the palette -- while the exchange thread may be re-running an earlier frame that drew the same model
        const uint8_t pd[2] = {1, 1};
        RREQ(mtr_model_set_parts_disp(R->model, pd, 1 + (size_t)(i & 1)));
        pal[12] = 0.001f * (float)i;
        RREQ(mtr_model_set_palette(R->model, pal, 1));
        mtr_frame* f = nullptr;
        int32_t rc = mtr_frame_begin(dev, W, H, clear, 1.0f, &f);
        if (rc == MTR_E_OVERFLOW) { i--; continue; }  // a latched report of an earlier frame: not this test's subject
        RREQ(rc);
        RREQ(mtr_frame_set_shard_map(f, (uint32_t)R->rank, WORLD, MTR_OWN_BANDS, 0, nullptr));
        RREQ(mtr_frame_draw_model(f, R->model, M));
        RREQ(mtr_frame_submit_exchange(f));
        R->handed++;
    }
    R->drain_rc = mtr_device_exchange_drain(dev);
#undef RREQ
}

int main(int argc, char** argv) {
    const long frames = argc > 1 ? strtol(argv[1], nullptr, 10) : 400;
    const float verts[9] = {-0.5f, -0.5f, 0.5f, 0.5f, -0.5f, 0.5f, 0.0f, 0.5f, 0.5f};
    const uint16_t idx[3] = {0, 1, 2};
    mtr_primitive pr;
    memset(&pr, 0, sizeof pr);
    pr.w[0] = 3u << 16; pr.w[2] = 1 | (12u << 16) | (3u << 24); pr.w[7] = 3;
    mtr_layout l;
    memset(&l, 0, sizeof l);
    l.elements[l.num_elements++] = mtr_element{MTR_SEM_POSITION, MTR_IEF_F32, 3, 0, 0, 0};
    const size_t nbytes = mtr_shard_bytes_map(W, H, WORLD, MTR_OWN_BANDS, 0, nullptr);
    Rank R[2];
    for (int r = 0; r < 2; r++) {
        R[r].rank = r; R[r].comm.rank = r;
        if (mtr_device_create(0, &R[r].dev)) return 3;
        if (mtr_model_create(R[r].dev, verts, sizeof verts, idx, 3, &pr, 1, &l, nullptr, nullptr, 0, nullptr, &R[r].model)) return 3;
        R[r].send.resize(nbytes); R[r].gathered.resize(nbytes * WORLD); R[r].final_.resize((size_t)W * H * 4);
        if (mtr_device_exchange_start(R[r].dev, barrier_allgather, &R[r].comm, 1, R[r].send.data(), nbytes, R[r].gathered.data(),
                                      R[r].final_.data(), WORLD, &R[r].fake_stream)) return 3;
    }
    g_fail_send = R[1].send.data();
    g_fail_at = frames / 3;
    std::thread t0(render_loop, &R[0], frames), t1(render_loop, &R[1], frames);
    t0.join(); t1.join();
    int rc = 0;
    if (R[0].fatal || R[1].fatal) rc = 4;
    { std::lock_guard<std::mutex> g(g_coll.m); if (g_coll.hung) { fprintf(stderr, "a collective hung: issued %ld vs %ld\n", g_coll.issued[0], g_coll.issued[1]); rc = 5; } }
    if (!rc && (g_coll.issued[0] != R[0].handed || g_coll.issued[1] != R[1].handed || R[0].handed != frames)) {
        fprintf(stderr, "all-gathers issued %ld / %ld, frames handed over %ld / %ld\n", g_coll.issued[0], g_coll.issued[1], R[0].handed, R[1].handed);
        rc = 6;
    }
    if (!rc && !(R[0].drain_rc == MTR_OK && R[1].drain_rc != MTR_OK)) {
        fprintf(stderr, "drain status: rank 0 %d (want 0), rank 1 %d (want an error)\n", R[0].drain_rc, R[1].drain_rc);
        rc = 7;
    }
    const int32_t agreed = std::max(R[0].drain_rc, R[1].drain_rc);  // what the host's MIN-of-ok all-reduce yields on every rank
    for (int r = 0; r < 2; r++) {
        (void)mtr_device_exchange_stop(R[r].dev);
        mtr_model_destroy(R[r].model);
        mtr_device_destroy(R[r].dev);
    }
    printf("handed=%ld reruns=%ld agreed_status=%d rank1_error=\"%s\"\n", R[0].handed + R[1].handed, g_reruns.load(), agreed, rc ? "" : "reported");
    return rc;
}
