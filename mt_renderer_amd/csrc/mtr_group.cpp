// mtr_group: one host thread, N devices (include/mtr.h, "one host thread, N devices").
//
// The reference has one event loop and one wgpu::Device (src/renderer_app_manager.rs:202-272); a host that keeps that
// shape and owns several GPUs drives them through a group: a group frame is N sharded frames, one per device, each
// drawn with that device's own models (HBM residency is per device), and ending it gathers the colour of every part
// into one RGBA8 image on the device of rank 0 -- pack on each device, one peer copy per rank over xGMI, unpack on
// rank 0.  No collective library: inside one process a gather is N copies.
//
// Written against the public C ABI only (plus the HIP runtime for the peer copies): it holds no renderer state of its
// own and a host could have written it.  The throughput path for N GPUs stays one process per GPU with the exchange
// thread (mtr_device_exchange_start); a group frame is ended synchronously, one at a time.
#include <hip/hip_runtime.h>

#include <memory>
#include <string>
#include <vector>

#include "../../include/mtr.h"

struct GroupMember {
    int hip_dev = 0;
    mtr_device* dev = nullptr;
    hipStream_t stream = nullptr;  // the device's public stream (ours: pack / unpack are enqueued on it)
    hipEvent_t packed = nullptr;
    void* send = nullptr;          // this rank's packed shard
    size_t send_cap = 0;
};

struct mtr_group {
    std::vector<GroupMember> m;
    std::string err;
    void* gathered = nullptr;   // on rank 0's device: world blocks of shard_bytes
    size_t gathered_cap = 0;
    void* image = nullptr;      // on rank 0's device: the assembled RGBA8 frame of the group frame that ended last
    size_t image_cap = 0;
    uint64_t generation = 0;    // group frames ended so far
};

struct mtr_group_frame {
    mtr_group* g = nullptr;
    uint32_t w = 0, h = 0;
    std::vector<mtr_frame*> parts;
    uint64_t ended_as = 0;  // the group's generation when this frame ended (0: not ended)
};

static std::string g_group_create_error;

static int32_t gfail(mtr_group* g, int32_t code, const std::string& msg) {
    (g ? g->err : g_group_create_error) = msg;
    return code;
}

static int32_t gfail_dev(mtr_group* g, int32_t code, int rank) {
    const char* m = mtr_last_error(g->m[(size_t)rank].dev);
    return gfail(g, code, "rank " + std::to_string(rank) + ": " + (m ? m : ""));
}

#define GHIP(g, expr)                                                                                                \
    do {                                                                                                             \
        hipError_t e_ = (expr);                                                                                      \
        if (e_ != hipSuccess) return gfail(g, MTR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

static int32_t grow(mtr_group* g, int hip_dev, void** p, size_t* cap, size_t need) {
    if (*cap >= need) return MTR_OK;
    GHIP(g, hipSetDevice(hip_dev));
    if (*p) {
        // the buffer may still be read by the copies / the unpack of the previous group frame
        GHIP(g, hipDeviceSynchronize());
        GHIP(g, hipFree(*p));
        *p = nullptr;
        *cap = 0;
    }
    GHIP(g, hipMalloc(p, need));
    *cap = need;
    return MTR_OK;
}

extern "C" {

const char* mtr_group_last_error(const mtr_group* g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }

void mtr_group_destroy(mtr_group* g) {
    if (!g) return;
    for (GroupMember& mb : g->m) {
        if (mb.dev) mtr_device_destroy(mb.dev);  // waits for everything in flight on the device
        (void)hipSetDevice(mb.hip_dev);
        if (mb.stream) (void)hipStreamSynchronize(mb.stream);
        if (mb.send) (void)hipFree(mb.send);
        if (mb.packed) (void)hipEventDestroy(mb.packed);
    }
    if (!g->m.empty()) {
        (void)hipSetDevice(g->m[0].hip_dev);
        if (g->gathered) (void)hipFree(g->gathered);
        if (g->image) (void)hipFree(g->image);
    }
    for (GroupMember& mb : g->m)
        if (mb.stream) {
            (void)hipSetDevice(mb.hip_dev);
            (void)hipStreamDestroy(mb.stream);
        }
    delete g;
}

int32_t mtr_group_create(const int32_t* hip_devices, int32_t n, mtr_group** out) {
    if (!out) return gfail(nullptr, MTR_E_INVALID, "out is NULL");
    *out = nullptr;
    if (!hip_devices || n < 1 || n > 64) return gfail(nullptr, MTR_E_INVALID, "a group has 1 to 64 devices");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return gfail(nullptr, MTR_E_HIP, std::string("no HIP device: ") + hipGetErrorString(e));
    for (int32_t i = 0; i < n; i++)
        if (hip_devices[i] < 0 || hip_devices[i] >= ndev) return gfail(nullptr, MTR_E_INVALID, "hip device index out of range");
    mtr_group* g = new mtr_group();
    g->m.resize((size_t)n);
    int32_t rc = MTR_OK;
    for (int32_t i = 0; i < n && !rc; i++) {
        GroupMember& mb = g->m[(size_t)i];
        mb.hip_dev = hip_devices[i];
        if (hipSetDevice(mb.hip_dev) != hipSuccess || hipStreamCreateWithFlags(&mb.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&mb.packed, hipEventDisableTiming) != hipSuccess) {
            rc = gfail(nullptr, MTR_E_HIP, "stream / event creation failed on device " + std::to_string(mb.hip_dev));
            break;
        }
        rc = mtr_device_create_on_stream(mb.hip_dev, mb.stream, &mb.dev);
        if (rc) gfail(nullptr, rc, std::string("rank ") + std::to_string(i) + ": " + mtr_last_error(nullptr));
    }
    if (rc) {
        const std::string keep = g_group_create_error;
        mtr_group_destroy(g);
        g_group_create_error = keep;
        return rc;
    }
    // direct peer copies into rank 0's memory where the link allows them (xGMI inside a node); without peer access the
    // runtime stages hipMemcpyPeerAsync through the host, which is slower and still correct
    (void)hipSetDevice(g->m[0].hip_dev);
    for (int32_t i = 1; i < n; i++) {
        const int peer = g->m[(size_t)i].hip_dev;
        int can = 0;
        if (peer != g->m[0].hip_dev && hipDeviceCanAccessPeer(&can, g->m[0].hip_dev, peer) == hipSuccess && can)
            (void)hipDeviceEnablePeerAccess(peer, 0);
    }
    (void)hipGetLastError();  // "already enabled" is not an error of ours
    *out = g;
    return MTR_OK;
}

int32_t mtr_group_size(const mtr_group* g) { return g ? (int32_t)g->m.size() : 0; }

mtr_device* mtr_group_device(mtr_group* g, int32_t rank) {
    if (!g || rank < 0 || (size_t)rank >= g->m.size()) return nullptr;
    return g->m[(size_t)rank].dev;
}

void mtr_group_frame_destroy(mtr_group_frame* gf) {
    if (!gf) return;
    for (mtr_frame* f : gf->parts)
        if (f) mtr_frame_destroy(f);
    delete gf;
}

int32_t mtr_group_frame_begin(mtr_group* g, uint32_t w, uint32_t h, const float clear_rgba[4], float clear_depth, uint32_t map,
                              uint32_t param, const uint32_t* band_rows, mtr_group_frame** out) {
    if (!g || !out) return MTR_E_INVALID;
    *out = nullptr;
    const uint32_t world = (uint32_t)g->m.size();
    if (!clear_rgba) return gfail(g, MTR_E_INVALID, "clear colour is NULL");
    if (mtr_shard_bytes_map(w, h, world, map, param, band_rows) == 0)
        return gfail(g, MTR_E_INVALID, "bad frame size or ownership map for a group of " + std::to_string(world));
    auto gf = std::make_unique<mtr_group_frame>();
    gf->g = g; gf->w = w; gf->h = h;
    gf->parts.assign(world, nullptr);
    for (uint32_t r = 0; r < world; r++) {
        int32_t rc = mtr_frame_begin(g->m[r].dev, w, h, clear_rgba, clear_depth, &gf->parts[r]);
        if (!rc) rc = mtr_frame_set_shard_map(gf->parts[r], r, world, map, param, band_rows);
        if (rc) {
            gfail_dev(g, rc, (int)r);
            mtr_group_frame_destroy(gf.release());
            return rc;
        }
    }
    *out = gf.release();
    return MTR_OK;
}

mtr_frame* mtr_group_frame_part(mtr_group_frame* gf, int32_t rank) {
    if (!gf || rank < 0 || (size_t)rank >= gf->parts.size()) return nullptr;
    return gf->parts[(size_t)rank];
}

int32_t mtr_group_frame_end(mtr_group_frame* gf) {
    if (!gf) return MTR_E_INVALID;
    mtr_group* g = gf->g;
    if (gf->ended_as) return gfail(g, MTR_E_INVALID, "group frame already ended");
    const uint32_t world = (uint32_t)g->m.size();
    // every device starts its part before the host waits for any of them
    for (uint32_t r = 0; r < world; r++) {
        const int32_t rc = mtr_frame_submit(gf->parts[r]);
        if (rc) return gfail_dev(g, rc, (int)r);
    }
    // the wait settles a part whose bounded bin queues overflowed (it is re-run through the exact queues) BEFORE its
    // pixels are packed: a gathered frame never misses triangles
    for (uint32_t r = 0; r < world; r++) {
        const int32_t rc = mtr_frame_wait(gf->parts[r]);
        if (rc) return gfail_dev(g, rc, (int)r);
    }
    const size_t bytes = mtr_frame_shard_bytes(gf->parts[0]);
    GroupMember& root = g->m[0];
    int32_t rc;
    if ((rc = grow(g, root.hip_dev, &g->gathered, &g->gathered_cap, bytes * world))) return rc;
    if ((rc = grow(g, root.hip_dev, &g->image, &g->image_cap, (size_t)gf->w * gf->h * 4))) return rc;
    for (uint32_t r = 0; r < world; r++) {
        GroupMember& mb = g->m[r];
        if ((rc = grow(g, mb.hip_dev, &mb.send, &mb.send_cap, bytes))) return rc;
        if ((rc = mtr_frame_pack_color_shard(gf->parts[r], mb.send, bytes))) return gfail_dev(g, rc, (int)r);  // on mb.stream
        GHIP(g, hipSetDevice(mb.hip_dev));
        GHIP(g, hipEventRecord(mb.packed, mb.stream));
    }
    GHIP(g, hipSetDevice(root.hip_dev));
    for (uint32_t r = 0; r < world; r++) {
        GroupMember& mb = g->m[r];
        uint8_t* dst = static_cast<uint8_t*>(g->gathered) + (size_t)r * bytes;
        if (r) GHIP(g, hipStreamWaitEvent(root.stream, mb.packed, 0));
        if (mb.hip_dev == root.hip_dev)
            GHIP(g, hipMemcpyAsync(dst, mb.send, bytes, hipMemcpyDeviceToDevice, root.stream));
        else
            GHIP(g, hipMemcpyPeerAsync(dst, root.hip_dev, mb.send, mb.hip_dev, bytes, root.stream));
    }
    if ((rc = mtr_frame_unpack_color_shards_on_stream(gf->parts[0], g->gathered, g->image, nullptr))) return gfail_dev(g, rc, 0);
    GHIP(g, hipSetDevice(root.hip_dev));
    GHIP(g, hipStreamSynchronize(root.stream));
    gf->ended_as = ++g->generation;
    return MTR_OK;
}

void* mtr_group_frame_color_devptr(mtr_group_frame* gf) {
    if (!gf || !gf->ended_as || gf->ended_as != gf->g->generation) return nullptr;
    return gf->g->image;
}

int32_t mtr_group_frame_read_color(mtr_group_frame* gf, void* rgba8, size_t len) {
    if (!gf || !rgba8) return MTR_E_INVALID;
    mtr_group* g = gf->g;
    if (!gf->ended_as) return gfail(g, MTR_E_INVALID, "group frame not ended");
    if (gf->ended_as != g->generation) return gfail(g, MTR_E_INVALID, "a later group frame has ended: the gathered image is that frame's");
    const size_t need = (size_t)gf->w * gf->h * 4;
    if (len < need) return gfail(g, MTR_E_INVALID, "output too small");
    GHIP(g, hipSetDevice(g->m[0].hip_dev));
    GHIP(g, hipMemcpy(rgba8, g->image, need, hipMemcpyDeviceToHost));
    return MTR_OK;
}

}  // extern "C"
