// mtr_api.cpp -- host side of libmtr.so: handles, validation, HBM residency, launch orchestration.
// Mirrors the reference's object model: Texture::new (src/texture.rs:11), Model::new / render /
// set_parts_disp (src/model.rs:36-363) and the render pass of src/bin/modelviewer.rs:190-234.
#include "../../include/mtr.h"
#include "mtr_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <thread>
#include <string>
#include <utility>
#include <vector>

namespace {

thread_local std::string g_create_error = "";

struct ColorDepth {
    uint32_t w, h;
    uint8_t* color;
    float* depth;
    uint32_t* counters;  // two blocks of CTR_NUM words: a frame counts in block ctr_live and its tile kernel zeroes the other
    uint32_t ctr_live = 0;
    bool ctr_dirty = true;     // the live block must be zeroed by a fill before the next run (fresh memory, or a re-run)
    bool next_zeroed = false;  // a tile kernel has been queued that zeroes the other block
    uint32_t* live() const { return counters + (size_t)ctr_live * CTR_NUM; }
    uint32_t* other() const { return counters + (size_t)(ctr_live ^ 1u) * CTR_NUM; }
    // recorded after the tile kernel of the last frame that rendered into these buffers; a frame that recycles
    // them (possibly on another internal stream) waits on it before its first write
    hipEvent_t done = nullptr;
    bool used = false;
};

}  // namespace

// Intermediate buffers of one frame in flight, and the internal stream its kernels run on.  The device keeps
// `nslots` slots and deals frames to them round-robin: the kernels of one frame follow each other on one stream
// with no cross-stream dependency in between, and the frames of different slots overlap (frame k+1's geometry runs
// while frame k's tile kernel is still rasterising; DESIGN.md "frames in flight").  Grow-only.
// 3 slots by default (headline scene, ms per frame: 1 slot 0.094, 2: 0.066, 3: 0.056, 4: 0.071, 6: 0.058); the environment
// variable MTR_NSLOTS (1..MTR_MAX_SLOTS) overrides it at device creation.
#define MTR_MAX_SLOTS 8
struct Slot {
    RecHdr* rec_hdr = nullptr;
    RecP* rec_a = nullptr;
    int4* rec_l = nullptr;
    RecB* rec_b = nullptr;
    ChunkInfo* chunk_info = nullptr;
    uint32_t rec_cap = 0, chunk_cap = 0;
    unsigned long long* bin_count = nullptr;
    unsigned long long* bin_fill = nullptr;
    uint32_t* bin_start = nullptr;
    uint32_t* seg_start = nullptr;
    uint8_t* bin_flag = nullptr;   // mixed frames: 1 = the bin holds a translucent triangle (ordered kernel's)
    uint32_t* inst_list = nullptr;   // sharded batch draws: compacted instance lists, draw after draw
    uint32_t* inst_count = nullptr;  // one counter per draw
    uint32_t inst_cap = 0, draw_cap = 0;
    uint16_t* work_mask = nullptr;   // sharded draws: which chunks survive culling, one bit each (k_cull_chunks -> k_geom), draw after draw
    uint32_t work_cap = 0;
    bool cull_counts_dirty = true;   // inst_count (MTR_CULL_CTR_WORDS per draw: word 0 = instance-list length) needs a fill
    uint32_t ctr_clean_draws = 0;    // draws whose counters the last tile kernel cleared
    CompMat* comp = nullptr;         // sharded batch draws: per (instance, joint) composites, k_cull_instances -> k_cull_chunks
    uint32_t comp_cap = 0;
    uint32_t bin_cap = 0;
    uint32_t* entries = nullptr;  // submission order of every (triangle, bin) pair
    Seg* segs = nullptr;
    uint32_t entry_cap = 0, seg_cap = 0;
    DMat* mats = nullptr;
    uint32_t mat_cap = 0;
    bool bin_fill_dirty = true;      // direct frames leave bin_fill zeroed (the tile kernels clean up); others do not
    std::vector<DMat> mats_uploaded;  // what `mats` currently holds: steady-state frames skip the upload
    hipStream_t stream = nullptr;     // a slot's frames are ordered by this stream: reuse needs no event
};

// Which rank owns which bin, for one (frame size, map, world) combination: host lists + their device image.
struct OwnTable {
    uint32_t w = 0, h = 0, map = 0, param = 0, world = 1;
    std::vector<uint32_t> bands;  // BANDS: world + 1 bin rows
    std::vector<uint32_t> offs;   // world + 1: rank r's bins are lists[offs[r] .. offs[r+1])
    uint32_t stride_bins = 0;     // the largest share = bins per rank in an all-gather buffer
    uint32_t nsx = 0, st_shift = 0;
    std::vector<uint32_t> lists;       // host copy of d_lists (mtr_frame_read_bin_counts masks the bins a rank does not own)
    uint32_t* d_lists = nullptr;       // nbins bin ids, rank after rank, each in tile-kernel order
    uint32_t* d_src_of_bin = nullptr;  // nbins: rank * stride_bins + k
    uint32_t refs = 0;                 // live frames that use it (submit_mu)
};

struct mtr_device {
    int hip_dev = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool profiling = false;
    uint32_t texture_residency = MTR_TEXRES_DECODED;
    int tile_mode = MTR_TILE_AUTO;
    std::string err;
    Slot slots[MTR_MAX_SLOTS];
    uint32_t nslots = 3;
    // host run-ahead bound: submitting frame i first waits (on the host) for frame i - max_inflight.  Without it a
    // host that never waits queues thousands of commands and the runtime's per-call cost grows with the backlog
    // and memory held by queued frames is unbounded.  Default 16 (no measurable cost); MTR_MAX_INFLIGHT overrides.
    static constexpr uint32_t kMaxInflight = 64;
    hipEvent_t inflight[kMaxInflight] = {};
    uint32_t max_inflight = 16;
    uint64_t frames_submitted = 0;  // index of the next frame; frame i records inflight[i % max_inflight]
    struct Garbage { void* p; uint64_t last_frame; };
    std::vector<Garbage> garbage;   // device buffers of destroyed batches whose last frame may still be in flight
    // Everything a frame submission touches (slots, the in-flight ring, frames_submitted, garbage, the models' chunk
    // tables and palette rings) is guarded by submit_mu: the render thread submits, but the exchange thread re-runs a
    // frame whose bin queues overflowed and destroys frames (which may own batches).  Uncontended in steady state.
    std::mutex submit_mu;
    // Frame status words, pinned host memory, one per frame in flight (frame i uses word i % max_inflight): the tile
    // kernel stores 0x80000000 | overflow flags there when it starts.  status_pending[i]: the frame that used word i
    // was released (destroyed, or handed to a consumer) without anyone having looked at its flags; they are examined
    // when the word is next polled / recycled, and a set flag is latched in sticky_err for the next API call to report.
    uint32_t* status_host = nullptr;
    uint32_t* status_dev = nullptr;
    bool status_checked[kMaxInflight] = {};   // somebody (mtr_frame_wait, the exchange thread) has looked at the word
    bool status_released[kMaxInflight] = {};  // its frame was destroyed without that: examine it at the next poll
    uint64_t status_owner[kMaxInflight] = {}; // index of the frame the word belongs to
    int status_slot_of[kMaxInflight] = {};    // which Slot that frame ran on
    int32_t sticky_err = MTR_OK;
    std::string sticky_msg;
    uint32_t queue_scale = 1;  // two-pass queues: multiplier on the default sizes, doubled when an un-waited frame overflowed them
    hipStream_t s_copy = nullptr;  // small read-backs of finished frames (statistics), independent of frames in flight
    uint32_t frame_counter = 0;
    // single-pass binning (bounded per-bin queues); a frame that overflows them is re-run with the exact
    // two-pass queues and the bound is doubled for later frames
    bool direct_enabled = true;
    uint32_t qcap = 1024, scap = 128;
    // parked colour / depth sets.  The one piece of device state a second host thread may touch: a frame can be packed
    // (mtr_frame_pack_color_shard_on_stream) and destroyed on an exchange thread while the render thread begins others.
    std::mutex pool_mu;
    std::vector<ColorDepth> free_fb;
    std::vector<std::pair<uint64_t, uint32_t>> fb_allocated;  // (w << 32 | h) -> colour / depth sets ever allocated
    std::vector<std::unique_ptr<OwnTable>> own_tables;  // grow-only cache (submit_mu)
    bool cull_enabled = true;   // sharded frames cull chunks / instances against the rank's bins
    bool cull_unsharded = false;  // MTR_GEOM_CULL_ALL_FRAMES: unsharded frames cull against the target too (frustum culling)
    uint32_t vis_waves = 0;     // MTR_VIS_WAVES: waves per bin of the visibility kernel, 0 = by the number of bins
    // timing-ablation hooks, read ONCE at device creation (never in the submit path): MTR_CULL_DEBUG in {0, 1, 3, 4, 5}
    // replaces the culling mode of maps that cull (k_geom.hip: k_cull_instances), MTR_GEOM_SLOTS bounds the instance
    // slots the full-rate sharded geometry launch covers (tests force k_geom_rest with it); 0xFFFFFFFF / 0: not set
    uint32_t cull_debug = 0xFFFFFFFFu;
    uint32_t geom_slots = 0;
    // launch-size feedback of sharded batch draws (TileParams::hint_out): two words of pinned host memory per hint slot; a
    // batch takes a slot at its first culled draw and gives it back when it is destroyed.  A late write of a frame still
    // in flight into a slot that has changed hands only mis-sizes a launch (the second geometry launch covers the rest).
    static constexpr uint32_t kHintSlots = 256;
    uint32_t* hint_host = nullptr;
    uint32_t* hint_dev = nullptr;
    bool hint_used[kHintSlots] = {};

    // Tile-kernel bin order across the 8 XCDs.  One contiguous eighth of the bins per XCD keeps the records of
    // neighbouring bins in one L2 and gives the shortest stand-alone kernel (48.9 us), but the XCDs that own the empty top
    // and bottom of a frame run dry while the middle ones work; dealing runs of a quarter bin row to the XCDs in turn
    // costs the stand-alone kernel 2-3 us (locality) and gains 4-5 % of pipelined throughput on the headline scene
    // (0.0541 -> 0.0516 ms per frame, four runs each; runs of 16 / 60 bins: 0.0518 / 0.0514), neutral on C3-C5.
    // Unsharded frames only: a rank's band is a few rows (N = 4, 8: 32.6 -> 35.5, 30.3 -> 35.0 us per frame with runs).
    // And only while the previous frame is still on the GPU (one event query per frame): a frame that has the GPU to
    // itself keeps the contiguous order and its shorter kernel.  MTR_TILE_RUN overrides (0 = contiguous eighths).
    static constexpr uint32_t kXcdRunAuto = 0xFFFFFFFFu;
    uint32_t xcd_run = kXcdRunAuto;
    mtr_model* cube = nullptr;  // debug-overlay cube, created lazily
    struct Exchange* xchg = nullptr;  // exchange thread of a sharded device (mtr_device_exchange_start)
};

// The exchange of a sharded frame (pack -> the host's all-gather -> unpack -> frame destroy) issued by a second host
// thread: a rank's share of a small frame is ~30 us of GPU time, the render submission alone costs the host ~30 us, and
// the three exchange calls another ~15 us -- on one thread they add, on two they overlap.
struct Exchange {
    mtr_allgather_fn fn = nullptr;
    void* comm = nullptr;
    int dtype_u8 = 0;
    uint8_t *send = nullptr, *gathered = nullptr, *dst = nullptr;
    size_t send_bytes = 0;
    uint32_t world = 1;
    hipStream_t stream = nullptr;
    // further lanes (mtr_device_exchange_add_lane): frames are dealt to the lanes in turn, lane 0 being the fields above.
    // A lane is an in-order stream, so one lane completes one (pack + all-gather + unpack) latency per frame; two lanes
    // with a communicator each keep two collectives in flight.
    struct Lane { void* comm; uint8_t *send, *gathered, *dst; hipStream_t stream; };
    std::vector<Lane> lanes;
    uint64_t dealt = 0;  // frames taken by the thread so far
    std::thread th;
    std::mutex mu;
    std::condition_variable cv_items, cv_idle;
    std::deque<mtr_frame*> q;
    std::atomic<uint32_t> pending{0};  // queued + being processed
    bool stop = false;
    int32_t err = MTR_OK;
    std::string err_msg;
    static constexpr size_t kDepth = 8;  // frames handed over and not yet issued
};

struct mtr_texture {
    mtr_device* dev;
    uint32_t w, h, fmt;
    uint32_t levels = 1;  // mip levels in d_rgba, level 0 first
    uint8_t* d_rgba;          // decoded RGBA8 texels (MTR_TR_RGBA8) or the BC blocks as uploaded (MTR_TR_BC1 / MTR_TR_BC7)
    uint32_t resident = MTR_TR_RGBA8;
    bool opaque;  // every decoded texel (of every level) has alpha == 255: sampling it yields a == 1 exactly
};

// The chunk table of a model under one parts_disp: immutable once built.  A draw holds the table that was current when
// it was recorded (Model::render reads parts_disp while it records, src/model.rs:318-320), so a frame that is re-run
// after a queue overflow -- possibly by the exchange thread, possibly after the host has changed parts_disp for a later
// frame -- reproduces exactly what was submitted, and nobody rewrites a table a kernel or another thread is reading.
struct ChunkTable {
    int hip_dev = 0;
    std::vector<DChunk> chunks;
    DChunk* d_chunks = nullptr;
    uint64_t ntris_visible = 0;
    ~ChunkTable() {
        if (!d_chunks) return;
        (void)hipSetDevice(hip_dev);
        (void)hipDeviceSynchronize();  // frames that drew with it may still be in flight; tables die rarely (parts_disp changed)
        (void)hipFree(d_chunks);
    }
};

struct mtr_model {
    mtr_device* dev;
    uint8_t* d_vbuf = nullptr;
    uint16_t* d_ibuf = nullptr;
    DPrim* d_prims = nullptr;
    std::shared_ptr<const ChunkTable> table;  // for the current parts_disp; rebuilt by the next draw when chunks_dirty (submit_mu)
    float* d_palette = nullptr;   // the current palette: one buffer of pal_ring
    uint32_t npal = 0;
    // mtr_model_set_palette does not wait for frames in flight: every call uploads into the next buffer of a ring
    // (max_inflight + 1 of them) on the copy stream and records an event; a frame captures pointer + event when the
    // model is drawn and its stream waits on the event.  A ring buffer comes round again only after max_inflight + 1
    // palette changes; if the last frame that read it can still be in flight (many changes, few frames) the call waits
    // for exactly that frame first.
    // pinned: frames that drew the model with this buffer and may still (re-)run: recorded and not yet submitted, or
    // submitted and their overflow flags not yet examined (a frame whose bin queues overflowed is run again); the pin is
    // dropped when the flags turn out clean, or when the frame is destroyed
    struct PalBuf { float* d = nullptr; uint32_t cap = 0; hipEvent_t ready = nullptr; uint64_t last_frame = 0; bool used = false; uint32_t pinned = 0; };
    std::vector<PalBuf> pal_ring;
    size_t pal_next = 0;
    hipEvent_t pal_ready = nullptr;  // of the current palette
    int pal_slot = -1;               // its index in pal_ring
    std::vector<DPrim> prims;
    std::vector<uint16_t> indices;
    std::vector<uint32_t> run;  // consecutive non-restart indices ending at each position
    std::vector<uint8_t> parts_disp;
    std::vector<int32_t> prim_to_texture;
    std::vector<mtr_texture*> textures;
    std::vector<uint32_t> debug_rgba8;
    std::vector<mtr_prim_state> states;  // material state per primitive, empty: the reference's pipeline state
    std::vector<float> joint_cubes;      // one instance matrix per joint: scale 0.005, translation = offset * 0.01 (src/model.rs:309-315)
    bool chunks_dirty = true;
    size_t vertex_len = 0;
    // culling bounds (multi-GPU v2), computed once at creation over every chunk of every primitive, whatever parts_disp
    // says: chunk k of primitive p is static chunk prim_chunk_base[p] + k
    std::vector<uint32_t> prim_chunk_base;
    std::vector<uint32_t> cb_first, cb_count, cb_flags;  // per static chunk: its boxes in d_boxes
    BoneBox* d_boxes = nullptr;
    // whole-model boxes for instance culling: [0, n_inst_unskinned) what a draw without palette transforms (one box),
    // [n_inst_unskinned, +n_inst_skinned) what a skinned draw does (the unskinnable primitives' box, then one per joint)
    BoneBox* d_inst_boxes = nullptr;
    uint32_t n_inst_unskinned = 0, n_inst_skinned = 0;
    bool inst_skinned_boundable = false;  // every skinned vertex's weights sum to 255
};

struct mtr_batch {
    mtr_device* dev;
    mtr_model* model;
    uint32_t n = 0;
    float* d_model_mats = nullptr;
    float* d_palettes = nullptr;
    uint32_t npal = 0;
    std::vector<int32_t> tex_override;
    hipEvent_t ready = nullptr;  // the uploads (copy stream); frames that draw the batch wait on it
    uint64_t last_frame = 0;     // last frame that drew it: its buffers are freed only once that frame has left the GPU
    bool used = false;
    int hint_slot = -1;          // mtr_device::hint_host slot, or -1
    uint64_t hint_key = 0;       // the ownership (table, rank) the slot's numbers were reported under
};

struct BatchDeleter {
    void operator()(mtr_batch* b) const { mtr_batch_destroy(b); }
};

struct Draw {
    mtr_model* model;
    std::shared_ptr<const ChunkTable> table;  // the model's chunk table when the draw was recorded
    const float* d_model_mats;  // nullptr: M = view_proj
    const float* d_palettes;
    hipEvent_t pal_ready;  // upload of d_palettes / d_model_mats on the copy stream (model palette ring, or the batch)
    mtr_batch* batch;      // drawn batch (not owned unless owned_batch), for its last-use bookkeeping
    int pal_slot;          // ring buffer of the model palette, or -1
    uint32_t npal, pal_stride, ninst;
    float vp[16];
    std::vector<int32_t> tex_override;  // per instance or empty
    int shader_override;                // -1 or MTR_SH_CONST (overlay)
    uint32_t const_rgba8;
    bool blend;
    bool pal_pinned = false;  // holds a pin on the model's palette ring buffer pal_slot until the frame is submitted
    std::unique_ptr<mtr_batch, BatchDeleter> owned_batch;
};

struct mtr_frame {
    mtr_device* dev;
    uint32_t w, h;
    uint32_t clear_rgba8;
    float clear_depth;
    ColorDepth fb;
    uint32_t shard_rank = 0, shard_world = 1;
    const OwnTable* own = nullptr;  // ownership map of a sharded frame (cached in the device), nullptr: not sharded
    std::vector<Draw> draws;
    std::vector<DMat> mats_host;  // kept alive until the async upload has certainly been consumed
    bool submitted = false, waited = false, all_opaque = true, force_two_pass = false, ran_direct = false;
    mtr_frame_stats stats{};
    hipEvent_t ev[MTR_STAGE_COUNT + 1] = {};
    bool have_events = false;
    float ms[MTR_STAGE_COUNT] = {};
    int slot = 0;
    uint64_t total_chunks = 0;
    uint64_t min_entries = 0, min_segs = 0;  // queue sizes measured by a previous, overflowed attempt
    bool for_exchange = false;  // submitted through mtr_frame_submit_exchange: no public-stream consumer
    int status_idx = -1;        // this frame's word of mtr_device::status_host (set by run_frame)
    uint64_t frame_index = 0;   // its index in submission order (of the last run)
    bool flags_checked = false; // somebody has examined this run's overflow flags
    bool stats_valid = false;   // f->stats holds the device counters of the last run
};

namespace {

std::mutex g_err_mu;  // two host threads (render + exchange) may fail at once

int32_t fail(mtr_device* d, int32_t code, const std::string& msg) {
    std::lock_guard<std::mutex> g(g_err_mu);
    if (d) d->err = msg; else g_create_error = msg;
    return code;
}

#define HIPCHK(dev, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail((dev), MTR_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));      \
    } while (0)

uint32_t quant8(float x) {
    if (!(x > 0.0f)) x = 0.0f;
    if (x > 1.0f) x = 1.0f;
    return (uint32_t)std::rint(x * 255.0f);
}

uint32_t pack_rgba8(const float c[4]) {
    return quant8(c[0]) | (quant8(c[1]) << 8) | (quant8(c[2]) << 16) | (quant8(c[3]) << 24);
}

// src/shaders/debug_ids.wgsl:23-44
const uint8_t kDebugPalette[20][3] = {
    {215, 62, 103}, {95, 190, 80},  {133, 95, 213},  {180, 184, 53},  {213, 87, 180}, {72, 138, 55},  {145, 79, 158},
    {91, 196, 153}, {206, 78, 55},  {74, 174, 209},  {225, 133, 58},  {92, 122, 198}, {207, 162, 81}, {188, 144, 216},
    {152, 173, 92}, {161, 71, 103}, {53, 133, 98},   {225, 131, 152}, {111, 111, 40}, {162, 99, 55},
};

// bytes a float-class element reads, 0 = not in the reference's table (src/rshader2.rs:516-564)
uint32_t elem_bytes(uint8_t fmt, uint8_t cnt) {
    switch (fmt) {
    case MTR_IEF_U8N: return cnt == 1 ? 2 : cnt == 4 ? 4 : 0;
    case MTR_IEF_S8N: return cnt == 1 ? 2 : (cnt == 3 || cnt == 4) ? 4 : 0;
    case MTR_IEF_S16N: return cnt == 1 ? 4 : cnt == 3 ? 8 : 0;
    case MTR_IEF_F16: return cnt == 2 ? 4 : 0;
    case MTR_IEF_F32: return cnt == 3 ? 12 : 0;
    case MTR_IEF_U8NL: return cnt == 3 ? 4 : 0;
    case MTR_IEF_SCMP3N: return 4;  // only reached with MTR_ELEM_DECODE_SCMP3N
    default: return 0;
    }
}

template <class T>
int32_t dev_alloc(mtr_device* d, T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
    if (e != hipSuccess) return fail(d, MTR_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return MTR_OK;
}

template <class T>
int32_t dev_grow(mtr_device* d, T** p, uint32_t* cap, size_t need) {
    if (need <= *cap && *p) return MTR_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    int32_t rc = dev_alloc(d, p, need);
    if (rc) return rc;
    *cap = (uint32_t)need;
    return MTR_OK;
}

int32_t set_device(mtr_device* d) {
    HIPCHK(d, hipSetDevice(d->hip_dev));
    return MTR_OK;
}

// Waits for everything the library has queued: the public stream and every slot stream (frames handed to the exchange
// thread are not waited for by the public stream).  Used before freeing or overwriting what a frame in flight may read.
int32_t drain_all(mtr_device* d) {
    HIPCHK(d, hipStreamSynchronize(d->stream));
    for (uint32_t i = 0; i < d->nslots; i++)
        if (d->slots[i].stream) HIPCHK(d, hipStreamSynchronize(d->slots[i].stream));
    return MTR_OK;
}

uint32_t status_load(const mtr_device* d, int i) { return __atomic_load_n(&d->status_host[i], __ATOMIC_ACQUIRE); }

// bounded per-bin queues overflowed: later frames get twice the bound (up to 16384 entries per bin, then exact two-pass)
void grow_direct_queues(mtr_device* d) {
    if (d->qcap < 16384) { d->qcap *= 2; d->scap *= 2; }
    else d->direct_enabled = false;
}

// A frame nobody waited for raised an overflow flag: its pixels are missing triangles and it is gone.  Latch an error
// for the next API call that can report one, and raise the bounds so the frames that follow fit.  submit_mu held.
void latch_overflow(mtr_device* d, uint32_t flags, uint64_t frame_index) {
    if (flags & 4u) grow_direct_queues(d);
    if (flags & 2u) d->queue_scale = std::min<uint32_t>(d->queue_scale * 2, 1024);
    if (d->sticky_err == MTR_OK) {
        d->sticky_err = MTR_E_OVERFLOW;
        d->sticky_msg = "frame " + std::to_string(frame_index) + " overflowed its bin queues (flags " + std::to_string(flags) +
                        ") and was never waited for: it is missing triangles; queue bounds raised for later frames";
    }
}

// Looks at status word i if nobody has.  force: the word's frame is known to have left the GPU (a word that is still
// invalid then belongs to a frame without a tile workgroup).  submit_mu held.
void examine_status(mtr_device* d, int i, bool force) {
    if (d->status_checked[i]) return;
    const uint32_t v = status_load(d, i);
    if (!(v & 0x80000000u) && !force) return;
    d->status_checked[i] = true;
    d->status_released[i] = false;
    if (v & 0x7fffffffu) latch_overflow(d, v & 0x7fffffffu, d->status_owner[i]);
}

void poll_released(mtr_device* d) {
    for (uint32_t i = 0; i < d->max_inflight; i++)
        if (d->status_released[i]) examine_status(d, (int)i, false);
}

int32_t report_sticky(mtr_device* d) {
    if (d->sticky_err == MTR_OK) return MTR_OK;
    const int32_t rc = d->sticky_err;
    const std::string msg = d->sticky_msg;
    d->sticky_err = MTR_OK;
    d->sticky_msg.clear();
    return fail(d, rc, msg);
}

// ---- ownership maps (mtr_internal.h: Ownership) ----
bool valid_own_args(uint32_t w, uint32_t h, uint32_t world, uint32_t map, uint32_t param, const uint32_t* band_rows) {
    if (w == 0 || h == 0 || w > 16384 || h > 16384 || world == 0 || world > 4096) return false;
    const uint32_t nby = (h + MTR_BIN - 1) / MTR_BIN;
    if (map == MTR_OWN_INTERLEAVED) return true;
    if (map == MTR_OWN_SUPERTILES) return param <= 6;
    if (map != MTR_OWN_BANDS) return false;
    if (band_rows) {
        if (band_rows[0] != 0 || band_rows[world] != nby) return false;
        for (uint32_t r = 0; r < world; r++)
            if (band_rows[r] > band_rows[r + 1]) return false;
    }
    return true;
}

// host lists of a map: lists = every bin, rank after rank; offs[r] = where rank r's share starts
void build_own_lists(uint32_t w, uint32_t h, uint32_t world, uint32_t map, uint32_t param, const uint32_t* band_rows,
                     std::vector<uint32_t>& bands, std::vector<uint32_t>& lists, std::vector<uint32_t>& offs) {
    const uint32_t nbx = (w + MTR_BIN - 1) / MTR_BIN, nby = (h + MTR_BIN - 1) / MTR_BIN, nbins = nbx * nby;
    bands.clear();
    if (map == MTR_OWN_BANDS) {
        bands.resize(world + 1);
        for (uint32_t r = 0; r <= world; r++) bands[r] = band_rows ? band_rows[r] : (uint32_t)((uint64_t)r * nby / world);
    }
    std::vector<std::vector<uint32_t>> per(world);
    if (map == MTR_OWN_BANDS) {
        for (uint32_t r = 0; r < world; r++)
            for (uint32_t b = bands[r] * nbx; b < bands[r + 1] * nbx; b++) per[r].push_back(b);
    } else if (map == MTR_OWN_SUPERTILES) {
        const uint32_t S = 1u << param, nsx = (nbx + S - 1) >> param, nsy = (nby + S - 1) >> param;
        for (uint32_t st = 0; st < nsx * nsy; st++) {
            const uint32_t sx = st % nsx, sy = st / nsx;
            for (uint32_t by = sy * S; by < std::min(nby, (sy + 1) * S); by++)
                for (uint32_t bx = sx * S; bx < std::min(nbx, (sx + 1) * S); bx++) per[st % world].push_back(by * nbx + bx);
        }
    } else {
        for (uint32_t b = 0; b < nbins; b++) per[b % world].push_back(b);
    }
    lists.clear();
    offs.assign(world + 1, 0);
    for (uint32_t r = 0; r < world; r++) {
        lists.insert(lists.end(), per[r].begin(), per[r].end());
        offs[r + 1] = (uint32_t)lists.size();
    }
}

uint32_t stride_of(const std::vector<uint32_t>& offs) {
    uint32_t s = 0;
    for (size_t r = 0; r + 1 < offs.size(); r++) s = std::max(s, offs[r + 1] - offs[r]);
    return s;
}

// the device's cached table for a map (built and uploaded on first use); submit_mu held
int32_t get_own_table(mtr_device* d, uint32_t w, uint32_t h, uint32_t world, uint32_t map, uint32_t param, const uint32_t* band_rows,
                      const OwnTable** out) {
    *out = nullptr;
    if (!valid_own_args(w, h, world, map, param, band_rows)) return fail(d, MTR_E_INVALID, "bad ownership map arguments");
    if (map != MTR_OWN_SUPERTILES) param = 0;
    const uint32_t nby = (h + MTR_BIN - 1) / MTR_BIN;
    std::vector<uint32_t> bands;
    if (map == MTR_OWN_BANDS) {
        bands.resize(world + 1);
        for (uint32_t r = 0; r <= world; r++) bands[r] = band_rows ? band_rows[r] : (uint32_t)((uint64_t)r * nby / world);
    }
    for (auto& t : d->own_tables)
        if (t->w == w && t->h == h && t->map == map && t->param == param && t->world == world && t->bands == bands) { *out = t.get(); return MTR_OK; }
    if (d->own_tables.size() >= 64) {  // a host that keeps changing the map: drop the tables no live frame uses
        int32_t rc = drain_all(d);        // (nothing in flight may still read them)
        if (rc) return rc;
        size_t keep = 0;
        for (auto& t : d->own_tables) {
            if (t->refs) { d->own_tables[keep++] = std::move(t); continue; }
            (void)hipFree(t->d_lists); (void)hipFree(t->d_src_of_bin);
        }
        d->own_tables.resize(keep);
    }
    auto t = std::make_unique<OwnTable>();
    t->w = w; t->h = h; t->map = map; t->param = param; t->world = world;
    std::vector<uint32_t> lists;
    build_own_lists(w, h, world, map, param, band_rows, t->bands, lists, t->offs);
    t->stride_bins = stride_of(t->offs);
    t->st_shift = param; t->nsx = (((w + MTR_BIN - 1) / MTR_BIN) + (1u << param) - 1) >> param;
    std::vector<uint32_t> src(lists.size());
    for (uint32_t r = 0; r < world; r++)
        for (uint32_t k = t->offs[r]; k < t->offs[r + 1]; k++) src[lists[k]] = r * t->stride_bins + (k - t->offs[r]);
    int32_t rc = dev_alloc(d, &t->d_lists, lists.size());
    if (!rc) rc = dev_alloc(d, &t->d_src_of_bin, src.size());
    if (rc) return rc;
    HIPCHK(d, hipMemcpy(t->d_lists, lists.data(), lists.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(d, hipMemcpy(t->d_src_of_bin, src.data(), src.size() * 4, hipMemcpyHostToDevice));
    t->lists = std::move(lists);
    *out = t.get();
    d->own_tables.push_back(std::move(t));
    return MTR_OK;
}

// ---- host mirror of the position decode (csrc/geom_common.h: decode_elem), for the culling bounds ----
float h_half(uint16_t h) {
    const uint32_t sign = (uint32_t)(h >> 15) << 31, ex = (h >> 10) & 0x1f, man = h & 0x3ff;
    uint32_t bits;
    if (ex == 0) {
        if (man == 0) bits = sign;
        else {  // subnormal: man * 2^-24
            float f = (float)man * 5.9604644775390625e-08f;
            memcpy(&bits, &f, 4);
            bits |= sign;
        }
    } else if (ex == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((ex + 112) << 23) | (man << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}
float h_snorm16(uint16_t v) { float f = (float)(int16_t)v / 32767.0f; return f < -1.0f ? -1.0f : f; }
float h_snorm8(uint8_t v) { float f = (float)(int8_t)v / 127.0f; return f < -1.0f ? -1.0f : f; }
float h_unorm8(uint8_t v) { return (float)v / 255.0f; }
uint16_t h_ld16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

void decode_pos_host(uint32_t fmt, uint32_t cnt, const uint8_t* p, float (&o)[3]) {
    o[0] = o[1] = o[2] = 0.0f;
    switch (fmt) {
    case MTR_IEF_U8N: case MTR_IEF_U8NL:
        o[0] = h_unorm8(p[0]); o[1] = h_unorm8(p[1]);
        if (!(fmt == MTR_IEF_U8N && cnt == 1)) o[2] = h_unorm8(p[2]);
        break;
    case MTR_IEF_S8N:
        o[0] = h_snorm8(p[0]); o[1] = h_snorm8(p[1]);
        if (cnt != 1) o[2] = h_snorm8(p[2]);
        break;
    case MTR_IEF_S16N:
        o[0] = h_snorm16(h_ld16(p)); o[1] = h_snorm16(h_ld16(p + 2));
        if (cnt == 3) o[2] = h_snorm16(h_ld16(p + 4));
        break;
    case MTR_IEF_F16:
        o[0] = h_half(h_ld16(p)); o[1] = h_half(h_ld16(p + 2));
        break;
    case MTR_IEF_F32:
        memcpy(&o[0], p, 4); memcpy(&o[1], p + 4, 4); memcpy(&o[2], p + 8, 4);
        break;
    case MTR_IEF_SCMP3N: {
        uint32_t w;
        memcpy(&w, p, 4);
        for (int k = 0; k < 3; k++) {
            const float f = (float)((int32_t)((w >> (10 * k)) << 22) >> 22) / 511.0f;
            o[k] = f < -1.0f ? -1.0f : f;
        }
        break;
    }
    default: break;
    }
}

struct BoxAcc {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    bool any = false, bad = false;
    void add(const float (&p)[3]) {
        for (int k = 0; k < 3; k++) {
            if (!(std::fabs(p[k]) < 3.0e38f)) bad = true;  // NaN / inf position: the box cannot hold it
            lo[k] = std::min(lo[k], (double)p[k]); hi[k] = std::max(hi[k], (double)p[k]);
        }
        any = true;
    }
    void merge(const BoxAcc& o) {
        if (!o.any) return;
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], o.lo[k]); hi[k] = std::max(hi[k], o.hi[k]); }
        any = true; bad = bad || o.bad;
    }
    BoneBox box(uint32_t joint) const {
        BoneBox b{};
        float* c = &b.cx; float* e = &b.ex;
        for (int k = 0; k < 3; k++) {
            const float cf = (float)((lo[k] + hi[k]) * 0.5);
            const double ext = std::max(hi[k] - (double)cf, (double)cf - lo[k]);
            c[k] = cf;
            e[k] = std::nextafter((float)ext, INFINITY);  // rounded up
            if (bad) e[k] = INFINITY;                      // the test sees a non-finite interval and keeps the geometry
        }
        b.joint = joint;
        return b;
    }
};

// Chunk and whole-model bounds of a new model (vertex bytes still on the host).  boxes: the d_boxes image.
void build_bounds(mtr_model* m, const uint8_t* vbuf, std::vector<BoneBox>& boxes, std::vector<BoneBox>& inst_boxes) {
    const size_t nprims = m->prims.size();
    m->prim_chunk_base.assign(nprims + 1, 0);
    for (size_t p = 0; p < nprims; p++)
        m->prim_chunk_base[p + 1] = m->prim_chunk_base[p] + (m->prims[p].index_num + MTR_CHUNK_NEW - 1) / MTR_CHUNK_NEW;
    const size_t nstatic = m->prim_chunk_base[nprims];
    m->cb_first.assign(nstatic, 0); m->cb_count.assign(nstatic, 0); m->cb_flags.assign(nstatic, 0);
    BoxAcc all_unskinned, rigid_part;   // every vertex / the vertices of primitives that cannot be skinned
    std::vector<BoxAcc> joint_acc(256);
    bool weights_ok = true;
    struct JA { uint32_t joint; BoxAcc acc; };
    std::vector<JA> ja;
    for (size_t p = 0; p < nprims; p++) {
        const DPrim& pr = m->prims[p];
        const uint8_t* vb = vbuf + pr.vertex_base;
        for (uint32_t start = 0, k = 0; start < pr.index_num; start += MTR_CHUNK_NEW, k++) {
            const size_t sc = m->prim_chunk_base[p] + k;
            BoxAcc whole;
            ja.clear();
            uint32_t flags = 0;
            const uint32_t lo = start >= 2 ? start - 2 : 0, hi = std::min(pr.index_num, start + MTR_CHUNK_NEW);
            for (uint32_t q = lo; q < hi; q++) {
                const uint32_t idx = m->indices[pr.index_ofs + q];
                if (pr.topology == 4 && idx == 0xFFFFu) continue;
                const uint32_t vid = idx + pr.index_base;
                if (vid >= pr.vertex_num) continue;
                const uint8_t* vp = vb + (size_t)vid * pr.stride;
                float pos[3];
                decode_pos_host(pr.pos_fmt, pr.pos_cnt, vp + pr.pos_off, pos);
                whole.add(pos);
                if (!pr.skinnable) continue;
                const uint8_t* jp = vp + pr.joint_off; const uint8_t* wp = vp + pr.weight_off;
                if ((uint32_t)wp[0] + wp[1] + wp[2] + wp[3] != 255u) { flags |= 1u; weights_ok = false; }
                for (int t = 0; t < 4; t++) {
                    if (wp[t] == 0) continue;  // contributes exactly nothing to the blend
                    joint_acc[jp[t]].add(pos);
                    size_t e = 0;
                    while (e < ja.size() && ja[e].joint != jp[t]) e++;
                    if (e == ja.size()) ja.push_back({jp[t], BoxAcc()});
                    ja[e].acc.add(pos);
                }
            }
            all_unskinned.merge(whole);
            if (!pr.skinnable) rigid_part.merge(whole);
            if (!whole.any) continue;  // no vertex: nothing to bound (the chunk is kept, it has no triangle anyway)
            if (ja.size() > MTR_CHUNK_MAX_BOXES) { flags |= 1u; ja.clear(); }
            m->cb_first[sc] = (uint32_t)boxes.size();
            m->cb_count[sc] = 1u + (uint32_t)ja.size();
            m->cb_flags[sc] = flags;
            boxes.push_back(whole.box(MTR_BOX_UNSKINNED));
            for (const JA& e : ja) boxes.push_back(e.acc.box(e.joint));
        }
    }
    inst_boxes.clear();
    if (all_unskinned.any) inst_boxes.push_back(all_unskinned.box(MTR_BOX_UNSKINNED));
    m->n_inst_unskinned = (uint32_t)inst_boxes.size();
    if (rigid_part.any) inst_boxes.push_back(rigid_part.box(MTR_BOX_UNSKINNED));
    for (uint32_t j = 0; j < 256; j++)
        if (joint_acc[j].any) inst_boxes.push_back(joint_acc[j].box(j));
    m->n_inst_skinned = (uint32_t)inst_boxes.size() - m->n_inst_unskinned;
    m->inst_skinned_boundable = weights_ok;
}

void rebuild_chunks(const mtr_model* m, ChunkTable* t) {
    t->chunks.clear();
    t->ntris_visible = 0;
    for (size_t p = 0; p < m->prims.size(); p++) {
        const DPrim& pr = m->prims[p];
        if (pr.parts_no >= m->parts_disp.size() || !m->parts_disp[pr.parts_no]) continue;  // src/model.rs:318-320
        for (uint32_t start = 0; start < pr.index_num; start += MTR_CHUNK_NEW) {
            DChunk c;
            c.prim = (uint32_t)p;
            c.start = start;
            c.q_before = start >= 3 ? m->run[pr.index_ofs + start - 3] : 0;
            // the run must not reach back before this primitive's first index
            if (start >= 3 && c.q_before > start - 2) c.q_before = start - 2;
            uint32_t nt = 0, end = std::min(pr.index_num, start + MTR_CHUNK_NEW);
            for (uint32_t q = start; q < end; q++) {
                if (pr.topology == 4) {
                    uint32_t r = std::min(m->run[pr.index_ofs + q], q + 1);
                    nt += r >= 3;
                } else {
                    nt += (q % 3 == 2);
                }
            }
            c.ntris = nt;
            const size_t sc = m->prim_chunk_base[p] + start / MTR_CHUNK_NEW;
            c.b_first = m->cb_first[sc]; c.b_count = m->cb_count[sc]; c.b_flags = m->cb_flags[sc] | (pr.skinnable ? 2u : 0u); c.pad = 0;
            t->ntris_visible += nt;
            t->chunks.push_back(c);
        }
    }
}

// The model's chunk table for its current parts_disp, built and uploaded on first use.  submit_mu held.
int32_t current_table(mtr_model* m, std::shared_ptr<const ChunkTable>* out) {
    mtr_device* d = m->dev;
    if (m->chunks_dirty || !m->table) {
        auto t = std::make_shared<ChunkTable>();
        t->hip_dev = d->hip_dev;
        rebuild_chunks(m, t.get());
        int32_t rc = dev_alloc(d, &t->d_chunks, t->chunks.size());
        if (rc) return rc;
        if (!t->chunks.empty()) HIPCHK(d, hipMemcpy(t->d_chunks, t->chunks.data(), t->chunks.size() * sizeof(DChunk), hipMemcpyHostToDevice));
        m->table = std::move(t);
        m->chunks_dirty = false;
    }
    *out = m->table;
    return MTR_OK;
}

}  // namespace

extern "C" {

int32_t mtr_abi_version(void) { return MTR_ABI_VERSION; }

const char* mtr_last_error(const mtr_device* dev) { return dev ? dev->err.c_str() : g_create_error.c_str(); }

int32_t mtr_device_create_on_stream(int32_t hip_device, void* hip_stream, mtr_device** out) {
    if (!out) return fail(nullptr, MTR_E_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, MTR_E_HIP, std::string("no HIP device: ") + hipGetErrorString(e));
    if (hip_device < 0 || hip_device >= n) return fail(nullptr, MTR_E_INVALID, "hip_device out of range");
    auto d = std::make_unique<mtr_device>();
    d->hip_dev = hip_device;
    HIPCHK(nullptr, hipSetDevice(hip_device));
    if (hip_stream) {
        d->stream = reinterpret_cast<hipStream_t>(hip_stream);
    } else {
        HIPCHK(nullptr, hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
        d->own_stream = true;
    }
    if (const char* e = getenv("MTR_NSLOTS")) {
        const long v = strtol(e, nullptr, 10);
        if (v >= 1 && v <= MTR_MAX_SLOTS) d->nslots = (uint32_t)v;
    }
    for (uint32_t i = 0; i < d->nslots; i++) HIPCHK(nullptr, hipStreamCreateWithFlags(&d->slots[i].stream, hipStreamNonBlocking));
    HIPCHK(nullptr, hipStreamCreateWithFlags(&d->s_copy, hipStreamNonBlocking));
    HIPCHK(nullptr, hipHostMalloc(reinterpret_cast<void**>(&d->status_host), mtr_device::kMaxInflight * sizeof(uint32_t),
                                  hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(nullptr, hipHostGetDevicePointer(reinterpret_cast<void**>(&d->status_dev), d->status_host, 0));
    for (uint32_t i = 0; i < mtr_device::kMaxInflight; i++) { d->status_host[i] = 0x80000000u; d->status_checked[i] = true; }
    HIPCHK(nullptr, hipHostMalloc(reinterpret_cast<void**>(&d->hint_host), mtr_device::kHintSlots * 2 * sizeof(uint32_t),
                                  hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(nullptr, hipHostGetDevicePointer(reinterpret_cast<void**>(&d->hint_dev), d->hint_host, 0));
    memset(d->hint_host, 0, mtr_device::kHintSlots * 2 * sizeof(uint32_t));
    if (const char* e = getenv("MTR_VIS_WAVES")) {
        const long v = strtol(e, nullptr, 10);
        if (v == 2 || v == 4 || v == 8) d->vis_waves = (uint32_t)v;
    }
    if (const char* e = getenv("MTR_TILE_RUN")) {
        const long v = strtol(e, nullptr, 10);
        if (v >= 0 && v <= 65536) d->xcd_run = (uint32_t)v;
    }
    if (const char* e = getenv("MTR_CULL_DEBUG")) {
        const long v = strtol(e, nullptr, 10);
        if (v == 0 || v == 1 || v == 3 || v == 4 || v == 5) d->cull_debug = (uint32_t)v;
    }
    if (const char* e = getenv("MTR_GEOM_SLOTS")) {
        const long v = strtol(e, nullptr, 10);
        if (v >= 1 && v <= 0xFFFF) d->geom_slots = (uint32_t)v;
    }
    if (const char* e = getenv("MTR_MAX_INFLIGHT")) {
        const long v = strtol(e, nullptr, 10);
        if (v >= 1 && v <= (long)mtr_device::kMaxInflight) d->max_inflight = (uint32_t)v;
    }
    *out = d.release();
    return MTR_OK;
}

int32_t mtr_device_create(int32_t hip_device, mtr_device** out) {
    return mtr_device_create_on_stream(hip_device, nullptr, out);
}

void mtr_device_destroy(mtr_device* d) {
    if (!d) return;
    (void)mtr_device_exchange_stop(d);
    (void)hipSetDevice(d->hip_dev);
    (void)hipStreamSynchronize(d->stream);
    for (Slot& sl : d->slots)
        if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    for (hipEvent_t e : d->inflight)
        if (e) (void)hipEventDestroy(e);
    for (auto& g : d->garbage) (void)hipFree(g.p);
    if (d->s_copy) (void)hipStreamDestroy(d->s_copy);
    if (d->cube) mtr_model_destroy(d->cube);
    for (auto& t : d->own_tables) {
        if (t->d_lists) (void)hipFree(t->d_lists);
        if (t->d_src_of_bin) (void)hipFree(t->d_src_of_bin);
    }
    for (auto& f : d->free_fb) {
        (void)hipFree(f.color);
        (void)hipFree(f.depth);
        (void)hipFree(f.counters);
        if (f.done) (void)hipEventDestroy(f.done);
    }
    for (Slot& sl : d->slots) {
        void* ptrs[] = {sl.rec_hdr, sl.rec_a, sl.rec_l, sl.rec_b, sl.chunk_info, sl.bin_count, sl.bin_fill,
                        sl.bin_start, sl.seg_start, sl.entries, sl.segs, sl.mats, sl.bin_flag, sl.inst_list, sl.inst_count, sl.work_mask, sl.comp};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
    }
    if (d->own_stream) (void)hipStreamDestroy(d->stream);
    if (d->status_host) (void)hipHostFree(d->status_host);
    if (d->hint_host) (void)hipHostFree(d->hint_host);
    delete d;
}

int32_t mtr_device_synchronize(mtr_device* d) {
    if (!d) return MTR_E_INVALID;
    int32_t rc = set_device(d);
    if (rc) return rc;
    if ((rc = mtr_device_exchange_drain(d))) return rc;
    if ((rc = drain_all(d))) return rc;
    std::lock_guard<std::mutex> submit_lock(d->submit_mu);
    for (uint32_t i = 0; i < d->max_inflight; i++) examine_status(d, (int)i, true);  // every frame has left the GPU
    return report_sticky(d);
}

int32_t mtr_device_set_tile_mode(mtr_device* d, int32_t mode) {
    if (!d) return MTR_E_INVALID;
    if (mode != MTR_TILE_AUTO && mode != MTR_TILE_ORDERED && mode != MTR_TILE_VISIBILITY)
        return fail(d, MTR_E_INVALID, "unknown tile mode");
    d->tile_mode = mode;
    return MTR_OK;
}

int32_t mtr_device_set_binning(mtr_device* d, int32_t single_pass, uint32_t queue_capacity) {
    if (!d) return MTR_E_INVALID;
    if (queue_capacity && (queue_capacity < 64 || queue_capacity > 65536)) return fail(d, MTR_E_INVALID, "queue capacity out of range");
    std::lock_guard<std::mutex> g(d->submit_mu);  // the exchange thread grows the bound when it re-runs an overflowed frame
    d->direct_enabled = single_pass != 0;
    if (queue_capacity) { d->qcap = queue_capacity; d->scap = std::max<uint32_t>(16, queue_capacity / 8); }
    return MTR_OK;
}

int32_t mtr_device_set_profiling(mtr_device* d, int32_t enable) {
    if (!d) return MTR_E_INVALID;
    d->profiling = enable != 0;
    return MTR_OK;
}

// ---------------------------------------------------------------------------------------------
// Texture::new
// ---------------------------------------------------------------------------------------------
int32_t mtr_texture_create_mips(mtr_device* d, uint32_t w, uint32_t h, uint32_t fmt, uint32_t levels, const void* data, size_t len,
                                mtr_texture** out) {
    if (!d || !out) return MTR_E_INVALID;
    *out = nullptr;
    if (!data || w == 0 || h == 0 || w > 16384 || h > 16384) return fail(d, MTR_E_INVALID, "bad texture size/data");
    if (levels == 0 || levels > 15 || (levels > 1 && (w >> (levels - 1)) == 0 && (h >> (levels - 1)) == 0))
        return fail(d, MTR_E_INVALID, "more mip levels than the texture size allows");
    if (fmt != MTR_TEX_RGBA8 && fmt != MTR_TEX_BC1 && fmt != MTR_TEX_BC7 && fmt != MTR_TEX_BC7_ALT)
        return fail(d, MTR_E_UNSUPPORTED, "unhandled texture format " + std::to_string(fmt));  // src/rtexture.rs:159
    // level l: max(1, w >> l) x max(1, h >> l), stored level after level in both the source and the decoded image
    size_t need = 0, texels = 0;
    for (uint32_t l = 0; l < levels; l++) {
        const size_t lw = std::max(1u, w >> l), lh = std::max(1u, h >> l);
        need += fmt == MTR_TEX_RGBA8 ? lw * lh * 4 : ((lw + 3) / 4) * ((lh + 3) / 4) * (fmt == MTR_TEX_BC1 ? 8 : 16);
        texels += lw * lh;
    }
    if (len < need) return fail(d, MTR_E_INVALID, "texture data too short");
    int32_t rc = set_device(d);
    if (rc) return rc;
    auto t = std::make_unique<mtr_texture>();
    t->dev = d; t->w = w; t->h = h; t->fmt = fmt; t->levels = levels; t->d_rgba = nullptr;
    uint8_t* d_rgba = nullptr;    // the decoded image: the resident one, or (blocks resident) a temporary for the alpha scan
    uint8_t* d_blocks = nullptr;
    const bool keep_blocks = fmt != MTR_TEX_RGBA8 && d->texture_residency == MTR_TEXRES_BLOCKS;
    rc = dev_alloc(d, &d_rgba, texels * 4);
    if (rc) return rc;
    if (fmt == MTR_TEX_RGBA8) {
        HIPCHK(d, hipMemcpyAsync(d_rgba, data, need, hipMemcpyHostToDevice, d->stream));
        HIPCHK(d, hipStreamSynchronize(d->stream));
    } else {
        rc = dev_alloc(d, &d_blocks, need);
        if (rc) { (void)hipFree(d_rgba); return rc; }
        HIPCHK(d, hipMemcpyAsync(d_blocks, data, need, hipMemcpyHostToDevice, d->stream));
        size_t src_off = 0, dst_off = 0;
        for (uint32_t l = 0; l < levels; l++) {
            const uint32_t lw = std::max(1u, w >> l), lh = std::max(1u, h >> l);
            if (fmt == MTR_TEX_BC1) mtr_launch_bc1_decode(d_blocks + src_off, d_rgba + dst_off, lw, lh, d->stream);
            else mtr_launch_bc7_decode(d_blocks + src_off, d_rgba + dst_off, lw, lh, d->stream);
            src_off += (size_t)((lw + 3) / 4) * ((lh + 3) / 4) * (fmt == MTR_TEX_BC1 ? 8 : 16);
            dst_off += (size_t)lw * lh * 4;
        }
        HIPCHK(d, hipGetLastError());
        HIPCHK(d, hipStreamSynchronize(d->stream));
    }
    {
        uint32_t* d_min = nullptr;
        uint32_t h_min = 255;
        if ((rc = dev_alloc(d, &d_min, 1))) return rc;
        HIPCHK(d, hipMemcpyAsync(d_min, &h_min, 4, hipMemcpyHostToDevice, d->stream));
        mtr_launch_alpha_min(d_rgba, texels, d_min, d->stream);
        HIPCHK(d, hipGetLastError());
        HIPCHK(d, hipMemcpyAsync(&h_min, d_min, 4, hipMemcpyDeviceToHost, d->stream));
        HIPCHK(d, hipStreamSynchronize(d->stream));
        (void)hipFree(d_min);
        t->opaque = h_min == 255;
    }
    if (keep_blocks) {  // the sampler decodes the texel's block per fetch (csrc/bc_sample.h): 1/4 (BC7) or 1/8 (BC1) of the bytes
        (void)hipFree(d_rgba);
        t->d_rgba = d_blocks;
        t->resident = fmt == MTR_TEX_BC1 ? MTR_TR_BC1 : MTR_TR_BC7;
    } else {
        if (d_blocks) (void)hipFree(d_blocks);
        t->d_rgba = d_rgba;
    }
    *out = t.release();
    return MTR_OK;
}

int32_t mtr_texture_create(mtr_device* d, uint32_t w, uint32_t h, uint32_t fmt, const void* data, size_t len, mtr_texture** out) {
    return mtr_texture_create_mips(d, w, h, fmt, 1, data, len, out);
}

void mtr_texture_destroy(mtr_texture* t) {
    if (!t) return;
    (void)hipSetDevice(t->dev->hip_dev);
    (void)drain_all(t->dev);  // frames in flight (on any slot stream) may still sample it
    (void)hipFree(t->d_rgba);
    delete t;
}

int32_t mtr_texture_read_rgba8(mtr_texture* t, void* out, size_t len) {
    if (!t || !out) return MTR_E_INVALID;
    mtr_device* d = t->dev;
    if (len < (size_t)t->w * t->h * 4) return fail(d, MTR_E_INVALID, "output too small");
    if (t->resident != MTR_TR_RGBA8) return fail(d, MTR_E_UNSUPPORTED, "the texture is resident as BC blocks (mtr_device_set_texture_residency): there is no decoded image to read");
    int32_t rc = set_device(d);
    if (rc) return rc;
    HIPCHK(d, hipMemcpyAsync(out, t->d_rgba, (size_t)t->w * t->h * 4, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(d, hipStreamSynchronize(d->stream));
    return MTR_OK;
}

// ---------------------------------------------------------------------------------------------
// Model::new
// ---------------------------------------------------------------------------------------------
int32_t mtr_model_create(mtr_device* d, const void* vertex_buf, size_t vertex_len, const uint16_t* index_buf,
                         size_t index_num, const mtr_primitive* prims, size_t nprims, const mtr_layout* layouts,
                         const int32_t* prim_to_texture, mtr_texture* const* textures, size_t ntextures,
                         const uint32_t* prim_debug_id, mtr_model** out) {
    if (!d || !out) return MTR_E_INVALID;
    *out = nullptr;
    if (!vertex_buf || !index_buf || !prims || !layouts || nprims == 0 || nprims > 0xFFFF)
        return fail(d, MTR_E_INVALID, "null or empty model input");
    if (index_num > 0x7FFFFFFFu || vertex_len > 0xFFFFFFFFu) return fail(d, MTR_E_INVALID, "model too large");
    auto m = std::make_unique<mtr_model>();
    m->dev = d;
    m->vertex_len = vertex_len;
    m->prims.resize(nprims);
    m->prim_to_texture.assign(nprims, -1);
    m->debug_rgba8.resize(nprims);
    for (size_t t = 0; t < ntextures; t++) {
        if (!textures || !textures[t] || textures[t]->dev != d) return fail(d, MTR_E_INVALID, "bad texture handle");
        m->textures.push_back(textures[t]);
    }
    for (size_t p = 0; p < nprims; p++) {
        const uint32_t* w = prims[p].w;  // bit-fields: src/rmodel.rs:173-225
        DPrim& pr = m->prims[p];
        memset(&pr, 0, sizeof pr);
        pr.vertex_num = (w[0] >> 16) & 0xffff;
        pr.parts_no = w[1] & 0xfff;
        pr.stride = (w[2] >> 16) & 0xff;
        pr.topology = (w[2] >> 24) & 0x3f;
        pr.vertex_base = w[4];
        pr.index_ofs = w[6];
        pr.index_num = w[7];
        pr.index_base = w[8];
        if (pr.topology != 4 && pr.topology != 3)  // PrimitiveTopology::from_repr().unwrap(), src/rmodel.rs:215
            return fail(d, MTR_E_UNSUPPORTED, "primitive " + std::to_string(p) + ": topology " + std::to_string(pr.topology));
        const mtr_layout& l = layouts[p];
        if (l.num_elements > 8) return fail(d, MTR_E_INVALID, "layout has more than 8 elements");
        bool has_pos = false, has_joint = false, has_weight = false;
        uint32_t align_or = pr.vertex_base | pr.stride;
        for (uint32_t i = 0; i < l.num_elements; i++) {
            const mtr_element& e = l.elements[i];
            if (e.format == MTR_IEF_SCMP3N && !(e.flags & MTR_ELEM_DECODE_SCMP3N)) continue;  // src/rshader2.rs:509-512
            if (e.semantic == MTR_SEM_POSITION || e.semantic == MTR_SEM_TEXCOORD) {
                uint32_t nb = elem_bytes(e.format, e.count);
                if (nb == 0)  // todo!() arms of src/rshader2.rs:516-564 and integer formats
                    return fail(d, MTR_E_UNSUPPORTED, "primitive " + std::to_string(p) + ": unhandled element format " +
                                                         std::to_string(e.format) + " x" + std::to_string(e.count));
                if ((uint32_t)e.offset + nb > pr.stride) return fail(d, MTR_E_INVALID, "element outside the vertex stride");
                align_or |= e.offset;
                if (e.semantic == MTR_SEM_POSITION) {
                    has_pos = true; pr.pos_fmt = e.format; pr.pos_cnt = e.count; pr.pos_off = e.offset;
                } else {
                    pr.has_uv = 1; pr.uv_fmt = e.format; pr.uv_cnt = e.count; pr.uv_off = e.offset;
                }
            } else if (e.semantic == MTR_SEM_JOINT) {
                if (e.format != MTR_IEF_U8 || e.count != 4) return fail(d, MTR_E_UNSUPPORTED, "Joint must be U8 x4");
                if ((uint32_t)e.offset + 4 > pr.stride) return fail(d, MTR_E_INVALID, "element outside the vertex stride");
                has_joint = true; pr.joint_off = e.offset; align_or |= e.offset;
            } else if (e.semantic == MTR_SEM_WEIGHT) {
                if (e.format != MTR_IEF_U8N || e.count != 4) return fail(d, MTR_E_UNSUPPORTED, "Weight must be U8N x4");
                if ((uint32_t)e.offset + 4 > pr.stride) return fail(d, MTR_E_INVALID, "element outside the vertex stride");
                has_weight = true; pr.weight_off = e.offset; align_or |= e.offset;
            }  // other names: `_ => continue`, src/rshader2.rs:506
        }
        if (!has_pos) return fail(d, MTR_E_UNSUPPORTED, "primitive " + std::to_string(p) + ": no Position element");
        pr.skinnable = has_joint && has_weight;
        pr.aligned4 = (align_or & 3) == 0;
        if ((size_t)pr.vertex_base + (size_t)pr.vertex_num * pr.stride > vertex_len)
            return fail(d, MTR_E_INVALID, "primitive " + std::to_string(p) + ": vertex slice outside the buffer");
        if ((size_t)pr.index_ofs + pr.index_num > index_num)
            return fail(d, MTR_E_INVALID, "primitive " + std::to_string(p) + ": index range outside the buffer");
        int32_t tex = prim_to_texture ? prim_to_texture[p] : -1;
        if (tex >= (int32_t)ntextures) return fail(d, MTR_E_INVALID, "prim_to_texture out of range");
        m->prim_to_texture[p] = tex < 0 ? -1 : tex;
        uint32_t id = prim_debug_id ? prim_debug_id[p] : 0;
        float c[4];
        for (int k = 0; k < 3; k++) c[k] = (float)kDebugPalette[id % 20][k] / 255.0f;  // debug_ids.wgsl:46
        c[3] = 1.0f;
        m->debug_rgba8[p] = pack_rgba8(c);
    }
    m->parts_disp.assign(nprims, 1);  // src/model.rs:270
    m->indices.resize(index_num);  // the caller's pointer may be an unaligned view into a file image: bytes only
    if (index_num) memcpy(m->indices.data(), index_buf, index_num * sizeof(uint16_t));
    m->run.resize(index_num);
    {
        // runs restart at every primitive's first index so a chunk never looks outside its primitive
        std::vector<uint8_t> is_first(index_num + 1, 0);
        for (auto& pr : m->prims) is_first[pr.index_ofs] = 1;
        uint32_t r = 0;
        for (size_t i = 0; i < index_num; i++) {
            if (is_first[i]) r = 0;
            r = m->indices[i] == 0xFFFF ? 0 : r + 1;
            m->run[i] = r;
        }
    }
    int32_t rc = set_device(d);
    if (rc) return rc;
    if ((rc = dev_alloc(d, &m->d_vbuf, vertex_len + 16))) return rc;
    if ((rc = dev_alloc(d, &m->d_ibuf, index_num + 2))) return rc;
    if ((rc = dev_alloc(d, &m->d_prims, nprims))) return rc;
    {
        std::vector<BoneBox> boxes, inst_boxes;
        build_bounds(m.get(), static_cast<const uint8_t*>(vertex_buf), boxes, inst_boxes);
        if ((rc = dev_alloc(d, &m->d_boxes, boxes.size()))) return rc;
        if ((rc = dev_alloc(d, &m->d_inst_boxes, inst_boxes.size()))) return rc;
        if (!boxes.empty()) HIPCHK(d, hipMemcpy(m->d_boxes, boxes.data(), boxes.size() * sizeof(BoneBox), hipMemcpyHostToDevice));
        if (!inst_boxes.empty()) HIPCHK(d, hipMemcpy(m->d_inst_boxes, inst_boxes.data(), inst_boxes.size() * sizeof(BoneBox), hipMemcpyHostToDevice));
    }
    HIPCHK(d, hipMemcpyAsync(m->d_vbuf, vertex_buf, vertex_len, hipMemcpyHostToDevice, d->stream));
    HIPCHK(d, hipMemcpyAsync(m->d_ibuf, index_buf, index_num * 2, hipMemcpyHostToDevice, d->stream));
    HIPCHK(d, hipMemcpyAsync(m->d_prims, m->prims.data(), nprims * sizeof(DPrim), hipMemcpyHostToDevice, d->stream));
    HIPCHK(d, hipStreamSynchronize(d->stream));
    *out = m.release();
    return MTR_OK;
}

void mtr_model_destroy(mtr_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->dev->hip_dev);
    (void)drain_all(m->dev);  // frames in flight (on any slot stream) may still read its buffers
    for (auto& pb : m->pal_ring) {
        if (pb.d) (void)hipFree(pb.d);
        if (pb.ready) (void)hipEventDestroy(pb.ready);
    }
    void* ptrs[] = {m->d_vbuf, m->d_ibuf, m->d_prims, m->d_boxes, m->d_inst_boxes};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete m;
}

int32_t mtr_model_set_prim_states(mtr_model* m, const mtr_prim_state* states, size_t nprims) {
    if (!m) return MTR_E_INVALID;
    mtr_device* d = m->dev;
    if (states && nprims != m->prims.size()) return fail(d, MTR_E_INVALID, "one state per primitive");
    for (size_t p = 0; states && p < nprims; p++)
        if (states[p].blend > MTR_BLEND_ADD || states[p].cull > MTR_CULL_FRONT) return fail(d, MTR_E_INVALID, "unknown blend / cull mode");
    int32_t rc = set_device(d);
    if (rc) return rc;
    std::lock_guard<std::mutex> submit_lock(d->submit_mu);
    if (states) m->states.assign(states, states + nprims); else m->states.clear();
    bool changed = false;
    for (size_t p = 0; p < m->prims.size(); p++) {
        const uint32_t cull = states ? states[p].cull : (uint32_t)MTR_CULL_BACK;
        changed = changed || m->prims[p].cull != cull;
        m->prims[p].cull = cull;
    }
    if (changed) {  // the cull mode lives in the device copy of the primitive table: frames in flight may be reading it
        if ((rc = drain_all(d))) return rc;
        HIPCHK(d, hipMemcpy(m->d_prims, m->prims.data(), m->prims.size() * sizeof(DPrim), hipMemcpyHostToDevice));
    }
    return MTR_OK;
}

int32_t mtr_model_set_parts_disp(mtr_model* m, const uint8_t* parts_disp, size_t n) {
    if (!m || (!parts_disp && n)) return MTR_E_INVALID;
    // the exchange thread may be re-running a frame that drew this model: the table a recorded draw holds is immutable,
    // and what the next draw will see changes under the lock
    std::lock_guard<std::mutex> submit_lock(m->dev->submit_mu);
    m->parts_disp.assign(parts_disp, parts_disp + n);
    m->chunks_dirty = true;
    return MTR_OK;
}

int32_t mtr_model_set_palette(mtr_model* m, const float* mats, size_t n) {
    if (!m) return MTR_E_INVALID;
    mtr_device* d = m->dev;
    if (n > 256 || (!mats && n)) return fail(d, MTR_E_INVALID, "palette: at most 256 matrices (u8 joint indices)");
    int32_t rc = set_device(d);
    if (rc) return rc;
    std::lock_guard<std::mutex> submit_lock(d->submit_mu);
    m->npal = (uint32_t)n;
    m->d_palette = nullptr;
    m->pal_ready = nullptr;
    m->pal_slot = -1;
    if (n) {
        if (m->pal_ring.size() < (size_t)d->max_inflight + 1)  // first use (the bound is fixed at device creation)
            m->pal_ring.resize((size_t)d->max_inflight + 1);
        size_t slot = m->pal_next++ % m->pal_ring.size();
        for (size_t tries = 0; m->pal_ring[slot].pinned && tries < m->pal_ring.size(); tries++) slot = m->pal_next++ % m->pal_ring.size();
        if (m->pal_ring[slot].pinned) {
            // every buffer is held by a live frame that may still (re-)run: the host keeps more un-waited frames alive than
            // the ring has buffers.  The ring grows by one (indices held by recorded draws stay valid).
            if (m->pal_ring.size() >= 4096) return fail(d, MTR_E_NOMEM, "more than 4096 live frames hold a palette of this model");
            m->pal_ring.emplace_back();
            slot = m->pal_ring.size() - 1;
        }
        mtr_model::PalBuf& pb = m->pal_ring[slot];
        // the last frame that read this buffer: finished for sure once max_inflight later frames have been submitted
        if (pb.used && d->frames_submitted < pb.last_frame + 1 + d->max_inflight && d->inflight[pb.last_frame % d->max_inflight])
            HIPCHK(d, hipEventSynchronize(d->inflight[pb.last_frame % d->max_inflight]));
        if (pb.cap < n) {  // grow: nothing in flight may still read the old buffer
            if ((rc = drain_all(d))) return rc;
            if (pb.d) (void)hipFree(pb.d);
            pb.d = nullptr; pb.cap = 0;
            if ((rc = dev_alloc(d, &pb.d, std::max<size_t>(n, 64) * 16))) return rc;
            pb.cap = (uint32_t)std::max<size_t>(n, 64);
        }
        if (!pb.ready) HIPCHK(d, hipEventCreateWithFlags(&pb.ready, hipEventDisableTiming));
        HIPCHK(d, hipMemcpyAsync(pb.d, mats, n * 64, hipMemcpyHostToDevice, d->s_copy));
        HIPCHK(d, hipEventRecord(pb.ready, d->s_copy));
        m->d_palette = pb.d;
        m->pal_ready = pb.ready;
        m->pal_slot = (int)slot;
    }
    return MTR_OK;
}

// ---------------------------------------------------------------------------------------------
// instance batches
// ---------------------------------------------------------------------------------------------
int32_t mtr_batch_create(mtr_device* d, mtr_model* model, size_t n, const float* model_mats, const float* palettes,
                         size_t npal, const int32_t* texture_override, mtr_batch** out) {
    if (!d || !out) return MTR_E_INVALID;
    *out = nullptr;
    if (!model || model->dev != d || !model_mats || n == 0 || n > 0xFFFFu)
        return fail(d, MTR_E_INVALID, "bad batch arguments");
    if (npal > 256 || (npal && !palettes)) return fail(d, MTR_E_INVALID, "palette: at most 256 matrices");
    auto b = std::make_unique<mtr_batch>();
    b->dev = d; b->model = model; b->n = (uint32_t)n; b->npal = palettes ? (uint32_t)npal : 0;
    if (texture_override) {
        b->tex_override.assign(texture_override, texture_override + n);
        for (int32_t t : b->tex_override)
            if (t >= (int32_t)model->textures.size()) return fail(d, MTR_E_INVALID, "texture_override out of range");
    }
    int32_t rc = set_device(d);
    if (rc) return rc;
    if ((rc = dev_alloc(d, &b->d_model_mats, n * 16))) return rc;
    // uploads go through the copy stream and an event: creating a batch does not wait for frames in flight
    HIPCHK(d, hipMemcpyAsync(b->d_model_mats, model_mats, n * 64, hipMemcpyHostToDevice, d->s_copy));
    if (b->npal) {
        if ((rc = dev_alloc(d, &b->d_palettes, n * npal * 16))) return rc;
        HIPCHK(d, hipMemcpyAsync(b->d_palettes, palettes, n * npal * 64, hipMemcpyHostToDevice, d->s_copy));
    }
    HIPCHK(d, hipEventCreateWithFlags(&b->ready, hipEventDisableTiming));
    HIPCHK(d, hipEventRecord(b->ready, d->s_copy));
    *out = b.release();
    return MTR_OK;
}

void mtr_batch_destroy(mtr_batch* b) {
    if (!b) return;
    mtr_device* d = b->dev;
    (void)hipSetDevice(d->hip_dev);
    // a frame that drew the batch may still be in flight: park the buffers until that frame has left the GPU
    // (collected at a later submit); otherwise free them now.  No stream is drained either way.  The exchange thread
    // destroys the batches its frames own while the render thread submits: submit_mu guards the list and the index.
    std::lock_guard<std::mutex> submit_lock(d->submit_mu);
    const bool busy = b->used && d->frames_submitted <= b->last_frame + d->max_inflight;
    if (b->hint_slot >= 0) { d->hint_used[b->hint_slot] = false; d->hint_host[2 * b->hint_slot] = d->hint_host[2 * b->hint_slot + 1] = 0u; }
    for (void* p : {(void*)b->d_model_mats, (void*)b->d_palettes})
        if (p) {
            if (busy) d->garbage.push_back({p, b->last_frame});
            else (void)hipFree(p);
        }
    if (b->ready) (void)hipEventDestroy(b->ready);
    delete b;
}

// ---------------------------------------------------------------------------------------------
// frame
// ---------------------------------------------------------------------------------------------
int32_t mtr_frame_begin(mtr_device* d, uint32_t w, uint32_t h, const float clear_rgba[4], float clear_depth,
                        mtr_frame** out) {
    if (!d || !out) return MTR_E_INVALID;
    *out = nullptr;
    if (w == 0 || h == 0 || w > 16384 || h > 16384 || !clear_rgba) return fail(d, MTR_E_INVALID, "bad frame size");
    int32_t rc = set_device(d);
    if (rc) return rc;
    {
        // an earlier frame that was released without a wait and turns out to have dropped triangles is reported here
        std::lock_guard<std::mutex> submit_lock(d->submit_mu);
        poll_released(d);
        if ((rc = report_sticky(d))) return rc;
    }
    auto f = std::make_unique<mtr_frame>();
    f->dev = d; f->w = w; f->h = h;
    f->clear_rgba8 = pack_rgba8(clear_rgba);
    f->clear_depth = clear_depth;
    // recycle colour / depth buffers: prefer a set whose last frame has finished; while fewer than nslots + 2 sets
    // of this size exist, allocate another rather than wait for one still in flight (a host that begins and
    // destroys a frame per step would otherwise chain every frame to its predecessor)
    bool found = false;
    size_t same = 0, oldest = SIZE_MAX, ready = SIZE_MAX;
    std::unique_lock<std::mutex> pool_lock(d->pool_mu);
    for (size_t i = 0; i < d->free_fb.size(); i++)
        if (d->free_fb[i].w == w && d->free_fb[i].h == h) {
            same++;
            if (oldest == SIZE_MAX) oldest = i;
            if (ready == SIZE_MAX && (!d->free_fb[i].used || hipEventQuery(d->free_fb[i].done) == hipSuccess)) ready = i;
        }
    (void)hipGetLastError();  // hipEventQuery reports "not ready" as an error code
    // total sets of this size, parked or held by live frames: past nslots + 1 the pool stops growing as long as a
    // parked set exists (its last frame is waited for on the device, not on the host)
    uint32_t* total = nullptr;
    for (auto& e : d->fb_allocated)
        if (e.first == (((uint64_t)w << 32) | h)) total = &e.second;
    if (!total) { d->fb_allocated.push_back({((uint64_t)w << 32) | h, 0u}); total = &d->fb_allocated.back().second; }
    const size_t pick = ready != SIZE_MAX ? ready : ((same > 0 && *total > d->nslots) ? oldest : SIZE_MAX);
    if (pick != SIZE_MAX) {
        f->fb = d->free_fb[pick];
        d->free_fb.erase(d->free_fb.begin() + (long)pick);
        found = true;
        // the last frame on this set zeroed the other counter block in its tile kernel: count there
        if (f->fb.next_zeroed) { f->fb.ctr_live ^= 1u; f->fb.next_zeroed = false; f->fb.ctr_dirty = false; }
    }
    if (!found) ++*total;
    pool_lock.unlock();
    if (!found) {
        f->fb.w = w; f->fb.h = h;
        if (!(rc = dev_alloc(d, &f->fb.color, (size_t)w * h * 4)) && !(rc = dev_alloc(d, &f->fb.depth, (size_t)w * h)) &&
            !(rc = dev_alloc(d, &f->fb.counters, (size_t)CTR_NUM * 2)) &&
            hipEventCreateWithFlags(&f->fb.done, hipEventDisableTiming) != hipSuccess)
            rc = fail(d, MTR_E_HIP, "hipEventCreate failed");
        if (rc) {  // give the partial set back
            if (f->fb.color) (void)hipFree(f->fb.color);
            if (f->fb.depth) (void)hipFree(f->fb.depth);
            if (f->fb.counters) (void)hipFree(f->fb.counters);
            std::lock_guard<std::mutex> g(d->pool_mu);
            for (auto& e : d->fb_allocated)
                if (e.first == (((uint64_t)w << 32) | h)) --e.second;
            return rc;
        }
    }
    *out = f.release();
    return MTR_OK;
}

static void release_palette_pins(mtr_frame* f);

void mtr_frame_destroy(mtr_frame* f) {
    if (!f) return;
    mtr_device* d = f->dev;
    (void)hipSetDevice(d->hip_dev);
    if (f->have_events)
        for (auto& e : f->ev)
            if (e) (void)hipEventDestroy(e);
    if (f->own) {
        std::lock_guard<std::mutex> g(d->submit_mu);
        const_cast<OwnTable*>(f->own)->refs--;
    }
    {
        std::lock_guard<std::mutex> g(d->submit_mu);
        release_palette_pins(f);  // drawn and never submitted, or submitted and never waited for
    }
    if (f->submitted && !f->flags_checked && f->status_idx >= 0) {
        // nobody looked at this frame's overflow flags: they are examined when its status word is polled or recycled,
        // and a frame that dropped triangles is then reported by the next call that can return an error
        std::lock_guard<std::mutex> g(d->submit_mu);
        if (d->status_owner[f->status_idx] == f->frame_index && !d->status_checked[f->status_idx]) d->status_released[f->status_idx] = true;
    }
    // the buffers may still be written by this frame's kernels: whoever recycles them waits on fb.done
    {
        std::lock_guard<std::mutex> g(d->pool_mu);
        d->free_fb.push_back(f->fb);
    }
    delete f;
}

int32_t mtr_frame_set_shard_map(mtr_frame* f, uint32_t rank, uint32_t world, uint32_t map, uint32_t param, const uint32_t* band_rows) {
    if (!f) return MTR_E_INVALID;
    mtr_device* d = f->dev;
    if (world == 0 || rank >= world) return fail(d, MTR_E_INVALID, "bad shard rank/world");
    if (f->submitted) return fail(d, MTR_E_INVALID, "frame already submitted");
    int32_t rc = set_device(d);
    if (rc) return rc;
    const OwnTable* t = nullptr;
    {
        std::lock_guard<std::mutex> submit_lock(d->submit_mu);
        if ((rc = get_own_table(d, f->w, f->h, world, map, param, band_rows, &t))) return rc;
        if (f->own) const_cast<OwnTable*>(f->own)->refs--;
        const_cast<OwnTable*>(t)->refs++;
    }
    f->shard_rank = rank; f->shard_world = world; f->own = t;
    return MTR_OK;
}

int32_t mtr_frame_set_shard(mtr_frame* f, uint32_t rank, uint32_t world) {
    return mtr_frame_set_shard_map(f, rank, world, MTR_OWN_INTERLEAVED, 0, nullptr);
}

int32_t mtr_device_set_texture_residency(mtr_device* d, uint32_t mode) {
    if (!d) return MTR_E_INVALID;
    if (mode != MTR_TEXRES_DECODED && mode != MTR_TEXRES_BLOCKS) return fail(d, MTR_E_INVALID, "unknown texture residency mode");
    d->texture_residency = mode;
    return MTR_OK;
}

int32_t mtr_device_set_culling(mtr_device* d, int32_t mode) {
    if (!d) return MTR_E_INVALID;
    if (mode < MTR_GEOM_CULL_OFF || mode > MTR_GEOM_CULL_ALL_FRAMES) return fail(d, MTR_E_INVALID, "unknown culling mode");
    std::lock_guard<std::mutex> g(d->submit_mu);
    d->cull_enabled = mode != MTR_GEOM_CULL_OFF;
    d->cull_unsharded = mode == MTR_GEOM_CULL_ALL_FRAMES;
    return MTR_OK;
}

// validates the draw and snapshots the model's chunk table (its visible primitives) as of now
static int32_t check_model_for_draw(mtr_frame* f, mtr_model* m, std::shared_ptr<const ChunkTable>* table) {
    mtr_device* d = f->dev;
    if (!m || m->dev != d) return fail(d, MTR_E_INVALID, "model belongs to another device");
    if (f->submitted) return fail(d, MTR_E_INVALID, "frame already submitted");
    int32_t rc = set_device(d);
    if (rc) return rc;
    std::lock_guard<std::mutex> submit_lock(d->submit_mu);
    for (size_t p = 0; p < m->prims.size(); p++)
        if (m->prims[p].parts_no >= m->parts_disp.size())  // self.parts_disp[parts_no] would panic, src/model.rs:318
            return fail(d, MTR_E_INVALID, "primitive " + std::to_string(p) + ": parts_no outside parts_disp");
    return current_table(m, table);
}

int32_t mtr_frame_draw_model(mtr_frame* f, mtr_model* m, const float view_proj[16]) {
    if (!f || !view_proj) return MTR_E_INVALID;
    std::shared_ptr<const ChunkTable> table;
    int32_t rc = check_model_for_draw(f, m, &table);
    if (rc) return rc;
    Draw dr{};
    dr.table = std::move(table);
    dr.model = m; dr.d_model_mats = nullptr; dr.d_palettes = m->d_palette; dr.npal = m->npal; dr.pal_ready = m->pal_ready; dr.pal_slot = m->pal_slot;
    dr.pal_stride = 0; dr.ninst = 1; dr.shader_override = -1; dr.blend = true;
    memcpy(dr.vp, view_proj, sizeof dr.vp);
    if (dr.pal_slot >= 0) {  // the ring buffer must not come round again before this frame has been submitted
        std::lock_guard<std::mutex> submit_lock(f->dev->submit_mu);
        m->pal_ring[(size_t)dr.pal_slot].pinned++;
        dr.pal_pinned = true;
    }
    f->draws.push_back(std::move(dr));
    return MTR_OK;
}

int32_t mtr_frame_draw_batch(mtr_frame* f, mtr_batch* b, const float view_proj[16]) {
    if (!f || !b || !view_proj) return MTR_E_INVALID;
    if (b->dev != f->dev) return fail(f->dev, MTR_E_INVALID, "batch belongs to another device");
    std::shared_ptr<const ChunkTable> table;
    int32_t rc = check_model_for_draw(f, b->model, &table);
    if (rc) return rc;
    Draw dr{};
    dr.table = std::move(table);
    dr.model = b->model; dr.d_model_mats = b->d_model_mats; dr.batch = b; dr.pal_ready = b->ready; dr.pal_slot = -1;
    dr.d_palettes = b->npal ? b->d_palettes : nullptr;
    dr.npal = b->npal; dr.pal_stride = b->npal * 16; dr.ninst = b->n;
    dr.tex_override = b->tex_override; dr.shader_override = -1; dr.blend = true;
    memcpy(dr.vp, view_proj, sizeof dr.vp);
    f->draws.push_back(std::move(dr));
    return MTR_OK;
}

int32_t mtr_frame_draw_instances(mtr_frame* f, mtr_model* m, const float* model_mats, const float* palettes,
                                 size_t npal, size_t n, const float view_proj[16]) {
    if (!f || !view_proj) return MTR_E_INVALID;
    mtr_batch* b = nullptr;
    int32_t rc = mtr_batch_create(f->dev, m, n, model_mats, palettes, npal, nullptr, &b);
    if (rc) return rc;
    rc = mtr_frame_draw_batch(f, b, view_proj);
    if (rc) { mtr_batch_destroy(b); return rc; }
    f->draws.back().owned_batch.reset(b);
    return MTR_OK;
}

int32_t mtr_model_set_joint_positions(mtr_model* m, const float* xyz, size_t n) {
    if (!m || (!xyz && n)) return MTR_E_INVALID;
    if (n > 0xFFFF) return fail(m->dev, MTR_E_INVALID, "too many joints");
    m->joint_cubes.assign(n * 16, 0.0f);
    for (size_t j = 0; j < n; j++) {
        float* M = &m->joint_cubes[j * 16];  // glam::Mat4::from_scale_rotation_translation(splat(0.005), IDENTITY, pos * 0.01)
        M[0] = M[5] = M[10] = 0.005f;
        M[12] = xyz[3 * j + 0] * 0.01f; M[13] = xyz[3 * j + 1] * 0.01f; M[14] = xyz[3 * j + 2] * 0.01f;
        M[15] = 1.0f;
    }
    return MTR_OK;
}

int32_t mtr_frame_draw_model_joints(mtr_frame* f, mtr_model* m, const float camera[16]) {
    if (!f || !m || !camera) return MTR_E_INVALID;
    if (m->dev != f->dev) return fail(f->dev, MTR_E_INVALID, "model belongs to another device");
    return mtr_frame_draw_overlay_cubes(f, camera, m->joint_cubes.data(), m->joint_cubes.size() / 16);
}

int32_t mtr_frame_draw_overlay_cubes(mtr_frame* f, const float camera[16], const float* inst_mats, size_t n) {
    if (!f || !camera || (!inst_mats && n)) return MTR_E_INVALID;
    mtr_device* d = f->dev;
    if (n == 0) return MTR_OK;
    if (!d->cube) {
        // src/debug_overlay.rs:10-35
        static const float verts[24] = {1, 1, -1, 1, -1, -1, 1, 1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1};
        static const uint16_t idx[36] = {4, 2, 0, 2, 7, 3, 6, 5, 7, 1, 7, 5, 0, 3, 1, 4, 1, 5,
                                         4, 6, 2, 2, 6, 7, 6, 4, 5, 1, 3, 7, 0, 2, 3, 4, 0, 1};
        mtr_primitive pr{};
        pr.w[0] = 8u << 16;
        pr.w[2] = (12u << 16) | (3u << 24);
        pr.w[7] = 36;
        mtr_layout lay{};
        lay.num_elements = 1;
        lay.elements[0].semantic = MTR_SEM_POSITION;
        lay.elements[0].format = MTR_IEF_F32;
        lay.elements[0].count = 3;
        int32_t rc = mtr_model_create(d, verts, sizeof verts, idx, 36, &pr, 1, &lay, nullptr, nullptr, 0, nullptr, &d->cube);
        if (rc) return rc;
    }
    mtr_batch* b = nullptr;
    int32_t rc = mtr_batch_create(d, d->cube, n, inst_mats, nullptr, 0, nullptr, &b);
    if (rc) return rc;
    rc = mtr_frame_draw_batch(f, b, camera);
    if (rc) { mtr_batch_destroy(b); return rc; }
    Draw& dr = f->draws.back();
    dr.owned_batch.reset(b);
    dr.shader_override = MTR_SH_CONST;
    const float c[4] = {0.1f, 0.2f, 0.3f, 1.0f};  // src/shaders/debug_overlay.wgsl:30
    dr.const_rgba8 = pack_rgba8(c);
    dr.blend = false;  // blend: None, src/debug_overlay.rs:174
    return MTR_OK;
}

// the frame can no longer (re-)run: its draws let go of the palette ring buffers they hold.  submit_mu held.
static void release_palette_pins(mtr_frame* f) {
    for (Draw& dr : f->draws)
        if (dr.pal_pinned && dr.pal_slot >= 0 && (size_t)dr.pal_slot < dr.model->pal_ring.size()) {
            dr.model->pal_ring[(size_t)dr.pal_slot].pinned--;
            dr.pal_pinned = false;
        }
}

// Enqueues every kernel of the frame.  The caller holds d->submit_mu.
static int32_t run_frame(mtr_frame* f) {
    mtr_device* d = f->dev;
    int32_t rc = set_device(d);
    if (rc) return rc;
    const uint32_t nbx = (f->w + MTR_BIN - 1) / MTR_BIN, nby = (f->h + MTR_BIN - 1) / MTR_BIN, nbins = nbx * nby;
    // ---- chunk tables, capacities ----
    uint64_t total_chunks = 0, nmats = 0, tris_in = 0;
    for (auto& dr : f->draws) {
        mtr_model* m = dr.model;
        total_chunks += (uint64_t)dr.table->chunks.size() * dr.ninst;
        nmats += (uint64_t)m->prims.size() * (dr.tex_override.empty() ? 1 : dr.ninst);
        tris_in += dr.table->ntris_visible * dr.ninst;
    }
    // a (triangle, bin) entry is the 32-bit submission order chunk * 128 + slot, and the visibility key stores order + 1:
    // fewer than 2^25 - 1 chunks (2 G triangles) per frame
    if (total_chunks >= (1ull << 25) - 1) return fail(d, MTR_E_OVERFLOW, "too many geometry chunks in one frame");
    // this frame's slot: the other slots may still be feeding earlier frames' tile kernels
    const uint64_t this_frame = d->frames_submitted++;
    hipEvent_t& ring = d->inflight[this_frame % d->max_inflight];
    if (ring) HIPCHK(d, hipEventSynchronize(ring));  // frame (i - max_inflight) has left the GPU
    else HIPCHK(d, hipEventCreateWithFlags(&ring, hipEventDisableTiming));
    // its status word is recycled for this frame: if nobody looked at that frame's overflow flags, do it now
    const int sidx = (int)(this_frame % d->max_inflight);
    examine_status(d, sidx, true);
    __atomic_store_n(&d->status_host[sidx], 0u, __ATOMIC_RELEASE);
    d->status_checked[sidx] = false; d->status_released[sidx] = false; d->status_owner[sidx] = this_frame;
    f->status_idx = sidx; f->frame_index = this_frame; f->flags_checked = false; f->stats_valid = false;
    // every frame up to this_frame - max_inflight has been waited for: collect what only they could still read
    if (!d->garbage.empty()) {
        size_t keep = 0;
        for (auto& g : d->garbage) {
            if (g.last_frame + d->max_inflight <= this_frame) (void)hipFree(g.p);
            else d->garbage[keep++] = g;
        }
        d->garbage.resize(keep);
    }
    f->slot = (int)(d->frame_counter++ % d->nslots);
    d->status_slot_of[sidx] = f->slot;
    Slot& sl = d->slots[f->slot];
    const uint64_t rec_need = total_chunks * MTR_CHUNK_SLOTS;
    if (rec_need > 0xFFFFFFF0ull) return fail(d, MTR_E_OVERFLOW, "too many triangles in one frame");
    if (rec_need > sl.rec_cap || !sl.rec_a) {
        HIPCHK(d, hipStreamSynchronize(sl.stream));
        uint32_t c0 = sl.rec_cap, c1 = sl.rec_cap, c2 = sl.rec_cap;
        if ((rc = dev_grow(d, &sl.rec_hdr, &c0, rec_need))) return rc;
        if ((rc = dev_grow(d, &sl.rec_a, &c1, rec_need))) return rc;
        { uint32_t c3 = sl.rec_cap; if ((rc = dev_grow(d, &sl.rec_l, &c3, rec_need))) return rc; }
        if ((rc = dev_grow(d, &sl.rec_b, &c2, rec_need))) return rc;
        sl.rec_cap = c0;
    }
    if (total_chunks > sl.chunk_cap || !sl.chunk_info) {
        HIPCHK(d, hipStreamSynchronize(sl.stream));
        if ((rc = dev_grow(d, &sl.chunk_info, &sl.chunk_cap, total_chunks))) return rc;
    }
    if (nbins + 1 > sl.bin_cap || !sl.bin_count) {
        HIPCHK(d, hipStreamSynchronize(sl.stream));
        HIPCHK(d, hipStreamSynchronize(d->stream));  // mtr_frame_read_bin_counts copies from them on the public stream
        uint32_t c0 = sl.bin_cap, c1 = sl.bin_cap, c2 = sl.bin_cap, c3 = sl.bin_cap;
        if ((rc = dev_grow(d, &sl.bin_count, &c0, nbins + 1))) return rc;
        if ((rc = dev_grow(d, &sl.bin_fill, &c1, nbins + 1))) return rc;
        if ((rc = dev_grow(d, &sl.bin_start, &c2, nbins + 1))) return rc;
        if ((rc = dev_grow(d, &sl.seg_start, &c3, nbins + 1))) return rc;
        uint32_t c4 = sl.bin_cap;
        if ((rc = dev_grow(d, &sl.bin_flag, &c4, nbins + 1))) return rc;
        sl.bin_cap = c0;
        sl.bin_fill_dirty = true;
    }
    {
        // direct mode: nbins bounded queues; the bound shrinks if the bin grid is so large that the queues would not
        // be addressable with 32 bits
        while ((uint64_t)nbins * d->qcap > 0xF0000000ull && d->qcap > 64) d->qcap /= 2;
        f->ran_direct = d->direct_enabled && !f->force_two_pass;
        uint64_t e_need = std::max<uint64_t>(1u << 20, rec_need / 2) * d->queue_scale, s_need = std::max<uint64_t>(1u << 18, total_chunks * 8) * d->queue_scale;
        e_need = std::max<uint64_t>(e_need, f->min_entries);
        s_need = std::max<uint64_t>(s_need, f->min_segs);
        if (f->ran_direct) {
            e_need = std::max<uint64_t>(e_need, (uint64_t)nbins * d->qcap);
            s_need = std::max<uint64_t>(s_need, (uint64_t)nbins * d->scap);
        }
        if (e_need > sl.entry_cap || !sl.entries) {
            HIPCHK(d, hipStreamSynchronize(sl.stream));
            if ((rc = dev_grow(d, &sl.entries, &sl.entry_cap, std::min<uint64_t>(e_need, 0xFFFFFFF0ull)))) return rc;
        }
        if (s_need > sl.seg_cap || !sl.segs) {
            HIPCHK(d, hipStreamSynchronize(sl.stream));
            if ((rc = dev_grow(d, &sl.segs, &sl.seg_cap, std::min<uint64_t>(s_need, 0xFFFFFFF0ull)))) return rc;
        }
    }
    // ---- material table ----
    std::vector<DMat>& mats = f->mats_host;
    mats.clear();
    f->all_opaque = true;
    mats.reserve(nmats);
    std::vector<uint32_t> mat_base(f->draws.size()), mat_stride(f->draws.size());
    for (size_t di = 0; di < f->draws.size(); di++) {
        Draw& dr = f->draws[di];
        mtr_model* m = dr.model;
        mat_base[di] = (uint32_t)mats.size();
        mat_stride[di] = dr.tex_override.empty() ? 0 : (uint32_t)m->prims.size();
        const uint32_t reps = dr.tex_override.empty() ? 1 : dr.ninst;
        for (uint32_t r = 0; r < reps; r++)
            for (size_t p = 0; p < m->prims.size(); p++) {
                DMat dm{};
                int32_t tex = m->prim_to_texture[p];
                if (tex >= 0 && !dr.tex_override.empty() && dr.tex_override[r] >= 0) tex = dr.tex_override[r];
                dm.blend = dr.blend ? MTR_DB_ALPHA : MTR_DB_OFF;
                dm.dstate = 3u;  // depth write | depth test << 1
                dm.tlevels = 1;
                if (!m->states.empty() && dr.shader_override != MTR_SH_CONST) {  // material state (row f-4)
                    const mtr_prim_state& st = m->states[p];
                    dm.blend = st.blend == MTR_BLEND_OFF ? MTR_DB_OFF : (st.blend == MTR_BLEND_ADD ? MTR_DB_ADD : MTR_DB_ALPHA);
                    dm.dstate = (st.depth_write ? 1u : 0u) | (st.depth_test ? 2u : 0u);
                }
                // order-dependent: an additive blend, or a depth state in which a fragment's fate depends on what came before
                bool order_dep = dm.blend == MTR_DB_ADD || dm.dstate != 3u;
                if (dr.shader_override == MTR_SH_CONST) {
                    dm.shader = MTR_SH_CONST; dm.rgba8 = dr.const_rgba8;
                } else if (tex >= 0 && m->prims[p].has_uv) {  // src/model.rs:212-216
                    dm.shader = MTR_SH_TEXTURED;
                    const mtr_texture* t = m->textures[(size_t)tex];
                    if (!t->opaque && dm.blend == MTR_DB_ALPHA) order_dep = true;  // a texel with alpha < 255 really blends
                    dm.tex = t->d_rgba; dm.tw = t->w; dm.th = t->h; dm.tlevels = t->levels | (t->resident << 8);
                } else {
                    dm.shader = MTR_SH_DEBUG; dm.rgba8 = m->debug_rgba8[p];
                }
                if (order_dep) { dm.translucent = 1; f->all_opaque = false; }
                mats.push_back(dm);
            }
    }
    if (mats.size() >= MTR_MAX_TEXTURED_MATERIALS) return fail(d, MTR_E_OVERFLOW, "too many materials in one frame (a record holds a 24-bit material id)");
    if (mats.size() > sl.mat_cap || !sl.mats) {
        HIPCHK(d, hipStreamSynchronize(sl.stream));
        if ((rc = dev_grow(d, &sl.mats, &sl.mat_cap, std::max<size_t>(mats.size(), 64)))) return rc;
        sl.mats_uploaded.clear();
    }
    // the material table is tiny; the copy is ordered on the stream before the kernels that read it
    if (mats.size() != sl.mats_uploaded.size() ||
        (!mats.empty() && memcmp(mats.data(), sl.mats_uploaded.data(), mats.size() * sizeof(DMat)) != 0)) {
        HIPCHK(d, hipMemcpyAsync(sl.mats, mats.data(), mats.size() * sizeof(DMat), hipMemcpyHostToDevice, sl.stream));
        sl.mats_uploaded = mats;
    }

    FrameBuffers fb{};
    fb.rec_hdr = sl.rec_hdr; fb.rec_a = sl.rec_a; fb.rec_l = sl.rec_l; fb.rec_b = sl.rec_b; fb.chunk_info = sl.chunk_info;
    fb.bin_count = sl.bin_count; fb.bin_fill = sl.bin_fill; fb.bin_start = sl.bin_start; fb.seg_start = sl.seg_start;
    fb.entries = sl.entries; fb.segs = sl.segs; fb.counters = f->fb.live();
    fb.rec_cap = sl.rec_cap; fb.entry_cap = sl.entry_cap; fb.seg_cap = sl.seg_cap;
    fb.W = f->w; fb.H = f->h; fb.nbx = nbx; fb.nby = nby;
    fb.own.map = MTR_OWN_INTERLEAVED; fb.own.rank = 0; fb.own.world = 1; fb.own.own_count = nbins; fb.own.own_list = nullptr;
    if (f->own && f->shard_world > 1) {
        const OwnTable& t = *f->own;
        fb.own.map = t.map; fb.own.rank = f->shard_rank; fb.own.world = f->shard_world;
        if (t.map == MTR_OWN_BANDS) { fb.own.y0 = t.bands[f->shard_rank]; fb.own.y1 = t.bands[f->shard_rank + 1]; }
        fb.own.st_shift = t.st_shift; fb.own.nsx = t.nsx;
        fb.own.own_count = t.offs[f->shard_rank + 1] - t.offs[f->shard_rank];
        fb.own.own_list = t.d_lists + t.offs[f->shard_rank];
        // interleaved bins: a chunk's rectangle holds a bin of every rank as soon as it is `world` bins wide, so there
        // is next to nothing to cull (and the work list would only cost): culling is for bands and super-tiles
        fb.own.cull = (d->cull_enabled && t.map != MTR_OWN_INTERLEAVED) ? 1u : 0u;
        if (fb.own.cull && d->cull_debug != 0xFFFFFFFFu) fb.own.cull = d->cull_debug;  // timing ablations only (MTR_CULL_DEBUG at device creation)
    }
    else if (d->cull_unsharded && d->cull_enabled) {
        fb.own.cull = 1u;  // world 1: "a bin of this rank" = a bin of the target, so what is culled is what is off the target
    }
    fb.direct = f->ran_direct ? 1u : 0u; fb.qcap = d->qcap; fb.scap = d->scap;
    // every material opaque (debug / overlay colours have a == 1; opaque textures sample a == 1): the frame is a
    // per-pixel (min z, latest) reduction and the visibility-key kernel applies; otherwise blend order matters
    const bool use_vis = f->all_opaque && d->tile_mode != MTR_TILE_ORDERED;
    fb.unordered = (f->ran_direct && use_vis) ? 1u : 0u;

    if (d->profiling && !f->have_events) {
        for (auto& e : f->ev) HIPCHK(d, hipEventCreate(&e));
        f->have_events = true;
    }
    const bool prof = d->profiling && f->have_events;
    // one stream per slot: the slot's previous frame is ordered before this one by the stream itself
    hipStream_t sg = sl.stream, st = sl.stream;
    // recycled colour / depth / counter buffers: their last frame may have run on another slot's stream
    if (f->fb.used) HIPCHK(d, hipStreamWaitEvent(sg, f->fb.done, 0));
    if (f->fb.ctr_dirty) HIPCHK(d, hipMemsetAsync(f->fb.live(), 0, CTR_NUM * sizeof(uint32_t), sg));
    f->fb.ctr_dirty = true;  // a second run of this frame (queue overflow) starts from a fill again
    if (!fb.direct) {
        HIPCHK(d, hipMemsetAsync(sl.bin_count, 0, (size_t)(nbins + 1) * sizeof(unsigned long long), sg));
        sl.bin_fill_dirty = true;
    } else if (sl.bin_fill_dirty) {
        HIPCHK(d, hipMemsetAsync(sl.bin_fill, 0, (size_t)sl.bin_cap * sizeof(unsigned long long), sg));
        sl.bin_fill_dirty = false;
    }
    // sharded draws: k_cull_instances compacts the instance list of a batch draw to the instances that may reach this
    // rank's bins, k_cull_chunks then bounds every chunk of the surviving instances and writes the work list of k_geom
    const size_t ndraws = f->draws.size();
    std::vector<uint32_t> inst_off(ndraws, 0xFFFFFFFFu), work_off(ndraws, 0u), comp_off(ndraws, 0u);
    uint32_t strad_base = 0;
    if (fb.own.cull) {
        uint64_t ninst_total = 0, work_total = 0, comp_total = 0;
        for (size_t di = 0; di < ndraws; di++) {
            const Draw& dr = f->draws[di];
            const mtr_model* m = dr.model;
            const bool sk = dr.d_palettes && dr.npal;
            // one 16-bit mask per (instance slot, group of 16 chunks)
            const uint64_t nx = ((uint64_t)dr.table->chunks.size() + 15) / 16;
            // one-dimensional launch of nx * 4 * slots workgroups of 256 threads: HIP rejects 2^32 threads or more per dimension
            if (nx * 4 * dr.ninst > 0xFFFFFFull) return fail(d, MTR_E_OVERFLOW, "too many geometry chunks in one sharded draw");
            work_off[di] = (uint32_t)work_total;
            work_total += (nx * dr.ninst + 1) & ~1ull;  // even: k_geom reads a mask through the aligned dword that holds it
            if (!dr.d_model_mats) continue;  // a single model: chunk culling only
            if (sk ? (!m->inst_skinned_boundable || m->n_inst_skinned == 0) : (m->n_inst_unskinned == 0)) continue;
            inst_off[di] = (uint32_t)ninst_total;
            ninst_total += dr.ninst;
            comp_off[di] = (uint32_t)comp_total;
            comp_total += (uint64_t)dr.ninst * (sk ? dr.npal + 1u : 1u);
        }
        if (work_total > 0xFFFFFFF0ull || comp_total > 0xFFFFFFF0ull) return fail(d, MTR_E_OVERFLOW, "too many geometry chunks in one sharded frame");
        if (comp_total > sl.comp_cap || !sl.comp) {
            HIPCHK(d, hipStreamSynchronize(sl.stream));
            if ((rc = dev_grow(d, &sl.comp, &sl.comp_cap, std::max<uint64_t>(comp_total, 64)))) return rc;
        }
        if (work_total > sl.work_cap || !sl.work_mask) {
            HIPCHK(d, hipStreamSynchronize(sl.stream));
            if ((rc = dev_grow(d, &sl.work_mask, &sl.work_cap, std::max<uint64_t>(work_total, 64)))) return rc;
        }
        strad_base = (uint32_t)ninst_total;  // the second half of inst_list: the slots of the instances that straddle the rank's border
        if (2 * ninst_total > sl.inst_cap || !sl.inst_list) {
            HIPCHK(d, hipStreamSynchronize(sl.stream));
            if ((rc = dev_grow(d, &sl.inst_list, &sl.inst_cap, std::max<uint64_t>(2 * ninst_total, 64)))) return rc;
        }
        if (ndraws > sl.draw_cap || !sl.inst_count) {
            HIPCHK(d, hipStreamSynchronize(sl.stream));
            uint32_t words = sl.draw_cap * MTR_CULL_CTR_WORDS;
            if ((rc = dev_grow(d, &sl.inst_count, &words, (size_t)MTR_CULL_CTR_WORDS * std::max<size_t>(ndraws, 4)))) return rc;
            sl.draw_cap = words / MTR_CULL_CTR_WORDS;
            sl.cull_counts_dirty = true;
        }
        // the counters start from zero: the tile kernel of the slot's previous frame cleared them (TileParams::zero_words)
        if (sl.cull_counts_dirty || ndraws > sl.ctr_clean_draws)
            HIPCHK(d, hipMemsetAsync(sl.inst_count, 0, (size_t)sl.draw_cap * MTR_CULL_CTR_WORDS * sizeof(uint32_t), sg));
        sl.cull_counts_dirty = true;  // until a tile kernel that clears them has been queued (below)
        // the exact two-pass fill walks every chunk's run descriptor: culled chunks write none
        if (!fb.direct) HIPCHK(d, hipMemsetAsync(sl.chunk_info, 0, total_chunks * sizeof(ChunkInfo), sg));
    }
    if (prof) HIPCHK(d, hipEventRecord(f->ev[0], sg));
    uint32_t nhint = 0;  // batch draws whose culling counters this frame's tile kernel reports to the host (launch sizing)
    uint16_t hint_word[4] = {}, hint_slot[4] = {};
    uint32_t chunk_base = 0;
    for (size_t di = 0; di < f->draws.size(); di++) {
        Draw& dr = f->draws[di];
        mtr_model* m = dr.model;
        GeomParams gp{};
        gp.vbuf = m->d_vbuf; gp.ibuf = m->d_ibuf; gp.prims = m->d_prims; gp.chunks = dr.table->d_chunks;
        gp.boxes = m->d_boxes;
        gp.nchunks = (uint32_t)dr.table->chunks.size(); gp.ninst = dr.ninst;
        if (dr.batch) { dr.batch->last_frame = this_frame; dr.batch->used = true; }
        if (dr.pal_ready) {  // uploads of a model palette (ring) or of a batch, made on the copy stream
            HIPCHK(d, hipStreamWaitEvent(sg, dr.pal_ready, 0));
            if (dr.pal_slot >= 0 && (size_t)dr.pal_slot < m->pal_ring.size()) {
                mtr_model::PalBuf& pb = m->pal_ring[(size_t)dr.pal_slot];
                pb.last_frame = this_frame;  // protects the buffer while this run's kernels are in flight
                pb.used = true;
                // the pin stays: until the frame's overflow flags have been looked at it may be re-run (settle_frame, by
                // mtr_frame_wait or the exchange thread, several submits later) and must then skin with the SAME palette
            }
        }
        gp.model_mats = dr.d_model_mats; gp.palettes = dr.d_palettes; gp.npal = dr.d_palettes ? dr.npal : 0;
        gp.pal_stride = dr.pal_stride;
        memcpy(gp.vp, dr.vp, sizeof gp.vp);
        gp.chunk_base = chunk_base; gp.mat_base = mat_base[di]; gp.mat_inst_stride = mat_stride[di];
        gp.fb = fb;
        gp.mats = sl.mats;
        if (fb.own.cull) {
            const bool sk = dr.d_palettes && dr.npal;
            uint32_t* inst_cnt = nullptr;
            if (inst_off[di] != 0xFFFFFFFFu) {
                CullParams cp{};
                cp.boxes = m->d_inst_boxes + (sk ? m->n_inst_unskinned : 0); cp.nboxes = sk ? m->n_inst_skinned : m->n_inst_unskinned;
                cp.ninst = dr.ninst; cp.model_mats = dr.d_model_mats; cp.palettes = gp.palettes; cp.npal = gp.npal; cp.pal_stride = gp.pal_stride;
                memcpy(cp.vp, dr.vp, sizeof cp.vp);
                cp.W = f->w; cp.H = f->h; cp.nbx = nbx; cp.nby = nby; cp.own = fb.own;
                cp.list = sl.inst_list + inst_off[di]; cp.count = inst_cnt = sl.inst_count + di * MTR_CULL_CTR_WORDS;
                cp.comp = sl.comp + comp_off[di]; cp.ncomp = sk ? dr.npal + 1u : 1u;
                cp.work_mask = sl.work_mask + work_off[di]; cp.strad = sl.inst_list + strad_base + inst_off[di]; cp.nchunks = gp.nchunks;
                cp.counters = fb.counters;
                mtr_launch_cull_instances(cp, sg);
                HIPCHK(d, hipGetLastError());  // a rejected launch must not pass as an empty frame (a later successful call clears the error)
            }
            ChunkCullParams cc{};
            cc.chunks = dr.table->d_chunks; cc.boxes = m->d_boxes; cc.nchunks = gp.nchunks; cc.ninst = dr.ninst;
            cc.inst_list = inst_cnt ? sl.inst_list + inst_off[di] : nullptr; cc.inst_count = inst_cnt;
            cc.strad = inst_cnt ? sl.inst_list + strad_base + inst_off[di] : nullptr;
            cc.model_mats = dr.d_model_mats; cc.palettes = gp.palettes; cc.npal = gp.npal; cc.pal_stride = gp.pal_stride;
            memcpy(cc.vp, dr.vp, sizeof cc.vp);
            cc.fb = fb;
            cc.comp = inst_cnt ? sl.comp + comp_off[di] : nullptr;
            cc.work_mask = sl.work_mask + work_off[di];
            cc.keep_all = (fb.own.cull == 3u || fb.own.cull == 4u) ? 1u : 0u;
            if (inst_cnt && dr.batch && !dr.owned_batch) {
                // launch sizes from what a recent frame of this batch kept under the same ownership (k_geom.hip); a camera that
                // moves changes the count gradually: the margin and the second geometry launch take what the hint misses
                mtr_batch* b = dr.batch;
                if (b->hint_slot < 0)
                    for (uint32_t i = 0; i < mtr_device::kHintSlots; i++)
                        if (!d->hint_used[i]) { d->hint_used[i] = true; b->hint_slot = (int)i; b->hint_key = 0; break; }
                if (b->hint_slot >= 0) {
                    uint64_t key = 0xcbf29ce484222325ull;  // FNV-1a over what the kept set depends on
                    auto mix = [&](const void* p, size_t n) { for (size_t i = 0; i < n; i++) key = (key ^ static_cast<const uint8_t*>(p)[i]) * 0x100000001b3ull; };
                    const void* own_id = f->own;
                    mix(&own_id, sizeof own_id); mix(&f->shard_rank, sizeof f->shard_rank); mix(&fb.own.cull, sizeof fb.own.cull);
                    volatile uint32_t* hw = d->hint_host + 2 * b->hint_slot;
                    if (key != b->hint_key) { b->hint_key = key; hw[0] = hw[1] = 0u; }  // other bands, another rank: start over
                    gp.slots_hint = hw[0]; cc.strad_hint = hw[1];
                    if (nhint < 4 && di * MTR_CULL_CTR_WORDS < 0xFFFFu) { hint_word[nhint] = (uint16_t)(di * MTR_CULL_CTR_WORDS); hint_slot[nhint++] = (uint16_t)b->hint_slot; }
                }
            }
            mtr_launch_cull_chunks(cc, sg);
            HIPCHK(d, hipGetLastError());
            gp.work_mask = cc.work_mask; gp.work_nx = (gp.nchunks + 15u) / 16u;
            gp.inst_list = cc.inst_list; gp.inst_count = cc.inst_count;
        }
        gp.slots_override = d->geom_slots;
        // a draw of fewer than ~64 k geometry waves (the headline model: 16 k) overlaps with the neighbouring frames' tile
        // kernels for most of its life: the build that leaves them a wave slot per SIMD (k_geom.hip: GEOM_OCC_SMALL)
        gp.small_draw = ((uint64_t)gp.nchunks * dr.ninst < 65536u) ? 1u : 0u;
        mtr_launch_geom(gp, sg);
        HIPCHK(d, hipGetLastError());
        chunk_base += gp.nchunks * dr.ninst;
    }
    if (prof) HIPCHK(d, hipEventRecord(f->ev[1], sg));
    // single-pass binning launches neither kernel: no event either (an event costs the stream ~5 us, which would count
    // as frame latency); mtr_frame_wait reports both stages as 0
    if (!fb.direct) { mtr_launch_scan(fb, sg); HIPCHK(d, hipGetLastError()); }
    if (prof && !fb.direct) HIPCHK(d, hipEventRecord(f->ev[2], sg));
    if (!fb.direct) { mtr_launch_fill(fb, (uint32_t)total_chunks, sg); HIPCHK(d, hipGetLastError()); }
    if (prof && !fb.direct) HIPCHK(d, hipEventRecord(f->ev[3], st));
    TileParams tp{};
    tp.fb = fb; tp.mats = sl.mats; tp.color = f->fb.color; tp.depth = f->fb.depth;
    tp.clear_rgba8 = f->clear_rgba8; tp.clear_depth = f->clear_depth;
    bool any_textured = false;
    for (const DMat& dm : mats) any_textured = any_textured || dm.shader == MTR_SH_TEXTURED;
    // some material translucent, some not: the visibility kernel takes the bins whose queue holds only opaque
    // triangles (order-free), flags the others, and the ordered kernel renders those in submission order
    // ... and also the bins whose translucent triangles are merely alpha-blended in the default depth state (prefix minima
    // of z, k_tile_vis.hip: STAIR).  Only when every material is HARD order-dependent (additive blend, depth write / test
    // off) is there nothing for it to do.
    bool any_soft = false;
    for (const DMat& dm : mats) any_soft = any_soft || !dm.translucent || !(dm.blend == MTR_DB_ADD || dm.dstate != 3u);
    const bool mixed = !use_vis && d->tile_mode == MTR_TILE_AUTO && any_soft;
    tp.bin_flag = sl.bin_flag; tp.mixed = mixed ? 1u : 0u;
    tp.zero_next = f->fb.other();
    f->fb.next_zeroed = fb.own.own_count != 0;  // a rank without a bin launches no tile workgroup
    tp.host_status = d->status_dev + sidx;
    tp.vis_waves = d->vis_waves;
    {
        // the previous frame still on the GPU: this one will share it, balance across the XCDs wins; otherwise the frame
        // has the GPU to itself and the contiguous order's locality gives the shorter kernel (latency)
        bool shared = false;
        if (this_frame > 0) {
            hipEvent_t prev = d->inflight[(this_frame - 1) % d->max_inflight];
            shared = prev && hipEventQuery(prev) == hipErrorNotReady;
        }
        tp.xcd_run = d->xcd_run != mtr_device::kXcdRunAuto ? d->xcd_run : ((fb.own.world <= 1 && shared) ? std::max(16u, nbx / 4u) : 0u);
        tp.quad_walk = shared ? 0u : 1u;  // k_tile_vis.hip: shorter alone (latency), slightly slower among other frames' kernels
    }
    if (fb.own.cull && fb.own.own_count) {  // this frame's tile kernel clears the slot's culling counters for the next one
        tp.zero_words = sl.inst_count; tp.zero_nwords = (uint32_t)ndraws * MTR_CULL_CTR_WORDS;
        tp.hint_out = d->hint_dev; tp.nhint = nhint;
        for (uint32_t k = 0; k < nhint; k++) { tp.hint_word[k] = hint_word[k]; tp.hint_slot[k] = hint_slot[k]; }
        sl.cull_counts_dirty = false;
        sl.ctr_clean_draws = (uint32_t)ndraws;  // what a frame with more draws than this one finds beyond is stale
    }
    f->stats.tile_kernel = use_vis ? MTR_TILE_VISIBILITY : (mixed ? MTR_TILE_MIXED : MTR_TILE_ORDERED);
    if (use_vis || mixed) { mtr_launch_tile_vis(tp, any_textured, st); HIPCHK(d, hipGetLastError()); }
    if (mixed) tp.nhint = 0;  // the visibility kernel of a mixed frame has reported (and cleared) the counters: the second kernel would report zeros
    if (!use_vis) { mtr_launch_tile(tp, any_textured, st); HIPCHK(d, hipGetLastError()); }
    // a rank without a bin launches no tile workgroup: nobody else would publish the (clean) status
    if (fb.own.own_count == 0) __atomic_store_n(&d->status_host[sidx], 0x80000000u, __ATOMIC_RELEASE);
    if (prof) HIPCHK(d, hipEventRecord(f->ev[4], st));
    HIPCHK(d, hipEventRecord(f->fb.done, st));
    HIPCHK(d, hipEventRecord(ring, st));
    f->fb.used = true;
    // the device's public stream (read-backs, shard packing, the caller's own work) sees the framebuffer complete
    if (!f->for_exchange) HIPCHK(d, hipStreamWaitEvent(d->stream, f->fb.done, 0));
    HIPCHK(d, hipGetLastError());
    {
        const uint32_t tk = f->stats.tile_kernel;
        f->stats = mtr_frame_stats{};
        f->stats.tile_kernel = tk;
        f->stats.binning = f->ran_direct ? 1u : 2u;
    }
    f->stats.tris_in = tris_in;
    f->total_chunks = total_chunks;
    f->stats.width = f->w; f->stats.height = f->h; f->stats.nbins = nbins; f->stats.ndraws = (uint32_t)f->draws.size();
    return MTR_OK;
}

int32_t mtr_frame_submit(mtr_frame* f) {
    if (!f) return MTR_E_INVALID;
    mtr_device* d = f->dev;
    if (f->submitted) return fail(d, MTR_E_INVALID, "frame already submitted");
    std::lock_guard<std::mutex> submit_lock(d->submit_mu);
    poll_released(d);
    int32_t rc = report_sticky(d);  // an earlier frame that nobody waited for dropped triangles
    if (rc) return rc;
    rc = run_frame(f);
    if (rc) return rc;
    f->submitted = true;
    return MTR_OK;
}

// Makes sure the frame's kernels ran with complete bin queues: reads the overflow flags its tile kernel published and,
// when one is set, re-runs the frame (bounded per-bin queue full: through the exact two-pass queues, and later frames get
// twice the bound; two-pass queues too small: grown to what the scan measured).  wait_done: block until the frame has
// left the GPU first (mtr_frame_wait); otherwise return as soon as the flags are known to be clean, which the tile
// kernel announces when it STARTS -- the exchange thread can then queue the pack behind the frame without a host-side
// wait for its completion.  Called by the render thread and by the exchange thread (run_frame under submit_mu).
static int32_t settle_frame(mtr_frame* f, bool wait_done) {
    mtr_device* d = f->dev;
    for (int attempt = 0; attempt < 6; attempt++) {
        uint32_t v = 0;
        if (!wait_done)
            for (int spin = 0; spin < 100000 && !((v = status_load(d, f->status_idx)) & 0x80000000u); spin++) __builtin_ia32_pause();
        if (!(v & 0x80000000u)) {
            HIPCHK(d, hipEventSynchronize(f->fb.done));
            v = status_load(d, f->status_idx);  // still 0: no tile workgroup ran (a rank that owns no bin), nothing to check
        }
        const uint32_t flags = v & 0x7fffffffu;
        {
            std::lock_guard<std::mutex> g(d->submit_mu);
            if (d->status_owner[f->status_idx] == f->frame_index) d->status_checked[f->status_idx] = true;
            if (!flags) release_palette_pins(f);  // this frame will not run again
        }
        f->flags_checked = true;
        if (!flags) return MTR_OK;
        if (flags & 1u) return fail(d, MTR_E_OVERFLOW, "a geometry chunk (62 strip positions) needed more than 124 records: guard-band clipping fanned too many of its triangles");
        if (!(flags & 4u)) {
            // exact queues too small: grow to what the scan measured
            uint32_t two[2] = {0, 0};
            HIPCHK(d, hipEventSynchronize(f->fb.done));
            HIPCHK(d, hipMemcpyAsync(two, f->fb.live() + CTR_ENTRIES, sizeof two, hipMemcpyDeviceToHost, d->s_copy));
            HIPCHK(d, hipStreamSynchronize(d->s_copy));
            const uint64_t e_need = (uint64_t)two[0] + two[0] / 4 + 1024, s_need = (uint64_t)two[1] + two[1] / 4 + 1024;
            if (e_need > 0xFFFFFFF0ull || s_need > 0xFFFFFFF0ull) return fail(d, MTR_E_OVERFLOW, "bin queues exceed 2^32 entries");
            f->min_entries = e_need;  // run_frame grows the queues of the slot it picks
            f->min_segs = s_need;
        }
        std::lock_guard<std::mutex> g(d->submit_mu);
        if (flags & 4u) {
            // a bounded per-bin queue filled up: this frame takes the exact two-pass path, later frames get twice the bound
            f->force_two_pass = true;
            grow_direct_queues(d);
        }
        const int32_t rc = run_frame(f);
        if (rc) return rc;
    }
    return fail(d, MTR_E_OVERFLOW, "bin queues still overflow after growing");
}

int32_t mtr_frame_wait(mtr_frame* f) {
    if (!f) return MTR_E_INVALID;
    mtr_device* d = f->dev;
    if (!f->submitted) return fail(d, MTR_E_INVALID, "frame not submitted");
    if (f->waited) return MTR_OK;
    int32_t rc = set_device(d);
    if (rc) return rc;
    // wait for THIS frame only (the public stream also carries the completion of every later frame)
    if ((rc = settle_frame(f, true))) return rc;
    HIPCHK(d, hipEventSynchronize(f->fb.done));  // of the last run
    if (d->profiling && f->have_events) {
        if (f->ran_direct) {
            HIPCHK(d, hipEventElapsedTime(&f->ms[MTR_STAGE_GEOM], f->ev[0], f->ev[1]));
            f->ms[MTR_STAGE_SCAN] = f->ms[MTR_STAGE_FILL] = 0.0f;
            HIPCHK(d, hipEventElapsedTime(&f->ms[MTR_STAGE_TILE], f->ev[1], f->ev[4]));
        } else {
            for (int s = 0; s < MTR_STAGE_COUNT; s++) HIPCHK(d, hipEventElapsedTime(&f->ms[s], f->ev[s], f->ev[s + 1]));
        }
    }
    f->waited = true;
    return MTR_OK;
}

// device counters of the finished frame -> f->stats (read back on demand: mtr_frame_wait itself copies nothing)
static int32_t fetch_stats(mtr_frame* f) {
    mtr_device* d = f->dev;
    int32_t rc = mtr_frame_wait(f);
    if (rc) return rc;
    if (f->stats_valid) return MTR_OK;
    uint32_t ctr[CTR_NUM];
    HIPCHK(d, hipMemcpyAsync(ctr, f->fb.live(), sizeof ctr, hipMemcpyDeviceToHost, d->s_copy));
    HIPCHK(d, hipStreamSynchronize(d->s_copy));
    f->stats.tris_setup = 0;
    for (int k = 0; k < CTR_NSHARDS; k++) f->stats.tris_setup += ctr[MTR_CTR(CTR_REC, k)];
    f->stats.bin_entries = ctr[CTR_ENTRIES];
    f->stats.segments = ctr[CTR_SEGS];
    if (f->ran_direct) {  // no scan in direct mode: the tile kernels counted the queues
        f->stats.bin_entries = f->stats.segments = 0;
        for (int k = 0; k < CTR_NSHARDS; k++) {
            f->stats.bin_entries += ctr[MTR_CTR(CTR_ENT, k)];
            f->stats.segments += ctr[MTR_CTR(CTR_SEG, k)];
        }
    }
    f->stats.binning = f->ran_direct ? 1u : 2u;
    f->stats.chunks = f->total_chunks;
    f->stats.chunks_culled = 0;
    for (int k = 0; k < CTR_NSHARDS; k++) f->stats.chunks_culled += ctr[MTR_CTR(CTR_CULL, k)];
    f->stats.shard_map = f->own ? f->own->map : 0u;
    f->stats.shard_bins = f->own ? f->own->offs[f->shard_rank + 1] - f->own->offs[f->shard_rank] : f->stats.nbins;
    f->stats_valid = true;
    return MTR_OK;
}

int32_t mtr_frame_end(mtr_frame* f) {
    int32_t rc = mtr_frame_submit(f);
    if (rc) return rc;
    return mtr_frame_wait(f);
}

int32_t mtr_frame_read_color(mtr_frame* f, void* rgba8, size_t len) {
    if (!f || !rgba8) return MTR_E_INVALID;
    mtr_device* d = f->dev;
    if (len < (size_t)f->w * f->h * 4) return fail(d, MTR_E_INVALID, "output too small");
    int32_t rc = mtr_frame_wait(f);
    if (rc) return rc;
    // the frame is complete (mtr_frame_wait): read it on the copy stream, not behind the later frames in flight
    HIPCHK(d, hipMemcpyAsync(rgba8, f->fb.color, (size_t)f->w * f->h * 4, hipMemcpyDeviceToHost, d->s_copy));
    HIPCHK(d, hipStreamSynchronize(d->s_copy));
    return MTR_OK;
}

int32_t mtr_frame_read_depth(mtr_frame* f, float* depth, size_t count) {
    if (!f || !depth) return MTR_E_INVALID;
    mtr_device* d = f->dev;
    if (count < (size_t)f->w * f->h) return fail(d, MTR_E_INVALID, "output too small");
    int32_t rc = mtr_frame_wait(f);
    if (rc) return rc;
    HIPCHK(d, hipMemcpyAsync(depth, f->fb.depth, (size_t)f->w * f->h * 4, hipMemcpyDeviceToHost, d->s_copy));
    HIPCHK(d, hipStreamSynchronize(d->s_copy));
    return MTR_OK;
}

void* mtr_frame_color_devptr(mtr_frame* f) { return f ? f->fb.color : nullptr; }
void* mtr_frame_depth_devptr(mtr_frame* f) { return f ? f->fb.depth : nullptr; }

size_t mtr_shard_bytes(uint32_t w, uint32_t h, uint32_t world) {
    if (world == 0) return 0;
    const size_t nbins = (size_t)((w + MTR_BIN - 1) / MTR_BIN) * ((h + MTR_BIN - 1) / MTR_BIN);
    return (nbins + world - 1) / world * (MTR_BIN * MTR_BIN * 4);
}

size_t mtr_shard_bytes_map(uint32_t w, uint32_t h, uint32_t world, uint32_t map, uint32_t param, const uint32_t* band_rows) {
    if (!valid_own_args(w, h, world, map, param, band_rows)) return 0;
    std::vector<uint32_t> bands, lists, offs;
    build_own_lists(w, h, world, map, map == MTR_OWN_SUPERTILES ? param : 0, band_rows, bands, lists, offs);
    return (size_t)stride_of(offs) * (MTR_BIN * MTR_BIN * 4);
}

// the frame's ownership table; an unsharded frame packs / unpacks as a world of one
static int32_t frame_table(mtr_frame* f, const OwnTable** t) {
    *t = f->own;
    if (*t) return MTR_OK;
    std::lock_guard<std::mutex> submit_lock(f->dev->submit_mu);
    return get_own_table(f->dev, f->w, f->h, 1, MTR_OWN_INTERLEAVED, 0, nullptr, t);
}

size_t mtr_frame_shard_bytes(mtr_frame* f) {
    if (!f) return 0;
    if (!f->own) return mtr_shard_bytes(f->w, f->h, 1);
    return (size_t)f->own->stride_bins * (MTR_BIN * MTR_BIN * 4);
}

static int32_t pack_shard_on(mtr_frame* f, void* dst_dev, size_t dst_bytes, hipStream_t s, bool wait_frame) {
    if (!f || !dst_dev) return MTR_E_INVALID;
    mtr_device* d = f->dev;
    if (!f->submitted) return fail(d, MTR_E_INVALID, "frame not submitted");
    if (dst_bytes < mtr_frame_shard_bytes(f)) return fail(d, MTR_E_INVALID, "shard buffer too small");
    int32_t rc = set_device(d);
    if (rc) return rc;
    const OwnTable* t = nullptr;
    if ((rc = frame_table(f, &t))) return rc;
    if (wait_frame) HIPCHK(d, hipStreamWaitEvent(s, f->fb.done, 0));  // the public stream already waits for every frame
    const uint32_t r = f->own ? f->shard_rank : 0;
    mtr_launch_pack_shard(f->fb.color, static_cast<uint8_t*>(dst_dev), f->w, f->h, t->d_lists + t->offs[r], t->offs[r + 1] - t->offs[r],
                          t->stride_bins, s);
    HIPCHK(d, hipGetLastError());
    // the colour buffer now has a reader after the tile kernel: whoever recycles it (the frame may be destroyed at
    // once) must wait for the pack too, so the buffer's completion event moves behind it
    HIPCHK(d, hipEventRecord(f->fb.done, s));
    return MTR_OK;
}

int32_t mtr_frame_pack_color_shard(mtr_frame* f, void* dst_dev, size_t dst_bytes) {
    return pack_shard_on(f, dst_dev, dst_bytes, f ? f->dev->stream : nullptr, false);
}

int32_t mtr_frame_pack_color_shard_on_stream(mtr_frame* f, void* dst_dev, size_t dst_bytes, void* hip_stream) {
    return pack_shard_on(f, dst_dev, dst_bytes, reinterpret_cast<hipStream_t>(hip_stream), true);
}

static int32_t unpack_table_on(mtr_device* d, const OwnTable* t, const void* gathered_dev, void* dst_dev, hipStream_t s) {
    mtr_launch_unpack_shards(static_cast<const uint8_t*>(gathered_dev), static_cast<uint8_t*>(dst_dev), t->w, t->h, t->d_src_of_bin, s);
    HIPCHK(d, hipGetLastError());
    return MTR_OK;
}

static int32_t unpack_shards_on(mtr_device* d, const void* gathered_dev, uint32_t world, uint32_t w, uint32_t h, void* dst_dev, hipStream_t s) {
    if (!d || !gathered_dev || !dst_dev) return MTR_E_INVALID;
    if (world == 0 || w == 0 || h == 0 || w > 16384 || h > 16384) return fail(d, MTR_E_INVALID, "bad unpack arguments");
    int32_t rc = set_device(d);
    if (rc) return rc;
    const OwnTable* t = nullptr;
    {
        std::lock_guard<std::mutex> submit_lock(d->submit_mu);
        if ((rc = get_own_table(d, w, h, world, MTR_OWN_INTERLEAVED, 0, nullptr, &t))) return rc;
    }
    return unpack_table_on(d, t, gathered_dev, dst_dev, s);
}

int32_t mtr_device_unpack_color_shards(mtr_device* d, const void* gathered_dev, uint32_t world, uint32_t w, uint32_t h,
                                       void* dst_dev) {
    return unpack_shards_on(d, gathered_dev, world, w, h, dst_dev, d ? d->stream : nullptr);
}

int32_t mtr_device_unpack_color_shards_on_stream(mtr_device* d, const void* gathered_dev, uint32_t world, uint32_t w, uint32_t h,
                                                 void* dst_dev, void* hip_stream) {
    return unpack_shards_on(d, gathered_dev, world, w, h, dst_dev, reinterpret_cast<hipStream_t>(hip_stream));
}

int32_t mtr_frame_unpack_color_shards_on_stream(mtr_frame* f, const void* gathered_dev, void* dst_dev, void* hip_stream) {
    if (!f || !gathered_dev || !dst_dev) return MTR_E_INVALID;
    mtr_device* d = f->dev;
    int32_t rc = set_device(d);
    if (rc) return rc;
    const OwnTable* t = nullptr;
    if ((rc = frame_table(f, &t))) return rc;
    return unpack_table_on(d, t, gathered_dev, dst_dev, hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : d->stream);
}

int32_t mtr_frame_get_stats(mtr_frame* f, mtr_frame_stats* out) {
    if (!f || !out) return MTR_E_INVALID;
    int32_t rc = fetch_stats(f);
    if (rc) return rc;
    *out = f->stats;
    return MTR_OK;
}

int32_t mtr_frame_get_timings(mtr_frame* f, float ms[MTR_STAGE_COUNT]) {
    if (!f || !ms) return MTR_E_INVALID;
    if (!f->have_events) return fail(f->dev, MTR_E_INVALID, "profiling was not enabled for this frame");
    int32_t rc = mtr_frame_wait(f);
    if (rc) return rc;
    memcpy(ms, f->ms, sizeof f->ms);
    return MTR_OK;
}

// ---------------------------------------------------------------------------------------------
// unit-test hooks
// ---------------------------------------------------------------------------------------------
int32_t mtr_frame_read_bin_counts(mtr_frame* f, uint32_t* entries, uint32_t* segments, size_t nbins) {
    if (!f || !entries || !segments) return MTR_E_INVALID;
    mtr_device* d = f->dev;
    if (nbins != f->stats.nbins) return fail(d, MTR_E_INVALID, "nbins mismatch");
    Slot& sl = d->slots[f->slot];
    int32_t rc = mtr_frame_wait(f);
    if (rc) return rc;
    std::vector<uint32_t> bs(nbins + 1), ss(nbins + 1);
    HIPCHK(d, hipMemcpyAsync(bs.data(), sl.bin_start, (nbins + 1) * 4, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(d, hipMemcpyAsync(ss.data(), sl.seg_start, (nbins + 1) * 4, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(d, hipStreamSynchronize(d->stream));
    if (f->ran_direct) {
        std::vector<unsigned long long> bf(nbins);  // the tile kernels moved the counts here when they cleaned bin_fill
        HIPCHK(d, hipMemcpyAsync(bf.data(), sl.bin_count, nbins * 8, hipMemcpyDeviceToHost, d->stream));
        HIPCHK(d, hipStreamSynchronize(d->stream));
        for (size_t b = 0; b < nbins; b++) { entries[b] = (uint32_t)bf[b]; segments[b] = (uint32_t)(bf[b] >> 32); }
        if (f->own && f->shard_world > 1) {
            // only the tile workgroups of the rank's own bins park a count: the words of the other bins hold whatever an
            // earlier frame on this slot left there
            std::vector<uint8_t> mine(nbins, 0);
            for (uint32_t k = f->own->offs[f->shard_rank]; k < f->own->offs[f->shard_rank + 1]; k++) mine[f->own->lists[k]] = 1;
            for (size_t b = 0; b < nbins; b++)
                if (!mine[b]) entries[b] = segments[b] = 0;
        }
        return MTR_OK;
    }
    for (size_t b = 0; b < nbins; b++) { entries[b] = bs[b + 1] - bs[b]; segments[b] = ss[b + 1] - ss[b]; }
    return MTR_OK;
}

int32_t mtr_model_vertex_stage(mtr_model* m, size_t prim, const float M[16], float* out_clip, float* out_uv) {
    if (!m || !M || !out_clip || !out_uv) return MTR_E_INVALID;
    mtr_device* d = m->dev;
    if (prim >= m->prims.size()) return fail(d, MTR_E_INVALID, "primitive out of range");
    int32_t rc = set_device(d);
    if (rc) return rc;
    const uint32_t nv = m->prims[prim].vertex_num;
    if (nv == 0) return MTR_OK;
    float *d_clip = nullptr, *d_uv = nullptr;
    if ((rc = dev_alloc(d, &d_clip, (size_t)nv * 4))) return rc;
    if ((rc = dev_alloc(d, &d_uv, (size_t)nv * 2))) return rc;
    GeomParams gp{};
    gp.vbuf = m->d_vbuf; gp.ibuf = m->d_ibuf; gp.prims = m->d_prims; gp.ninst = 1;
    gp.palettes = m->d_palette; gp.npal = m->d_palette ? m->npal : 0;
    memcpy(gp.vp, M, sizeof gp.vp);
    if (m->pal_ready) HIPCHK(d, hipStreamWaitEvent(d->stream, m->pal_ready, 0));
    mtr_launch_vertex_stage(gp, (uint32_t)prim, d_clip, d_uv, d->stream);
    HIPCHK(d, hipGetLastError());
    HIPCHK(d, hipMemcpyAsync(out_clip, d_clip, (size_t)nv * 16, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(d, hipMemcpyAsync(out_uv, d_uv, (size_t)nv * 8, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(d, hipStreamSynchronize(d->stream));
    (void)hipFree(d_clip);
    (void)hipFree(d_uv);
    return MTR_OK;
}

// ---------------------------------------------------------------------------------------------
// exchange thread
// ---------------------------------------------------------------------------------------------
static void exchange_main(mtr_device* d, Exchange* x) {
    (void)hipSetDevice(d->hip_dev);
    for (;;) {
        mtr_frame* f = nullptr;
        // a frame arrives every few tens of microseconds: poll briefly before sleeping on the condition variable
        for (int spin = 0; spin < 20000 && !f; spin++) {
            if (x->pending.load(std::memory_order_acquire)) {
                std::lock_guard<std::mutex> g(x->mu);
                if (!x->q.empty()) { f = x->q.front(); x->q.pop_front(); }
            } else {
                __builtin_ia32_pause();
            }
        }
        if (!f) {
            std::unique_lock<std::mutex> lk(x->mu);
            x->cv_items.wait(lk, [&] { return x->stop || !x->q.empty(); });
            if (x->q.empty()) return;  // stop requested and nothing left
            f = x->q.front(); x->q.pop_front();
        }
        x->cv_items.notify_all();  // room in the queue
        int32_t rc;
        { std::lock_guard<std::mutex> g(x->mu); rc = x->err; }
        std::string msg;
        // NO RANK MAY SKIP A COLLECTIVE.  The all-gather of frame k completes only when every rank has issued it: a rank
        // that failed (this frame, or an earlier one whose error has not been collected yet) still takes part, sending a
        // shard filled with the frame's clear colour, and carries its status out of band -- mtr_device_exchange_drain
        // returns it, and the host agrees on it across the ranks (bench.py: a MIN all-reduce after the drain).  Skipping
        // the call instead would leave the healthy ranks waiting in frame k's collective for ever.
        const Exchange::Lane ln = x->lanes[(size_t)(x->dealt++ % x->lanes.size())];
        // every rank sends exactly its shard of THIS frame: the unpack derives the per-rank stride from the frame size
        const size_t count = mtr_frame_shard_bytes(f);
        if (rc == MTR_OK) {
            // a frame whose bin queues overflowed is re-run (exact two-pass queues) BEFORE its colour is packed: the
            // gathered frame is never missing triangles.  The flags are known when the frame's tile kernel starts, so
            // in the normal case this does not wait for the frame to finish.
            rc = settle_frame(f, false);
            if (rc == MTR_OK) rc = mtr_frame_pack_color_shard_on_stream(f, ln.send, count, ln.stream);
            if (rc != MTR_OK) { std::lock_guard<std::mutex> g(g_err_mu); msg = d->err; }
        }
        if (rc != MTR_OK) {  // the shard of a rank in error: the clear colour (count is a multiple of 4)
            std::vector<uint32_t> fill(count / 4, f->clear_rgba8);
            (void)hipMemcpyAsync(ln.send, fill.data(), count, hipMemcpyHostToDevice, ln.stream);
            (void)hipStreamSynchronize(ln.stream);  // `fill` goes out of scope
        }
        {
            const int nrc = x->fn(ln.send, ln.gathered, count, x->dtype_u8, ln.comm, ln.stream);
            if (nrc != 0 && rc == MTR_OK) { rc = MTR_E_HIP; msg = "all-gather callback returned " + std::to_string(nrc); }
        }
        if (rc == MTR_OK) {
            rc = mtr_frame_unpack_color_shards_on_stream(f, ln.gathered, ln.dst, ln.stream);
            if (rc != MTR_OK) { std::lock_guard<std::mutex> g(g_err_mu); msg = d->err; }
        }
        mtr_frame_destroy(f);
        {
            std::lock_guard<std::mutex> g(x->mu);
            if (rc != MTR_OK && x->err == MTR_OK) { x->err = rc; x->err_msg = msg; }
            x->pending.fetch_sub(1, std::memory_order_release);
        }
        x->cv_idle.notify_all();
    }
}

int32_t mtr_device_exchange_start(mtr_device* d, mtr_allgather_fn fn, void* comm, int dtype_u8, void* send_dev, size_t send_bytes,
                                  void* gathered_dev, void* dst_dev, uint32_t world, void* hip_stream) {
    if (!d) return MTR_E_INVALID;
    if (!fn || !send_dev || !gathered_dev || !dst_dev || world == 0 || !hip_stream)
        return fail(d, MTR_E_INVALID, "bad exchange arguments");
    if (d->xchg) return fail(d, MTR_E_INVALID, "exchange already started");
    auto* x = new Exchange();
    x->fn = fn; x->comm = comm; x->dtype_u8 = dtype_u8;
    x->send = static_cast<uint8_t*>(send_dev); x->send_bytes = send_bytes;
    x->gathered = static_cast<uint8_t*>(gathered_dev); x->dst = static_cast<uint8_t*>(dst_dev);
    x->world = world; x->stream = reinterpret_cast<hipStream_t>(hip_stream);
    x->lanes.push_back({x->comm, x->send, x->gathered, x->dst, x->stream});
    d->xchg = x;
    x->th = std::thread(exchange_main, d, x);
    return MTR_OK;
}

int32_t mtr_device_exchange_add_lane(mtr_device* d, void* comm, void* send_dev, void* gathered_dev, void* dst_dev, void* hip_stream) {
    if (!d) return MTR_E_INVALID;
    Exchange* x = d->xchg;
    if (!x) return fail(d, MTR_E_INVALID, "no exchange thread (mtr_device_exchange_start)");
    if (!send_dev || !gathered_dev || !dst_dev || !hip_stream) return fail(d, MTR_E_INVALID, "bad exchange lane arguments");
    std::lock_guard<std::mutex> g(x->mu);
    if (x->pending.load(std::memory_order_acquire) != 0) return fail(d, MTR_E_INVALID, "exchange lanes change only while the thread is idle");
    if (x->lanes.size() >= 4) return fail(d, MTR_E_INVALID, "at most 4 exchange lanes");
    for (const Exchange::Lane& ln : x->lanes)
        if (ln.stream == hip_stream || ln.send == send_dev || ln.gathered == gathered_dev)
            return fail(d, MTR_E_INVALID, "an exchange lane needs a stream and buffers of its own");
    x->lanes.push_back({comm, static_cast<uint8_t*>(send_dev), static_cast<uint8_t*>(gathered_dev), static_cast<uint8_t*>(dst_dev),
                        reinterpret_cast<hipStream_t>(hip_stream)});
    return MTR_OK;
}

int32_t mtr_frame_submit_exchange(mtr_frame* f) {
    if (!f) return MTR_E_INVALID;
    mtr_device* d = f->dev;
    Exchange* x = d->xchg;
    if (!x) return fail(d, MTR_E_INVALID, "no exchange thread (mtr_device_exchange_start)");
    if (f->shard_world != x->world) return fail(d, MTR_E_INVALID, "frame shard world differs from the exchange's");
    if (x->send_bytes < mtr_frame_shard_bytes(f)) return fail(d, MTR_E_INVALID, "exchange send buffer too small");
    if (f->waited) f->flags_checked = true;
    if (!f->submitted) {
        f->for_exchange = true;  // its only consumer is the exchange thread, which waits for the frame on its own stream
        int32_t rc = mtr_frame_submit(f);
        if (rc) { f->for_exchange = false; return rc; }
    }
    {
        std::unique_lock<std::mutex> lk(x->mu);
        x->cv_items.wait(lk, [&] { return x->q.size() < Exchange::kDepth; });
        x->q.push_back(f);
        x->pending.fetch_add(1, std::memory_order_release);
    }
    x->cv_items.notify_all();
    return MTR_OK;
}

int32_t mtr_device_exchange_drain(mtr_device* d) {
    if (!d) return MTR_E_INVALID;
    Exchange* x = d->xchg;
    if (!x) return MTR_OK;
    std::unique_lock<std::mutex> lk(x->mu);
    x->cv_idle.wait(lk, [&] { return x->pending.load(std::memory_order_acquire) == 0; });
    if (x->err != MTR_OK) {
        const int32_t rc = x->err;
        const std::string msg = "exchange thread: " + x->err_msg;
        x->err = MTR_OK;
        lk.unlock();
        return fail(d, rc, msg);
    }
    return MTR_OK;
}

int32_t mtr_device_exchange_stop(mtr_device* d) {
    if (!d) return MTR_E_INVALID;
    Exchange* x = d->xchg;
    if (!x) return MTR_OK;
    const int32_t rc = mtr_device_exchange_drain(d);
    {
        std::lock_guard<std::mutex> g(x->mu);
        x->stop = true;
    }
    x->cv_items.notify_all();
    x->th.join();
    for (const Exchange::Lane& ln : x->lanes) (void)hipStreamSynchronize(ln.stream);
    d->xchg = nullptr;
    delete x;
    return rc;
}

uint32_t mtr_crc32(const uint8_t* bytes, size_t len, uint32_t init) {
    // src/util/crc.rs:36-50: reflected 0xEDB88320 table, no final xor, stops at the first NUL
    struct Table {
        uint32_t t[256];
        constexpr Table() : t() {
            for (uint32_t i = 0; i < 256; i++) {
                uint32_t c = i;
                for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
                t[i] = c;
            }
        }
    };
    static constexpr Table kTable{};  // constant-initialised: no lazy set-up for two threads to race on
    const uint32_t* table = kTable.t;
    uint32_t v = init;
    for (size_t i = 0; i < len && bytes[i] != 0; i++) v = table[(bytes[i] ^ v) & 0xff] ^ (v >> 8);
    return v;
}

}  // extern "C"
