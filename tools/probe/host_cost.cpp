// Host cost of one frame through the C ABI, with the GPU idle: a one-triangle model, thousands of frames, the time spent
// inside each call (what a native host -- the reference's Rust -- pays per frame; the Python harness adds ctypes on top).
//   g++ -O2 -std=c++17 -I include tools/probe/host_cost.cpp -L mt_renderer_amd -lmtr -Wl,-rpath,$PWD/mt_renderer_amd -o /tmp/host_cost
#include "mtr.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define REQ(x) do { int32_t rc_ = (x); if (rc_) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, mtr_last_error(dev)); return 2; } } while (0)

int main(int argc, char** argv) {
    const int frames = argc > 1 ? atoi(argv[1]) : 20000;
    const int world = argc > 2 ? atoi(argv[2]) : 1;
    mtr_device* dev = nullptr;
    if (mtr_device_create(0, &dev)) { fprintf(stderr, "no device: %s\n", mtr_last_error(nullptr)); return 77; }
    const float verts[9] = {-0.5f, -0.5f, 0.5f, 0.5f, -0.5f, 0.5f, 0.0f, 0.5f, 0.5f};
    const uint16_t idx[3] = {0, 1, 2};
    mtr_primitive pr; memset(&pr, 0, sizeof pr);
    pr.w[0] = 3u << 16; pr.w[2] = 1 | (12u << 16) | (3u << 24); pr.w[7] = 3;
    mtr_layout l; memset(&l, 0, sizeof l);
    l.elements[l.num_elements++] = mtr_element{MTR_SEM_POSITION, MTR_IEF_F32, 3, 0, 0, 0};
    mtr_model* model = nullptr;
    REQ(mtr_model_create(dev, verts, sizeof verts, idx, 3, &pr, 1, &l, nullptr, nullptr, 0, nullptr, &model));
    const float clear[4] = {1, 1, 1, 1}, M[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    const uint32_t W = 256, H = 256;
    double t_begin = 0, t_shard = 0, t_draw = 0, t_submit = 0, t_destroy = 0;
    for (int pass = 0; pass < 2; pass++) {
        t_begin = t_shard = t_draw = t_submit = t_destroy = 0;
        const double t0 = now();
        for (int i = 0; i < frames; i++) {
            mtr_frame* f = nullptr;
            double a = now();
            REQ(mtr_frame_begin(dev, W, H, clear, 1.0f, &f));
            double b = now(); t_begin += b - a;
            if (world > 1) REQ(mtr_frame_set_shard_map(f, 1, (uint32_t)world, MTR_OWN_BANDS, 0, nullptr));
            a = now(); t_shard += a - b;
            REQ(mtr_frame_draw_model(f, model, M));
            b = now(); t_draw += b - a;
            REQ(mtr_frame_submit(f));
            a = now(); t_submit += a - b;
            mtr_frame_destroy(f);
            b = now(); t_destroy += b - a;
        }
        const double t1 = now();
        REQ(mtr_device_synchronize(dev));
        if (pass == 1)
            printf("world=%d frames=%d: %.2f us/frame on the host (begin %.2f, set_shard %.2f, draw %.2f, submit %.2f, destroy %.2f); drain %.0f us\n",
                   world, frames, (t1 - t0) / frames, t_begin / frames, t_shard / frames, t_draw / frames, t_submit / frames, t_destroy / frames, now() - t1);
    }
    mtr_model_destroy(model);
    mtr_device_destroy(dev);
    return 0;
}
