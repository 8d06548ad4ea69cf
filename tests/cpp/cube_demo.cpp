// Compile check + GPU demo of include/mtr.hpp: config C1's cube through the C++ mirror.
#include "mtr.hpp"
#include <cstdio>
#include <cstring>
#include <vector>

int main() {
    static const float verts[24] = {1, 1, -1, 1, -1, -1, 1, 1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1};
    static const uint16_t idx[36] = {4, 2, 0, 2, 7, 3, 6, 5, 7, 1, 7, 5, 0, 3, 1, 4, 1, 5, 4, 6, 2, 2, 6, 7, 6, 4, 5, 1, 3, 7, 0, 2, 3, 4, 0, 1};
    try {
        mtr::Device dev(0);
        mtr_primitive p{};
        p.w[0] = 8u << 16;
        p.w[2] = (12u << 16) | (3u << 24);
        p.w[7] = 36;
        mtr_layout l{};
        l.num_elements = 1;
        l.elements[0].semantic = MTR_SEM_POSITION;
        l.elements[0].format = MTR_IEF_F32;
        l.elements[0].count = 3;
        mtr::Model model(dev, verts, sizeof verts, idx, 36, {p}, {l}, {-1}, {}, {3});
        const float clear[4] = {1, 1, 1, 1};
        mtr::Frame frame(dev, 256, 256, clear, 1.0f);
        // a rotated cube pushed to z = 0.5 in an orthographic-style transform
        const float vp[16] = {0.25f, 0.1f, 0.05f, 0, -0.1f, 0.25f, 0.05f, 0, 0.05f, -0.05f, 0.1f, 0, 0, 0, 0.5f, 1};
        model.render(frame, vp);
        frame.end();
        mtr_frame_stats s = frame.stats();
        std::printf("tris_in=%llu tris_setup=%llu\n", (unsigned long long)s.tris_in, (unsigned long long)s.tris_setup);
        if (s.tris_in != 12 || s.tris_setup != 6) return 1;
        std::vector<uint8_t> one(256 * 256 * 4), gathered(256 * 256 * 4);
        frame.read_color(one.data(), one.size());
        // the same cube through a group of two ranks (both on card 0), bands of bin rows: the gathered image is the same image
        mtr::Group group({0, 0});
        mtr::Model m0(group.device(0), verts, sizeof verts, idx, 36, {p}, {l}, {-1}, {}, {3});
        mtr::Model m1(group.device(1), verts, sizeof verts, idx, 36, {p}, {l}, {-1}, {}, {3});
        {
            mtr::GroupFrame gf(group, 256, 256, clear, 1.0f, MTR_OWN_BANDS);
            m0.render(gf.part(0), vp);
            m1.render(gf.part(1), vp);
            gf.end();
            gf.read_color(gathered.data(), gathered.size());
        }
        std::printf("group image %s\n", one == gathered ? "matches" : "DIFFERS");
        return one == gathered ? 0 : 3;
    } catch (const mtr::Error& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return e.code == MTR_E_HIP ? 77 : 2;  // 77: no GPU here
    }
}
