// Probe of v_mfma_f32_4x4x1_16b_f32 on gfx950: lane -> (block, row/col) maps and fma-chain exactness.
// hipcc --offload-arch=gfx950 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k(const float* a, const float* b, float* d, int steps) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < steps; s++) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s * 64 + threadIdx.x], b[s * 64 + threadIdx.x], acc, 0, 0, 0);
    for (int i = 0; i < 4; i++) d[threadIdx.x * 4 + i] = acc[i];
}

int main() {
    const int S = 4;
    float ha[S * 64], hb[S * 64], hd[256];
    // layout probe: A = 100*lane, B = lane -> D[v] of lane l should be A[lane of (block, row v)] * B[l]
    for (int l = 0; l < 64; l++) { ha[l] = 1.0f + l; hb[l] = 1000.0f + l; }
    float *da, *db, *dd;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dd, sizeof hd);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dd, 1);
    hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int v = 0; v < 4; v++) {
            float expect = ha[(l & ~3) + v] * hb[l];  // A from lane (block base + row v), B from own lane
            if (hd[l * 4 + v] != expect) bad++;
        }
    printf("layout D[lane][v] = A[block*4+v] * B[lane]: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    // exactness: 4-step chain vs fmaf chain on awkward values
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) - (1 << 23)) / 8388608.0f * 3.7f; };
    for (int i = 0; i < S * 64; i++) { ha[i] = rnd(); hb[i] = rnd() * 1e-3f; }
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dd, S);
    hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    bad = 0;
    for (int l = 0; l < 64; l++)
        for (int v = 0; v < 4; v++) {
            float acc = 0.f;
            for (int st = 0; st < S; st++) acc = fmaf(ha[st * 64 + (l & ~3) + v], hb[st * 64 + l], acc);
            if (memcmp(&acc, &hd[l * 4 + v], 4)) bad++;
        }
    printf("4-step MFMA chain == fmaf chain bitwise: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    return 0;
}
