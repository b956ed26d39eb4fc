// MI355X finding behind DESIGN.md "VALU time adds to MFMA time": fp32 MFMAs of one wave and VALU instructions of ANOTHER wave
// on the same SIMD do not overlap (both-kernel time = sum of the two, not max).  hipcc --offload-arch=gfx950 -O3 -o mix scripts/mfma_valu_overlap.hip
// Do VALU instructions of one wave overlap with MFMAs of another wave on the same SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <typename F> float timeit(F f, int n = 30) { for (int i = 0; i < 5; ++i) f(); hipDeviceSynchronize(); hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a); for (int i = 0; i < n; ++i) f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / n; }
// waves 0-3: MFMA (mf iterations of 16 MFMAs); waves 4-7: VALU (va iterations of 64 fma); kind selects the VALU op
template <int KIND>
__global__ void k_mix(float* out, int mf, int va) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) {
        v4f acc[4];
        for (int c = 0; c < 4; ++c) acc[c] = (v4f){0.f, 0.f, 0.f, 0.f};
        float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
        for (int i = 0; i < mf; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
        for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    } else {
        float x[8];
        for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * (j + 1) * 1e-3f;
        int y[8];
        for (int j = 0; j < 8; ++j) y[j] = threadIdx.x * (j + 1);
        for (int i = 0; i < va; ++i)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (KIND == 0) x[j] = fmaf(x[j], 1.0001f, 0.5f);
                    else y[j] = (y[j] ^ (y[j] >> 3)) + 12345;   // integer ops
                }
        for (int j = 0; j < 8; ++j) s += x[j] + y[j];
    }
    if (s == 123.456f) out[0] = s;
}
int main() {
    float* d; hipMalloc(&d, 1024);
    const int mf = 2000;
    for (int kind = 0; kind < 2; ++kind) {
        auto L = [&](int m, int v) { return kind == 0 ? timeit([&] { hipLaunchKernelGGL(k_mix<0>, dim3(256), dim3(512), 0, 0, d, m, v); })
                                                      : timeit([&] { hipLaunchKernelGGL(k_mix<1>, dim3(256), dim3(512), 0, 0, d, m, v); }); };
        const float tm = L(mf, 0);
        for (int va : {500, 1000, 2000, 4000}) {
            const float tv = L(0, va), tb = L(mf, va);
            printf("kind=%d mfma-only %.1f us | valu-only(%d) %.1f us | both %.1f us  (max %.1f, sum %.1f)\n", kind, tm, va, tv, tb, tm > tv ? tm : tv, tm + tv);
        }
    }
    return 0;
}
