// QuadDecoder<64>::run in isolation: 12 waves per CU, T tiles per wave, synthetic inputs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "pin_abi.h"
#include "mlp_quad.h"
using namespace pin;
template <typename F> float timeit(F f, int n = 50) { for (int i = 0; i < 5; ++i) f(); hipDeviceSynchronize(); hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a); for (int i = 0; i < n; ++i) f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / n; }

template <int MODE>
__global__ __launch_bounds__(GQ_BLOCK, 1) void k_dec(const float* dec, int L, int tiles, float* out) {
    using Q = QuadDecoder<64>;
    __shared__ __attribute__((aligned(16))) float lds[Q::TOTAL];
    Q::stage(dec, L, lds, threadIdx.x, blockDim.x);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float z[4] = {lane * 1e-3f, 0.1f, 0.2f, -0.3f};
    float s = 0.f;
    for (int t = 0; t < tiles; ++t) {
        float a[4];
        const float x = Q::run(lds, L, z, a);
        s += x + a[0] + a[1] + a[2] + a[3];
        z[0] += 1e-3f * x;
    }
    if (s == 123.456f) out[0] = s;
}
int main() {
    const int L = 4, H = 64;
    const int n = 4 * (H * 11 + H) + 3 * (H * H + H) + H + 1 + 4096;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = ((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
    float *d, *o; hipMalloc(&d, n * 4); hipMalloc(&o, 1024); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int waves : {4, 8, 12}) {
        const int tiles = 40;
        float t = timeit([&] { hipLaunchKernelGGL(k_dec<0>, dim3(256), dim3(waves * 64), 0, 0, d, L, tiles, o); });
        float t0 = timeit([&] { hipLaunchKernelGGL(k_dec<0>, dim3(256), dim3(waves * 64), 0, 0, d, L, 0, o); });
        const double per = (t - t0) / (tiles * waves / 4.0);
        printf("waves/CU=%2d: %.1f us (empty %.1f) -> %.2f us per tile per SIMD, MFMA share %.0f%%\n", waves, t, t0, per, 100.0 * 416 * 32 / 2400.0 / per);
    }
    return 0;
}
