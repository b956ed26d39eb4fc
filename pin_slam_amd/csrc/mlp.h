// Per-thread shallow-MLP decoder (Decoder.mlp, model/decoder.py:61-80) for gfx950.
//
// One thread evaluates one sample.  Weights are wave-uniform, so they are fetched with
// scalar loads (s_load_dwordx*) straight into SGPR operands of v_fmac -- no LDS traffic for
// weights.  The activation vector of the current layer lives in VGPRs (static indices);
// the layer being produced is written to a private LDS column (stride = block size, bank
// conflict free), because its index is the runtime loop variable.  ReLU masks (1 bit per
// unit per layer) stay in registers for the input-Jacobian / backward pass, so forward
// activations are never stored.
#pragma once
#include "pin_common.h"

namespace pin {

constexpr int MLP_IN = PIN_MLP_IN;  // 8 feature dims + 3 relative position
constexpr int MLP_MAX_LEVELS = 4;

// Wave-uniform, read-only parameters: loads through the constant address space are always
// selected as scalar (s_load_*) when the address is uniform.
#define PIN_CONST __attribute__((address_space(4)))
typedef const PIN_CONST float* cfloatp;
__device__ __forceinline__ cfloatp as_const(const float* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (cfloatp)p;
#pragma clang diagnostic pop
}

struct MlpMasks {
    // four named words, not an array: a runtime layer index must never become a dynamic
    // register-array index (that would be lowered to scratch memory)
    unsigned long long m0, m1, m2, m3;
    __device__ __forceinline__ unsigned long long get(int l) const {
        return l == 0 ? m0 : (l == 1 ? m1 : (l == 2 ? m2 : m3));
    }
    __device__ __forceinline__ void set(int l, unsigned long long v) {
        m0 = l == 0 ? v : m0; m1 = l == 1 ? v : m1; m2 = l == 2 ? v : m2; m3 = l == 3 ? v : m3;
    }
};

__host__ __device__ inline int mlp_param_count(int H, int L) {
    return H * MLP_IN + H + (L - 1) * (H * H + H) + H + 1;
}

// Forward.  `col` points at this thread's LDS column (element i at col[i * T]).
// Returns the raw MLP output (before sdf_scale).
template <int H, int T>
__device__ __forceinline__ float mlp_forward(cfloatp P, const int L,
                                             const float (&z)[MLP_IN], float* col, MlpMasks& mk) {
    {
        cfloatp W = P;
        cfloatp b = P + H * MLP_IN;
        unsigned long long m = 0;
        for (int i = 0; i < H; ++i) {
            float acc = b[i];
#pragma unroll
            for (int j = 0; j < MLP_IN; ++j) acc = fmaf(W[i * MLP_IN + j], z[j], acc);
            const bool on = acc > 0.f;
            m |= (unsigned long long)on << i;
            col[i * T] = on ? acc : 0.f;
        }
        mk.m0 = m; mk.m1 = 0; mk.m2 = 0; mk.m3 = 0;
        P += H * MLP_IN + H;
    }
    float h[H];
    for (int l = 1; l < L; ++l) {
#pragma unroll
        for (int j = 0; j < H; ++j) h[j] = col[j * T];
        cfloatp W = P;
        cfloatp b = P + H * H;
        unsigned long long m = 0;
        for (int i = 0; i < H; ++i) {
            float acc = b[i];
#pragma unroll
            for (int j = 0; j < H; ++j) acc = fmaf(W[i * H + j], h[j], acc);
            const bool on = acc > 0.f;
            m |= (unsigned long long)on << i;
            col[i * T] = on ? acc : 0.f;
        }
        mk.set(l, m);
        P += H * H + H;
    }
    float out = P[H];  // lout.bias
#pragma unroll
    for (int j = 0; j < H; ++j) out = fmaf(P[j], col[j * T], out);
    return out;
}

// d out / d z through the stored ReLU masks (the autograd path of get_gradient,
// utils/tools.py:247-260, restricted to the decoder).
template <int H, int T>
__device__ __forceinline__ void mlp_input_jacobian(cfloatp P, const int L,
                                                   const MlpMasks& mk, float* col, float (&a_in)[MLP_IN]) {
    cfloatp Wo = P + H * MLP_IN + H + (L - 1) * (H * H + H);
    // a = d out / d h_{L-1} masked by that layer's ReLU, staged in the LDS column
    {
        const unsigned long long m = mk.get(L - 1);
#pragma unroll
        for (int j = 0; j < H; ++j) col[j * T] = ((m >> j) & 1ull) ? Wo[j] : 0.f;
    }
    for (int l = L - 1; l >= 1; --l) {
        cfloatp W = P + H * MLP_IN + H + (l - 1) * (H * H + H);
        float ap[H];
#pragma unroll
        for (int j = 0; j < H; ++j) ap[j] = 0.f;
        for (int i = 0; i < H; ++i) {
            const float am = col[i * T];
#pragma unroll
            for (int j = 0; j < H; ++j) ap[j] = fmaf(W[i * H + j], am, ap[j]);
        }
        const unsigned long long m = mk.get(l - 1);
#pragma unroll
        for (int j = 0; j < H; ++j) col[j * T] = ((m >> j) & 1ull) ? ap[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < MLP_IN; ++j) a_in[j] = 0.f;
    for (int i = 0; i < H; ++i) {
        const float am = col[i * T];
#pragma unroll
        for (int j = 0; j < MLP_IN; ++j) a_in[j] = fmaf(P[i * MLP_IN + j], am, a_in[j]);
    }
}

}  // namespace pin
