// The shallow-MLP decoder on the fp32 matrix cores for ONE 16-query tile per wave, four lanes per query:
// lane (n, g) owns decoder-input components 4g..4g+3 of query n (see gn_quad.h for why).  The fp32 image is the
// A/B reference (PIN_MLP=f32) of the split-fp16 decoder in mlp_h2.h.
#pragma once
#include "mlp_mfma.h"

namespace pin {

#ifndef PIN_GQ_BLOCK
#define PIN_GQ_BLOCK 512
#endif
constexpr int GQ_BLOCK = PIN_GQ_BLOCK;  // 2 waves per SIMD: up to 256 VGPRs each (the tile kernels are bound by instruction issue, not occupancy)

template <int H>
struct QuadDecoder {
    using D = MfmaDecoder<H>;
    static constexpr int MT = H / 16;
    static constexpr int OFF_A0Q = D::weight_floats(MLP_MAX_LEVELS);  // [MT][4][64] layer-0 forward operand, k = 4g + r
    static constexpr int TOTAL = OFF_A0Q + MT * 4 * 64;

    __device__ static void stage(const float* __restrict__ dec, int L, float* __restrict__ w, int tid, int nthreads) {
        D::stage(dec, L, w, tid, nthreads, 1);
        D::copy_permuted(dec, w + OFF_A0Q, MT * 4 * 64, tid, nthreads, [](int e) {
            const int lane = e & 63, r = (e >> 6) & 3, mt = e >> 8;
            const int c = 4 * (lane >> 4) + r;
            return c < MLP_IN ? (16 * mt + (lane & 15)) * MLP_IN + c : -1;
        });
    }

    __device__ __forceinline__ static unsigned int relu16(const v4f_t (&acc)[MT], v4f_t (&h)[MT]) {
        unsigned int mm = 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool on = acc[mt][r] > 0.f;
                mm |= (unsigned int)on << (mt * 4 + r);
                h[mt][r] = on ? acc[mt][r] : 0.f;
            }
        return mm;
    }

    // one hidden layer: acc = bias + W h   (MT independent accumulator chains, A operands one K-tile ahead)
    __device__ __forceinline__ static void hidden_forward(const float* __restrict__ F, const v4f_t (&h)[MT], v4f_t (&acc)[MT]) {
        const int lane = threadIdx.x & 63, g = lane >> 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = *reinterpret_cast<const v4f_t*>(F + H * H + 16 * mt + 4 * g);
        v4f_t a4[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a4[mt] = *reinterpret_cast<const v4f_t*>(F + ((mt * MT + 0) * 64 + lane) * 4);
#pragma unroll
        for (int kt = 0; kt < MT; ++kt) {
            v4f_t nx[MT];
            if (kt + 1 < MT) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    nx[mt] = *reinterpret_cast<const v4f_t*>(F + ((mt * MT + kt + 1) * 64 + lane) * 4);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mt][r], h[kt][r], acc[mt], 0, 0, 0);
            if (kt + 1 < MT) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a4[mt] = nx[mt];
            }
        }
    }

    // one transposed hidden layer: h <- mask .* (W^T h), W read out of the forward image
    __device__ __forceinline__ static void hidden_backward(const float* __restrict__ F, unsigned int mm, v4f_t (&h)[MT],
                                                           v4f_t (&acc)[MT]) {
        const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
        // W_l[16*ki + 4*g + r][16*mj + n] out of the forward image
        const float* __restrict__ Ft = F + (16 * (n >> 2) + 4 * g) * 4 + (n & 3);
#pragma unroll
        for (int mj = 0; mj < MT; ++mj) acc[mj] = (v4f_t){0.f, 0.f, 0.f, 0.f};
        float at[MT][4], nx[MT][4];
#pragma unroll
        for (int mj = 0; mj < MT; ++mj)
#pragma unroll
            for (int r = 0; r < 4; ++r) at[mj][r] = Ft[((0 * MT + mj) * 64 + r) * 4];
#pragma unroll
        for (int ki = 0; ki < MT; ++ki) {
            if (ki + 1 < MT) {
#pragma unroll
                for (int mj = 0; mj < MT; ++mj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) nx[mj][r] = Ft[(((ki + 1) * MT + mj) * 64 + r) * 4];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mj = 0; mj < MT; ++mj)
                    acc[mj] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[mj][r], h[ki][r], acc[mj], 0, 0, 0);
            if (ki + 1 < MT) {
#pragma unroll
                for (int mj = 0; mj < MT; ++mj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) at[mj][r] = nx[mj][r];
            }
        }
#pragma unroll
        for (int mj = 0; mj < MT; ++mj)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[mj][r] = ((mm >> (mj * 4 + r)) & 1u) ? acc[mj][r] : 0.f;
    }

    // transposed layer 0 (two interleaved accumulation chains): a[r] = d out / d z[4g + r]
    __device__ __forceinline__ static void input_backward(const float* __restrict__ w, const v4f_t (&h)[MT], float (&a)[4]) {
        const int lane = threadIdx.x & 63;
        v4f_t ai0 = (v4f_t){0.f, 0.f, 0.f, 0.f}, ai1 = (v4f_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < MT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                ai0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[D::OFF_A0T + (kt * 4 + r) * 64 + lane], h[kt][r], ai0, 0, 0, 0);
                ai1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[D::OFF_A0T + (kt * 4 + r + 1) * 64 + lane], h[kt][r + 1], ai1, 0, 0, 0);
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = ai0[r] + ai1[r];
    }

    // forward + input Jacobian of one 16-query tile.  z[r] = component 4g + r of this lane's query;
    // returns the raw MLP output (complete in all four lanes of the query), a[r] = d out / d z[4g + r].
    __device__ __forceinline__ static float run(const float* __restrict__ w, int L, const float (&z)[4], float (&a)[4]) {
        const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
        v4f_t h[MT], acc[MT];
        unsigned int masks[MLP_MAX_LEVELS];
        // ---- layer 0: the MT accumulators are independent chains
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = *reinterpret_cast<const v4f_t*>(w + D::OFF_B0 + 16 * mt + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[OFF_A0Q + (mt * 4 + r) * 64 + lane], z[r], acc[mt], 0, 0, 0);
        masks[0] = relu16(acc, h);
#pragma unroll
        for (int l = 1; l < MLP_MAX_LEVELS; ++l) masks[l] = 0;
        // ---- hidden layers
        for (int l = 1; l < L; ++l) {
            hidden_forward(w + D::OFF_HID + (l - 1) * D::HID_SZ, h, acc);
            const unsigned int mm = relu16(acc, h);
#pragma unroll
            for (int q = 1; q < MLP_MAX_LEVELS; ++q) masks[q] = q == l ? mm : masks[q];
        }
        // ---- output head
        const float* __restrict__ O = w + D::off_out(L);
        float x = 0.f;
#pragma unroll
        for (int kt = 0; kt < MT; ++kt) {
            const v4f_t wo = *reinterpret_cast<const v4f_t*>(O + 16 * kt + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) x = fmaf(wo[r], h[kt][r], x);
        }
        x = rows_sum_lds(x);
        x += O[MF_OD_MAX * H];
        // ---- transposed sweep: seed with the output weights under the last ReLU mask
        unsigned int mlast = masks[0];
#pragma unroll
        for (int q = 1; q < MLP_MAX_LEVELS; ++q) mlast = q == L - 1 ? masks[q] : mlast;
#pragma unroll
        for (int kt = 0; kt < MT; ++kt) {
            const v4f_t wo = *reinterpret_cast<const v4f_t*>(O + 16 * kt + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) h[kt][r] = ((mlast >> (kt * 4 + r)) & 1u) ? wo[r] : 0.f;
        }
        for (int l = L - 1; l >= 1; --l) {
            unsigned int mm = masks[0];
#pragma unroll
            for (int q = 1; q < MLP_MAX_LEVELS; ++q) mm = q == l - 1 ? masks[q] : mm;
            hidden_backward(w + D::OFF_HID + (l - 1) * D::HID_SZ, mm, h, acc);
        }
        input_backward(w, h, a);
        return x;
    }
};

// sum over the 16 query lanes of a DPP row (result in every lane of the row)
__device__ __forceinline__ float row_sum_f32(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}
__device__ __forceinline__ float octet_sum(float v) {  // over aligned groups of 8 lanes (result in every lane of the group)
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    return v;
}
__device__ __forceinline__ float quad_lanes_sum(float v) {  // over the four lanes (n, g = 0..3) of a query
    v = rows_sum_lds(v);
    return v;
}

}  // namespace pin
