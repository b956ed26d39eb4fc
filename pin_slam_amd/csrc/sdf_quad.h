// Inference on the tile decoder: pin_sdf_query (Tracker.query_source_points, utils/tracker.py:297-354; Mesher.query_points,
// utils/mesher.py:60-140) for the interpolate-first mode with one SDF head, FOUR LANES PER QUERY as in gn_quad.h -- the same
// gather (quad_gather_pass), the same split-fp16 decoder image, the same chain rule (quad_chain), without the Gauss-Newton
// sums.  Replaces sdf_query_mfma_kernel (a thread per query: 3 x 11 + 11 + 9 accumulators and the neighbour records of a
// query in ONE lane -- 256 registers and 40-316 B of scratch at H = 64) on this path; per-neighbour decoding (which reports
// the spread of the k predictions) and the fp32-image A/B mode stay on that kernel.
//   GRAD = false (the mesher's 1e7..5e8 grid queries): the interpolated input only, forward sweep only.
//   certainty: sum_t w_t certainty[idx_t] (neural_points.py:726-729), accumulated by the lanes g == 3 of the query, which
//   the gather leaves idle.
#pragma once
#include "gn_quad.h"

namespace pin {

template <bool ORIENT, bool GRAD>
__device__ __forceinline__ void quad_gather_query(const pin_field& f, const float4* __restrict__ rp, int kk, int nn, float px,
                                                  float py, float pz, int g, QuadIn<ORIENT>& in, bool want_cert, float& cert) {
    float4 e[PIN_MAX_K];
#pragma unroll
    for (int t = 0; t < PIN_MAX_K; ++t) e[t] = rp[t < kk ? t : 0];
    const float4* __restrict__ rows = reinterpret_cast<const float4*>(f.feats) + (g & 1);
    float u[PIN_MAX_K];
    float4 ft[PIN_MAX_K];
    float cv[PIN_MAX_K];
    int raw[PIN_MAX_K];
    float S = 0.f;
    bool any_flag = false;
    const bool cert_lane = want_cert && g == 3;
#pragma unroll
    for (int t = 0; t < PIN_MAX_K; ++t) {  // (quad_gather's prologue: every load of the tile is issued here)
        raw[t] = __float_as_int(e[t].w);
        const bool val = t < kk && raw[t] >= 0;
        const int id = val ? (raw[t] & ~PIN_NBR_QUIRK_BIT) : 0;
        ft[t] = rows[2 * (size_t)(unsigned int)id];
        cv[t] = (cert_lane && val) ? f.certainty[id] : 0.f;
        const float ut = val ? __builtin_amdgcn_rcpf(dist2_exact(e[t].x, e[t].y, e[t].z) + IDW_EPS) : 0.f;
        u[t] = ut;
        S += (nn == 0 && t < kk) ? IDW_EPS : ut;
        any_flag = any_flag || (val && (raw[t] & PIN_NBR_QUIRK_BIT) != 0);
        raw[t] = val ? raw[t] : -1;
    }
    const bool general = ORIENT || __builtin_amdgcn_ballot_w64(any_flag) != 0ull;
    if constexpr (GRAD) {
        if constexpr (ORIENT) {
            quad_gather_pass<true, true>(f, e, ft, u, raw, S, px, py, pz, g, in);
        } else {
            // rare: a flagged neighbour, or a query with nearly all its weight on its nearest neighbour (the GENERAL pass is the
            // pivoted one: quad_gather_pass, PIVOT)
            if (general || __builtin_amdgcn_ballot_w64(u[0] > QUAD_PIVOT_SHARE * S) != 0ull)
                quad_gather_pass<false, true>(f, e, ft, u, raw, S, px, py, pz, g, in);
            else quad_gather_pass<false, false>(f, e, ft, u, raw, S, px, py, pz, g, in);
        }
    } else {
        const float invS = 1.0f / S;
        const bool is_feat = g < 2;
        const float mv = g == 2 ? 1.f : 0.f;
        float z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < PIN_MAX_K; ++t) {
            const float wt = u[t] * invS;
            float v[3] = {e[t].x, e[t].y, e[t].z};
            if (general) {  // (wave-uniform) after PGO, or a flagged neighbour somewhere in the wave
                float Rm[9];
                if (raw[t] >= 0)
                    neighbor_vector(f, raw[t] & ~PIN_NBR_QUIRK_BIT, (raw[t] & PIN_NBR_QUIRK_BIT) != 0, e[t].x, e[t].y, e[t].z, px, py,
                                    pz, v, Rm);
            }
            z[0] = fmaf(wt, is_feat ? ft[t].x : mv * v[0], z[0]);
            z[1] = fmaf(wt, is_feat ? ft[t].y : mv * v[1], z[1]);
            z[2] = fmaf(wt, is_feat ? ft[t].z : mv * v[2], z[2]);
            z[3] = fmaf(wt, is_feat ? ft[t].w : 0.f, z[3]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) in.z[r] = z[r];
        in.S = S;
    }
    cert = 0.f;
    if (want_cert) {  // (wave-uniform)
        const float invS = 1.0f / S;
#pragma unroll
        for (int t = 0; t < PIN_MAX_K; ++t) cert = fmaf(cv[t], u[t] * invS, cert);
    }
}

// threads per (persistent, one-per-CU) block: with the gradient 8 waves, 2 per SIMD (up to 256 registers); forward only 12
// waves, 3 per SIMD (the forward sweep needs ~140 registers, and a tile is a chain of memory round trips: waves in flight
// are what hides them)
template <bool GRAD>
constexpr int sq_block() { return GRAD ? 512 : 768; }

// OD = 3: the colour field (pin_color_query; Decoder.regress_color, model/decoder.py:112): `sdf_out` takes the value
// sum_c kappa_c sigmoid(head_c), `grad_out` its gradient, `color_out` [n][3] the three sigmoid outputs
template <int H, bool ORIENT, int LC, bool GRAD, int OD = 1>
__global__ __launch_bounds__(sq_block<GRAD>(), 1) void sdf_query_quad_kernel(pin_field f, const float* __restrict__ query,
                                                                    const float4* __restrict__ nbr,
                                                                    const int* __restrict__ nn_count, int n_q,
                                                                    float* __restrict__ sdf_out, float* __restrict__ grad_out,
                                                                    float* __restrict__ std_out, float* __restrict__ cert_out,
                                                                    Kappa kap, float* __restrict__ color_out) {
    using Q = QuadDecoderH<H>;
    constexpr int SQ_BLOCK = sq_block<GRAD>();
    // (the SAME dynamic-LDS symbol as the registration kernels of this translation unit: a second `extern __shared__` name
    // makes the compiler give up folding the first one's addresses into instruction offsets -- +30 address additions and
    // 187 -> 211 registers in gn_accumulate_quad_kernel<64, .., 4>, found when this header was added)
    extern __shared__ __attribute__((aligned(16))) unsigned char gq_smem[];  // the decoder image
    unsigned char* const lds = gq_smem;
    if (f.dec_image != nullptr && f.dec_image_bytes == Q::bytes(LC)) {  // staged by the caller (pin_stage_decoder): a linear copy
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(f.dec_image);
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(lds);
        constexpr int N16 = Q::bytes(LC) >> 4, TRIPS = (N16 + SQ_BLOCK - 1) / SQ_BLOCK, CH = 4;  // (CH loads in flight per lane)
#pragma unroll
        for (int base = 0; base < TRIPS; base += CH) {
            uint4 v[CH];
#pragma unroll
            for (int it = 0; it < CH; ++it) {
                const int i = (base + it) * SQ_BLOCK + threadIdx.x;
                v[it] = src[i < N16 ? i : 0];
            }
#pragma unroll
            for (int it = 0; it < CH; ++it) {
                const int i = (base + it) * SQ_BLOCK + threadIdx.x;
                if (i < N16) dst[i] = v[it];
            }
        }
    } else {
        Q::stage(f.dec, LC, lds, threadIdx.x, SQ_BLOCK, OD);
    }
    const int lane = threadIdx.x & 63, nq = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const int n_tiles = (n_q + 15) >> 4;
    const int n_simd = gridDim.x * 4;
    const int simd = blockIdx.x * 4 + (wave & 3);
    const bool want_cert = cert_out != nullptr && f.certainty != nullptr;
    const float s = f.sdf_scale;
    bool staged = false;
    for (int tile = simd + n_simd * (wave >> 2);; tile += n_simd * (SQ_BLOCK / 256)) {
        const bool work = tile < n_tiles;
        if (!work && staged) break;
        const int qi = (work ? tile : 0) * 16 + nq;
        const bool active = work && qi < n_q;
        const int qq = qi < n_q ? qi : n_q - 1;
        const float px = query[3 * qq], py = query[3 * qq + 1], pz = query[3 * qq + 2];
        const int nn = nn_count[qq];
        QuadIn<ORIENT> in;
        float cert;
        quad_gather_query<ORIENT, GRAD>(f, nbr + (size_t)qq * f.k, f.k, nn, px, py, pz, g, in, want_cert, cert);
        if (!staged) {  // the first gather overlaps the staging of the image
            __syncthreads();
            staged = true;
            if (!work) break;
        }
        if constexpr (OD == 3) {
            const float kk[3] = {kap.k[0], kap.k[1], kap.k[2]};
            float pc[3], value;
            if constexpr (GRAD) {
                float a[4], gx, gy, gz;
                value = Q::template run_color<LC>(lds, in.z, kk, true, a, &pc);
                quad_chain<ORIENT>(in, a, g, 1.0f, gx, gy, gz);  // (colour heads are not scaled, decoder.py:112)
                if (active && g == 0) {
                    grad_out[3 * (size_t)qi] = gx; grad_out[3 * (size_t)qi + 1] = gy; grad_out[3 * (size_t)qi + 2] = gz;
                }
            } else {
                int zero = 0;  // (as below)
                asm volatile("" : "+v"(zero));
                float o[3];
                Q::template forward<LC, 3>(lds + zero, in.z, o);
                value = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {  // (run_color's arithmetic)
                    pc[c] = 1.f / (1.f + expf(-o[c]));
                    value = fmaf(kk[c], pc[c], value);
                }
            }
            if (active && g == 1 && sdf_out) sdf_out[qi] = value;
            if (active && g == 2 && color_out) {
                color_out[3 * (size_t)qi] = pc[0]; color_out[3 * (size_t)qi + 1] = pc[1]; color_out[3 * (size_t)qi + 2] = pc[2];
            }
        } else if constexpr (GRAD) {
            float a[4], gx, gy, gz;
            const float x = Q::template run<LC>(lds, in.z, a);
            quad_chain<ORIENT>(in, a, g, s, gx, gy, gz);
            if (active && g == 0) {
                if (sdf_out) sdf_out[qi] = s * x;
                grad_out[3 * (size_t)qi] = gx; grad_out[3 * (size_t)qi + 1] = gy; grad_out[3 * (size_t)qi + 2] = gz;
            }
        } else {
            // (an opaque zero in the image's address: at 168 registers the compiler otherwise hoists weight fragments -- loop
            // invariant LDS reads -- out of the tile loop and then spills them to scratch memory)
            int zero = 0;
            asm volatile("" : "+v"(zero));
            float x[1];
            Q::template forward<LC, 1>(lds + zero, in.z, x);
            if (active && g == 0 && sdf_out) sdf_out[qi] = s * x[0];
        }
        if (active && g == 1 && std_out) std_out[qi] = 0.f;  // (one prediction per query: no spread, tracker.py:317-322)
        if (active && g == 3 && cert_out) cert_out[qi] = cert;
    }
}

}  // namespace pin
