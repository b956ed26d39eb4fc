// Fused SDF + Jacobian + Gauss-Newton sums with FOUR LANES PER QUERY (weighted_first, one SDF head).
//
// Why: a 100k-point scan is only ~1.5 waves per SIMD when a wave carries 64 queries -- too few
// to hide LDS / memory latency behind the fp32 MFMAs, and the thread-per-query phases hold
// 3x11 + 11 + 11 live floats per lane.  Here a wave carries ONE 16-query MFMA tile and the four
// lanes (n, g), g = 0..3, of query n split the decoder INPUT COMPONENTS:
//     lane g owns components 4g..4g+3 of [f_0..f_7, v_x, v_y, v_z, 0, ...]
// which is exactly (a) the B operand of the first layer (K-step r uses component 4g + r; the
// layer-0 weights are staged in that k order), and (b) the layout in which the transposed
// first layer returns the input Jacobian (D[row = 4g + r][query n] in lane (n, g), register r).
// So the interpolated input z, the Jacobian a and the 3x11 matrix Y = sum_t g_t (x) y_t never
// leave the registers of the lane that produced them: no LDS exchange, 12 + 4 + 4 live floats.
// Lanes g = 0,1 gather the two halves of each neighbour's 32-byte feature row, g = 2 handles the
// relative positions (and the after-PGO rotation), g = 3 the certainties.
//
// Scheduling: persistent blocks of 16 waves (one per CU, 4 waves per SIMD, <= 128 VGPRs); the
// 16-query tiles are dealt round-robin to the SIMDs, so every SIMD gets the same MFMA work
// within one tile.  The weight image is staged once per block.
#pragma once
#include "mlp_mfma.h"

namespace pin {

constexpr int GQ_BLOCK = 1024;

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
            const float* __restrict__ F = w + D::OFF_HID + (l - 1) * D::HID_SZ;
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
        x += __shfl_xor(x, 16, 64);
        x += __shfl_xor(x, 32, 64);
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
            const float* __restrict__ F = w + D::OFF_HID + (l - 1) * D::HID_SZ;
            unsigned int mm = masks[0];
#pragma unroll
            for (int q = 1; q < MLP_MAX_LEVELS; ++q) mm = q == l - 1 ? masks[q] : mm;
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
        // ---- transposed layer 0: two interleaved accumulation chains
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
__device__ __forceinline__ float quad_lanes_sum(float v) {  // over the four lanes (n, g = 0..3) of a query
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

template <int H, bool ORIENT>
__global__ __launch_bounds__(GQ_BLOCK, 1) void gn_accumulate_quad_kernel(pin_field f, pin_gn_params gp,
                                                                         const float* __restrict__ query,
                                                                         const float4* __restrict__ nbr,
                                                                         const int* __restrict__ nn_count,
                                                                         const float* __restrict__ labels, int n_q,
                                                                         double* __restrict__ sums, float* __restrict__ sdf_out,
                                                                         float* __restrict__ grad_out,
                                                                         const double* __restrict__ state) {
    using Q = QuadDecoder<H>;
    __shared__ __attribute__((aligned(16))) float lds[Q::TOTAL];
    __shared__ float red[GQ_BLOCK / 64][PIN_GN_NSUMS];
    if (state != nullptr && state[PIN_GN_STATE_DONE] != 0.0) return;
    Q::stage(f.dec, f.levels, lds, threadIdx.x, GQ_BLOCK);  // visible after the barrier that follows the first gather
    const int lane = threadIdx.x & 63, nq = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const int n_tiles = (n_q + 15) >> 4;
    const int n_simd = gridDim.x * 4;
    const int simd = blockIdx.x * 4 + (wave & 3);
    const float s = f.sdf_scale;
    // The four waves of a SIMD would otherwise run their phases in lock step (all gather, then all
    // queue on the MFMA pipe).  Different priorities let one wave finish its decoder first and
    // move on to its next gather while the others keep the matrix pipe busy.
    switch (wave >> 2) {  // s_setprio takes an immediate
        case 0: __builtin_amdgcn_s_setprio(3); break;
        case 1: __builtin_amdgcn_s_setprio(2); break;
        case 2: __builtin_amdgcn_s_setprio(1); break;
        default: __builtin_amdgcn_s_setprio(0); break;
    }
    // running sums: lane (n, g) keeps sums i = 4j + g (j = 0..7) of ITS queries; one row reduction at the end
    float tot[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool staged = false;

    for (int tile = simd + n_simd * (wave >> 2);; tile += n_simd * (GQ_BLOCK / 256)) {
        const bool work = tile < n_tiles;
        if (!work && staged) break;
        const int qi = (work ? tile : 0) * 16 + nq;
        const bool active = qi < n_q;
        const int qq = active ? qi : n_q - 1;
        const float px = query[3 * qq], py = query[3 * qq + 1], pz = query[3 * qq + 2];
        const int nn = nn_count[qq];
        // ---- neighbour records and IDW weights (all four lanes of the query; neural_points.py:660-683)
        float vx[PIN_MAX_K], vy[PIN_MAX_K], vz[PIN_MAX_K], u[PIN_MAX_K];
        int idx[PIN_MAX_K];
        float S = 0.f;
#pragma unroll
        for (int t = 0; t < PIN_MAX_K; ++t) {
            idx[t] = -1; u[t] = 0.f; vx[t] = vy[t] = vz[t] = 0.f;
            if (t < f.k) {
                const float4 e = nbr[(size_t)qq * f.k + t];
                const int raw = __float_as_int(e.w);
                if (raw >= 0) {
                    idx[t] = raw;  // quirk bit kept, stripped where used
                    vx[t] = e.x; vy[t] = e.y; vz[t] = e.z;
                    u[t] = 1.0f / (dist2_exact(e.x, e.y, e.z) + IDW_EPS);
                }
                if (nn == 0) u[t] = IDW_EPS;
                S += u[t];
            }
        }
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        float Y[3][4];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) Y[c][r] = 0.f;
        float Gx = 0.f, Gy = 0.f, Gz = 0.f, wsum = 0.f, cert = 0.f;
        float M[ORIENT ? 9 : 1] = {0.f};
        if (ORIENT) {
#pragma unroll
            for (int c = 0; c < (ORIENT ? 9 : 1); ++c) M[c] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < PIN_MAX_K; ++t) {
            if (idx[t] < 0) continue;
            const int id = idx[t] & ~PIN_NBR_QUIRK_BIT;
            const float wt = u[t] / S;
            const float cg = -2.f * u[t] * u[t];
            const float g0 = cg * vx[t], g1 = cg * vy[t], g2 = cg * vz[t];
            Gx += g0; Gy += g1; Gz += g2; wsum += wt;
            float y[4] = {0.f, 0.f, 0.f, 0.f};
            if (g < 2) {
                const float4 ft = reinterpret_cast<const float4*>(f.feats + (size_t)id * PIN_FEATURE_DIM)[g];
                y[0] = ft.x; y[1] = ft.y; y[2] = ft.z; y[3] = ft.w;
            } else if (g == 2) {
                float v[3], Rm[9];
                neighbor_vector(f, id, (idx[t] & PIN_NBR_QUIRK_BIT) != 0, vx[t], vy[t], vz[t], px, py, pz, v, Rm);
                y[0] = v[0]; y[1] = v[1]; y[2] = v[2];
                if constexpr (ORIENT) {  // d v_t / d q = R_t: accumulate w_t R_t^T
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) M[rr * 3 + cc] = fmaf(wt, Rm[cc * 3 + rr], M[rr * 3 + cc]);
                }
            } else if (f.certainty != nullptr) {
                cert = fmaf(f.certainty[id], wt, cert);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                z[r] = fmaf(wt, y[r], z[r]);
                Y[0][r] = fmaf(g0, y[r], Y[0][r]); Y[1][r] = fmaf(g1, y[r], Y[1][r]); Y[2][r] = fmaf(g2, y[r], Y[2][r]);
            }
        }
        if (!staged) {  // the first gather overlaps the weight staging of the block
            __syncthreads();
            staged = true;
            if (!work) break;
        }
        // ---- decoder on the matrix cores
        float a[4];
        const float x = Q::run(lds, f.levels, z, a);
        // ---- chain rule back to the query position (see eval_query)
        float cbar = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cbar = fmaf(a[r], z[r], cbar);
            ax = fmaf(Y[0][r], a[r], ax); ay = fmaf(Y[1][r], a[r], ay); az = fmaf(Y[2][r], a[r], az);
        }
        float dxs = 0.f, dys = 0.f, dzs = 0.f;  // the direct a_v term lives in lane g == 2
        if (g == 2) {
            if constexpr (ORIENT) {
                dxs = M[0] * a[0] + M[1] * a[1] + M[2] * a[2];
                dys = M[3] * a[0] + M[4] * a[1] + M[5] * a[2];
                dzs = M[6] * a[0] + M[7] * a[1] + M[8] * a[2];
            } else { dxs = a[0] * wsum; dys = a[1] * wsum; dzs = a[2] * wsum; }
        }
        cbar = quad_lanes_sum(cbar);
        ax = quad_lanes_sum(ax); ay = quad_lanes_sum(ay); az = quad_lanes_sum(az);
        dxs = quad_lanes_sum(dxs); dys = quad_lanes_sum(dys); dzs = quad_lanes_sum(dzs);
        const float invS = 1.0f / S;
        const float sdf = s * x;
        const float gx = s * (dxs + (ax - cbar * Gx) * invS);
        const float gy = s * (dys + (ay - cbar * Gy) * invS);
        const float gz = s * (dzs + (az - cbar * Gz) * invS);
        // ---- Gauss-Newton terms (tracker.py:409-524, 652-671).  All four lanes of a query hold the
        // result; each accumulates its quarter of the 31 sums (index i = 4j + g), no cross-lane work here.
        if (active) {
            if (g == 0) {
                if (sdf_out) sdf_out[qi] = sdf;
                if (grad_out) { grad_out[3 * qi] = gx; grad_out[3 * qi + 1] = gy; grad_out[3 * qi + 2] = gz; }
            }
            const float gn = sqrtf(gx * gx + gy * gy + gz * gz);
            const bool valid = nn >= gp.valid_nn_k && gn < gp.max_grad_norm && gn > gp.min_grad_norm && 0.f < gp.max_sdf_std;
            if (valid) {
                const float res = sdf - (labels ? labels[qi] : 0.f);
                float wgt = 1.f;
                if (gp.gm_grad > 0.f) { const float d = gn - 1.f; const float t = gp.gm_grad / (gp.gm_grad + d * d); wgt *= t * t; }
                if (gp.gm_dist > 0.f) { const float t = gp.gm_dist / (gp.gm_dist + res * res); wgt *= t * t; }
                float J[6];
                J[0] = py * gz - pz * gy; J[1] = pz * gx - px * gz; J[2] = px * gy - py * gx;
                J[3] = gx; J[4] = gy; J[5] = gz;
                float v[PIN_GN_NSUMS];
                int o = 0;
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = i; j < 6; ++j) v[o++] = wgt * J[i] * J[j];
#pragma unroll
                for (int i = 0; i < 6; ++i) v[21 + i] = wgt * J[i] * res;
                v[27] = wgt; v[28] = fabsf(res); v[29] = 1.f; v[30] = wgt * res * res; v[31] = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) tot[j] += g == 0 ? v[4 * j] : g == 1 ? v[4 * j + 1] : g == 2 ? v[4 * j + 2] : v[4 * j + 3];
            }
        }
    }
    __builtin_amdgcn_s_setprio(0);
    // wave: sum over the 16 queries of the row; lane (0, g) then holds sums 4j + g
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float t = row_sum_f32(tot[j]);
        if (nq == 0) red[wave][4 * j + g] = t;
    }
    // block reduction (16 waves -> one set of atomics; same-address f64 atomics serialise in L2)
    __syncthreads();
    if (threadIdx.x < PIN_GN_NSUMS) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < GQ_BLOCK / 64; ++w) t += (double)red[w][threadIdx.x];
        if (t != 0.0) atomicAdd(sums + (size_t)(blockIdx.x % GN_REPLICAS) * PIN_GN_NSUMS + threadIdx.x, t);
    }
}

// PIN_GN=wave keeps the 64-queries-per-wave kernel (A/B runs)
static inline bool use_quad_gn() {
    static const int on = [] {
        const char* e = getenv("PIN_GN");
        return (e != nullptr && strcmp(e, "wave") == 0) ? 0 : 1;
    }();
    return on != 0;
}

}  // namespace pin
