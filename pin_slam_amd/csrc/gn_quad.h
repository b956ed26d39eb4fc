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
// relative positions (and the after-PGO rotation); g = 3 only carries zero padding.
//
// Scheduling: persistent blocks of 16 waves (one per CU, 4 waves per SIMD, <= 128 VGPRs); the
// 16-query tiles are dealt round-robin to the SIMDs, so every SIMD gets the same MFMA work
// within one tile.  The weight image is staged once per block.
#pragma once
#include "brick.h"
#include "mlp_h2.h"

namespace pin {

// share of the IDW weight on the nearest neighbour from which a wave takes the pivoted gather (quad_gather_pass): the
// plain form's error is ~ 2^-24 rho |cbar| / |spread of c| of the gradient with rho = u_0 / (S - u_0); 0.98 <=> rho = 49
constexpr float QUAD_PIVOT_SHARE = 0.98f;
#ifndef PIN_AB_PIVOT2  // (A/B builds: the pivot of the colour variants' geometry rows -- see quad_gather_pass2)
#define PIN_AB_PIVOT2 false
#endif

// ---- decoder-phase building blocks (lane = query n + 16 * component group g) ---------------------------
template <bool ORIENT>
struct QuadIn {  // what the gather leaves in the registers of lane (n, g)
    float z[4];      // interpolated decoder input, components 4g..4g+3
    float zt[4];     // the same relative to the pivot row: z - y_pivot (PIVOT) or z itself; what the chain rule contracts with
    float Y[3][4];   // sum_t g_t (x) (y_t - y_pivot), the same components
    float Gx, Gy, Gz, wsum, S;
    float M[ORIENT ? 9 : 1];  // sum_t w_t R_t^T (after PGO), lane g == 2
};

// neighbour records -> IDW weights -> feature / position gather (neural_points.py:590-746)
//
// Straight-line: the kernel is bound by its vector-instruction count (PMC, r01: 27 of 54 us), and the r01 version of
// this function was 40 % of it -- per-neighbour branches on the validity / lane role (both sides execute in a wave
// that holds every lane role), two IEEE divisions per neighbour and 64-bit address arithmetic.  Now: the 8 record
// loads and the 8 row loads are each issued as one batch; an invalid neighbour keeps weight 0 and reads row 0; every
// lane loads a 16-byte half row (lanes g = 2, 3 re-read half 0, same cache line) and SELECTS its four input components
// (feature half / relative position / zero) instead of branching; 1 / (d2 + eps) is the hardware reciprocal
// (1 ulp) and w_t = u_t * (1 / S) with one division per query (the IDW weights are compared at 1e-4, not bit for
// bit, in this kernel).  After PGO (ORIENT) and for the rare flagged neighbours (non-local points, see PIN_NONLOCAL)
// the relative position comes from neighbor_vector as before.
//
// PIVOT (r06): the weight-derivative term of the chain rule is sum_t g_t (c_t - cbar) with c_t = a . y_t, cbar = a . z.  When
// one neighbour carries nearly all the weight (a query a millimetre from a neural point: u_0 ~ 1e6 against ~4e2 for the
// others) cbar is c_0 up to 1e-3..1e-4 of itself and g_0 ~ u_0^2: forming (sum_t g_t c_t) - cbar G, or even g_0 (c_0 - cbar),
// in fp32 loses those digits -- the reference's autograd and this kernel alike sat at 1e-4 of the gradient there (c5:
// 1.09e-4, the one comparison above the bar).  The sum does not change when every row is taken RELATIVE TO ONE OF THEM,
//     sum_t g_t (c_t - cbar) = a . Ytilde - (a . ztilde) G,   Ytilde = sum_t g_t (y_t - y_0),  ztilde = sum_t w_t (y_t - y_0),
// (sum_t w_t = 1), and with the pivot y_0 = the NEAREST neighbour's row (records are sorted by distance) the dominant term
// vanishes identically (y_0 - y_0), ztilde is a small number computed to full relative accuracy, and what is left does not
// cancel: a . Ytilde ~ sum_{t>0} u_t / d_t against (a . ztilde) G ~ u_0 / d_0 -- the second dominates by d_t / d_0.  The
// decoder input is rebuilt as z = wsum y_0 + ztilde.  Four subtractions per neighbour and lane: taken on the rare path only --
// a wave in which some query has u_0 > QUAD_PIVOT_SHARE S (quad_gather; ~1 % of the queries, wave-uniform: the same branch as
// `any_flag`, i.e. the GENERAL pass IS the pivoted one), and always after PGO.
template <bool ORIENT, bool GENERAL, bool PIVOT = GENERAL>
__device__ __forceinline__ void quad_gather_pass(const pin_field& f, const float4 (&e)[PIN_MAX_K], const float4 (&ft)[PIN_MAX_K],
                                                 const float (&u)[PIN_MAX_K], const int (&raw)[PIN_MAX_K], float S, float px,
                                                 float py, float pz, int g, QuadIn<ORIENT>& in) {
    const float invS = 1.0f / S;
    const bool is_feat = g < 2;
    const float mv = g == 2 ? 1.f : 0.f;
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    float Y[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[c][r] = 0.f;
    float Gx = 0.f, Gy = 0.f, Gz = 0.f, wsum = 0.f;
    float M[ORIENT ? 9 : 1] = {0.f};
    float y0[4] = {0.f, 0.f, 0.f, 0.f};  // (PIVOT) the nearest neighbour's four input components
#pragma unroll
    for (int t = 0; t < PIN_MAX_K; ++t) {
        const float wt = u[t] * invS;
        const float cg = -2.f * u[t] * u[t];
        const float g0 = cg * e[t].x, g1 = cg * e[t].y, g2 = cg * e[t].z;
        Gx += g0; Gy += g1; Gz += g2; wsum += wt;
        float v[3] = {e[t].x, e[t].y, e[t].z};
        if constexpr (GENERAL) {  // after PGO, or a flagged neighbour somewhere in the wave
            float Rm[9];
            if (raw[t] >= 0) {
                neighbor_vector(f, raw[t] & ~PIN_NBR_QUIRK_BIT, (raw[t] & PIN_NBR_QUIRK_BIT) != 0, e[t].x, e[t].y, e[t].z, px, py,
                                pz, v, Rm);
                if constexpr (ORIENT) {  // d v_t / d q = R_t: accumulate w_t R_t^T (used by lane g == 2)
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) M[rr * 3 + cc] = fmaf(wt, Rm[cc * 3 + rr], M[rr * 3 + cc]);
                }
            }
        }
        float y[4];
        y[0] = is_feat ? ft[t].x : mv * v[0];
        y[1] = is_feat ? ft[t].y : mv * v[1];
        y[2] = is_feat ? ft[t].z : mv * v[2];
        y[3] = is_feat ? ft[t].w : 0.f;
        if constexpr (PIVOT) {
            if (t == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) y0[r] = y[r];
                continue;  // (y_0 - y_0: the pivot's own terms vanish)
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] -= y0[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            z[r] = fmaf(wt, y[r], z[r]);
            Y[0][r] = fmaf(g0, y[r], Y[0][r]); Y[1][r] = fmaf(g1, y[r], Y[1][r]); Y[2][r] = fmaf(g2, y[r], Y[2][r]);
        }
    }
    in.Gx = Gx; in.Gy = Gy; in.Gz = Gz; in.wsum = wsum; in.S = S;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        in.zt[r] = z[r];
        in.z[r] = PIVOT ? fmaf(wsum, y0[r], z[r]) : z[r];
        in.Y[0][r] = Y[0][r]; in.Y[1][r] = Y[1][r]; in.Y[2][r] = Y[2][r];
    }
    if constexpr (ORIENT) {
#pragma unroll
        for (int c = 0; c < 9; ++c) in.M[c] = M[c];
    }
}

// The colour term's gather: BOTH tables in one sweep over the neighbours -- weights, gradient factors, relative position
// (and, after PGO, the rotation) computed once per neighbour, a neighbour's record and its two half rows dead as soon as
// it is done.  (Two calls of quad_gather_pass kept 96 registers of records and rows alive across the first one and, on
// the GENERAL path, went through neighbor_vector twice per neighbour: 256 registers + 76-232 B of scratch.)  The same
// arithmetic per accumulator in the same order: in / inc as two passes leave them.
// PIVOT (see quad_gather_pass) is written for the GEOMETRY rows here and is OFF: in these two-table variants the pivoted rare
// path costs the kernel 70 registers (214 -> 256 + 29 spilled at 1 x 64, measured with -DPIN_AB_PIVOT2=GENERAL), i.e. scratch
// memory in C5's registration kernel; their gradients are the plain form's (c5: 3.5e-5 on the tested points, the colour
// gradient is compared at 3e-4).  The single-table kernels -- every query kernel, the registration kernel without the colour
// term -- are pivoted.
template <bool ORIENT, bool GENERAL, bool PIVOT = PIN_AB_PIVOT2>
__device__ __forceinline__ void quad_gather_pass2(const pin_field& f, const float4 (&e)[PIN_MAX_K], const float4 (&ft)[PIN_MAX_K],
                                                  const float4 (&fc)[PIN_MAX_K], const float (&u)[PIN_MAX_K],
                                                  const int (&raw)[PIN_MAX_K], float S, float px, float py, float pz, int g,
                                                  QuadIn<ORIENT>& in, QuadIn<ORIENT>& inc) {
    const float invS = 1.0f / S;
    const bool is_feat = g < 2;
    const float mv = g == 2 ? 1.f : 0.f;
    float z[4] = {0.f, 0.f, 0.f, 0.f}, zc[4] = {0.f, 0.f, 0.f, 0.f};
    float Y[3][4], Yc[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) { Y[c][r] = 0.f; Yc[c][r] = 0.f; }
    float Gx = 0.f, Gy = 0.f, Gz = 0.f, wsum = 0.f;
    float M[ORIENT ? 9 : 1] = {0.f};
    float y0[4] = {0.f, 0.f, 0.f, 0.f};  // (PIVOT)
#pragma unroll
    for (int t = 0; t < PIN_MAX_K; ++t) {
        const float wt = u[t] * invS;
        const float cg = -2.f * u[t] * u[t];
        const float g0 = cg * e[t].x, g1 = cg * e[t].y, g2 = cg * e[t].z;
        Gx += g0; Gy += g1; Gz += g2; wsum += wt;
        float v[3] = {e[t].x, e[t].y, e[t].z};
        if constexpr (GENERAL) {
            float Rm[9];
            if (raw[t] >= 0) {
                neighbor_vector(f, raw[t] & ~PIN_NBR_QUIRK_BIT, (raw[t] & PIN_NBR_QUIRK_BIT) != 0, e[t].x, e[t].y, e[t].z, px, py,
                                pz, v, Rm);
                if constexpr (ORIENT) {
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) M[rr * 3 + cc] = fmaf(wt, Rm[cc * 3 + rr], M[rr * 3 + cc]);
                }
            }
        }
        float y[4], yc[4];
        y[0] = is_feat ? ft[t].x : mv * v[0];  yc[0] = is_feat ? fc[t].x : mv * v[0];
        y[1] = is_feat ? ft[t].y : mv * v[1];  yc[1] = is_feat ? fc[t].y : mv * v[1];
        y[2] = is_feat ? ft[t].z : mv * v[2];  yc[2] = is_feat ? fc[t].z : mv * v[2];
        y[3] = is_feat ? ft[t].w : 0.f;        yc[3] = is_feat ? fc[t].w : 0.f;
        if constexpr (PIVOT) {
            if (t == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { y0[r] = y[r]; y[r] = 0.f; }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] -= y0[r];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            z[r] = fmaf(wt, y[r], z[r]);
            Y[0][r] = fmaf(g0, y[r], Y[0][r]); Y[1][r] = fmaf(g1, y[r], Y[1][r]); Y[2][r] = fmaf(g2, y[r], Y[2][r]);
            zc[r] = fmaf(wt, yc[r], zc[r]);
            Yc[0][r] = fmaf(g0, yc[r], Yc[0][r]); Yc[1][r] = fmaf(g1, yc[r], Yc[1][r]); Yc[2][r] = fmaf(g2, yc[r], Yc[2][r]);
        }
    }
    in.Gx = Gx; in.Gy = Gy; in.Gz = Gz; in.wsum = wsum; in.S = S;
    inc.Gx = Gx; inc.Gy = Gy; inc.Gz = Gz; inc.wsum = wsum; inc.S = S;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        in.zt[r] = z[r]; in.z[r] = PIVOT ? fmaf(wsum, y0[r], z[r]) : z[r];
        in.Y[0][r] = Y[0][r]; in.Y[1][r] = Y[1][r]; in.Y[2][r] = Y[2][r];
        inc.z[r] = zc[r]; inc.zt[r] = zc[r]; inc.Y[0][r] = Yc[0][r]; inc.Y[1][r] = Yc[1][r]; inc.Y[2][r] = Yc[2][r];
    }
    if constexpr (ORIENT) {
#pragma unroll
        for (int c = 0; c < 9; ++c) { in.M[c] = M[c]; inc.M[c] = M[c]; }
    }
}

// COLOR: the same neighbours and weights over a second feature table (the colour features, `feats_c`) -> inc
// PIV = false: the one instantiation that cannot afford the pivot row's four registers (after PGO with a 3 x 64 decoder: 255
// registers without it)
template <bool ORIENT, bool COLOR = false, bool PIV = true>
__device__ __forceinline__ void quad_gather(const pin_field& f, const float4* __restrict__ rp, int kk, int nn, float px, float py,
                                            float pz, int g, QuadIn<ORIENT>& in, const float* __restrict__ feats_c = nullptr,
                                            QuadIn<ORIENT>* inc = nullptr) {
    float4 e[PIN_MAX_K];
#pragma unroll
    for (int t = 0; t < PIN_MAX_K; ++t) e[t] = rp[t < kk ? t : 0];
    const float4* __restrict__ rows = reinterpret_cast<const float4*>(f.feats) + (g & 1);
    float u[PIN_MAX_K];
    float4 ft[PIN_MAX_K];
    float4 fc[COLOR ? PIN_MAX_K : 1];  // (all row loads of a tile are issued together)
    int raw[PIN_MAX_K];
    float S = 0.f;
    bool any_flag = false;
#pragma unroll
    for (int t = 0; t < PIN_MAX_K; ++t) {
        raw[t] = __float_as_int(e[t].w);
        const bool val = t < kk && raw[t] >= 0;
        const int id = val ? (raw[t] & ~PIN_NBR_QUIRK_BIT) : 0;
        ft[t] = rows[2 * (size_t)(unsigned int)id];
        if constexpr (COLOR) fc[t] = (reinterpret_cast<const float4*>(feats_c) + (g & 1))[2 * (size_t)(unsigned int)id];
        const float ut = val ? __builtin_amdgcn_rcpf(dist2_exact(e[t].x, e[t].y, e[t].z) + IDW_EPS) : 0.f;
        u[t] = ut;  // (an invalid neighbour contributes nothing)
        S += (nn == 0 && t < kk) ? IDW_EPS : ut;  // no neighbour at all: S = k * eps, as the reference's weights then are
        any_flag = any_flag || (val && (raw[t] & PIN_NBR_QUIRK_BIT) != 0);
        raw[t] = val ? raw[t] : -1;
    }
    // (PIVOT, see quad_gather_pass) some query of the wave has nearly all its weight on its nearest neighbour: such a wave
    // takes the GENERAL pass, which is the pivoted one -- ONE rare variant beside the straight-line one
    const bool rare = __builtin_amdgcn_ballot_w64(any_flag || u[0] > QUAD_PIVOT_SHARE * S) != 0ull;
    if constexpr (COLOR) {
        if constexpr (ORIENT) {
            quad_gather_pass2<true, true, false>(f, e, ft, fc, u, raw, S, px, py, pz, g, in, *inc);
        } else {
            if (rare) quad_gather_pass2<false, true>(f, e, ft, fc, u, raw, S, px, py, pz, g, in, *inc);
            else quad_gather_pass2<false, false, false>(f, e, ft, fc, u, raw, S, px, py, pz, g, in, *inc);
        }
    } else if constexpr (ORIENT) {
        quad_gather_pass<true, true, PIV>(f, e, ft, u, raw, S, px, py, pz, g, in);
    } else {
        if (rare) quad_gather_pass<false, true>(f, e, ft, u, raw, S, px, py, pz, g, in);
        else quad_gather_pass<false, false>(f, e, ft, u, raw, S, px, py, pz, g, in);
    }
}

// chain rule from the decoder's input Jacobian a (d value / d z, this lane's four components) back to the query position
// (see eval_query): value gradient = scale * (direct term through the relative positions + (sum_t Y_t a - (a . z) G) / S)
// SWAP: the seven sums over the query's four lanes through the row swaps (rows_sum) instead of the LDS crossbar -- the
// colour variants, which run at the register limit, take it (no lane-index registers); the plain kernels are bound by their
// vector-ALU issue and keep the permutes (pin_common.h)
template <bool ORIENT, bool SWAP = false>
__device__ __forceinline__ void quad_chain(const QuadIn<ORIENT>& in, const float (&a)[4], int g, float scale, float& gx, float& gy,
                                           float& gz) {
    const float (&z)[4] = in.zt;  // (relative to the gather's pivot row, as Y is: see quad_gather_pass)
    const float (&Y)[3][4] = in.Y;
    const float (&M)[ORIENT ? 9 : 1] = in.M;
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
        } else { dxs = a[0] * in.wsum; dys = a[1] * in.wsum; dzs = a[2] * in.wsum; }
    }
    if constexpr (SWAP) {
        cbar = rows_sum(cbar);
        ax = rows_sum(ax); ay = rows_sum(ay); az = rows_sum(az);
        dxs = rows_sum(dxs); dys = rows_sum(dys); dzs = rows_sum(dzs);
    } else {
        cbar = quad_lanes_sum(cbar);
        ax = quad_lanes_sum(ax); ay = quad_lanes_sum(ay); az = quad_lanes_sum(az);
        dxs = quad_lanes_sum(dxs); dys = quad_lanes_sum(dys); dzs = quad_lanes_sum(dzs);
    }
    const float invS = 1.0f / in.S;
    gx = scale * (dxs + (ax - cbar * in.Gx) * invS);
    gy = scale * (dys + (ay - cbar * in.Gy) * invS);
    gz = scale * (dzs + (az - cbar * in.Gz) * invS);
}

// decoder on the matrix cores -> chain rule -> Gauss-Newton terms of the tile; tot[j] += sum 4j + g.
// COLOR: the colour term of the registration (tracker.py:493-542, 699-744) from the colour decoder's image `lds_c`
// over the colour inputs `inc`: intensity = 0.299 R + 0.587 G + 0.114 B of the regressed colour (tools.py:408) against
// the measured one -- a consistency weight exp(-|dI|) (mode 1) or the photometric rows J_c = [p x dI, dI] (mode 2).
template <int H, bool ORIENT, bool SPLIT = false, int LC = 0, bool COLOR = false>
__device__ __forceinline__ void quad_finish(const pin_field& f, const pin_gn_params& gp, const unsigned char* __restrict__ lds,
                                            const QuadIn<ORIENT>& in, int nn, float px, float py, float pz, bool active, int qi,
                                            int g, const float* __restrict__ labels, float* __restrict__ sdf_out,
                                            float* __restrict__ grad_out, float (&tot)[8],
                                            const unsigned char* __restrict__ lds_c = nullptr,
                                            const QuadIn<ORIENT>* inc = nullptr, const ColorTerm* ct = nullptr) {
    using Q = QuadDec<H, SPLIT>;
    const float s = f.sdf_scale;
    // ---- decoder on the matrix cores
    float a[4];
    const float x = Q::template run<LC>(lds, f.levels, in.z, a);
    const float sdf = s * x;
    float gx, gy, gz;
    quad_chain<ORIENT, COLOR>(in, a, g, s, gx, gy, gz);
    float ipred = 0.f, igx = 0.f, igy = 0.f, igz = 0.f;
    if constexpr (COLOR) {
        const float kappa[3] = {0.299f, 0.587f, 0.114f};
        float ac[4];
        ipred = QuadDecoderH<H>::template run_color<LC>(lds_c, inc->z, kappa, ct->mode == 2, ac);
        if (ct->mode == 2) quad_chain<ORIENT, true>(*inc, ac, g, 1.0f, igx, igy, igz);
    }
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
            const float res = (gp.dist_div_grad_norm ? sdf / gn : sdf) - (labels ? labels[qi] : 0.f);
            float wgt = 1.f;
            if (gp.gm_grad > 0.f) { const float d = gn - 1.f; const float t = gp.gm_grad / (gp.gm_grad + d * d); wgt *= t * t; }
            if (gp.gm_dist > 0.f) { const float t = gp.gm_dist / (gp.gm_dist + res * res); wgt *= t * t; }
            float cres = 0.f;
            if constexpr (COLOR) {
                const float* cm = ct->colors + 3 * (size_t)qi;
                const float imeas = 0.299f * cm[0] + 0.587f * cm[1] + 0.114f * cm[2];
                cres = ipred - imeas;
                if (ct->mode == 1) wgt *= expf(-fabsf(cres));  // consistency weight (tracker.py:509-514)
            }
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
            if constexpr (COLOR) {
                if (ct->mode == 2) {  // implicit_color_reg (tracker.py:699-744): + w_photo * Jc^T W Jc, Jc^T W rc
                    float Jc[6];
                    Jc[0] = py * igz - pz * igy; Jc[1] = pz * igx - px * igz; Jc[2] = px * igy - py * igx;
                    Jc[3] = igx; Jc[4] = igy; Jc[5] = igz;
                    const float wp = wgt * ct->photo_weight;
                    o = 0;
#pragma unroll
                    for (int i = 0; i < 6; ++i)
#pragma unroll
                        for (int j = i; j < 6; ++j) v[o++] += wp * Jc[i] * Jc[j];
#pragma unroll
                    for (int i = 0; i < 6; ++i) v[21 + i] += wp * Jc[i] * cres;
                    v[31] = fabsf(cres);
                }
            }
            // lane g keeps sums 4j + g.  Written as masked FMAs: a select chain over v[] is turned into a
            // dynamically indexed private array (scratch memory) by the compiler
            int gm = g;
            if constexpr (COLOR) asm volatile("" : "+v"(gm));  // (the colour variants run at the register limit: the four masks are rebuilt per tile there instead of living across the loop)
            const float m0 = gm == 0 ? 1.f : 0.f, m1 = gm == 1 ? 1.f : 0.f, m2 = gm == 2 ? 1.f : 0.f, m3 = gm == 3 ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                tot[j] += fmaf(m0, v[4 * j], fmaf(m1, v[4 * j + 1], fmaf(m2, v[4 * j + 2], m3 * v[4 * j + 3])));
        }
    }
}

__host__ __device__ constexpr int gq_red_offset(int image_bytes) { return (image_bytes + 15) & ~15; }
__host__ __device__ constexpr int gq_lds_bytes(int image_bytes, int block = GQ_BLOCK) {
    return gq_red_offset(image_bytes) + (block / 64) * PIN_GN_NSUMS * (int)sizeof(float);
}

// LC: the number of H-wide layers when the split-fp16 decoder is used (compile time: both sweeps unrolled);
// 0 with the fp32 image, which reads it from the field
// COLOR: + the colour term (a second image of LC layers and 3 heads behind the first, staged per block from ct.fc.dec)
// BLK: threads per (persistent, one-per-CU) block = 4 SIMDs x BLK / 256 waves; 768 = three waves per SIMD, which caps
// the kernel at 168 registers
template <int H, bool ORIENT, bool SPLIT, int LC, bool COLOR = false, int BLK = GQ_BLOCK>
__global__ __launch_bounds__(BLK, 1) void gn_accumulate_quad_kernel(pin_field f, pin_gn_params gp,
                                                                         const float* __restrict__ query,
                                                                         const float4* __restrict__ nbr,
                                                                         const int* __restrict__ nn_count,
                                                                         const float* __restrict__ labels, int n_q,
                                                                         double* __restrict__ sums, float* __restrict__ sdf_out,
                                                                         float* __restrict__ grad_out,
                                                                         const double* state, ColorTerm ct, int tail_on) {
    using Q = QuadDec<H, SPLIT>;
    static_assert(!COLOR || (SPLIT && LC >= 1), "the colour term runs on the split-fp16 images");
    extern __shared__ __attribute__((aligned(16))) unsigned char gq_smem[];  // decoder image(s), then the block reduction
    unsigned char* const lds = gq_smem;
    unsigned char* const lds_c = lds + gq_red_offset(Q::bytes(f.levels));  // (COLOR)
    float (*red)[PIN_GN_NSUMS] = reinterpret_cast<float (*)[PIN_GN_NSUMS]>(lds + (COLOR ? 2 : 1) * gq_red_offset(Q::bytes(f.levels)));
    if (state != nullptr && state[PIN_GN_STATE_DONE] != 0.0) return;
    // weights: copy the image staged once per registration (pin_stage_decoder) or split them here; either way the
    // image is visible after the barrier that follows the first gather.  COLOR: both images are staged by the caller
    // (the launcher takes another kernel otherwise), two linear copies
    if constexpr (COLOR) {
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(f.dec_image);
        const uint4* __restrict__ src_c = reinterpret_cast<const uint4*>(ct.fc.dec_image);
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(lds);
        uint4* __restrict__ dst_c = reinterpret_cast<uint4*>(lds_c);
        constexpr int n16 = Q::bytes(LC) >> 4;
        for (int i = threadIdx.x; i < n16; i += BLK) { dst[i] = src[i]; dst_c[i] = src_c[i]; }
    } else if (SPLIT && f.dec_image != nullptr && f.dec_image_bytes == Q::bytes(f.levels)) {
        // all loads of the copy in flight at once (the image size is a compile-time constant with the split decoder):
        // one memory round trip instead of one per unrolled chunk
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(f.dec_image);
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(lds);
        constexpr int N16 = Q::bytes(LC > 0 ? LC : 1) >> 4, TRIPS = (N16 + BLK - 1) / BLK;
        uint4 v[TRIPS];
#pragma unroll
        for (int it = 0; it < TRIPS; ++it) {
            const int i = it * BLK + threadIdx.x;
            v[it] = src[i < N16 ? i : 0];
        }
#pragma unroll
        for (int it = 0; it < TRIPS; ++it) {
            const int i = it * BLK + threadIdx.x;
            if (i < N16) dst[i] = v[it];
        }
    } else {
        Q::stage(f.dec, f.levels, lds, threadIdx.x, BLK);
    }
    const int lane = threadIdx.x & 63, nq = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const int n_tiles = (n_q + 15) >> 4;
    const int n_simd = gridDim.x * 4;
    const int simd = blockIdx.x * 4 + (wave & 3);
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
    // XCD-aware tile order (brick.h, xcd_logical_block): with a grid of a multiple of 8 blocks, XCD x = blockIdx.x & 7 takes the
    // x-th eighth of the (Morton-ordered) queries -- the range whose records the search kernel's blocks on that XCD have just
    // written -- and deals it out over its own SIMDs; otherwise the tiles are dealt out over all SIMDs as before
    int tile0 = simd + n_simd * (wave >> 2), tile_step = n_simd * (BLK / 256), tile_end = n_tiles;
    if ((gridDim.x & 7) == 0 && g_xcd_on) {
        const int per_xcd = 2 * ((((n_q + 31) >> 5) + 7) >> 3);  // tiles of 8-lanes-per-query search blocks (32 queries = 2 tiles)
        const int x = blockIdx.x & 7, ls = (blockIdx.x >> 3) * 4 + (wave & 3), ns = (gridDim.x >> 3) * 4;
        tile0 = x * per_xcd + ls + ns * (wave >> 2);
        tile_step = ns * (BLK / 256);
        tile_end = min(n_tiles, (x + 1) * per_xcd);
    }
    for (int tile = tile0;; tile += tile_step) {
        const bool work = tile < tile_end;
        if (!work && staged) break;
        const int qi = (work ? tile : 0) * 16 + nq;
        const bool active = qi < n_q;
        const int qq = active ? qi : n_q - 1;
        const float px = query[3 * qq], py = query[3 * qq + 1], pz = query[3 * qq + 2];
        const int nn = nn_count[qq];
        QuadIn<ORIENT> in;
        QuadIn<ORIENT> inc;  // (COLOR)
        if constexpr (COLOR) quad_gather<ORIENT, true>(f, nbr + (size_t)qq * f.k, f.k, nn, px, py, pz, g, in, ct.fc.feats, &inc);
        else quad_gather<ORIENT, false, !(ORIENT && H == 64 && LC == 3)>(f, nbr + (size_t)qq * f.k, f.k, nn, px, py, pz, g, in);
        if (!staged) {  // the first gather overlaps the weight staging of the block
            __syncthreads();
            staged = true;
            if (!work) break;
        }
        if constexpr (COLOR)
            quad_finish<H, ORIENT, SPLIT, LC, true>(f, gp, lds, in, nn, px, py, pz, active, qi, g, labels, sdf_out, grad_out, tot,
                                                    lds_c, &inc, &ct);
        else
            quad_finish<H, ORIENT, SPLIT, LC>(f, gp, lds, in, nn, px, py, pz, active, qi, g, labels, sdf_out, grad_out, tot);
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
        for (int w = 0; w < BLK / 64; ++w) t += (double)red[w][threadIdx.x];
        if (t != 0.0) atomicAdd(sums + (size_t)(blockIdx.x % GN_REPLICAS) * PIN_GN_NSUMS + threadIdx.x, t);
    }
    // pin_gn_accumulate_solve: the block whose atomics land last solves the normal equations and moves the loop state (gn_solve.h)
    // (not in the colour variants: they sit at the register limit, and four more live values put one of them into scratch memory)
    if constexpr (!COLOR) {
        if (tail_on && wave == 0) gn_tail_last_block(const_cast<double*>(state), sums);
    }
}



// ---- per-neighbour decoding (weighted_first = False: run_kitti.yaml and eight more shipped configs) ---------------
// The decoder runs once per NEIGHBOUR (k times the work of the interpolate-first mode) and the spread of the k
// predictions gates the registration (tracker.py:317-328).  In the quad layout a decoder column is a (query, neighbour)
// pair: a 16-column tile holds 2 queries x 8 neighbour slots, lane (n, g) = column n = 8 * (query of the tile) + t,
// component group g.  No interpolation before the decoder: the lane's input components ARE its neighbour's feature
// half-row / relative position.  Everything that mixes the neighbours of a query -- IDW normalisation, the weighted
// mean / spread of the predictions, the gradient terms -- is a sum over the 8 consecutive lanes of the query inside a
// DPP row (three DPP steps, no LDS).  The 32 lanes of a query hold its result; lane (t, g) keeps Gauss-Newton sum
// 4 t + g, picked by a per-lane selector that is built once.

constexpr int NWF_BLOCK = 512;  // 2 waves per SIMD (up to 256 VGPRs: the two-deep prefetch state needs ~190)

// MODE 0: the Gauss-Newton sums (pin_gn_accumulate).  MODE 1 / 2: pin_sdf_query on the same tiles (Tracker.query_source_points,
// Mesher.query_points with weighted_first = False) -- per query the weighted mean of the k predictions, their spread
// (`std_out`, tracker.py:317-322), the interpolated certainty (`cert_out`, neural_points.py:726-729; lanes g == 3) and, MODE 1,
// the gradient; MODE 2 runs the forward sweep only.  No sums, no loop state.
template <int H, bool ORIENT, bool SPLIT, int LC, int MODE = 0>
__global__ __launch_bounds__(NWF_BLOCK, 1) void gn_accumulate_quad_nwf_kernel(pin_field f, pin_gn_params gp,
                                                                             const float* __restrict__ query,
                                                                             const float4* __restrict__ nbr,
                                                                             const int* __restrict__ nn_count,
                                                                             const float* __restrict__ labels, int n_q,
                                                                             double* __restrict__ sums, float* __restrict__ sdf_out,
                                                                             float* __restrict__ grad_out,
                                                                             const double* state,
                                                                             float* __restrict__ std_out = nullptr,
                                                                             float* __restrict__ cert_out = nullptr, int tail_on = 0) {
    static_assert(MODE == 0 || SPLIT, "the query modes run on the split-fp16 image");
    using Q = QuadDec<H, SPLIT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char gq_smem[];
    unsigned char* const lds = gq_smem;
    float (*red)[PIN_GN_NSUMS] = reinterpret_cast<float (*)[PIN_GN_NSUMS]>(lds + gq_red_offset(Q::bytes(f.levels)));  // [NWF_BLOCK / 64]
    if (state != nullptr && state[PIN_GN_STATE_DONE] != 0.0) return;
    if (SPLIT && f.dec_image != nullptr && f.dec_image_bytes == Q::bytes(f.levels)) {
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(f.dec_image);
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(lds);
        const int n16 = f.dec_image_bytes >> 4;
#pragma unroll 4
        for (int i = threadIdx.x; i < n16; i += NWF_BLOCK) dst[i] = src[i];
    } else {
        Q::stage(f.dec, f.levels, lds, threadIdx.x, NWF_BLOCK);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, nq = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const int q2 = nq >> 3, t = nq & 7;
    const int kk = f.k;
    const float s = f.sdf_scale;
    // which Gauss-Newton sum this lane keeps: i = 4 t + g.  v_i = W * A * B with (tracker.py:652-671)
    //   i < 21: J_a J_b (upper triangle, row-major), W = w      21..26: J_a * res, W = w      27: w      28: |res|
    //   29: 1 (count)      30: w res^2      31: unused
    const int si = 4 * t + g;
    float ea[6], eb[6];
    float ea_res = 0.f, ea_one = 0.f, eb_res = 0.f, eb_abs = 0.f, eb_one = 0.f, wsel = 0.f;
    {
        int a = -1, b = -1;
        if (si < 21) {
            int o = 0;
            for (int i = 0; i < 6; ++i)
                for (int j = i; j < 6; ++j, ++o)
                    if (o == si) { a = i; b = j; }
            wsel = 1.f;
        } else if (si < 27) { a = si - 21; eb_res = 1.f; wsel = 1.f; }
        else if (si == 27) { ea_one = 1.f; eb_one = 1.f; wsel = 1.f; }
        else if (si == 28) { ea_one = 1.f; eb_abs = 1.f; }
        else if (si == 29) { ea_one = 1.f; eb_one = 1.f; }
        else if (si == 30) { ea_res = 1.f; eb_res = 1.f; wsel = 1.f; }
#pragma unroll
        for (int j = 0; j < 6; ++j) { ea[j] = a == j ? 1.f : 0.f; eb[j] = b == j ? 1.f : 0.f; }
    }
    const float4* __restrict__ rows = reinterpret_cast<const float4*>(f.feats) + (g & 1);
    const bool is_feat = g < 2;
    const float mv = g == 2 ? 1.f : 0.f;
    const int n_tiles = (n_q + 1) >> 1;
    const int n_simd = gridDim.x * 4;
    const int simd = blockIdx.x * 4 + (wave & 3);
    float tot = 0.f;
    // A tile is two dependent memory round trips (record -> feature row) followed by ~0.2 us of arithmetic: the loop is
    // software-pipelined two tiles deep (records of tile i + 2 and the feature rows of tile i + 1 are in flight while
    // tile i is decoded), otherwise the kernel is bound by memory latency (measured: 58.7 us -> see DESIGN).
    const int stride = n_simd * (NWF_BLOCK / 256);
    struct Head { float px, py, pz; int nn, qi; float4 e; bool active; };
    auto fetch_head = [&](int tile, Head& h) {
        h.qi = tile * 2 + q2;
        h.active = tile < n_tiles && h.qi < n_q;
        const int qq = h.active ? h.qi : n_q - 1;
        h.px = query[3 * qq]; h.py = query[3 * qq + 1]; h.pz = query[3 * qq + 2];
        h.nn = nn_count[qq];
        h.e = nbr[(size_t)qq * kk + (t < kk ? t : 0)];
    };
    auto row_of = [&](const Head& h) -> float4 {
        const int raw = __float_as_int(h.e.w);
        const int id = (t < kk && raw >= 0) ? (raw & ~PIN_NBR_QUIRK_BIT) : 0;
        return rows[2 * (size_t)(unsigned int)id];
    };
    const int tile0 = simd + n_simd * (wave >> 2);
    Head h0, h1, h2;
    fetch_head(tile0, h0);
    fetch_head(tile0 + stride, h1);
    float4 ft0 = row_of(h0), ft1;
    for (int tile = tile0; tile < n_tiles; tile += stride) {
        fetch_head(tile + 2 * stride, h2);
        ft1 = row_of(h1);
        const int qi = h0.qi;
        const bool active = h0.active;
        const float px = h0.px, py = h0.py, pz = h0.pz;
        const int nn = h0.nn;
        const float4 e = h0.e;
        const float4 ft = ft0;
        h0 = h1; h1 = h2; ft0 = ft1;
        const int raw = __float_as_int(e.w);
        const bool val = t < kk && raw >= 0;
        const int id = val ? (raw & ~PIN_NBR_QUIRK_BIT) : 0;
        const float u = val ? __builtin_amdgcn_rcpf(dist2_exact(e.x, e.y, e.z) + IDW_EPS) : 0.f;
        const float S = octet_sum((nn == 0 && t < kk) ? IDW_EPS : u);
        const float invS = 1.0f / S;
        const float wt = u * invS;
        float v[3] = {e.x, e.y, e.z};
        float Rm[9];
        const bool flagged = val && (raw & PIN_NBR_QUIRK_BIT) != 0;
        if (ORIENT || __builtin_amdgcn_ballot_w64(flagged) != 0ull) {  // after PGO / a flagged neighbour in the wave (rare)
            if (val) neighbor_vector(f, id, flagged, e.x, e.y, e.z, px, py, pz, v, Rm);
        }
        float z[4], a[4];
        z[0] = is_feat ? ft.x : mv * v[0];
        z[1] = is_feat ? ft.y : mv * v[1];
        z[2] = is_feat ? ft.z : mv * v[2];
        z[3] = is_feat ? ft.w : 0.f;
        float x;  // this neighbour's prediction (a: d x / d its input)
        if constexpr (MODE == 2) {
            float xo[1];
            QuadDecoderH<H>::template forward<LC, 1>(lds, z, xo);
            x = xo[0];
            a[0] = a[1] = a[2] = a[3] = 0.f;
        } else {
            x = Q::template run<LC>(lds, f.levels, z, a);
        }
        // ---- across the neighbours of the query (eval_query, weighted_first = False)
        const float st = s * x;
        const float mean = octet_sum(wt * st);
        const float dv = st - mean;
        const float sd = sqrtf(octet_sum(wt * dv * dv));  // tracker.py:317-322
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;  // w_t * d x_t / d q through the relative position (lane g == 2)
        if (g == 2) {
            if constexpr (ORIENT) {
                if (val) {
                    d0 = wt * (Rm[0] * a[0] + Rm[3] * a[1] + Rm[6] * a[2]);
                    d1 = wt * (Rm[1] * a[0] + Rm[4] * a[1] + Rm[7] * a[2]);
                    d2 = wt * (Rm[2] * a[0] + Rm[5] * a[1] + Rm[8] * a[2]);
                }
            } else { d0 = wt * a[0]; d1 = wt * a[1]; d2 = wt * a[2]; }
        }
        d0 = quad_lanes_sum(octet_sum(d0)); d1 = quad_lanes_sum(octet_sum(d1)); d2 = quad_lanes_sum(octet_sum(d2));
        const float cg = -2.f * u * u;  // d u_t / d q = cg * (q - P_t); through the weights: sum_t (s_t - mean) d w_t
        // ... with every prediction taken relative to the NEAREST neighbour's (t = 0): sum_t g_t (s_t - mean) =
        // sum_t g_t (s_t - s_0) - (sum_t w_t (s_t - s_0)) G, whose dominant term vanishes identically when one neighbour
        // carries nearly all the weight (quad_gather_pass, PIVOT: the plain form cancels to 1e-4 of the gradient there)
        const float sp = st - octet_sum(t == 0 ? st : 0.f);
        const float cs = cg * sp;
        const float mt = octet_sum(wt * sp);
        const float ax = octet_sum(cs * e.x), ay = octet_sum(cs * e.y), az = octet_sum(cs * e.z);
        const float Gx = octet_sum(cg * e.x), Gy = octet_sum(cg * e.y), Gz = octet_sum(cg * e.z);
        const float gx = s * d0 + (ax - mt * Gx) * invS;
        const float gy = s * d1 + (ay - mt * Gy) * invS;
        const float gz = s * d2 + (az - mt * Gz) * invS;
        if constexpr (MODE != 0) {
            float cert = 0.f;
            if (cert_out != nullptr && f.certainty != nullptr)  // (uniform)
                cert = octet_sum(wt * ((g == 3 && val) ? f.certainty[id] : 0.f));
            if (active && t == 0) {
                if (g == 0) {
                    if (sdf_out) sdf_out[qi] = mean;
                    if (MODE == 1 && grad_out) { grad_out[3 * qi] = gx; grad_out[3 * qi + 1] = gy; grad_out[3 * qi + 2] = gz; }
                } else if (g == 1) {
                    if (std_out) std_out[qi] = sd;
                } else if (g == 3) {
                    if (cert_out) cert_out[qi] = cert;
                }
            }
            continue;
        }
        if (active) {
            if (t == 0 && g == 0) {
                if (sdf_out) sdf_out[qi] = mean;
                if (grad_out) { grad_out[3 * qi] = gx; grad_out[3 * qi + 1] = gy; grad_out[3 * qi + 2] = gz; }
            }
            const float gn = sqrtf(gx * gx + gy * gy + gz * gz);
            const bool valid = nn >= gp.valid_nn_k && gn < gp.max_grad_norm && gn > gp.min_grad_norm && sd < gp.max_sdf_std;
            if (valid) {
                const float res = (gp.dist_div_grad_norm ? mean / gn : mean) - (labels ? labels[qi] : 0.f);
                float wgt = 1.f;
                if (gp.gm_grad > 0.f) { const float d = gn - 1.f; const float tt = gp.gm_grad / (gp.gm_grad + d * d); wgt *= tt * tt; }
                if (gp.gm_dist > 0.f) { const float tt = gp.gm_dist / (gp.gm_dist + res * res); wgt *= tt * tt; }
                float J[6];
                J[0] = py * gz - pz * gy; J[1] = pz * gx - px * gz; J[2] = px * gy - py * gx;
                J[3] = gx; J[4] = gy; J[5] = gz;
                float A = fmaf(ea_res, res, ea_one), B = fmaf(eb_res, res, fmaf(eb_abs, fabsf(res), eb_one));
#pragma unroll
                for (int j = 0; j < 6; ++j) { A = fmaf(ea[j], J[j], A); B = fmaf(eb[j], J[j], B); }
                tot += (wsel != 0.f ? wgt : 1.f) * A * B;
            }
        }
    }
    if constexpr (MODE != 0) return;
    // the two queries of a tile sit in lanes n and n ^ 8: add them, lane (q2 = 0, t, g) then holds sum 4 t + g of the wave
    tot += dpp_mov<0x128>(tot);  // row_ror:8
    if (q2 == 0) red[wave][si] = tot;
    __syncthreads();
    if (threadIdx.x < PIN_GN_NSUMS) {
        double tt = 0.0;
#pragma unroll
        for (int w = 0; w < NWF_BLOCK / 64; ++w) tt += (double)red[w][threadIdx.x];
        if (tt != 0.0) atomicAdd(sums + (size_t)(blockIdx.x % GN_REPLICAS) * PIN_GN_NSUMS + threadIdx.x, tt);
    }
    if (tail_on && threadIdx.x < 64) gn_tail_last_block(const_cast<double*>(state), sums);
}

}  // namespace pin
