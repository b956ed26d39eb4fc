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
#include "mlp_bf3.h"

namespace pin {

// ---- decoder-phase building blocks (lane = query n + 16 * component group g) ---------------------------
template <bool ORIENT>
struct QuadIn {  // what the gather leaves in the registers of lane (n, g)
    float z[4];      // interpolated decoder input, components 4g..4g+3
    float Y[3][4];   // sum_t g_t (x) y_t, the same components
    float Gx, Gy, Gz, wsum, S;
    float M[ORIENT ? 9 : 1];  // sum_t w_t R_t^T (after PGO), lane g == 2
};

// neighbour records -> IDW weights -> feature / position gather (neural_points.py:590-746)
template <bool ORIENT>
__device__ __forceinline__ void quad_gather(const pin_field& f, const float4* __restrict__ rp, int kk, int nn, float px, float py,
                                            float pz, int g, QuadIn<ORIENT>& in) {
    // ---- neighbour records and IDW weights (all four lanes of the query; neural_points.py:660-683)
    float vx[PIN_MAX_K], vy[PIN_MAX_K], vz[PIN_MAX_K], u[PIN_MAX_K];
    int idx[PIN_MAX_K];
    float S = 0.f;
#pragma unroll
    for (int t = 0; t < PIN_MAX_K; ++t) {
        idx[t] = -1; u[t] = 0.f; vx[t] = vy[t] = vz[t] = 0.f;
        if (t < kk) {
            const float4 e = rp[t];
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
    float Gx = 0.f, Gy = 0.f, Gz = 0.f, wsum = 0.f;
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
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            z[r] = fmaf(wt, y[r], z[r]);
            Y[0][r] = fmaf(g0, y[r], Y[0][r]); Y[1][r] = fmaf(g1, y[r], Y[1][r]); Y[2][r] = fmaf(g2, y[r], Y[2][r]);
        }
    }
    in.Gx = Gx; in.Gy = Gy; in.Gz = Gz; in.wsum = wsum; in.S = S;
#pragma unroll
    for (int r = 0; r < 4; ++r) { in.z[r] = z[r]; in.Y[0][r] = Y[0][r]; in.Y[1][r] = Y[1][r]; in.Y[2][r] = Y[2][r]; }
    if constexpr (ORIENT) {
#pragma unroll
        for (int c = 0; c < 9; ++c) in.M[c] = M[c];
    }
}

// decoder on the matrix cores -> chain rule -> Gauss-Newton terms of the tile; tot[j] += sum 4j + g
template <int H, bool ORIENT, bool BF = false>
__device__ __forceinline__ void quad_finish(const pin_field& f, const pin_gn_params& gp, const unsigned char* __restrict__ lds,
                                            const QuadIn<ORIENT>& in, int nn, float px, float py, float pz, bool active, int qi,
                                            int g, const float* __restrict__ labels, float* __restrict__ sdf_out,
                                            float* __restrict__ grad_out, float (&tot)[8]) {
    using Q = QuadDec<H, BF>;
    const float s = f.sdf_scale;
    const float (&z)[4] = in.z;
    const float (&Y)[3][4] = in.Y;
    const float Gx = in.Gx, Gy = in.Gy, Gz = in.Gz, wsum = in.wsum, S = in.S;
    const float (&M)[ORIENT ? 9 : 1] = in.M;
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
            // lane g keeps sums 4j + g.  Written as masked FMAs: a select chain over v[] is turned into a
            // dynamically indexed private array (scratch memory) by the compiler
            const float m0 = g == 0 ? 1.f : 0.f, m1 = g == 1 ? 1.f : 0.f, m2 = g == 2 ? 1.f : 0.f, m3 = g == 3 ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                tot[j] += fmaf(m0, v[4 * j], fmaf(m1, v[4 * j + 1], fmaf(m2, v[4 * j + 2], m3 * v[4 * j + 3])));
        }
    }
}

__host__ __device__ constexpr int gq_red_offset(int image_bytes) { return (image_bytes + 15) & ~15; }
__host__ __device__ constexpr int gq_lds_bytes(int image_bytes) {
    return gq_red_offset(image_bytes) + (GQ_BLOCK / 64) * PIN_GN_NSUMS * (int)sizeof(float);
}

template <int H, bool ORIENT, bool BF>
__global__ __launch_bounds__(GQ_BLOCK, 1) void gn_accumulate_quad_kernel(pin_field f, pin_gn_params gp,
                                                                         const float* __restrict__ query,
                                                                         const float4* __restrict__ nbr,
                                                                         const int* __restrict__ nn_count,
                                                                         const float* __restrict__ labels, int n_q,
                                                                         double* __restrict__ sums, float* __restrict__ sdf_out,
                                                                         float* __restrict__ grad_out,
                                                                         const double* __restrict__ state) {
    using Q = QuadDec<H, BF>;
    extern __shared__ __attribute__((aligned(16))) unsigned char gq_smem[];  // decoder image, then the block reduction
    unsigned char* const lds = gq_smem;
    float (*red)[PIN_GN_NSUMS] = reinterpret_cast<float (*)[PIN_GN_NSUMS]>(lds + gq_red_offset(Q::bytes(f.levels)));
    if (state != nullptr && state[PIN_GN_STATE_DONE] != 0.0) return;
    // weights: copy the image staged once per registration (pin_stage_decoder) or split them here; either way the
    // image is visible after the barrier that follows the first gather
    if (BF && f.dec_image != nullptr && f.dec_image_bytes == Q::bytes(f.levels)) {
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(f.dec_image);
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(lds);
        const int n16 = f.dec_image_bytes >> 4;
#pragma unroll 4
        for (int i = threadIdx.x; i < n16; i += GQ_BLOCK) dst[i] = src[i];
    } else {
        Q::stage(f.dec, f.levels, lds, threadIdx.x, GQ_BLOCK);
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
    for (int tile = simd + n_simd * (wave >> 2);; tile += n_simd * (GQ_BLOCK / 256)) {
        const bool work = tile < n_tiles;
        if (!work && staged) break;
        const int qi = (work ? tile : 0) * 16 + nq;
        const bool active = qi < n_q;
        const int qq = active ? qi : n_q - 1;
        const float px = query[3 * qq], py = query[3 * qq + 1], pz = query[3 * qq + 2];
        const int nn = nn_count[qq];
        QuadIn<ORIENT> in;
        quad_gather<ORIENT>(f, nbr + (size_t)qq * f.k, f.k, nn, px, py, pz, g, in);
        if (!staged) {  // the first gather overlaps the weight staging of the block
            __syncthreads();
            staged = true;
            if (!work) break;
        }
        quad_finish<H, ORIENT, BF>(f, gp, lds, in, nn, px, py, pz, active, qi, g, labels, sdf_out, grad_out, tot);
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


// ---- search front-end of the fused tile kernel: the brick-cache kNN with FOUR lanes per query --------
// Same candidates, same float32 distances, same (d2, candidate order) ranking as knn_brick_kernel
// (bit-identical records), laid out for the quad decoder: lane (n, g) probes candidates c = 4r + g.
// Brick headers go through a wave-private LDS table, the k winners' records land in a second one
// that all four lanes of the query read back.  Returns the number of accepted candidates.
__device__ __forceinline__ unsigned int quad_min_u32(unsigned int v) {
    v = min(v, (unsigned int)__shfl_xor((int)v, 16, 64));
    v = min(v, (unsigned int)__shfl_xor((int)v, 32, 64));
    return v;
}

struct QuadCell {  // what one candidate cell resolves to
    bool ok;
    float4 E;
    int l;
};

__device__ __forceinline__ QuadCell quad_probe(const pin_search_params& sp, const pin_brick_cache& bc,
                                               const float4* __restrict__ bricks, int nq, bool far, long long gx, long long gy,
                                               long long gz, int b0x, int b0y, int b0z, int c, float d_cur) {
    QuadCell r;
    r.ok = false; r.l = -1; r.E = make_float4(0.f, 0.f, 0.f, 0.f);
    const int dxc = bc.cand_dx[3 * c], dyc = bc.cand_dx[3 * c + 1], dzc = bc.cand_dx[3 * c + 2];
    const int cx = (int)gx + dxc, cy = (int)gy + dyc, cz = (int)gz + dzc;
    const int sel = (((cx >> 2) - b0x) << 2) | (((cy >> 2) - b0y) << 1) | ((cz >> 2) - b0z);
    const float4 bi = bricks[nq * 8 + (sel & 7)];
    const int base = far ? -1 : __float_as_int(bi.x);
    if (base >= 0) {
        const unsigned int lo = __float_as_uint(bi.y), hi = __float_as_uint(bi.z);
        const int bit = ((cx & 3) << 4) | ((cy & 3) << 2) | (cz & 3);
        const unsigned int word = bit < 32 ? lo : hi;
        if ((word >> (bit & 31)) & 1u) {
            const unsigned int below = word & ((1u << (bit & 31)) - 1u);
            r.E = reinterpret_cast<const float4*>(bc.entries)[base + __popc(below) + (bit < 32 ? 0 : __popc(lo))];
            r.l = __float_as_int(r.E.w);
            r.ok = true;
        }
    } else {
        r.ok = lookup_cell(sp, gx + dxc, gy + dyc, gz + dzc, d_cur, r.E, r.l);  // exact slow path for uncached bricks
    }
    return r;
}

// quad (4 consecutive lanes) all-reduce on the DPP path
__device__ __forceinline__ unsigned int quadperm_min_u32(unsigned int v) {
    v = min(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
    v = min(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
    return v;
}

constexpr int GQ_ROW = 10;  // float4 per query in the wave's record table: 8 records, (qx, qy, qz, count), pad

// Search phase layout: lane = 4 * query + sub (a quad per query), so the per-round reductions are
// two quad_perm DPP steps.  Results go to the wave-private LDS table `recs` ([16][GQ_ROW] float4),
// from where the decoder phase (lane = query + 16 * component group) reads them back.
template <int R>
__device__ __forceinline__ void knn_quad(const pin_search_params& sp, const pin_brick_cache& bc, float qx, float qy, float qz,
                                         int k, float4* __restrict__ recs, float4* __restrict__ bricks, int nq, int g) {
    const long long gx = voxel_coord(qx, sp.resolution), gy = voxel_coord(qy, sp.resolution),
                    gz = voxel_coord(qz, sp.resolution);
    const int nd = bc.n_dilate;
    const long long lim = 1LL << 29;
    const bool far = gx >= lim || gx < -lim || gy >= lim || gy < -lim || gz >= lim || gz < -lim;
    const int b0x = ((int)gx - nd) >> 2, b0y = ((int)gy - nd) >> 2, b0z = ((int)gz - nd) >> 2;
    // lane g resolves bricks 2g and 2g + 1 of the 2x2x2 bricks that cover the candidate window
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
        const int b = 2 * g + sb;
        int base = -1;
        unsigned int lo = 0, hi = 0;
        if (!far) {
            const BrickInfo bi = dir_lookup(bc, brick_key(b0x + (b >> 2), b0y + ((b >> 1) & 1), b0z + (b & 1)));
            base = bi.base; lo = bi.lo; hi = bi.hi;
        }
        bricks[nq * 8 + b] = make_float4(__int_as_float(base), __uint_as_float(lo), __uint_as_float(hi), 0.f);
    }
    wave_lds_sync();
    const float d_cur = sp.travel_dist ? sp.travel_dist[sp.cur_ts] : 0.f;
    // Each lane keeps the best PIN_MAX_K of its own candidates, sorted by (d2 bits, candidate order): the
    // query's k nearest are among the four lanes' lists.  Rolled loop (three probes in flight per turn), so
    // the code stays small -- the fully unrolled form does not fit the instruction cache next to the decoder.
    unsigned int ld[PIN_MAX_K];
    int lc[PIN_MAX_K];
#pragma unroll
    for (int i = 0; i < PIN_MAX_K; ++i) { ld[i] = 0xffffffffu; lc[i] = -1; }
    int cnt = 0;
    constexpr int U = 3;
#pragma unroll 1
    for (int r0 = 0; r0 < R; r0 += U) {
        unsigned int nd2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = 4 * (r0 + u) + g;
            nd2[u] = 0xffffffffu;
            if (r0 + u < R && c < sp.n_cand) {
                const QuadCell q = quad_probe(sp, bc, bricks, nq, far, gx, gy, gz, b0x, b0y, b0z, c, d_cur);
                if (q.ok) {
                    const float d2 = dist2_exact(q.E.x - qx, q.E.y - qy, q.E.z - qz);
                    if (!(d2 > sp.max_valid_dist2)) { nd2[u] = __float_as_uint(d2); ++cnt; }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // insert (candidates arrive in increasing order, `<` keeps equal distances in that order)
            unsigned int d = nd2[u];
            int c = 4 * (r0 + u) + g;
#pragma unroll
            for (int i = 0; i < PIN_MAX_K; ++i) {
                const bool sw = d < ld[i];
                const unsigned int td = ld[i];
                const int tc = lc[i];
                ld[i] = sw ? d : td; lc[i] = sw ? c : tc;
                d = sw ? td : d; c = sw ? tc : c;
            }
        }
    }
    cnt += __builtin_amdgcn_update_dpp(0, cnt, 0xB1, 0xf, 0xf, true);
    cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x4E, 0xf, 0xf, true);
    // k rounds over the heads of the four lists; only the winners' candidate numbers are kept, their records
    // are fetched afterwards, all at once
    int mine0 = -1, mine1 = -1;  // winners t = 2g and t = 2g + 1 (this lane publishes them)
#pragma unroll 1
    for (int t = 0; t < k; ++t) {
        const unsigned int wd = quadperm_min_u32(ld[0]);
        if (wd == 0xffffffffu) break;
        const unsigned int myc = ld[0] == wd ? (unsigned int)lc[0] : 0xffffffffu;
        const unsigned int wc = quadperm_min_u32(myc);
        if (myc == wc) {  // pop
#pragma unroll
            for (int i = 0; i + 1 < PIN_MAX_K; ++i) { ld[i] = ld[i + 1]; lc[i] = lc[i + 1]; }
            ld[PIN_MAX_K - 1] = 0xffffffffu;
        }
        if ((t >> 1) == g) { if (t & 1) mine1 = (int)wc; else mine0 = (int)wc; }
    }
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
        const int t = 2 * g + sb;
        const int c = sb ? mine1 : mine0;
        if (t < k) {
            float4 rec = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            if (c >= 0) {
                const QuadCell q = quad_probe(sp, bc, bricks, nq, far, gx, gy, gz, b0x, b0y, b0z, c, d_cur);
                const float dx = q.E.x - qx, dy = q.E.y - qy, dz = q.E.z - qz;
                rec = make_float4(-dx, -dy, -dz, __int_as_float(q.l));
            }
            recs[nq * GQ_ROW + t] = rec;
        }
    }
    if (g == 0) recs[nq * GQ_ROW + 8] = make_float4(qx, qy, qz, __int_as_float(cnt));
    wave_lds_sync();
}

struct QuadPose { float m[12]; int on; };

// ---- one Gauss-Newton iteration as ONE kernel: search producers + decoder consumers --------------------------
// STATUS: correct (bit-identical records, tests/test_gpu_bricks.py) but NOT faster than the two separate
// kernels at C3: 135 us vs 55 + 85 us.  The search is latency bound and wants ~24 waves per CU; inside a
// 16-wave block that shares the CU with the decoder it gets 8.  Measured splits 8/8, 12/4 producers/consumers
// and 3 / 7 probes in flight: 135 / 143 / 147 us.  Kept as an opt-in (PIN_GN_FUSED=1) entry point.
// Per CU one persistent block of 16 waves: waves 0-7 run the brick-cache search for the CU's tiles
// (VALU / memory latency bound) and leave the kNN records in LDS slots, waves 8-15 take the slots
// through the decoder (fp32 MFMA bound).  The two halves run on different pipes of the same SIMDs
// at the same time -- with one role per wave their phases cannot fall into lock step, which is what
// happens when every wave searches, gathers and decodes its own tile.  Slots form a ring; per-slot
// sequence numbers in LDS (ready / done) are the only synchronisation, all 16 waves are resident.
constexpr int GI_SLOTS = 24;            // tile slots per CU (one tile = 16 queries)
constexpr int GI_PRODUCERS = 6;
template <int H>
__host__ __device__ constexpr int gi_lds_floats() {
    return QuadDecoder<H>::TOTAL + (GQ_BLOCK / 64) * PIN_GN_NSUMS + GI_SLOTS * 16 * GQ_ROW * 4 + GI_PRODUCERS * 16 * 8 * 4 +
           2 * GI_SLOTS;
}

__device__ __forceinline__ void lds_wait_eq(volatile int* flag, int value) {
    while (*flag != value) __builtin_amdgcn_s_sleep(1);
    __threadfence_block();  // acquire: nothing that follows may be read before the flag
}

template <int H, int R>
__global__ __launch_bounds__(GQ_BLOCK, 1) void gn_iteration_kernel(pin_field f, pin_gn_params gp, pin_search_params sp,
                                                                   pin_brick_cache bc, const float* __restrict__ src,
                                                                   const float* __restrict__ labels, int n_q, int knn_k,
                                                                   double* __restrict__ sums, const double* __restrict__ state,
                                                                   QuadPose pose, float* __restrict__ cur_out,
                                                                   float4* __restrict__ nbr_out, int* __restrict__ nn_out) {
    using Q = QuadDecoder<H>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float (*red)[PIN_GN_NSUMS] = reinterpret_cast<float (*)[PIN_GN_NSUMS]>(lds + Q::TOTAL);
    float4* slots = reinterpret_cast<float4*>(lds + Q::TOTAL + (GQ_BLOCK / 64) * PIN_GN_NSUMS);
    float4* btab = slots + GI_SLOTS * 16 * GQ_ROW;
    volatile int* ready = reinterpret_cast<volatile int*>(btab + GI_PRODUCERS * 16 * 8);
    volatile int* done = ready + GI_SLOTS;
    if (state != nullptr && state[PIN_GN_STATE_DONE] != 0.0) return;
    if (state != nullptr) {
#pragma unroll
        for (int i = 0; i < 12; ++i) pose.m[i] = (float)state[i];
        pose.on = 1;
    }
    Q::stage(f.dec, f.levels, lds, threadIdx.x, GQ_BLOCK);
    for (int i = threadIdx.x; i < 2 * GI_SLOTS; i += GQ_BLOCK) ready[i] = 0;
    for (int i = threadIdx.x; i < (GQ_BLOCK / 64) * PIN_GN_NSUMS; i += GQ_BLOCK) (&red[0][0])[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_tiles = (n_q + 15) >> 4;
    // this CU's tiles: blockIdx.x + gridDim.x * i, i = 0 .. my_tiles - 1
    const int my_tiles = n_tiles > (int)blockIdx.x ? (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (wave < GI_PRODUCERS) {
        // ------------------------------------------------------------------ producers: a quad of lanes per query
        const int sq = lane >> 2, sg = lane & 3;
        float4* bricks = btab + wave * (16 * 8);
        for (int i = wave; i < my_tiles; i += GI_PRODUCERS) {
            const int tile = blockIdx.x + gridDim.x * i;
            const int slot = i % GI_SLOTS;
            if (i >= GI_SLOTS) lds_wait_eq(done + slot, i - GI_SLOTS + 1);  // previous occupant consumed
            const int qi = tile * 16 + sq;
            const int qq = min(qi, n_q - 1);
            float sx = src[3 * qq], sy = src[3 * qq + 1], sz = src[3 * qq + 2];
            if (pose.on) {  // transform_torch folded in, as in the search kernels
                const float* m = pose.m;
                const float tx = fmaf(sz, m[2], fmaf(sy, m[1], sx * m[0])) + m[3];
                const float ty = fmaf(sz, m[6], fmaf(sy, m[5], sx * m[4])) + m[7];
                const float tz = fmaf(sz, m[10], fmaf(sy, m[9], sx * m[8])) + m[11];
                sx = tx; sy = ty; sz = tz;
            }
            float4* recs = slots + slot * (16 * GQ_ROW);
            knn_quad<R>(sp, bc, sx, sy, sz, knn_k, recs, bricks, sq, sg);  // ends with a wave-level LDS sync
            if (qi < n_q) {  // optional dumps (tests): transformed point, records, accepted-candidate count
                const float4* rp = recs + sq * GQ_ROW;
                if (cur_out != nullptr && sg == 0) { cur_out[3 * qi] = sx; cur_out[3 * qi + 1] = sy; cur_out[3 * qi + 2] = sz; }
                if (nn_out != nullptr && sg == 0) nn_out[qi] = __float_as_int(rp[8].w);
                if (nbr_out != nullptr) {
                    for (int t = 2 * sg; t < 2 * sg + 2; ++t)
                        if (t < knn_k) nbr_out[(size_t)qi * knn_k + t] = rp[t];
                }
            }
            __threadfence_block();
            if (lane == 0) ready[slot] = i + 1;
        }
    } else {
        // ------------------------------------------------------------------ consumers: lane = query + 16 * component group
        const int nq = lane & 15, g = lane >> 4;
        float tot[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i = wave - GI_PRODUCERS; i < my_tiles; i += GQ_BLOCK / 64 - GI_PRODUCERS) {
            const int tile = blockIdx.x + gridDim.x * i;
            const int slot = i % GI_SLOTS;
            lds_wait_eq(ready + slot, i + 1);
            const float4* rp = slots + slot * (16 * GQ_ROW) + nq * GQ_ROW;
            const float4 qc = rp[8];
            const int qi = tile * 16 + nq;
            QuadIn<false> in;
            quad_gather<false>(f, rp, knn_k, __float_as_int(qc.w), qc.x, qc.y, qc.z, g, in);
            __threadfence_block();
            if (lane == 0) done[slot] = i + 1;  // records are in registers: the slot can be refilled
            quad_finish<H, false, false>(f, gp, reinterpret_cast<const unsigned char*>(lds), in, __float_as_int(qc.w), qc.x, qc.y, qc.z, qi < n_q, qi, g, labels, nullptr, nullptr,
                                  tot);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = row_sum_f32(tot[j]);
            if (nq == 0) red[wave][4 * j + g] = t;
        }
    }
    __syncthreads();
    if (threadIdx.x < PIN_GN_NSUMS) {
        double t = 0.0;
#pragma unroll
        for (int w = GI_PRODUCERS; w < GQ_BLOCK / 64; ++w) t += (double)red[w][threadIdx.x];
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
