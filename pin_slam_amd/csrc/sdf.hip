// Fused feature interpolation -> shallow-MLP SDF -> analytic Jacobian -> Gauss-Newton sums.
//
// Replaces (reference paths relative to PRBonn/PIN_SLAM):
//   NeuralPoints.query_feature        model/neural_points.py:590-746   (gather, IDW, certainty)
//   Decoder.sdf                       model/decoder.py:83-85
//   get_gradient (autograd)           utils/tools.py:247-260 via utils/tracker.py:331
//   Tracker.registration_step/implicit_reg  utils/tracker.py:409-524, 652-671
//
// One thread per query.  Input is the kNN record written by knn.hip (k x 16 B, coalesced
// per thread), so the only random traffic here is the k feature rows (32 B each).
#include <mutex>
#include "mlp_mfma.h"

namespace pin {

constexpr int SDF_BLOCK = 128;
constexpr int GN_REPLICAS = PIN_GN_REPLICAS;  // sums are scattered over a few replicas to spread the atomics (the solve reads them all)

struct Nbrs {
    float vx[PIN_MAX_K], vy[PIN_MAX_K], vz[PIN_MAX_K];  // q - P (global position)
    float u[PIN_MAX_K], w[PIN_MAX_K];
    int idx[PIN_MAX_K];  // index into the field arrays, -1 invalid
    bool quirk[PIN_MAX_K];
    float S;
    int nn;
};

__device__ __forceinline__ void load_neighbors(const float4* __restrict__ nbr, const int* __restrict__ nn_count,
                                               int qi, int k, Nbrs& nb) {
    nb.nn = nn_count[qi];
    float S = 0.f;
#pragma unroll
    for (int t = 0; t < PIN_MAX_K; ++t) {
        nb.idx[t] = -1; nb.u[t] = 0.f; nb.w[t] = 0.f; nb.quirk[t] = false;
        nb.vx[t] = nb.vy[t] = nb.vz[t] = 0.f;
        if (t < k) {
            const float4 e = nbr[(size_t)qi * k + t];
            const int raw = __float_as_int(e.w);
            if (raw >= 0) {
                nb.idx[t] = raw & ~PIN_NBR_QUIRK_BIT;
                nb.quirk[t] = (raw & PIN_NBR_QUIRK_BIT) != 0;
                nb.vx[t] = e.x; nb.vy[t] = e.y; nb.vz[t] = e.z;
                const float d2 = dist2_exact(e.x, e.y, e.z);
                nb.u[t] = 1.0f / (d2 + IDW_EPS);  // neural_points.py:667
            }
            if (nb.nn == 0) nb.u[t] = IDW_EPS;  // neural_points.py:672-674
            S += nb.u[t];
        }
    }
    nb.S = S;
#pragma unroll
    for (int t = 0; t < PIN_MAX_K; ++t)
        if (t < k && nb.idx[t] >= 0) nb.w[t] = nb.u[t] / S;  // invalid keep 0 (neural_points.py:683)
}

// neighbour vector fed to the decoder: q - P[idx] (rotated into the point frame after PGO)
__device__ __forceinline__ void neighbor_vector(const pin_field& f, int idx, bool quirk, float vgx, float vgy,
                                                float vgz, float qx, float qy, float qz, float (&v)[3],
                                                float (&Rm)[9]) {
    v[0] = vgx; v[1] = vgy; v[2] = vgz;
    if (quirk) {  // reference gathers local point #1 for non-local neighbours
        const float* p = f.pos + 3 * (size_t)idx;
        v[0] = qx - p[0]; v[1] = qy - p[1]; v[2] = qz - p[2];
    }
    if (f.orient != nullptr) {  // apply_quaternion_rotation (utils/tools.py:428-437) rotates by the conjugate: Rm = R(q)^T
        const float4 q = reinterpret_cast<const float4*>(f.orient)[idx];
        const float q0 = q.x, q1 = q.y, q2 = q.z, q3 = q.w;
        Rm[0] = 1 - 2 * (q2 * q2 + q3 * q3); Rm[3] = 2 * (q1 * q2 - q0 * q3); Rm[6] = 2 * (q1 * q3 + q0 * q2);
        Rm[1] = 2 * (q1 * q2 + q0 * q3); Rm[4] = 1 - 2 * (q1 * q1 + q3 * q3); Rm[7] = 2 * (q2 * q3 - q0 * q1);
        Rm[2] = 2 * (q1 * q3 - q0 * q2); Rm[5] = 2 * (q2 * q3 + q0 * q1); Rm[8] = 1 - 2 * (q1 * q1 + q2 * q2);
        const float x = v[0], y = v[1], z = v[2];
        v[0] = Rm[0] * x + Rm[1] * y + Rm[2] * z;
        v[1] = Rm[3] * x + Rm[4] * y + Rm[5] * z;
        v[2] = Rm[6] * x + Rm[7] * y + Rm[8] * z;
    }
}

__device__ __forceinline__ void load_feature(const pin_field& f, int idx, float (&ft)[PIN_FEATURE_DIM]) {
    const float4* row = reinterpret_cast<const float4*>(f.feats + (size_t)idx * PIN_FEATURE_DIM);
    const float4 a = row[0], b = row[1];
    ft[0] = a.x; ft[1] = a.y; ft[2] = a.z; ft[3] = a.w;
    ft[4] = b.x; ft[5] = b.y; ft[6] = b.z; ft[7] = b.w;
}

struct SdfResult {
    float sdf, gx, gy, gz, std, cert;
    float p[MF_OD_MAX];  // colour heads: sigmoid outputs (interpolated over neighbours if per-neighbour decode)
};

struct ColorTerm {
    pin_field fc;          // colour feature table + colour decoder (3 heads)
    const float* colors;   // [n][3] measured colours of the source points
    int mode;              // 0 none, 1 consistency weight, 2 photometric term
    float photo_weight;
};

struct Kappa {
    float k[MF_OD_MAX];  // value = sum_c k[c] * sigmoid(out_c) for the colour decoder
};

// The fused per-query evaluation shared by pin_sdf_query and pin_gn_accumulate: decoder on the matrix cores
// (block weights `col`, wave scratch `xb`).
template <int H, bool GRAD, bool MFMA, int OD = 1>
__device__ __forceinline__ float decode(const pin_field& f, const float (&z)[MLP_IN], float (&a)[MLP_IN], float* col,
                                        float* xb, const Kappa& kap, float (&p)[MF_OD_MAX]) {
    static_assert(MFMA, "the thread-per-query vector decoder was removed in round 2");
    float kk[OD], pp[OD];
#pragma unroll
    for (int c = 0; c < OD; ++c) kk[c] = kap.k[c];
    const float v = MfmaDecoder<H>::template run_heads<GRAD, 2, OD>(col, f.levels, xb, z, kk, pp, a);
#pragma unroll
    for (int c = 0; c < OD; ++c) p[c] = pp[c];
    return v;
}

template <int H, bool WF, bool GRAD, bool MFMA = true, int OD = 1>
__device__ __forceinline__ SdfResult eval_query(const pin_field& f, const float4* __restrict__ nbr,
                                                const int* __restrict__ nn_count, int qi, float qx, float qy,
                                                float qz, float* col, float* xb = nullptr, Kappa kap = Kappa()) {
    Nbrs nb;
    load_neighbors(nbr, nn_count, qi, f.k, nb);
    SdfResult r;
    r.std = 0.f; r.gx = r.gy = r.gz = 0.f;
#pragma unroll
    for (int c = 0; c < MF_OD_MAX; ++c) r.p[c] = 0.f;
    float cert = 0.f;
    if (f.certainty != nullptr) {
#pragma unroll
        for (int t = 0; t < PIN_MAX_K; ++t)
            if (nb.idx[t] >= 0) cert = fmaf(f.certainty[nb.idx[t]], nb.w[t], cert);
    }
    r.cert = cert;
    const float s = OD == 1 ? f.sdf_scale : 1.f;  // colour heads are not scaled (decoder.py:112)
    float Rm[9];
    // G = sum_t d u_t / d q,  d u_t/d q = -2 u_t^2 (q - P_t)
    float Gx = 0.f, Gy = 0.f, Gz = 0.f, wsum = 0.f;
    if (GRAD) {
#pragma unroll
        for (int t = 0; t < PIN_MAX_K; ++t)
            if (nb.idx[t] >= 0) {
                const float c = -2.f * nb.u[t] * nb.u[t];
                Gx = fmaf(c, nb.vx[t], Gx); Gy = fmaf(c, nb.vy[t], Gy); Gz = fmaf(c, nb.vz[t], Gz);
                wsum += nb.w[t];
            }
    }
    if (WF) {
        float z[MLP_IN];
#pragma unroll
        for (int j = 0; j < MLP_IN; ++j) z[j] = 0.f;
        // One pass over the neighbours also accumulates what the gradient needs afterwards, so the
        // feature rows are gathered once:  sum_t c_t g_t = Y a  with  Y = sum_t g_t (x) [f_t; v_t]
        // (c_t = a . [f_t; v_t], g_t = d u_t / d q), and after PGO  sum_t w_t R_t^T a_v = M a_v.
        float Y[3][MLP_IN];
        float M[9];
        if (GRAD) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int j = 0; j < MLP_IN; ++j) Y[c][j] = 0.f;
#pragma unroll
            for (int c = 0; c < 9; ++c) M[c] = 0.f;
        }
        // (GRAD) every row relative to the nearest neighbour's, y_t - y_0: the weight-derivative term sum_t g_t (c_t - cbar)
        // keeps its digits when one neighbour dominates (gn_quad.h, quad_gather_pass PIVOT); zt = sum_t w_t (y_t - y_0)
        float y0[MLP_IN], zt[MLP_IN];
#pragma unroll
        for (int j = 0; j < MLP_IN; ++j) y0[j] = zt[j] = 0.f;
#pragma unroll
        for (int t = 0; t < PIN_MAX_K; ++t)
            if (nb.idx[t] >= 0) {
                float y[MLP_IN], ft[PIN_FEATURE_DIM], v[3];
                load_feature(f, nb.idx[t], ft);
                neighbor_vector(f, nb.idx[t], nb.quirk[t], nb.vx[t], nb.vy[t], nb.vz[t], qx, qy, qz, v, Rm);
#pragma unroll
                for (int j = 0; j < PIN_FEATURE_DIM; ++j) y[j] = ft[j];
                y[8] = v[0]; y[9] = v[1]; y[10] = v[2];
#pragma unroll
                for (int j = 0; j < MLP_IN; ++j) z[j] = fmaf(nb.w[t], y[j], z[j]);
                if (GRAD) {
                    if (t == 0) {
#pragma unroll
                        for (int j = 0; j < MLP_IN; ++j) y0[j] = y[j];
                    }
#pragma unroll
                    for (int j = 0; j < MLP_IN; ++j) { y[j] -= y0[j]; zt[j] = fmaf(nb.w[t], y[j], zt[j]); }
                    const float cg = -2.f * nb.u[t] * nb.u[t];
                    const float g0 = cg * nb.vx[t], g1 = cg * nb.vy[t], g2 = cg * nb.vz[t];
#pragma unroll
                    for (int j = 0; j < MLP_IN; ++j) {
                        Y[0][j] = fmaf(g0, y[j], Y[0][j]); Y[1][j] = fmaf(g1, y[j], Y[1][j]); Y[2][j] = fmaf(g2, y[j], Y[2][j]);
                    }
                    if (f.orient != nullptr) {  // d v_t / d q = R_t: accumulate w_t R_t^T
#pragma unroll
                        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                            for (int cc = 0; cc < 3; ++cc) M[rr * 3 + cc] = fmaf(nb.w[t], Rm[cc * 3 + rr], M[rr * 3 + cc]);
                    }
                }
            }
        float a[MLP_IN];
        const float x = decode<H, GRAD, MFMA, OD>(f, z, a, col, xb, kap, r.p);
        r.sdf = s * x;
        if (GRAD) {
            float cbar = 0.f;
#pragma unroll
            for (int j = 0; j < MLP_IN; ++j) cbar = fmaf(a[j], zt[j], cbar);  // = sum_t w_t (c_t - c_0), as Y holds the g_t (x) (y_t - y_0)
            float ax = 0.f, ay = 0.f, az = 0.f;  // sum_t c_t g_t
#pragma unroll
            for (int j = 0; j < MLP_IN; ++j) { ax = fmaf(Y[0][j], a[j], ax); ay = fmaf(Y[1][j], a[j], ay); az = fmaf(Y[2][j], a[j], az); }
            float dxs, dys, dzs;  // the direct a_v term
            if (f.orient != nullptr) {
                dxs = M[0] * a[8] + M[1] * a[9] + M[2] * a[10];
                dys = M[3] * a[8] + M[4] * a[9] + M[5] * a[10];
                dzs = M[6] * a[8] + M[7] * a[9] + M[8] * a[10];
            } else { dxs = a[8] * wsum; dys = a[9] * wsum; dzs = a[10] * wsum; }
            const float invS = 1.0f / nb.S;
            r.gx = s * (dxs + (ax - cbar * Gx) * invS);
            r.gy = s * (dys + (ay - cbar * Gy) * invS);
            r.gz = s * (dzs + (az - cbar * Gz) * invS);
        }
    } else {
        // decode every neighbour, then weight (weighted_first = False, run_kitti.yaml:25)
        float sk[PIN_MAX_K];
        float mean = 0.f, dxs = 0.f, dys = 0.f, dzs = 0.f;
        float ax = 0.f, ay = 0.f, az = 0.f, s0 = 0.f, mt = 0.f;
#pragma unroll 1
        for (int t = 0; t < f.k; ++t) {
            // static-index copies of neighbour t
            int idx = -1; float wt = 0.f, ut = 0.f, vgx = 0.f, vgy = 0.f, vgz = 0.f;
#pragma unroll
            for (int u = 0; u < PIN_MAX_K; ++u)
                if (u == t) { idx = nb.idx[u]; wt = nb.w[u]; ut = nb.u[u]; vgx = nb.vx[u]; vgy = nb.vy[u]; vgz = nb.vz[u]; }
            float st = 0.f;
            {   // every lane decodes (the MFMA back-end is wave-wide); invalid neighbours decode
                // zeros and are discarded: their weight is zero in the reference too
                float z[MLP_IN], ft[PIN_FEATURE_DIM] = {0, 0, 0, 0, 0, 0, 0, 0}, v[3] = {0, 0, 0};
                if (idx >= 0) {
                    load_feature(f, idx, ft);
                    bool qk = false;
#pragma unroll
                    for (int u = 0; u < PIN_MAX_K; ++u) if (u == t) qk = nb.quirk[u];
                    neighbor_vector(f, idx, qk, vgx, vgy, vgz, qx, qy, qz, v, Rm);
                }
#pragma unroll
                for (int j = 0; j < PIN_FEATURE_DIM; ++j) z[j] = ft[j];
                z[8] = v[0]; z[9] = v[1]; z[10] = v[2];
                float a[MLP_IN];
                float pt[MF_OD_MAX];
                const float xt = decode<H, GRAD, MFMA, OD>(f, z, a, col, xb, kap, pt);
                if (idx >= 0) {
#pragma unroll
                    for (int c = 0; c < OD; ++c) r.p[c] = fmaf(wt, pt[c], r.p[c]);
                    st = s * xt;
                    mean = fmaf(wt, st, mean);
                    if (GRAD) {
                        if (f.orient != nullptr) {
                            dxs += wt * (Rm[0] * a[8] + Rm[3] * a[9] + Rm[6] * a[10]);
                            dys += wt * (Rm[1] * a[8] + Rm[4] * a[9] + Rm[7] * a[10]);
                            dzs += wt * (Rm[2] * a[8] + Rm[5] * a[9] + Rm[8] * a[10]);
                        } else {
                            dxs = fmaf(wt, a[8], dxs); dys = fmaf(wt, a[9], dys); dzs = fmaf(wt, a[10], dzs);
                        }
                        if (t == 0) s0 = st;  // (predictions relative to the nearest neighbour's: see the weighted-first branch)
                        const float sp = st - s0;
                        mt = fmaf(wt, sp, mt);
                        const float cg = -2.f * ut * ut * sp;
                        ax = fmaf(cg, vgx, ax); ay = fmaf(cg, vgy, ay); az = fmaf(cg, vgz, az);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < PIN_MAX_K; ++u) if (u == t) sk[u] = st;
        }
        r.sdf = mean;
        float var = 0.f;
#pragma unroll
        for (int t = 0; t < PIN_MAX_K; ++t)
            if (t < f.k && nb.idx[t] >= 0) { const float d = sk[t] - mean; var = fmaf(nb.w[t], d * d, var); }
        r.std = sqrtf(var);  // tracker.py:317-322
        if (GRAD) {
            const float invS = 1.0f / nb.S;
            r.gx = s * dxs + (ax - mt * Gx) * invS;
            r.gy = s * dys + (ay - mt * Gy) * invS;
            r.gz = s * dzs + (az - mt * Gz) * invS;
        }
    }
    return r;
}

// ---- the fused query / Gauss-Newton kernels, 64 queries per wave, decoder on the fp32 matrix cores (mlp_mfma.h):
// per-neighbour decoding (weighted_first = False), the colour term and plain forward queries -----------

template <int H, bool WF, int OD = 1>
__global__ __launch_bounds__(MF_BLOCK, 1) void sdf_query_mfma_kernel(pin_field f, const float* __restrict__ query,
                                                                  const float4* __restrict__ nbr,
                                                                  const int* __restrict__ nn_count, int n,
                                                                  float* __restrict__ sdf_out, float* __restrict__ grad_out,
                                                                  float* __restrict__ std_out, float* __restrict__ cert_out,
                                                                  Kappa kap = Kappa(), float* __restrict__ color_out = nullptr) {
    __shared__ __attribute__((aligned(16))) float lds[MfmaLds<H>::TOTAL];
    float* xb = lds + MfmaLds<H>::W + (threadIdx.x >> 6) * MfmaDecoder<H>::scratch_floats();
    MfmaDecoder<H>::stage(f.dec, f.levels, lds, threadIdx.x, MF_BLOCK, OD);
    __syncthreads();
    const int qi = blockIdx.x * MF_BLOCK + threadIdx.x;
    const bool active = qi < n;
    const int qq = active ? qi : n - 1;  // every lane takes part in the wave-wide MFMAs
    const float qx = query[3 * qq], qy = query[3 * qq + 1], qz = query[3 * qq + 2];
    SdfResult r;
    if (grad_out != nullptr) r = eval_query<H, WF, true, true, OD>(f, nbr, nn_count, qq, qx, qy, qz, lds, xb, kap);
    else r = eval_query<H, WF, false, true, OD>(f, nbr, nn_count, qq, qx, qy, qz, lds, xb, kap);
    if (!active) return;
    if (color_out) {
#pragma unroll
        for (int c = 0; c < OD; ++c) color_out[(size_t)qi * OD + c] = r.p[c];
    }
    if (sdf_out) sdf_out[qi] = r.sdf;
    if (grad_out) { grad_out[3 * qi] = r.gx; grad_out[3 * qi + 1] = r.gy; grad_out[3 * qi + 2] = r.gz; }
    if (std_out) std_out[qi] = r.std;
    if (cert_out) cert_out[qi] = r.cert;
}

template <int H, bool WF>
__global__ __launch_bounds__(MF_BLOCK, 1) void gn_accumulate_mfma_kernel(pin_field f, pin_gn_params gp,
                                                                      const float* __restrict__ query,
                                                                      const float4* __restrict__ nbr,
                                                                      const int* __restrict__ nn_count,
                                                                      const float* __restrict__ labels, int n,
                                                                      double* __restrict__ sums, float* __restrict__ sdf_out,
                                                                      float* __restrict__ grad_out,
                                                                      const double* __restrict__ state, ColorTerm ct) {
    __shared__ __attribute__((aligned(16))) float lds[MfmaLds<H>::TOTAL];
    if (state != nullptr && state[PIN_GN_STATE_DONE] != 0.0) return;
    float* xb = lds + MfmaLds<H>::W + (threadIdx.x >> 6) * MfmaDecoder<H>::scratch_floats();
    MfmaDecoder<H>::stage(f.dec, f.levels, lds, threadIdx.x, MF_BLOCK);
    __syncthreads();
    const int qi = blockIdx.x * MF_BLOCK + threadIdx.x;
    const bool active = qi < n;
    const int qq = active ? qi : n - 1;
    float v[PIN_GN_NSUMS];
#pragma unroll
    for (int i = 0; i < PIN_GN_NSUMS; ++i) v[i] = 0.f;
    const float px = query[3 * qq], py = query[3 * qq + 1], pz = query[3 * qq + 2];
    const SdfResult r = eval_query<H, WF, true, true>(f, nbr, nn_count, qq, px, py, pz, lds, xb);
    // colour term (tracker.py:493-542): the colour decoder reuses the LDS image after the SDF pass
    float ipred = 0.f, igx = 0.f, igy = 0.f, igz = 0.f;
    if (ct.mode != 0) {
        __syncthreads();
        MfmaDecoder<H>::stage(ct.fc.dec, ct.fc.levels, lds, threadIdx.x, MF_BLOCK, 3);
        __syncthreads();
        Kappa kap;
        kap.k[0] = 0.299f; kap.k[1] = 0.587f; kap.k[2] = 0.114f;  // color_to_intensity, tools.py:408
        SdfResult rc;
        if (ct.mode == 2) rc = eval_query<H, WF, true, true, 3>(ct.fc, nbr, nn_count, qq, px, py, pz, lds, xb, kap);
        else rc = eval_query<H, WF, false, true, 3>(ct.fc, nbr, nn_count, qq, px, py, pz, lds, xb, kap);
        ipred = rc.sdf; igx = rc.gx; igy = rc.gy; igz = rc.gz;
    }
    if (active) {
        if (sdf_out) sdf_out[qi] = r.sdf;
        if (grad_out) { grad_out[3 * qi] = r.gx; grad_out[3 * qi + 1] = r.gy; grad_out[3 * qi + 2] = r.gz; }
        const float gn = sqrtf(r.gx * r.gx + r.gy * r.gy + r.gz * r.gz);
        const bool valid = nn_count[qi] >= gp.valid_nn_k && gn < gp.max_grad_norm && gn > gp.min_grad_norm &&
                           r.std < gp.max_sdf_std;
        if (valid) {
            const float res = (gp.dist_div_grad_norm ? r.sdf / gn : r.sdf) - (labels ? labels[qi] : 0.f);
            float w = 1.f;
            if (gp.gm_grad > 0.f) { const float a = gn - 1.f; const float t = gp.gm_grad / (gp.gm_grad + a * a); w *= t * t; }
            if (gp.gm_dist > 0.f) { const float t = gp.gm_dist / (gp.gm_dist + res * res); w *= t * t; }
            float cres = 0.f;
            if (ct.mode != 0) {
                const float* cm = ct.colors + 3 * (size_t)qi;
                const float imeas = 0.299f * cm[0] + 0.587f * cm[1] + 0.114f * cm[2];
                cres = ipred - imeas;
                if (ct.mode == 1) w *= expf(-fabsf(cres));  // consistency weight (tracker.py:509-514)
            }
            float J[6];
            J[0] = py * r.gz - pz * r.gy; J[1] = pz * r.gx - px * r.gz; J[2] = px * r.gy - py * r.gx;
            J[3] = r.gx; J[4] = r.gy; J[5] = r.gz;
            int o = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) v[o++] = w * J[a] * J[b];
#pragma unroll
            for (int a = 0; a < 6; ++a) v[21 + a] = w * J[a] * res;
            v[27] = w; v[28] = fabsf(res); v[29] = 1.f; v[30] = w * res * res;
            if (ct.mode == 2) {  // implicit_color_reg (tracker.py:699-744): + w_photo * Jc^T W Jc, Jc^T W rc
                float Jc[6];
                Jc[0] = py * igz - pz * igy; Jc[1] = pz * igx - px * igz; Jc[2] = px * igy - py * igx;
                Jc[3] = igx; Jc[4] = igy; Jc[5] = igz;
                const float wp = w * ct.photo_weight;
                o = 0;
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int b = a; b < 6; ++b) v[o++] += wp * Jc[a] * Jc[b];
#pragma unroll
                for (int a = 0; a < 6; ++a) v[21 + a] += wp * Jc[a] * cres;
                v[31] = fabsf(cres);
            }
        }
    }
    // block reduction first (4 waves -> one set of atomics): same-address f64 atomics serialise in L2
    __shared__ double red[MF_BLOCK / 64][PIN_GN_NSUMS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < PIN_GN_NSUMS; ++i) {
        const double t = (double)wave_sum_f32(v[i]);  // the reference sums these in float32 (torch mm)
        if (lane == 0) red[wave][i] = t;
    }
    __syncthreads();
    if (threadIdx.x < PIN_GN_NSUMS) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < MF_BLOCK / 64; ++w) t += red[w][threadIdx.x];
        if (t != 0.0) atomicAdd(sums + (size_t)(blockIdx.x % GN_REPLICAS) * PIN_GN_NSUMS + threadIdx.x, t);
    }
}


}  // namespace pin
#include "gn_solve.h"
#include "gn_quad.h"
#include "sdf_quad.h"
namespace pin {

// ---- device-side normal-equation solve + loop control (one wave) ---------------------------
// implicit_reg (utils/tracker.py:656-679) and the bookkeeping of Tracker.tracking (:147-184).
// (the system spread over the lanes of the wave: gn_solve.h; PIN_GN_SOLVE=seq launches the r01-r05 form below for A/B runs)
__global__ __launch_bounds__(64) void gn_solve_kernel(double* __restrict__ sums, double* __restrict__ st,
                                                      pin_gn_loop_params lp, const int* __restrict__ status) {
    gn_solve_wave<false>(sums, st, lp, status);
}

__global__ __launch_bounds__(64) void gn_solve_seq_kernel(double* __restrict__ sums, double* __restrict__ st,
                                                          pin_gn_loop_params lp, const int* __restrict__ status) {
    __shared__ double s[PIN_GN_NSUMS];
    const int lane = threadIdx.x;
    // this kernel is a chain of memory round trips around ~1 us of arithmetic: everything it reads is requested up
    // front (the stop flag, the sums, the loop state the end of the kernel needs)
    const double done = st[PIN_GN_STATE_DONE];
    const double st_pre = st[lane < PIN_GN_STATE_MSE ? lane : 0];  // pose (0..15) and loop scalars (16..22), one per lane
    const double nsrc = st[PIN_GN_STATE_NSRC];
    {   // replica sum: two lanes per sum, GN_REPLICAS / 2 independent loads each
        const int i = lane & 31, h = lane >> 5;
        double a = 0.0;
#pragma unroll 8
        for (int r = 0; r < GN_REPLICAS / 2; ++r) a += sums[(h * (GN_REPLICAS / 2) + r) * PIN_GN_NSUMS + i];
        a += __shfl_xor(a, 32, 64);
        if (done != 0.0) return;  // (uniform)
        if (h == 0) s[i] = a;
    }
    __shared__ double stv[PIN_GN_STATE_MSE];
    if (lane < PIN_GN_STATE_MSE) stv[lane] = st_pre;
    __syncthreads();
    for (int i = lane; i < GN_REPLICAS * PIN_GN_NSUMS; i += 64) sums[i] = 0.0;  // ready for the next iteration
    if (lane != 0) return;
    if (status != nullptr) st[PIN_GN_STATE_STATUS] = (double)*status;  // (sticky flags of the library, e.g. a decoder outside the fp16 range)
    const double cnt = rint(s[29]);
    double dT[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double res_cm = 0.0;
    if (cnt >= 10.0) {  // tracker.py:430-432
        const double scale = cnt / (2.0 * s[27]);  // w /= 2*mean(w)
        // every loop below is fully unrolled with compile-time indices: the 6x7 system lives in registers
        // (dynamic indexing would put it in scratch memory, which also costs at launch)
        double N[6][7];
        {
            int o = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) { N[a][b] = N[b][a] = scale * s[o++]; }
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int b = 0; b < 6; ++b) st[PIN_GN_STATE_NRAW + a * 6 + b] = N[a][b];
            N[a][6] = -scale * s[21 + a];
        }
        st[PIN_GN_STATE_MSE] = scale * s[30] / cnt;
#pragma unroll
        for (int a = 0; a < 6; ++a) N[a][a] += lp.lm_lambda * N[a][a];
        // Gaussian elimination with partial pivoting (float64); the pivot row is brought up by conditional
        // row exchanges (select instructions), no data-dependent indexing
#pragma unroll
        for (int c = 0; c < 6; ++c) {
#pragma unroll
            for (int r = c + 1; r < 6; ++r) {
                const bool sw = fabs(N[r][c]) > fabs(N[c][c]);
#pragma unroll
                for (int b = c; b < 7; ++b) {
                    const double x = N[c][b], y = N[r][b];
                    N[c][b] = sw ? y : x;
                    N[r][b] = sw ? x : y;
                }
            }
            const double inv = 1.0 / N[c][c];
#pragma unroll
            for (int r = c + 1; r < 6; ++r) {
                const double f = N[r][c] * inv;
#pragma unroll
                for (int b = c; b < 7; ++b) N[r][b] -= f * N[c][b];
            }
        }
        double t[6];
#pragma unroll
        for (int r = 5; r >= 0; --r) {
            double acc = N[r][6];
#pragma unroll
            for (int b = r + 1; b < 6; ++b) acc -= N[r][b] * t[b];
            t[r] = acc / N[r][r];
        }
        // expmap (tracker.py:784-795)
        const double ang = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
        const double ax = t[0] / ang, ay = t[1] / ang, az = t[2] / ang;
        const double sn = sin(ang), cs = 1.0 - cos(ang);
        const double S[9] = {0, -az, ay, az, 0, -ax, -ay, ax, 0};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                double ss = 0.0;
#pragma unroll
                for (int c = 0; c < 3; ++c) ss += S[a * 3 + c] * S[c * 3 + b];
                dT[a * 4 + b] = (a == b ? 1.0 : 0.0) + S[a * 3 + b] * sn + ss * cs;
            }
        dT[3] = t[3]; dT[7] = t[4]; dT[11] = t[5];
        res_cm = s[28] / cnt * 100.0;
    }
    // T = dT @ T
    double Tn[16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < 4; ++c) acc += dT[a * 4 + c] * stv[c * 4 + b];
            Tn[a * 4 + b] = acc;
        }
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = Tn[i];
    st[PIN_GN_STATE_RES] = res_cm;
    st[PIN_GN_STATE_CNT] = cnt;
    const double last = stv[PIN_GN_STATE_LAST_RES];
    bool valid = stv[PIN_GN_STATE_VALID] != 0.0;
    if ((res_cm - last) / last > lp.max_increment_ratio) valid = false;  // tracker.py:150-159
    else st[PIN_GN_STATE_LAST_RES] = res_cm;
    if (cnt < lp.min_valid_points || cnt / nsrc < lp.min_valid_ratio) valid = false;  // :161-169
    st[PIN_GN_STATE_VALID] = valid ? 1.0 : 0.0;
    const int i = (int)stv[PIN_GN_STATE_ITERS];
    st[PIN_GN_STATE_ITERS] = i + 1;
    const bool converged = stv[PIN_GN_STATE_CONVERGED] != 0.0;
    if (!valid || converged || i + 1 >= lp.iter_n) { st[PIN_GN_STATE_DONE] = 1.0; return; }  // :171-172
    const double rot_deg = acos((dT[0] + dT[5] + dT[10] - 1.0) / 2.0) * 180.0 / 3.14159265358979323846;
    const double tran = sqrt(dT[3] * dT[3] + dT[7] * dT[7] + dT[11] * dT[11]);
    if ((lp.early_exit && fabs(rot_deg) < lp.term_thre_deg && tran < lp.term_thre_m) || i == lp.iter_n - 2)
        st[PIN_GN_STATE_CONVERGED] = 1.0;  // :179-184
}

__global__ void gn_state_init_kernel(double* st, int n_src) {
    const int i = threadIdx.x;
    if (i >= 16 && i < PIN_GN_STATE_DOUBLES) st[i] = 0.0;
    __syncthreads();
    if (i == 0) { st[PIN_GN_STATE_LAST_RES] = 1e5; st[PIN_GN_STATE_VALID] = 1.0; st[PIN_GN_STATE_NSRC] = (double)n_src; }
}

// pin_gn_loop_init: the same with the pose by value (no copy in front of the launch) and the loop parameters in the state
struct Pose16 { double m[16]; };
__global__ __launch_bounds__(128) void gn_loop_init_kernel(double* st, Pose16 T, int n_src, pin_gn_loop_params lp, const int* status) {
    const int i = threadIdx.x;
    if (i >= PIN_GN_STATE_DOUBLES) return;
    double v = 0.0;
    if (i < 16) v = T.m[i];
    switch (i) {
        case PIN_GN_STATE_LAST_RES: v = 1e5; break;
        case PIN_GN_STATE_VALID: v = 1.0; break;
        case PIN_GN_STATE_NSRC: v = (double)n_src; break;
        case PIN_GN_STATE_LP + 0: v = lp.lm_lambda; break;
        case PIN_GN_STATE_LP + 1: v = lp.term_thre_deg; break;
        case PIN_GN_STATE_LP + 2: v = lp.term_thre_m; break;
        case PIN_GN_STATE_LP + 3: v = lp.min_valid_ratio; break;
        case PIN_GN_STATE_LP + 4: v = lp.max_increment_ratio; break;
        case PIN_GN_STATE_LP + 5: v = (double)lp.min_valid_points; break;
        case PIN_GN_STATE_LP + 6: v = (double)lp.iter_n; break;
        case PIN_GN_STATE_LP + 7: v = (double)lp.early_exit; break;
        case PIN_GN_STATE_STATUS_PTR: v = __longlong_as_double((long long)(unsigned long long)status); break;
        default: break;
    }
    st[i] = v;
}

// ---- tensor-API kernels (Mesher / drop-in query_feature, Decoder.sdf) --------------------
__global__ __launch_bounds__(256) void query_feature_kernel(pin_field f, const float* __restrict__ query,
                                                            const float4* __restrict__ nbr,
                                                            const int* __restrict__ nn_count, int n,
                                                            float* __restrict__ feat_out, float* __restrict__ weight_out,
                                                            float* __restrict__ cert_out, int training,
                                                            float* __restrict__ cert_rw, int* __restrict__ ts_rw,
                                                            const int* __restrict__ query_ts) {
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= n) return;
    Nbrs nb;
    load_neighbors(nbr, nn_count, qi, f.k, nb);
    float z[MLP_IN], Rm[9];
#pragma unroll
    for (int j = 0; j < MLP_IN; ++j) z[j] = 0.f;
    float cert = 0.f;
#pragma unroll
    for (int t = 0; t < PIN_MAX_K; ++t) {
        if (t >= f.k) continue;
        float ft[PIN_FEATURE_DIM] = {0, 0, 0, 0, 0, 0, 0, 0}, v[3] = {0, 0, 0};
        if (nb.idx[t] >= 0) {
            load_feature(f, nb.idx[t], ft);
            neighbor_vector(f, nb.idx[t], nb.quirk[t], nb.vx[t], nb.vy[t], nb.vz[t], query[3 * qi], query[3 * qi + 1],
                            query[3 * qi + 2], v, Rm);
            if (f.certainty) cert = fmaf(f.certainty[nb.idx[t]], nb.w[t], cert);  // pre-scatter values
        }
        if (f.weighted_first) {
#pragma unroll
            for (int j = 0; j < PIN_FEATURE_DIM; ++j) z[j] = fmaf(nb.w[t], ft[j], z[j]);
#pragma unroll
            for (int j = 0; j < 3; ++j) z[8 + j] = fmaf(nb.w[t], v[j], z[8 + j]);
        } else {
            float* o = feat_out + ((size_t)qi * f.k + t) * MLP_IN;
#pragma unroll
            for (int j = 0; j < PIN_FEATURE_DIM; ++j) o[j] = ft[j];
            o[8] = v[0]; o[9] = v[1]; o[10] = v[2];
        }
        weight_out[(size_t)qi * f.k + t] = nb.w[t];
    }
    if (f.weighted_first) {
#pragma unroll
        for (int j = 0; j < MLP_IN; ++j) feat_out[(size_t)qi * MLP_IN + j] = z[j];
    }
    if (cert_out) cert_out[qi] = cert;
    if (training) {  // neural_points.py:685-710
#pragma unroll
        for (int t = 0; t < PIN_MAX_K; ++t)
            if (nb.idx[t] >= 0) {
                atomicAdd(cert_rw + nb.idx[t], nb.w[t]);
                if (ts_rw && query_ts) atomicMax(ts_rw + nb.idx[t], query_ts[qi]);
            }
    }
}

template <int H>
__global__ __launch_bounds__(SDF_BLOCK) void decoder_sdf_kernel(pin_field f, const float* __restrict__ feat, int n,
                                                                float* __restrict__ out) {
    __shared__ float lds[H * SDF_BLOCK];
    const int i = blockIdx.x * SDF_BLOCK + threadIdx.x;
    if (i >= n) return;
    float z[MLP_IN];
#pragma unroll
    for (int j = 0; j < MLP_IN; ++j) z[j] = feat[(size_t)i * MLP_IN + j];
    MlpMasks mk;
    out[i] = f.sdf_scale * mlp_forward<H, SDF_BLOCK>(as_const(f.dec), f.levels, z, lds + threadIdx.x, mk);
}

static int check_field(const pin_field* f) {
    PIN_CHECK_ARG(f != nullptr, "field NULL");
    PIN_CHECK_ARG(f->k >= 1 && f->k <= PIN_MAX_K, "k must be in [1, 8]");
    PIN_CHECK_ARG(f->hidden == 32 || f->hidden == 64, "hidden must be 32 or 64");
    PIN_CHECK_ARG(f->levels >= 1 && f->levels <= MLP_MAX_LEVELS, "levels must be in [1, 4]");
    PIN_CHECK_ARG(f->dec != nullptr, "decoder parameters NULL");
    PIN_CHECK_ARG(f->out_dim == 0 || f->out_dim == 1 || f->out_dim == 3, "out_dim must be 1 (sdf) or 3 (colour)");
    return 0;
}

}  // namespace pin

using namespace pin;

#define PIN_DISPATCH_HW(f, KERNEL, ...)                                                      \
    do {                                                                                     \
        if ((f)->hidden == 64) {                                                             \
            if ((f)->weighted_first) hipLaunchKernelGGL((KERNEL<64, true>), __VA_ARGS__);    \
            else hipLaunchKernelGGL((KERNEL<64, false>), __VA_ARGS__);                       \
        } else {                                                                             \
            if ((f)->weighted_first) hipLaunchKernelGGL((KERNEL<32, true>), __VA_ARGS__);    \
            else hipLaunchKernelGGL((KERNEL<32, false>), __VA_ARGS__);                       \
        }                                                                                    \
    } while (0)

#define PIN_DISPATCH_FIELD(f, KERNEL, n, stream, ...)                                               \
    do {                                                                                            \
        const dim3 grid_(cdiv(n, MF_BLOCK)), block_(MF_BLOCK);                                      \
        PIN_DISPATCH_HW(f, KERNEL##_mfma_kernel, grid_, block_, 0, stream, __VA_ARGS__);            \
    } while (0)

// persistent blocks, one per CU (gn_quad.h)
static int gq_cu_count() {
    static const int n_cu = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n_cu;
}

// pin_gn_accumulate_solve tells the tile kernels that can finish the iteration themselves (gn_solve.h) to do so through this flag
// instead of through five layers of launchers; a launcher that takes it says so in `tl_tail_taken`
static thread_local bool tl_tail = false;
static thread_local bool tl_tail_taken = false;
static int take_tail() {
    if (tl_tail) tl_tail_taken = true;
    return tl_tail ? 1 : 0;
}

template <int H, bool ORIENT, bool SPLIT, int LC, int BLK>
static int launch_quad_blk(const pin_field* f, const pin_gn_params* gp, const float* pts, const float4* nb4,
                           const int32_t* nn_count, const float* labels, int32_t n, double* sums, float* sdf_out,
                           float* grad_out, const double* state, hipStream_t s) {
    constexpr int max_bytes = gq_lds_bytes(QuadDec<H, SPLIT>::bytes(LC > 0 ? LC : MLP_MAX_LEVELS), BLK);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_accumulate_quad_kernel<H, ORIENT, SPLIT, LC, false, BLK>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, max_bytes);
    if (attr != hipSuccess) return fail(-2, "gn tile kernel: cannot reserve %d bytes of LDS: %s", max_bytes, hipGetErrorString(attr));
    const int tiles = cdiv(n, 16);
    // (measured on C3, same box: the search kernel gains 1.5 us per launch from the XCD-aware order, this kernel nothing --
    // 37.7 vs 38.0 us -- so its tiles stay dealt out over all SIMDs unless PIN_XCD_GN=1)
    xcd_mode_init("PIN_XCD_GN", 0);
    const dim3 grid(min(gq_cu_count(), tiles)), block(BLK);  // all CUs, even when there are fewer tiles than waves
    ColorTerm none;
    memset(&none, 0, sizeof(none));
    hipLaunchKernelGGL((gn_accumulate_quad_kernel<H, ORIENT, SPLIT, LC, false, BLK>), grid, block,
                       gq_lds_bytes(QuadDec<H, SPLIT>::bytes(f->levels), BLK), s, *f, *gp, pts, nb4, nn_count, labels, n, sums, sdf_out,
                       grad_out, state, none, take_tail());
    return 0;
}

template <int H, bool ORIENT, bool SPLIT, int LC>
static int launch_quad_inst(const pin_field* f, const pin_gn_params* gp, const float* pts, const float4* nb4,
                            const int32_t* nn_count, const float* labels, int32_t n, double* sums, float* sdf_out,
                            float* grad_out, const double* state, hipStream_t s) {
    // (three waves per SIMD -- 768 threads, 168 registers -- was measured in r03 and is not built: slower, and it spills)
    return launch_quad_blk<H, ORIENT, SPLIT, LC, GQ_BLOCK>(f, gp, pts, nb4, nn_count, labels, n, sums, sdf_out, grad_out, state, s);
}

// the same kernel with the colour term of the registration: two split-fp16 images (sdf, colour) of LC layers each
constexpr int GQ_COLOR_MAX_LEVELS = 2;  // (2 x 42 KB at 2 x 64; three layers each would not fit the 160 KB of a CU)
template <int H, bool ORIENT, int LC>
static int launch_quad_color_inst(const pin_field* f, const pin_gn_params* gp, const ColorTerm& ct, const float* pts,
                                  const float4* nb4, const int32_t* nn_count, const float* labels, int32_t n, double* sums,
                                  float* sdf_out, float* grad_out, const double* state, hipStream_t s) {
    // (after PGO with two 64-wide layers per decoder the tile's state -- both inputs, the rotation sums, two decoders' masks --
    // exceeds 256 registers: that one variant runs one wave per SIMD with the other half of the register file instead of
    // spilling 75 registers to scratch memory)
    constexpr int BLK = (H == 64 && ORIENT && LC == 2) ? 256 : GQ_BLOCK;
    constexpr int lds_bytes = 2 * gq_red_offset(QuadDecoderH<H>::bytes(LC)) + (BLK / 64) * PIN_GN_NSUMS * (int)sizeof(float);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_accumulate_quad_kernel<H, ORIENT, true, LC, true, BLK>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (attr != hipSuccess) return fail(-2, "gn tile kernel: cannot reserve %d bytes of LDS: %s", lds_bytes, hipGetErrorString(attr));
    const dim3 grid(min(gq_cu_count(), cdiv(n, 16))), block(BLK);
    hipLaunchKernelGGL((gn_accumulate_quad_kernel<H, ORIENT, true, LC, true, BLK>), grid, block, lds_bytes, s, *f, *gp, pts, nb4, nn_count,
                       labels, n, sums, sdf_out, grad_out, state, ct, 0);  // (no tail in the colour variants: gn_quad.h)
    return 0;
}

static bool quad_color_ok(const pin_field* f, const ColorTerm& ct) {
    if (!(f->weighted_first && use_split_decoder() && ct.mode != 0 && ct.fc.levels == f->levels && ct.fc.hidden == f->hidden &&
          f->levels <= GQ_COLOR_MAX_LEVELS))
        return false;
    // both decoders staged by the caller (pin_stage_decoder on the sdf and on the colour field): the kernel only copies
    const int64_t bytes = pin_decoder_image_bytes(f->hidden, f->levels);
    return f->dec_image != nullptr && ct.fc.dec_image != nullptr && f->dec_image_bytes == bytes && ct.fc.dec_image_bytes == bytes;
}

static int launch_quad_color(const pin_field* f, const pin_gn_params* gp, const ColorTerm& ct, const float* pts, const float4* nb4,
                             const int32_t* nn_count, const float* labels, int32_t n, double* sums, float* sdf_out,
                             float* grad_out, const double* state, hipStream_t s) {
#define PIN_LQC(HH, OO, LL) \
    return launch_quad_color_inst<HH, OO, LL>(f, gp, ct, pts, nb4, nn_count, labels, n, sums, sdf_out, grad_out, state, s)
#define PIN_LQC_L(HH, OO) do { if (f->levels == 1) PIN_LQC(HH, OO, 1); else PIN_LQC(HH, OO, 2); } while (0)
    if (f->hidden == 64) { if (f->orient) PIN_LQC_L(64, true); else PIN_LQC_L(64, false); }
    if (f->orient) PIN_LQC_L(32, true); else PIN_LQC_L(32, false);
#undef PIN_LQC_L
#undef PIN_LQC
}

template <int H, bool ORIENT>
static int launch_quad_ho(const pin_field* f, const pin_gn_params* gp, const float* pts, const float4* nb4,
                          const int32_t* nn_count, const float* labels, int32_t n, double* sums, float* sdf_out,
                          float* grad_out, const double* state, hipStream_t s) {
#define PIN_LQ(BB, LL) \
    return launch_quad_inst<H, ORIENT, BB, LL>(f, gp, pts, nb4, nn_count, labels, n, sums, sdf_out, grad_out, state, s)
    if (!use_split_decoder()) PIN_LQ(false, 0);  // PIN_MLP=f32: the fp32 MFMA image (A/B runs)
    switch (f->levels) {
        case 1: PIN_LQ(true, 1);
        case 2: PIN_LQ(true, 2);
        case 3: PIN_LQ(true, 3);
        default: PIN_LQ(true, 4);
    }
#undef PIN_LQ
}

// weighted_first = False: a decoder column per (query, neighbour) pair, 2 queries per 16-column tile (gn_quad.h)
template <int H, bool ORIENT, bool SPLIT, int LC>
static int launch_quad_nwf_inst(const pin_field* f, const pin_gn_params* gp, const float* pts, const float4* nb4,
                                const int32_t* nn_count, const float* labels, int32_t n, double* sums, float* sdf_out,
                                float* grad_out, const double* state, hipStream_t s) {
    constexpr int lds_bytes = gq_red_offset(QuadDec<H, SPLIT>::bytes(1)) + (NWF_BLOCK / 64) * PIN_GN_NSUMS * (int)sizeof(float);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_accumulate_quad_nwf_kernel<H, ORIENT, SPLIT, LC>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (attr != hipSuccess) return fail(-2, "gn tile kernel: cannot reserve %d bytes of LDS: %s", lds_bytes, hipGetErrorString(attr));
    const int tiles = cdiv(n, 2);
    const dim3 grid(min(gq_cu_count(), cdiv(tiles, NWF_BLOCK / 64))), block(NWF_BLOCK);
    hipLaunchKernelGGL((gn_accumulate_quad_nwf_kernel<H, ORIENT, SPLIT, LC>), grid, block, lds_bytes, s, *f, *gp, pts, nb4, nn_count,
                       labels, n, sums, sdf_out, grad_out, state, (float*)nullptr, (float*)nullptr, take_tail());
    return 0;
}

// one H-wide layer only (the class default 1x64 that every shipped weighted_first = False config uses); deeper
// decoders with per-neighbour decoding stay on the 64-queries-per-wave kernel
static int launch_quad_nwf(const pin_field* f, const pin_gn_params* gp, const float* pts, const float4* nb4,
                           const int32_t* nn_count, const float* labels, int32_t n, double* sums, float* sdf_out,
                           float* grad_out, const double* state, hipStream_t s) {
#define PIN_LQ(HH, OO) \
    return use_split_decoder() ? launch_quad_nwf_inst<HH, OO, true, 1>(f, gp, pts, nb4, nn_count, labels, n, sums, sdf_out, grad_out, state, s) \
                             : launch_quad_nwf_inst<HH, OO, false, 0>(f, gp, pts, nb4, nn_count, labels, n, sums, sdf_out, grad_out, state, s)
    if (f->hidden == 64) { if (f->orient) PIN_LQ(64, true); else PIN_LQ(64, false); }
    if (f->orient) PIN_LQ(32, true); else PIN_LQ(32, false);
#undef PIN_LQ
}

static int launch_quad(const pin_field* f, const pin_gn_params* gp, const float* pts, const float4* nb4,
                       const int32_t* nn_count, const float* labels, int32_t n, double* sums, float* sdf_out,
                       float* grad_out, const double* state, hipStream_t s) {
    if (f->hidden == 64) {
        if (f->orient) return launch_quad_ho<64, true>(f, gp, pts, nb4, nn_count, labels, n, sums, sdf_out, grad_out, state, s);
        return launch_quad_ho<64, false>(f, gp, pts, nb4, nn_count, labels, n, sums, sdf_out, grad_out, state, s);
    }
    if (f->orient) return launch_quad_ho<32, true>(f, gp, pts, nb4, nn_count, labels, n, sums, sdf_out, grad_out, state, s);
    return launch_quad_ho<32, false>(f, gp, pts, nb4, nn_count, labels, n, sums, sdf_out, grad_out, state, s);
}

static int launch_gn(const pin_field* f, const pin_gn_params* gp, const pin_color_term* color, const float* pts,
                     const float* nbr, const int32_t* nn_count, const float* labels, int32_t n, double* sums,
                     float* sdf_out, float* grad_out, const double* state, hipStream_t s) {
    ColorTerm ct;
    memset(&ct, 0, sizeof(ct));
    if (color != nullptr && color->mode != 0) {
        PIN_CHECK_ARG(color->field && color->colors, "colour term: field / colors NULL");
        if (int e = check_field(color->field)) return e;
        PIN_CHECK_ARG(color->field->out_dim == 3 && color->field->hidden == f->hidden && color->field->k == f->k &&
                          color->field->weighted_first == f->weighted_first,
                      "colour decoder must have 3 heads and the sdf decoder's hidden width / k / weighting mode");
        ct.fc = *color->field;
        ct.colors = color->colors;
        ct.mode = color->mode;
        ct.photo_weight = color->photo_weight;
    }
    const float4* nb4 = reinterpret_cast<const float4*>(nbr);
    if (f->weighted_first && ct.mode == 0) {  // four lanes per query, persistent blocks (gn_quad.h)
        if (int e = launch_quad(f, gp, pts, nb4, nn_count, labels, n, sums, sdf_out, grad_out, state, s)) return e;
    } else if (quad_color_ok(f, ct)) {  // the same with the colour term (shallow decoders: both images in LDS)
        if (int e = launch_quad_color(f, gp, ct, pts, nb4, nn_count, labels, n, sums, sdf_out, grad_out, state, s)) return e;
    } else if (!f->weighted_first && ct.mode == 0 && f->levels == 1) {  // per-neighbour decoding: a column per (query, neighbour) pair
        if (int e = launch_quad_nwf(f, gp, pts, nb4, nn_count, labels, n, sums, sdf_out, grad_out, state, s)) return e;
    } else {  // colour term with deeper decoders, per-neighbour decoding with a deeper decoder: 64 queries per wave
        const dim3 grid(cdiv(n, MF_BLOCK)), block(MF_BLOCK);
        PIN_DISPATCH_HW(f, gn_accumulate_mfma_kernel, grid, block, 0, s, *f, *gp, pts, nb4, nn_count, labels, n, sums,
                        sdf_out, grad_out, state, ct);
    }
    return 0;
}

// pin_sdf_query on the tile decoder (sdf_quad.h): interpolate-first, one head, split-fp16 image
template <int H, bool ORIENT, int LC, bool GRAD, int OD>
static int launch_query_quad_inst(const pin_field* f, const float* query, const float4* nb4, const int32_t* nn_count, int32_t n,
                                  float* sdf_out, float* grad_out, float* std_out, float* cert_out, const Kappa& kap,
                                  float* color_out, hipStream_t s) {
    constexpr int lds_bytes = gq_red_offset(QuadDecoderH<H>::bytes(LC));
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&sdf_query_quad_kernel<H, ORIENT, LC, GRAD, OD>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (attr != hipSuccess) return fail(-2, "query tile kernel: cannot reserve %d bytes of LDS: %s", lds_bytes, hipGetErrorString(attr));
    constexpr int SQ_BLOCK = sq_block<GRAD>();
    const dim3 grid(min(gq_cu_count(), cdiv(cdiv(n, 16), SQ_BLOCK / 64))), block(SQ_BLOCK);
    hipLaunchKernelGGL((sdf_query_quad_kernel<H, ORIENT, LC, GRAD, OD>), grid, block, lds_bytes, s, *f, query, nb4, nn_count, n,
                       sdf_out, grad_out, std_out, cert_out, kap, color_out);
    return 0;
}

template <int H, bool ORIENT, bool GRAD, int OD>
static int launch_query_quad_l(const pin_field* f, const float* query, const float4* nb4, const int32_t* nn_count, int32_t n,
                               float* sdf_out, float* grad_out, float* std_out, float* cert_out, const Kappa& kap,
                               float* color_out, hipStream_t s) {
#define PIN_LQQ(LL) \
    return launch_query_quad_inst<H, ORIENT, LL, GRAD, OD>(f, query, nb4, nn_count, n, sdf_out, grad_out, std_out, cert_out, kap, color_out, s)
    switch (f->levels) {
        case 1: PIN_LQQ(1);
        case 2: PIN_LQQ(2);
        case 3: PIN_LQQ(3);
        default: PIN_LQQ(4);
    }
#undef PIN_LQQ
}

// OD = 1: pin_sdf_query; OD = 3: pin_color_query (value -> sdf_out)
template <int OD>
static int launch_query_quad(const pin_field* f, const float* query, const float4* nb4, const int32_t* nn_count, int32_t n,
                             float* sdf_out, float* grad_out, float* std_out, float* cert_out, const Kappa& kap, float* color_out,
                             hipStream_t s) {
#define PIN_LQQ(HH, OO)                                                                                                          \
    do {                                                                                                                         \
        if (grad_out != nullptr)                                                                                                 \
            return launch_query_quad_l<HH, OO, true, OD>(f, query, nb4, nn_count, n, sdf_out, grad_out, std_out, cert_out, kap, color_out, s); \
        return launch_query_quad_l<HH, OO, false, OD>(f, query, nb4, nn_count, n, sdf_out, grad_out, std_out, cert_out, kap, color_out, s);    \
    } while (0)
    if (f->hidden == 64) { if (f->orient) PIN_LQQ(64, true); else PIN_LQQ(64, false); }
    if (f->orient) PIN_LQQ(32, true); else PIN_LQQ(32, false);
#undef PIN_LQQ
}

// weighted_first = False with the one-layer decoder every shipped configuration of that mode uses: the query modes of the
// per-neighbour tile kernel (gn_quad.h)
template <int H, bool ORIENT, int MODE>
static int launch_query_nwf_inst(const pin_field* f, const float* query, const float4* nb4, const int32_t* nn_count, int32_t n,
                                 float* sdf_out, float* grad_out, float* std_out, float* cert_out, hipStream_t s) {
    constexpr int lds_bytes = gq_red_offset(QuadDec<H, true>::bytes(1)) + (NWF_BLOCK / 64) * PIN_GN_NSUMS * (int)sizeof(float);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_accumulate_quad_nwf_kernel<H, ORIENT, true, 1, MODE>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (attr != hipSuccess) return fail(-2, "query tile kernel: cannot reserve %d bytes of LDS: %s", lds_bytes, hipGetErrorString(attr));
    const int tiles = cdiv(n, 2);
    const dim3 grid(min(gq_cu_count(), cdiv(tiles, NWF_BLOCK / 64))), block(NWF_BLOCK);
    pin_gn_params none;
    memset(&none, 0, sizeof(none));
    hipLaunchKernelGGL((gn_accumulate_quad_nwf_kernel<H, ORIENT, true, 1, MODE>), grid, block, lds_bytes, s, *f, none, query, nb4, nn_count,
                       (const float*)nullptr, n, (double*)nullptr, sdf_out, grad_out, (const double*)nullptr, std_out, cert_out);
    return 0;
}

static int launch_query_nwf(const pin_field* f, const float* query, const float4* nb4, const int32_t* nn_count, int32_t n,
                            float* sdf_out, float* grad_out, float* std_out, float* cert_out, hipStream_t s) {
#define PIN_LQN(HH, OO)                                                                                                                \
    do {                                                                                                                               \
        if (grad_out != nullptr) return launch_query_nwf_inst<HH, OO, 1>(f, query, nb4, nn_count, n, sdf_out, grad_out, std_out, cert_out, s); \
        return launch_query_nwf_inst<HH, OO, 2>(f, query, nb4, nn_count, n, sdf_out, grad_out, std_out, cert_out, s);                 \
    } while (0)
    if (f->hidden == 64) { if (f->orient) PIN_LQN(64, true); else PIN_LQN(64, false); }
    if (f->orient) PIN_LQN(32, true); else PIN_LQN(32, false);
#undef PIN_LQN
}

// PIN_QUERY_QUAD=0: the thread-per-query kernel everywhere (A/B runs)
static bool query_quad_on() {
    static const bool on = [] { const char* e = getenv("PIN_QUERY_QUAD"); return !(e && e[0] == '0'); }();
    return on;
}

extern "C" int pin_sdf_query(const pin_field* f, const float* query, const float* nbr, const int32_t* nn_count,
                             int32_t n, float* sdf_out, float* grad_out, float* std_out, float* certainty_out,
                             void* stream) {
    PIN_ENTER();
    if (int e = check_field(f)) return e;
    PIN_CHECK_ARG(n >= 0, "n < 0");
    if (n == 0) return 0;
    PIN_CHECK_ARG(query && nbr && nn_count && f->feats, "NULL pointer");
    if (f->weighted_first && f->out_dim <= 1 && use_split_decoder() && query_quad_on()) {  // four lanes per query (sdf_quad.h)
        if (int e = launch_query_quad<1>(f, query, reinterpret_cast<const float4*>(nbr), nn_count, n, sdf_out, grad_out, std_out,
                                         certainty_out, Kappa(), nullptr, as_stream(stream)))
            return e;
        PIN_CHECK_LAUNCH();
        return 0;
    }
    if (!f->weighted_first && f->levels == 1 && f->out_dim <= 1 && use_split_decoder() && query_quad_on()) {  // a column per (query, neighbour)
        if (int e = launch_query_nwf(f, query, reinterpret_cast<const float4*>(nbr), nn_count, n, sdf_out, grad_out, std_out,
                                     certainty_out, as_stream(stream)))
            return e;
        PIN_CHECK_LAUNCH();
        return 0;
    }
    PIN_DISPATCH_FIELD(f, sdf_query, n, as_stream(stream), *f, query, reinterpret_cast<const float4*>(nbr), nn_count, n,
                       sdf_out, grad_out, std_out, certainty_out);
    PIN_CHECK_LAUNCH();
    return 0;
}

// Decoder.regress_color on caller-provided features: sigmoid(mlp) for the 3 colour heads
template <int H>
__global__ __launch_bounds__(MF_BLOCK, 2) void decoder_color_mfma_kernel(pin_field f, const float* __restrict__ feat, int n,
                                                                         float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[MfmaLds<H>::TOTAL];
    float* xb = lds + MfmaLds<H>::W + (threadIdx.x >> 6) * MfmaDecoder<H>::scratch_floats();
    MfmaDecoder<H>::stage(f.dec, f.levels, lds, threadIdx.x, MF_BLOCK, 3);
    __syncthreads();
    const int i = blockIdx.x * MF_BLOCK + threadIdx.x;
    const int ii = i < n ? i : n - 1;
    float z[MLP_IN], a[MLP_IN], p[3];
#pragma unroll
    for (int j = 0; j < MLP_IN; ++j) z[j] = feat[(size_t)ii * MLP_IN + j];
    const float kk[3] = {0.f, 0.f, 0.f};
    MfmaDecoder<H>::template run_heads<false, 2, 3>(lds, f.levels, xb, z, kk, p, a);
    if (i < n) { out[3 * (size_t)i] = p[0]; out[3 * (size_t)i + 1] = p[1]; out[3 * (size_t)i + 2] = p[2]; }
}

extern "C" int pin_decoder_color(const pin_field* f, const float* feat_in, int32_t n, float* color_out, void* stream) {
    PIN_ENTER();
    if (int e = check_field(f)) return e;
    PIN_CHECK_ARG(f->out_dim == 3, "colour decoder must have 3 output heads");
    PIN_CHECK_ARG(n >= 0, "n < 0");
    if (n == 0) return 0;
    PIN_CHECK_ARG(feat_in && color_out, "NULL pointer");
    const dim3 grid(cdiv(n, MF_BLOCK)), block(MF_BLOCK);
    if (f->hidden == 64) hipLaunchKernelGGL(decoder_color_mfma_kernel<64>, grid, block, 0, as_stream(stream), *f, feat_in, n, color_out);
    else hipLaunchKernelGGL(decoder_color_mfma_kernel<32>, grid, block, 0, as_stream(stream), *f, feat_in, n, color_out);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_color_query(const pin_field* fc, const float* query, const float* nbr, const int32_t* nn_count,
                               int32_t n, const float* kappa_host, float* color_out, float* value_out, float* grad_out,
                               void* stream) {
    PIN_ENTER();
    if (int e = check_field(fc)) return e;
    PIN_CHECK_ARG(fc->out_dim == 3, "colour field must have 3 output heads");
    PIN_CHECK_ARG(n >= 0, "n < 0");
    if (n == 0) return 0;
    PIN_CHECK_ARG(query && nbr && nn_count && fc->feats && kappa_host, "NULL pointer");
    Kappa kap;
    for (int c = 0; c < 3; ++c) kap.k[c] = kappa_host[c];
    const dim3 grid(cdiv(n, MF_BLOCK)), block(MF_BLOCK);
    const float4* nb4 = reinterpret_cast<const float4*>(nbr);
    hipStream_t s = as_stream(stream);
    if (fc->weighted_first && use_split_decoder() && query_quad_on()) {  // four lanes per query (sdf_quad.h)
        if (int e = launch_query_quad<3>(fc, query, nb4, nn_count, n, value_out, grad_out, nullptr, nullptr, kap, color_out, s)) return e;
        PIN_CHECK_LAUNCH();
        return 0;
    }
#define PIN_COLOR_Q(HH, WFV) \
    hipLaunchKernelGGL((sdf_query_mfma_kernel<HH, WFV, 3>), grid, block, 0, s, *fc, query, nb4, nn_count, n, value_out, \
                       grad_out, (float*)nullptr, (float*)nullptr, kap, color_out)
    if (fc->hidden == 64) { if (fc->weighted_first) PIN_COLOR_Q(64, true); else PIN_COLOR_Q(64, false); }
    else { if (fc->weighted_first) PIN_COLOR_Q(32, true); else PIN_COLOR_Q(32, false); }
#undef PIN_COLOR_Q
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_gn_accumulate(const pin_field* f, const pin_gn_params* gp, const pin_color_term* color,
                                 const float* query, const float* nbr, const int32_t* nn_count, const float* sdf_labels,
                                 int32_t n, double* sums_out, float* sdf_out, float* grad_out, void* stream) {
    PIN_ENTER();
    if (int e = check_field(f)) return e;
    PIN_CHECK_ARG(gp && sums_out, "NULL pointer");
    PIN_CHECK_ARG(n >= 0, "n < 0");
    hipStream_t s = as_stream(stream);
    PIN_CHECK_HIP(hipMemsetAsync(sums_out, 0, sizeof(double) * PIN_GN_NSUMS * GN_REPLICAS, s));
    if (n == 0) return 0;
    PIN_CHECK_ARG(query && nbr && nn_count && f->feats, "NULL pointer");
    if (int e = launch_gn(f, gp, color, query, nbr, nn_count, sdf_labels, n, sums_out, sdf_out, grad_out, nullptr, s)) return e;
    PIN_CHECK_LAUNCH();
    return 0;
}

// States that hold their loop parameters (pin_gn_loop_init), by address: pin_gn_accumulate_solve lets the tile kernel finish the
// iteration only on one of these and only with the parameters it holds.  A handful of trackers per process: a short table.
namespace {
struct LoopState { const double* state; pin_gn_loop_params lp; };
std::mutex g_loop_mutex;
LoopState g_loop_states[8];
int g_loop_next = 0;
void loop_state_forget(const double* state) {
    std::lock_guard<std::mutex> lock(g_loop_mutex);
    for (auto& e : g_loop_states) if (e.state == state) e.state = nullptr;
}
void loop_state_remember(const double* state, const pin_gn_loop_params& lp) {
    std::lock_guard<std::mutex> lock(g_loop_mutex);
    LoopState* slot = nullptr;
    for (auto& e : g_loop_states) if (e.state == state) slot = &e;
    if (slot == nullptr) { slot = &g_loop_states[g_loop_next]; g_loop_next = (g_loop_next + 1) % 8; }
    slot->state = state;
    slot->lp = lp;
}
bool same_lp(const pin_gn_loop_params& a, const pin_gn_loop_params& b) {
    return a.lm_lambda == b.lm_lambda && a.term_thre_deg == b.term_thre_deg && a.term_thre_m == b.term_thre_m &&
           a.min_valid_ratio == b.min_valid_ratio && a.max_increment_ratio == b.max_increment_ratio &&
           a.min_valid_points == b.min_valid_points && a.iter_n == b.iter_n && a.early_exit == b.early_exit;
}
}  // namespace
static bool loop_state_holds(const double* state, const pin_gn_loop_params* lp) {
    std::lock_guard<std::mutex> lock(g_loop_mutex);
    for (const auto& e : g_loop_states) if (e.state == state && e.state != nullptr) return same_lp(e.lp, *lp);
    return false;
}

extern "C" int pin_gn_loop_init(double* state, const double* T_init_host, int32_t n_src, const pin_gn_loop_params* lp, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(state && T_init_host && lp && n_src > 0, "bad arguments");
    Pose16 T;
    memcpy(T.m, T_init_host, sizeof(T.m));
    hipLaunchKernelGGL(gn_loop_init_kernel, dim3(1), dim3(128), 0, as_stream(stream), state, T, n_src, *lp, (const int*)status_word());
    PIN_CHECK_LAUNCH();
    loop_state_remember(state, *lp);
    return 0;
}

extern "C" int pin_gn_state_init(double* state, const double* T_init_host, int32_t n_src, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(state && T_init_host && n_src > 0, "bad arguments");
    loop_state_forget(state);
    hipStream_t s = as_stream(stream);
    PIN_CHECK_HIP(hipMemcpyAsync(state, T_init_host, 16 * sizeof(double), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(gn_state_init_kernel, dim3(1), dim3(64), 0, s, state, n_src);
    PIN_CHECK_LAUNCH();
    return 0;
}

// ---- decoder image for the GN tile kernel (pin_field.dec_image) ------------------------------------------------
template <int H>
__global__ __launch_bounds__(GQ_BLOCK) void stage_decoder_kernel(pin_field f, unsigned char* __restrict__ out, int* __restrict__ status) {
    const int od = f.out_dim > 1 ? f.out_dim : 1;
    // (every loop of stage() strides over (thread, thread count) and no thread waits for another: the blocks of the launch share
    // the slots -- one memory round trip per thread instead of four at one block: 18.6 -> 6 us, twice per frame)
    const int tid = blockIdx.x * GQ_BLOCK + threadIdx.x, nthreads = gridDim.x * GQ_BLOCK;
    QuadDecoderH<H>::stage(f.dec, f.levels, out, tid, nthreads, od);  // (1 or 3 heads)
    // Range guard of the split-fp16 image (mlp_h2.h "Range"): a parameter of magnitude >= 65504 (or a non-finite one) has no
    // fp16 high piece -- the tile kernels would turn it into inf / NaN outputs.  The host hears about it through the sticky
    // status word (pin_status, PIN_GN_STATE_STATUS) and raises; PIN_MLP=f32 selects the fp32 image for such a decoder.
    if (status == nullptr) return;
    const int n_dec = H * MLP_IN + H + (f.levels - 1) * (H * H + H) + od * H + od;
    int bad = 0;
    for (int i = tid; i < n_dec; i += nthreads) bad |= !(fabsf(f.dec[i]) < 65504.f);
    if (__syncthreads_or(bad) && threadIdx.x == 0) atomicOr(status, PIN_STATUS_FP16_RANGE);
}

extern "C" int64_t pin_decoder_image_bytes(int32_t hidden, int32_t levels) {
    if (!use_split_decoder() || levels < 1 || levels > MLP_MAX_LEVELS) return 0;
    if (hidden == 64) return QuadDecoderH<64>::bytes(levels);
    if (hidden == 32) return QuadDecoderH<32>::bytes(levels);
    return 0;
}

extern "C" int pin_stage_decoder(const pin_field* f, void* image_out, int64_t image_bytes, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(f && f->dec && image_out, "NULL pointer");
    PIN_CHECK_ARG(image_bytes > 0 && image_bytes == pin_decoder_image_bytes(f->hidden, f->levels),
                  "image size does not match pin_decoder_image_bytes(hidden, levels)");
    PIN_CHECK_ARG(((uintptr_t)image_out & 15) == 0, "image must be 16-byte aligned");
    pin_field g = *f;
    g.dec_image = nullptr;
    g.dec_image_bytes = 0;
    unsigned char* out = reinterpret_cast<unsigned char*>(image_out);
    const dim3 grid(f->levels > 1 ? 8 : 2);
    if (f->hidden == 64) hipLaunchKernelGGL(stage_decoder_kernel<64>, grid, dim3(GQ_BLOCK), 0, as_stream(stream), g, out, status_word());
    else hipLaunchKernelGGL(stage_decoder_kernel<32>, grid, dim3(GQ_BLOCK), 0, as_stream(stream), g, out, status_word());
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_gn_knn(const pin_search_params* sp, const pin_brick_cache* bc, const float* src, int32_t n, int32_t k,
                          const double* state, float* cur_out, float* nbr_out, int32_t* nn_count_out, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(state && cur_out, "state / cur_out NULL");
    if (bc != nullptr) return knn_bricks_dev(sp, bc, src, n, k, state, cur_out, nbr_out, nn_count_out, stream);
    return knn_direct_dev(sp, src, n, k, state, cur_out, nbr_out, nn_count_out, stream);
}

extern "C" int pin_gn_accumulate_dev(const pin_field* f, const pin_gn_params* gp, const pin_color_term* color,
                                    const float* cur, const float* nbr, const int32_t* nn_count, const float* sdf_labels,
                                    int32_t n, double* sums, const double* state, void* stream) {
    PIN_ENTER();
    if (int e = check_field(f)) return e;
    PIN_CHECK_ARG(gp && sums && state && n > 0, "bad arguments");
    PIN_CHECK_ARG(cur && nbr && nn_count && f->feats, "NULL pointer");
    hipStream_t s = as_stream(stream);
    if (int e = launch_gn(f, gp, color, cur, nbr, nn_count, sdf_labels, n, sums, nullptr, nullptr, state, s)) return e;
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_gn_solve(double* sums, double* state, const pin_gn_loop_params* lp, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(sums && state && lp, "NULL pointer");
    static const bool seq = [] { const char* e = getenv("PIN_GN_SOLVE"); return e && e[0] == 's'; }();
    if (seq) hipLaunchKernelGGL(gn_solve_seq_kernel, dim3(1), dim3(64), 0, as_stream(stream), sums, state, *lp, (const int*)status_word());
    else hipLaunchKernelGGL(gn_solve_kernel, dim3(1), dim3(64), 0, as_stream(stream), sums, state, *lp, (const int*)status_word());
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_gn_accumulate_solve(const pin_field* f, const pin_gn_params* gp, const pin_color_term* color,
                                       const pin_gn_loop_params* lp, const float* cur, const float* nbr,
                                       const int32_t* nn_count, const float* sdf_labels, int32_t n, double* sums,
                                       double* state, void* stream) {
    // On a state from pin_gn_loop_init with these parameters the tile kernels finish the iteration in their last block
    // (gn_solve.h: solve, pose, loop state, sums cleared): two launches per iteration instead of three.  The 64-queries-per-wave
    // kernel (deep decoders with a colour term or per-neighbour decoding), other parameters than the state holds, a state from
    // pin_gn_state_init and PIN_GN_FUSE=0 (A/B runs) keep the solve kernel behind the tile kernel.
    static const bool fuse = [] { const char* e = getenv("PIN_GN_FUSE"); return !(e && e[0] == '0'); }();
    PIN_CHECK_ARG(lp != nullptr && state != nullptr, "NULL pointer");
    tl_tail = fuse && loop_state_holds(state, lp);
    tl_tail_taken = false;
    const int e = pin_gn_accumulate_dev(f, gp, color, cur, nbr, nn_count, sdf_labels, n, sums, state, stream);
    tl_tail = false;
    if (e) return e;
    if (tl_tail_taken) return 0;
    return pin_gn_solve(sums, state, lp, stream);
}

extern "C" int pin_query_feature(const pin_field* f, const float* query, const float* nbr, const int32_t* nn_count, int32_t n,
                                 float* feat_out, float* weight_out, float* certainty_out, int32_t training,
                                 float* certainty_rw, int32_t* ts_update_rw, const int32_t* query_ts, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(f != nullptr && f->k >= 1 && f->k <= PIN_MAX_K, "bad field");
    PIN_CHECK_ARG(n >= 0, "n < 0");
    if (n == 0) return 0;
    PIN_CHECK_ARG(query && nbr && nn_count && feat_out && weight_out && f->feats, "NULL pointer");
    PIN_CHECK_ARG(!training || certainty_rw, "training mode needs certainty_rw");
    hipLaunchKernelGGL(query_feature_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), *f, query,
                       reinterpret_cast<const float4*>(nbr), nn_count, n, feat_out, weight_out, certainty_out, training,
                       certainty_rw, ts_update_rw, query_ts);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_decoder_sdf(const pin_field* f, const float* feat_in, int32_t n, float* sdf_out, void* stream) {
    PIN_ENTER();
    if (int e = check_field(f)) return e;
    PIN_CHECK_ARG(n >= 0, "n < 0");
    if (n == 0) return 0;
    PIN_CHECK_ARG(feat_in && sdf_out, "NULL pointer");
    const dim3 grid(cdiv(n, SDF_BLOCK)), block(SDF_BLOCK);
    if (f->hidden == 64) hipLaunchKernelGGL(decoder_sdf_kernel<64>, grid, block, 0, as_stream(stream), *f, feat_in, n, sdf_out);
    else hipLaunchKernelGGL(decoder_sdf_kernel<32>, grid, block, 0, as_stream(stream), *f, feat_in, n, sdf_out);
    PIN_CHECK_LAUNCH();
    return 0;
}

// pin_warmup (common.hip): asking for a kernel's attributes makes the runtime load this translation unit's code object now
// instead of inside the first frame that launches one of its kernels
namespace pin {
int pin_warm_sdf() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&gn_solve_kernel)) == hipSuccess ? 0 : -2;
}
}  // namespace pin
