// Map-training step of Mapper.mapping (utils/mapper.py:645-818) on gfx950:
//   forward (interpolate + decode) over the batch and its 6*n_e central-difference queries
//   (get_numerical_gradient, mapper.py:986-1036), BCE-with-logits + Eikonal loss
//   (utils/loss.py:45-63, mapper.py:732-780), backward to neural-point features (atomic
//   scatter) and decoder parameters, and Adam (utils/tools.py:198-199).
//
// Structure (one C-ABI call = one stream-ordered sequence of launches):
//   weighted_first (class default):  [queries + kNN by the caller] -> train_stage -> train_fused -> train_dw_stream ->
//       train_finalize   (train_fused.h: one tile kernel from the gather to the gradients, decoder on the split-fp16
//       matrix cores, weight gradient streamed over identity-MFMA-transposed operands)
//   per-neighbour decoding (weighted_first = False), one-layer decoder:  train_stage -> train_fused_nwf -> train_dw_stream
//       -> train_finalize
//   per-neighbour decoding, deeper decoders:  train_fwd_mfma -> train_loss -> train_bwd_mfma -> train_dw
//       (this file: 64 queries per wave, fp32 MFMA 16x16x4, activations and layer deltas through a caller-provided
//       workspace laid out unit-major ([row][Q], query contiguous))
// followed by the optimiser kernels (dense / row-flagged / exact lazy Adam).
#include "mlp_quad.h"

namespace pin {


constexpr int DW_SLOTS = 32;            // (train_fused.h)
// pin_train_params.defer_dec_reduce: where the last training call of this thread left the decoder's gradient (pin_train_deferred_partial)
struct DeferredPartial { const float* partial; int64_t n; float scale; };
static thread_local DeferredPartial tl_deferred = {nullptr, 0, 0.f};
constexpr int FUSED_NDEC_MAX = 16384;   // >= parameters of the largest decoder (4 x 64: 13 313)

struct TrainWs {
    float* z;      // [12][Qs]  interpolated decoder input (row 11 unused)
    float* h;      // [L*H][Qs] post-ReLU activations
    float* d;      // [L*H + OD][QsT] deltas; last rows = d loss / d head outputs
    float* pred;   // [OD][Qs]
    float* dpred;  // [OD][Qs]
    float* xraw;   // [OD][QsT] raw head outputs per neighbour sample (colour, per-neighbour decode)
    unsigned long long* mask;  // [L][QsT] ReLU masks of the MFMA decoder (one word per lane and layer)
    int Qs;        // padded query count (multiple of 64)
    int QsT;       // sample columns of z / h / d / mask: Qs (weighted_first) or k * Qs (one decode per neighbour)
};

__host__ __device__ inline size_t train_ws_floats(int Q, int H, int L, int expand) {
    const size_t Qs = (size_t)((Q + 63) / 64) * 64;
    const size_t QsT = Qs * (size_t)expand;
    const size_t unit_major = QsT * (12 + (size_t)L * H + (size_t)L * H + MF_OD_MAX + 2 * (size_t)L + MF_OD_MAX) + 2 * MF_OD_MAX * Qs;
    // operand stream of the fused paths (train_fused.h): two streams of (L * H / 16 + 1) blocks of 1 KiB per tile + slot
    // partials of the weight gradient, per-block loss sums, decoder image.  Tiles: <= Q / 14 + 1 of 16 queries whatever the
    // main / Eikonal split (fused_tiles); per-neighbour decoding: <= Q / 2 + 3 tiles of 2 queries x 8 neighbours
    const size_t tiles = expand == 1 ? (size_t)(Q + 11) / 12 + 4 : (size_t)Q / 2 + 4;
    const size_t stream = tiles * 128 * ((size_t)L * (H / 16) + 1) * 2 * 2 + (size_t)DW_SLOTS * FUSED_NDEC_MAX + 2048 + 32768;
    return unit_major > stream ? unit_major : stream;
}

static void carve_ws(TrainWs& ws, float* w, int H, int L) {
    ws.pred = w; w += (size_t)MF_OD_MAX * ws.Qs;
    ws.dpred = w; w += (size_t)MF_OD_MAX * ws.Qs;
    ws.z = w; w += (size_t)12 * ws.QsT;
    ws.h = w; w += (size_t)L * H * ws.QsT;
    ws.d = w; w += ((size_t)L * H + MF_OD_MAX) * ws.QsT;
    ws.xraw = w; w += (size_t)MF_OD_MAX * ws.QsT;
    ws.mask = reinterpret_cast<unsigned long long*>(w);
}

__global__ void make_queries_kernel(const float* __restrict__ coord, int n_main, int n_eik, int dec, int first,
                                    float eps, float* __restrict__ q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = n_main + 6 * n_eik;
    if (i >= total) return;
    if (i < n_main) {
        q[3 * i] = coord[3 * i]; q[3 * i + 1] = coord[3 * i + 1]; q[3 * i + 2] = coord[3 * i + 2];
        return;
    }
    const int e = i - n_main, s = e / 6, a = e - 6 * s;
    const int src = first + s * dec;  // coord[::dec] of the GLOBAL batch (first = shard phase)
    float x = coord[3 * src], y = coord[3 * src + 1], z = coord[3 * src + 2];
    const float d = (a & 1) ? -eps : eps;  // order x+, x-, y+, y-, z+, z-
    if ((a >> 1) == 0) x += d; else if ((a >> 1) == 1) y += d; else z += d;
    q[3 * i] = x; q[3 * i + 1] = y; q[3 * i + 2] = z;
}

// Mapper.get_batch gathers (utils/mapper.py:482-488): pool rows selected by `index`
__global__ void gather_batch_kernel(const float* __restrict__ pc, const float* __restrict__ pl,
                                    const float* __restrict__ pw, const int* __restrict__ pt,
                                    const int* __restrict__ index, int n, float* __restrict__ coord,
                                    float* __restrict__ label, float* __restrict__ weight, int* __restrict__ ts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = index[i];
    coord[3 * i] = pc[3 * (size_t)s]; coord[3 * i + 1] = pc[3 * (size_t)s + 1]; coord[3 * i + 2] = pc[3 * (size_t)s + 2];
    label[i] = pl[s];
    if (weight) weight[i] = pw ? pw[s] : 1.f;
    if (ts) ts[i] = pt ? pt[s] : 0;
}

// The whole of Mapper.get_batch after its two torch.randint draws (utils/mapper.py:462-500): rows
// index_history[i] for i < n_hist, rows new_idx[index_new_batch[i - n_hist]] after that
__global__ void gather_batch_drawn_kernel(const float* __restrict__ pc, const float* __restrict__ pl,
                                          const float* __restrict__ pw, const int* __restrict__ pt,
                                          const float* __restrict__ pcol, int cw, const long long* __restrict__ index_hist,
                                          int n_hist, const long long* __restrict__ index_new_batch,
                                          const long long* __restrict__ new_idx, int n, float* __restrict__ coord,
                                          float* __restrict__ label, float* __restrict__ weight, int* __restrict__ ts,
                                          float* __restrict__ color, float* __restrict__ q, int n_eik, int dec, int first,
                                          float eps, long hist_stride, long new_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    {   // blockIdx.y = which of the batches of one launch (pin_gather_batches_drawn: the iterations of a mapping call)
        const size_t b = blockIdx.y;
        index_hist += b * hist_stride;
        if (index_new_batch != nullptr) index_new_batch += b * new_stride;
        coord += b * 3 * (size_t)n; label += b * (size_t)n; weight += b * (size_t)n; ts += b * (size_t)n;
        if (color != nullptr) color += b * (size_t)n * cw;
        if (q != nullptr) q += b * 3 * ((size_t)n + 6 * (size_t)n_eik);
    }
    const size_t s = (size_t)(i < n_hist ? index_hist[i] : new_idx[index_new_batch[i - n_hist]]);
    const float x = pc[3 * s], y = pc[3 * s + 1], z = pc[3 * s + 2];
    coord[3 * i] = x; coord[3 * i + 1] = y; coord[3 * i + 2] = z;
    label[i] = pl[s];
    weight[i] = pw[s];
    ts[i] = pt[s];
    for (int c = 0; c < cw; ++c) color[(size_t)i * cw + c] = pcol[s * cw + c];
    if (q == nullptr) return;
    // the training queries of make_queries_kernel in the same pass: the sample itself, and for every dec-th
    // sample the six +-eps probes of the numerical gradient
    q[3 * i] = x; q[3 * i + 1] = y; q[3 * i + 2] = z;
    const int r = i - first;
    if (r >= 0 && r % dec == 0 && r / dec < n_eik) {
        float* e = q + 3 * ((size_t)n + 6 * (size_t)(r / dec));
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const float d = (a & 1) ? -eps : eps;  // order x+, x-, y+, y-, z+, z-
            e[3 * a] = (a >> 1) == 0 ? x + d : x;
            e[3 * a + 1] = (a >> 1) == 1 ? y + d : y;
            e[3 * a + 2] = (a >> 1) == 2 ? z + d : z;
        }
    }
}

// ---- forward -----------------------------------------------------------------------------
struct NbrW {
    float w[PIN_MAX_K];
    int idx[PIN_MAX_K];
};

// weights only (no gradient terms), same arithmetic as sdf.hip load_neighbors
__device__ __forceinline__ void neighbor_weights(const float4* __restrict__ nbr, int nn, int qi, int k, NbrW& nb,
                                                 float (&vx)[PIN_MAX_K], float (&vy)[PIN_MAX_K],
                                                 float (&vz)[PIN_MAX_K], bool (&quirk)[PIN_MAX_K], float* u_out = nullptr) {
    // straight-line: the k record loads are issued together (one memory round trip, not k)
    float4 e[PIN_MAX_K];
#pragma unroll
    for (int t = 0; t < PIN_MAX_K; ++t) e[t] = nbr[(size_t)qi * k + (t < k ? t : 0)];
    float u[PIN_MAX_K];
    float S = 0.f;
#pragma unroll
    for (int t = 0; t < PIN_MAX_K; ++t) {
        const int raw = __float_as_int(e[t].w);
        const bool val = t < k && raw >= 0;
        nb.idx[t] = val ? (raw & ~PIN_NBR_QUIRK_BIT) : -1;
        nb.w[t] = 0.f;
        quirk[t] = val && (raw & PIN_NBR_QUIRK_BIT) != 0;
        vx[t] = val ? e[t].x : 0.f; vy[t] = val ? e[t].y : 0.f; vz[t] = val ? e[t].z : 0.f;
        float ut = val ? 1.0f / (dist2_exact(vx[t], vy[t], vz[t]) + IDW_EPS) : 0.f;
        if (nn == 0 && t < k) ut = IDW_EPS;
        u[t] = ut;
        S += ut;
    }
#pragma unroll
    for (int t = 0; t < PIN_MAX_K; ++t)
        if (nb.idx[t] >= 0) nb.w[t] = u[t] / S;
    if (u_out != nullptr) {  // (the analytic Eikonal term differentiates the weights: u_t of the valid neighbours, then S)
#pragma unroll
        for (int t = 0; t < PIN_MAX_K; ++t) u_out[t] = nb.idx[t] >= 0 ? u[t] : 0.f;
        u_out[PIN_MAX_K] = S;
    }
}

// ---- forward / backward with the decoder on the matrix cores (mlp_mfma.h) -------------------
// one neighbour's decoder input [f_t; v_t]
__device__ __forceinline__ void neighbor_input(const pin_field& f, int idx, bool quirk, float vgx, float vgy, float vgz,
                                               float qx, float qy, float qz, float (&ft)[PIN_FEATURE_DIM], float (&v)[3]) {
    const float4* row = reinterpret_cast<const float4*>(f.feats + (size_t)idx * PIN_FEATURE_DIM);
    const float4 a = row[0], b = row[1];
    ft[0] = a.x; ft[1] = a.y; ft[2] = a.z; ft[3] = a.w; ft[4] = b.x; ft[5] = b.y; ft[6] = b.z; ft[7] = b.w;
    v[0] = vgx; v[1] = vgy; v[2] = vgz;
    if (quirk) {
        const float* p = f.pos + 3 * (size_t)idx;
        v[0] = qx - p[0]; v[1] = qy - p[1]; v[2] = qz - p[2];
    }
    if (f.orient != nullptr) {  // apply_quaternion_rotation (utils/tools.py:428-437): conjugate rotation
        const float4 q4 = reinterpret_cast<const float4*>(f.orient)[idx];
        const float q0_ = q4.x, q1 = q4.y, q2 = q4.z, q3 = q4.w;
        const float r0 = 1 - 2 * (q2 * q2 + q3 * q3), r3 = 2 * (q1 * q2 - q0_ * q3), r6 = 2 * (q1 * q3 + q0_ * q2);
        const float r1 = 2 * (q1 * q2 + q0_ * q3), r4 = 1 - 2 * (q1 * q1 + q3 * q3), r7 = 2 * (q2 * q3 - q0_ * q1);
        const float r2 = 2 * (q1 * q3 - q0_ * q2), r5 = 2 * (q2 * q3 + q0_ * q1), r8 = 1 - 2 * (q1 * q1 + q2 * q2);
        const float x = v[0], y = v[1], zz = v[2];
        v[0] = r0 * x + r1 * y + r2 * zz; v[1] = r3 * x + r4 * y + r5 * zz; v[2] = r6 * x + r7 * y + r8 * zz;
    }
}

// only the relative-position part of neighbor_input (the feature row is read by other lanes)
__device__ __forceinline__ void neighbor_vector_only(const pin_field& f, int idx, bool quirk, float vgx, float vgy, float vgz,
                                                     float qx, float qy, float qz, float (&v)[3]) {
    v[0] = vgx; v[1] = vgy; v[2] = vgz;
    if (quirk) {
        const float* p = f.pos + 3 * (size_t)idx;
        v[0] = qx - p[0]; v[1] = qy - p[1]; v[2] = qz - p[2];
    }
    if (f.orient != nullptr) {  // apply_quaternion_rotation (utils/tools.py:428-437): conjugate rotation
        const float4 q4 = reinterpret_cast<const float4*>(f.orient)[idx];
        const float q0_ = q4.x, q1 = q4.y, q2 = q4.z, q3 = q4.w;
        const float r0 = 1 - 2 * (q2 * q2 + q3 * q3), r3 = 2 * (q1 * q2 - q0_ * q3), r6 = 2 * (q1 * q3 + q0_ * q2);
        const float r1 = 2 * (q1 * q2 + q0_ * q3), r4 = 1 - 2 * (q1 * q1 + q3 * q3), r7 = 2 * (q2 * q3 - q0_ * q1);
        const float r2 = 2 * (q1 * q3 - q0_ * q2), r5 = 2 * (q2 * q3 + q0_ * q1), r8 = 1 - 2 * (q1 * q1 + q2 * q2);
        const float x = v[0], y = v[1], zz = v[2];
        v[0] = r0 * x + r1 * y + r2 * zz; v[1] = r3 * x + r4 * y + r5 * zz; v[2] = r6 * x + r7 * y + r8 * zz;
    }
}

template <int H, bool WF, int OD = 1>
__global__ __launch_bounds__(MF_BLOCK) void train_fwd_mfma_kernel(pin_field f, const float* __restrict__ query,
                                                                  const float4* __restrict__ nbr,
                                                                  const int* __restrict__ nn_count, int Q, int n_main,
                                                                  TrainWs ws, float* __restrict__ cert_rw,
                                                                  int* __restrict__ ts_rw, const int* __restrict__ sample_ts) {
    __shared__ __attribute__((aligned(16))) float lds[MfmaLds<H>::TOTAL];
    float* xb = lds + MfmaLds<H>::W + (threadIdx.x >> 6) * MfmaDecoder<H>::scratch_floats();
    MfmaDecoder<H>::stage(f.dec, f.levels, lds, threadIdx.x, MF_BLOCK, OD);
    __syncthreads();
    const int q0 = (blockIdx.x * (MF_BLOCK / 64) + (threadIdx.x >> 6)) * 64;
    if (q0 >= ws.Qs) return;  // whole wave
    const int qi = q0 + (threadIdx.x & 63);
    const bool active = qi < Q;
    const int qq = active ? qi : Q - 1;
    NbrW nb;
    float vx[PIN_MAX_K], vy[PIN_MAX_K], vz[PIN_MAX_K];
    bool quirk[PIN_MAX_K];
    neighbor_weights(nbr, nn_count[qq], qq, f.k, nb, vx, vy, vz, quirk);
    const float qx = query[3 * qq], qy = query[3 * qq + 1], qz = query[3 * qq + 2];
    const size_t QsT = ws.QsT;
    float pred[OD];  // OD = 1: sdf (scaled); OD = 3: colour = sigmoid(head) (decoder.py:112)
#pragma unroll
    for (int c = 0; c < OD; ++c) pred[c] = 0.f;
    if (WF) {
        float z[MLP_IN];
#pragma unroll
        for (int j = 0; j < MLP_IN; ++j) z[j] = 0.f;
#pragma unroll
        for (int t = 0; t < PIN_MAX_K; ++t)
            if (nb.idx[t] >= 0) {
                float ft[PIN_FEATURE_DIM], v[3];
                neighbor_input(f, nb.idx[t], quirk[t], vx[t], vy[t], vz[t], qx, qy, qz, ft, v);
                const float w = nb.w[t];
#pragma unroll
                for (int j = 0; j < PIN_FEATURE_DIM; ++j) z[j] = fmaf(w, ft[j], z[j]);
                z[8] = fmaf(w, v[0], z[8]); z[9] = fmaf(w, v[1], z[9]); z[10] = fmaf(w, v[2], z[10]);
            }
#pragma unroll
        for (int j = 0; j < MLP_IN; ++j) ws.z[(size_t)j * QsT + qi] = z[j];
        ws.z[(size_t)11 * QsT + qi] = 0.f;
        float o[OD];
        MfmaDecoder<H>::template forward_store<OD>(lds, f.levels, xb, z, ws.h, QsT, (size_t)q0, ws.mask + q0, QsT, o);
#pragma unroll
        for (int c = 0; c < OD; ++c) pred[c] = OD == 1 ? f.sdf_scale * o[c] : sigmoidf_(o[c]);
    } else {
        // weighted_first = False (run_kitti.yaml:25): decode every neighbour, then weight the
        // predictions (mapper.py:658-662); neighbour t of all queries forms sample block t
#pragma unroll 1
        for (int t = 0; t < f.k; ++t) {
            int idx = -1; float wt = 0.f, gx = 0.f, gy = 0.f, gz = 0.f; bool qk = false;
#pragma unroll
            for (int u = 0; u < PIN_MAX_K; ++u)
                if (u == t) { idx = nb.idx[u]; wt = nb.w[u]; gx = vx[u]; gy = vy[u]; gz = vz[u]; qk = quirk[u]; }
            float z[MLP_IN];
#pragma unroll
            for (int j = 0; j < MLP_IN; ++j) z[j] = 0.f;
            if (idx >= 0) {
                float ft[PIN_FEATURE_DIM], v[3];
                neighbor_input(f, idx, qk, gx, gy, gz, qx, qy, qz, ft, v);
#pragma unroll
                for (int j = 0; j < PIN_FEATURE_DIM; ++j) z[j] = ft[j];
                z[8] = v[0]; z[9] = v[1]; z[10] = v[2];
            }
            const size_t col0 = (size_t)t * ws.Qs + q0;
#pragma unroll
            for (int j = 0; j < MLP_IN; ++j) ws.z[(size_t)j * QsT + col0 + (threadIdx.x & 63)] = z[j];
            ws.z[(size_t)11 * QsT + col0 + (threadIdx.x & 63)] = 0.f;
            float o[OD];
            MfmaDecoder<H>::template forward_store<OD>(lds, f.levels, xb, z, ws.h, QsT, col0, ws.mask + col0, QsT, o);
#pragma unroll
            for (int c = 0; c < OD; ++c) {
                if (OD > 1) ws.xraw[(size_t)c * QsT + col0 + (threadIdx.x & 63)] = o[c];
                if (idx >= 0) pred[c] = fmaf(wt, OD == 1 ? f.sdf_scale * o[c] : sigmoidf_(o[c]), pred[c]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < OD; ++c) ws.pred[(size_t)c * ws.Qs + qi] = pred[c];
    if (active && qi < n_main && cert_rw != nullptr) {
#pragma unroll
        for (int t = 0; t < PIN_MAX_K; ++t)
            if (nb.idx[t] >= 0) {
                atomicAdd(cert_rw + nb.idx[t], nb.w[t]);
                if (ts_rw != nullptr && sample_ts != nullptr) atomicMax(ts_rw + nb.idx[t], sample_ts[qi]);
            }
    }
}

template <int H, bool WF, int OD = 1>
__global__ __launch_bounds__(MF_BLOCK) void train_bwd_mfma_kernel(pin_field f, const float4* __restrict__ nbr,
                                                                  const int* __restrict__ nn_count, int Q, TrainWs ws,
                                                                  float* __restrict__ feat_grad, int want_dec) {
    __shared__ __attribute__((aligned(16))) float lds[MfmaLds<H>::TOTAL];
    float* xb = lds + MfmaLds<H>::W + (threadIdx.x >> 6) * MfmaDecoder<H>::scratch_floats();
    MfmaDecoder<H>::stage(f.dec, f.levels, lds, threadIdx.x, MF_BLOCK, OD);
    __syncthreads();
    const int q0 = (blockIdx.x * (MF_BLOCK / 64) + (threadIdx.x >> 6)) * 64;
    if (q0 >= ws.Qs) return;
    const int lane = threadIdx.x & 63;
    const int qi = q0 + lane;
    const bool active = qi < Q;
    const size_t QsT = ws.QsT;
    // d loss / d prediction per head; sdf: the prediction is sdf_scale * head
    float dpq[OD];
    bool any = false;
#pragma unroll
    for (int c = 0; c < OD; ++c) {
        dpq[c] = active ? ws.dpred[(size_t)c * ws.Qs + qi] * (OD == 1 ? f.sdf_scale : 1.f) : 0.f;
        any = any || dpq[c] != 0.f;
    }
    NbrW nb;
    {
        float vx[PIN_MAX_K], vy[PIN_MAX_K], vz[PIN_MAX_K];
        bool quirk[PIN_MAX_K];
        const int qq = active ? qi : Q - 1;
        neighbor_weights(nbr, nn_count[qq], qq, f.k, nb, vx, vy, vz, quirk);
    }
    const bool live = active && any;
    float* sdz = xb;                 // [32][8] (WF) / [64][8] (per-neighbour mode)
    float* sw = xb + 256;            // [32][8]
    int* sidx = reinterpret_cast<int*>(xb + 512);  // [32][8] / [64]
    if (WF) {
        float dxa[OD];
#pragma unroll
        for (int c = 0; c < OD; ++c) {
            float dh = dpq[c];  // through the sigmoid of the colour heads: p (1 - p)
            if (OD > 1) { const float pc = ws.pred[(size_t)c * ws.Qs + (active ? qi : 0)]; dh *= pc * (1.f - pc); }
            dxa[c] = dh;
            if (want_dec) ws.d[(size_t)(f.levels * H + c) * QsT + qi] = dh;
        }
        float dz[MLP_IN];
        MfmaDecoder<H>::template backward_store<OD>(lds, f.levels, xb, dxa, ws.mask + q0, QsT, ws.d, QsT, (size_t)q0,
                                                    want_dec != 0, dz);
        // Feature-gradient scatter.  One atomic instruction per QUERY: its 64 lanes are the 8
        // neighbours x 8 feature dims, so every instruction touches 8 whole 32-byte rows instead
        // of 64 different rows (the L2 atomic units work per cache line; measured 3x on this kernel).
        for (int half = 0; half < 2; ++half) {
            if ((lane >> 5) == half) {
                const int ql = lane & 31;
#pragma unroll
                for (int j = 0; j < PIN_FEATURE_DIM; ++j) sdz[ql * 8 + j] = dz[j];
#pragma unroll
                for (int t = 0; t < PIN_MAX_K; ++t) {
                    sw[ql * 8 + t] = nb.w[t];
                    sidx[ql * 8 + t] = live ? nb.idx[t] : -1;
                }
            }
            wave_lds_sync();
            const int t = lane >> 3, j = lane & 7;
            for (int i = 0; i < 32; ++i) {
                const int idx = sidx[i * 8 + t];
                if (idx >= 0) atomicAdd(feat_grad + (size_t)idx * PIN_FEATURE_DIM + j, sw[i * 8 + t] * sdz[i * 8 + j]);
            }
            wave_lds_sync();
        }
    } else {
#pragma unroll 1
        for (int t = 0; t < f.k; ++t) {
            int idx = -1; float wt = 0.f;
#pragma unroll
            for (int u = 0; u < PIN_MAX_K; ++u)
                if (u == t) { idx = nb.idx[u]; wt = nb.w[u]; }
            const size_t col0 = (size_t)t * ws.Qs + q0;
            float dxa[OD];
            bool nz = false;
#pragma unroll
            for (int c = 0; c < OD; ++c) {
                float dh = (live && idx >= 0) ? dpq[c] * wt : 0.f;  // d loss / d head_c of neighbour t
                if (OD > 1) { const float pc = sigmoidf_(ws.xraw[(size_t)c * QsT + col0 + lane]); dh *= pc * (1.f - pc); }
                dxa[c] = dh;
                nz = nz || dh != 0.f;
                if (want_dec) ws.d[(size_t)(f.levels * H + c) * QsT + col0 + lane] = dh;
            }
            float dz[MLP_IN];
            MfmaDecoder<H>::template backward_store<OD>(lds, f.levels, xb, dxa, ws.mask + col0, QsT, ws.d, QsT, col0,
                                                        want_dec != 0, dz);
            // one neighbour per query here: 8 queries x 8 feature dims per atomic instruction
#pragma unroll
            for (int j = 0; j < PIN_FEATURE_DIM; ++j) sdz[lane * 8 + j] = dz[j];
            sidx[lane] = nz ? idx : -1;
            wave_lds_sync();
            const int qo = lane >> 3, j = lane & 7;
            for (int i = 0; i < 8; ++i) {
                const int ql = i * 8 + qo;
                const int id = sidx[ql];
                if (id >= 0) atomicAdd(feat_grad + (size_t)id * PIN_FEATURE_DIM + j, sdz[ql * 8 + j]);
            }
            wave_lds_sync();
        }
    }
}

// ---- loss --------------------------------------------------------------------------------
// d loss / d prediction of query qi with the arithmetic of train_loss_kernel below (same bits), for kernels that
// fold the loss into their own prologue; l_bce / l_eik receive this query's loss terms (the Eikonal term of a
// sample is reported by its first query).
__device__ __forceinline__ float loss_dpred(const pin_train_params& tp, const float* __restrict__ label,
                                            const float* __restrict__ weight, const float* __restrict__ pred, int qi,
                                            double& l_bce, double& l_eik) {
    l_bce = 0.0; l_eik = 0.0;
    if (qi < tp.n_main) {
        const float xl = pred[qi] / tp.sigma;
        const float y = 1.f / (1.f + expf(-label[qi] / tp.sigma));
        float l = fmaxf(xl, 0.f) - xl * y + log1pf(expf(-fabsf(xl)));
        float g = 1.f / (1.f + expf(-xl)) - y;
        if (tp.loss_weight_on) { const float w = fabsf(weight[qi]); l *= w; g *= w; }
        l_bce = (double)l;
        return g * tp.inv_n_main / tp.sigma;
    }
    const int e = qi - tp.n_main, s = e / 6, a = e - 6 * s;
    if (s >= tp.n_eik) return 0.f;
    const float* P = pred + tp.n_main + 6 * s;
    const float two_eps = 2.f * tp.eik_eps;
    const float gx = (P[0] - P[1]) / two_eps, gy = (P[2] - P[3]) / two_eps, gz = (P[4] - P[5]) / two_eps;
    const float n = sqrtf(gx * gx + gy * gy + gz * gz);
    const float r = n - 1.f;
    if (a == 0) l_eik = (double)(r * r);
    const float c = n > 0.f ? tp.weight_e * 2.f * r * tp.inv_n_eik / (n * two_eps) : 0.f;
    const float ga = a < 2 ? gx : (a < 4 ? gy : gz);
    return (a & 1) ? -c * ga : c * ga;
}

__global__ __launch_bounds__(256) void train_loss_kernel(pin_train_params tp, const float* __restrict__ label,
                                                         const float* __restrict__ weight, TrainWs ws,
                                                         double* __restrict__ loss_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double l_bce = 0.0, l_eik = 0.0;
    if (i < tp.n_main) {
        const float xl = ws.pred[i] / tp.sigma;
        const float y = 1.f / (1.f + expf(-label[i] / tp.sigma));
        float l = fmaxf(xl, 0.f) - xl * y + log1pf(expf(-fabsf(xl)));
        float g = 1.f / (1.f + expf(-xl)) - y;
        if (tp.loss_weight_on) { const float w = fabsf(weight[i]); l *= w; g *= w; }
        ws.dpred[i] = g * tp.inv_n_main / tp.sigma;
        l_bce = (double)l;
    } else if (i < tp.n_main + tp.n_eik) {
        const int s = i - tp.n_main;
        const float* P = ws.pred + tp.n_main + 6 * s;
        float* D = ws.dpred + tp.n_main + 6 * s;
        const float two_eps = 2.f * tp.eik_eps;
        const float gx = (P[0] - P[1]) / two_eps, gy = (P[2] - P[3]) / two_eps, gz = (P[4] - P[5]) / two_eps;
        const float n = sqrtf(gx * gx + gy * gy + gz * gz);
        const float r = n - 1.f;
        l_eik = (double)(r * r);
        // d/dg of weight_e * mean((|g|-1)^2); torch's norm backward is 0 at |g| = 0
        const float c = n > 0.f ? tp.weight_e * 2.f * r * tp.inv_n_eik / (n * two_eps) : 0.f;
        D[0] = c * gx; D[1] = -c * gx; D[2] = c * gy; D[3] = -c * gy; D[4] = c * gz; D[5] = -c * gz;
    }
    l_bce = wave_sum(l_bce);
    l_eik = wave_sum(l_eik);
    if ((threadIdx.x & 63) == 0) {
        if (l_bce != 0.0) atomicAdd(loss_out + 0, l_bce);
        if (l_eik != 0.0) atomicAdd(loss_out + 1, l_eik);
    }
}

// ---- colour loss: weight_i * mean over surface samples and channels of |pred - label| ---------
// (color_diff_loss, utils/loss.py:31-42; mapper.py:673-675, 804-812)
__global__ __launch_bounds__(256) void color_count_kernel(const float* __restrict__ label, int n, float range,
                                                          int* __restrict__ count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool on = i < n && fabsf(label[i]) < range;
    const int c = __popcll(__ballot(on));
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

__global__ __launch_bounds__(256) void train_color_loss_kernel(pin_train_color_params tp, const float* __restrict__ label,
                                                               const float* __restrict__ color,
                                                               const float* __restrict__ weight, TrainWs ws,
                                                               const int* __restrict__ count, double* __restrict__ loss_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double l = 0.0;
    if (i < tp.n_main) {
        const bool on = fabsf(label[i]) < tp.surface_range;
        const float wt = tp.loss_weight_on ? fabsf(weight[i]) : 1.f;
        const float scale = tp.weight_i * wt / fmaxf((float)(*count) * 3.f, 1.f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float diff = ws.pred[(size_t)c * ws.Qs + i] - color[3 * (size_t)i + c];
            ws.dpred[(size_t)c * ws.Qs + i] = on ? (diff > 0.f ? scale : (diff < 0.f ? -scale : 0.f)) : 0.f;
            if (on) l += (double)(wt * fabsf(diff));
        }
    }
    l = wave_sum(l);
    if ((threadIdx.x & 63) == 0 && l != 0.0) atomicAdd(loss_out, l);
}

// ---- decoder weight gradients: G[i][j] = sum_q D[i][q] * X[j][q]  (fp32 MFMA 16x16x4) ------
// A skinny GEMM whose reduction dimension is the batch.  grid = (K-slices, layers); a block is
// 4 waves, each wave owns a contiguous run of queries and ALL (<= 4x4) 16x16 output tiles.
// Operands are read K-contiguous as float4 (lane (i, g) takes queries q+4g..q+4g+3 of row i):
// the MFMA k index is a free permutation as long as A and B agree, so component c of the
// float4 is k-step c and no cross-lane shuffle is needed.  The 4 waves are summed through
// LDS and one wave issues the float atomics.
typedef float v4f __attribute__((ext_vector_type(4)));

struct DwLayers {
    int H, L, Q, Qs, per_wave, OD;  // Q / Qs: sample columns (valid / row stride); OD output heads
};

__device__ __forceinline__ float4 load_row4(const float* __restrict__ base, int row, int nrows, int Qs, int q, int Q) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrows && q < Q) {
        v = *reinterpret_cast<const float4*>(base + (size_t)row * Qs + q);
        if (q + 1 >= Q) v.y = 0.f;
        if (q + 2 >= Q) v.z = 0.f;
        if (q + 3 >= Q) v.w = 0.f;
    }
    return v;
}

__global__ __launch_bounds__(256) void train_dw_kernel(TrainWs ws, DwLayers dl, float* __restrict__ dec_grad) {
    __shared__ float red[3][16 * 256 + 64];
    const int H = dl.H, L = dl.L, Qs = dl.Qs, Q = dl.Q;
    const int l = blockIdx.y;
    const int rows = l < L ? H : dl.OD;
    const int cols = l == 0 ? 12 : H;       // z carries 12 rows (row 11 is zero)
    const int cols_out = l == 0 ? MLP_IN : H;
    const float* __restrict__ D = ws.d + (size_t)l * H * Qs;
    const float* __restrict__ X = l == 0 ? ws.z : ws.h + (size_t)(l - 1) * H * Qs;
    size_t off = 0;
    for (int u = 0; u < l; ++u) off += (size_t)H * (u == 0 ? MLP_IN : H) + H;
    float* __restrict__ gW = dec_grad + off;
    float* __restrict__ gb = gW + (size_t)rows * cols_out;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 15, g = lane >> 4;
    const int q0 = (blockIdx.x * 4 + wave) * dl.per_wave;
    const int q1 = min(q0 + dl.per_wave, Q);
    const int ntr = (rows + 15) >> 4, ntc = (cols + 15) >> 4;
    v4f acc[4][4];
    float bsum[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        bsum[a] = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (v4f){0.f, 0.f, 0.f, 0.f};
    }
    for (int q = q0; q < q1; q += 16) {
        float4 av[4], bv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            av[t] = t < ntr ? load_row4(D, t * 16 + li, rows, Qs, q + 4 * g, q1) : make_float4(0.f, 0.f, 0.f, 0.f);
            bv[t] = t < ntc ? load_row4(X, t * 16 + li, cols, Qs, q + 4 * g, q1) : make_float4(0.f, 0.f, 0.f, 0.f);
            bsum[t] += (av[t].x + av[t].y) + (av[t].z + av[t].w);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (a >= ntr) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b >= ntc) continue;
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a].x, bv[b].x, acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a].y, bv[b].y, acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a].z, bv[b].z, acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a].w, bv[b].w, acc[a][b], 0, 0, 0);
            }
        }
    }
    // row sums over the 4 k-groups -> bias gradient (kept in lanes g == 0)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        bsum[t] = rows_sum(bsum[t]);
    }
    // block reduction: waves 1..3 park their tiles in LDS, wave 0 adds and publishes
    if (wave > 0) {
        float* r = red[wave - 1];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int c = 0; c < 4; ++c) r[((a * 4 + b) * 4 + c) * 64 + lane] = acc[a][b][c];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (g == 0) r[16 * 256 + t * 16 + li] = bsum[t];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (a >= ntr) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b >= ntc) continue;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int e = ((a * 4 + b) * 4 + c) * 64 + lane;
                    const float v = acc[a][b][c] + red[0][e] + red[1][e] + red[2][e];
                    // D layout: C[i = 4*(lane>>4) + c][j = lane&15] of tile (a, b)
                    const int i = a * 16 + 4 * g + c, j = b * 16 + li;
                    if (i < rows && j < cols_out && v != 0.f) atomicAdd(gW + (size_t)i * cols_out + j, v);
                }
            }
            if (g == 0) {
                const int i = a * 16 + li, e = 16 * 256 + a * 16 + li;
                const float v = bsum[a] + red[0][e] + red[1][e] + red[2][e];
                if (i < rows && v != 0.f) atomicAdd(gb + i, v);
            }
        }
    }
}

}  // namespace pin
#include "sem.h"
#include "train_fused.h"

namespace pin {
// tiles per block of the weight-gradient launch: whole rounds of blocks.  The grid is (chunks, L + 1 layers) and a block
// fills a CU (16 waves, 128 registers), so `rounds * n_cu / (L + 1)` chunks make exactly `rounds` rounds -- r03a's fixed 32
// tiles gave 52 x 5 = 260 blocks on 256 CUs at the reference's batch: a second round for four blocks.  Large batches take
// chunks of up to ~DW_MAX_CHUNK tiles (fewer block tails: the reduction over the phases and the slot atomics).
// PIN_DW_CHUNK / PIN_DW_MAXCHUNK override for A/B runs.
constexpr int DW_MAX_CHUNK = 256;
static int dw_chunk(int n_tiles, int layers_plus_one, int n_cu) {
    static const int forced = [] { const char* e = getenv("PIN_DW_CHUNK"); return e ? atoi(e) : 0; }();
    static const int max_chunk = [] { const char* e = getenv("PIN_DW_MAXCHUNK"); return e ? atoi(e) : DW_MAX_CHUNK; }();
    if (forced > 0) return forced;
    const long work = (long)n_tiles * layers_plus_one;
    const int rounds = (int)((work + (long)n_cu * max_chunk - 1) / ((long)n_cu * max_chunk));
    const int chunks = max(1, rounds * n_cu / layers_plus_one);
    return max(1, cdiv(n_tiles, chunks));
}
// the weight gradient recomputes the layers' inputs instead of streaming them (train_dw_recompute_kernel) from this many
// tiles on; PIN_DW_RECOMPUTE=0 / 1 forces the choice (read per call: tests switch it)
// With the forward pass recomputed, the deltas are what is left of the operand stream (1.0 of 1.15 KB per query at 4 x 64), half of
// it their LOW fp16 pieces: bits 12-23 of numbers that are summed over >= 131 072 queries per weight.  PIN_DW_DELTA=hi streams
// the high pieces only (an experiment: see DESIGN section 8 "Round 6" for what it does to the gradient and to the iteration);
// read per call (tests switch it)
static bool dw_delta_hi_only() {
    const char* e = getenv("PIN_DW_DELTA");
    return e != nullptr && e[0] == 'h';
}
// Compute units the persistent training tile kernels take.  They hold a unit's whole register file (12 waves x 168 registers), so while
// one runs nothing else does -- a stream beside the mapper (the next frame's scan chain, the brick build) advances only between
// tile kernels.  At the reference's batch there are fewer tiles than waves (1 639 for 3 072), so an eighth of the units is left to
// the other streams at no cost in rounds (same box, two runs each: mapping 1.28 / 1.27 -> 1.28 / 1.26 ms, the scan chain beside it
// 0.71 -> 0.60 ms, the main stream's wait for it 0.06 -> 0.045 ms; it matters on boxes whose host queues the training loop
// slowly: there the wait was 0.14 ms).  Large batches take every unit.  PIN_TRAIN_CU_RESERVE=<units> overrides (read once).
static int train_grid_cus(int n_cu, int n_tiles) {
    static const int reserve = [] { const char* e = getenv("PIN_TRAIN_CU_RESERVE"); const int v = e ? atoi(e) : 32; return v < 0 ? 0 : v; }();
    const int cus = n_cu - reserve;
    return (cus >= 64 && n_tiles <= cus * 8) ? cus : n_cu;
}
constexpr int DW_RECOMPUTE_MIN_TILES = 8192;
static bool dw_recompute(int n_tiles) {
    const char* e = getenv("PIN_DW_RECOMPUTE");
    if (e != nullptr && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
    return n_tiles >= DW_RECOMPUTE_MIN_TILES;
}
}  // namespace pin
namespace pin {

// ---- Adam --------------------------------------------------------------------------------
// (adam_elem: one element, one step -- pin_common.h, shared with the data-parallel halo step in dp.hip)

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr_over_bc1, float inv_sqrt_bc2,
                                                   float b1, float b2, float eps, int zero_grad) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (; i < n; i += stride) {
        float pi = p[i], mi = m[i], vi = v[i];
        adam_elem(pi, mi, vi, g[i], lr_over_bc1, inv_sqrt_bc2, b1, b2, eps);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (zero_grad) g[i] = 0.f;
    }
}

// Exact sparse form of the same step for the neural-point feature tables.  Adam state is reset at every
// Mapper.mapping call (mapper.py:615), so a row no query has touched since then has g = m = v = 0 and its
// dense update is exactly p -= lr * 0 / (0 + eps) = 0: skipping it changes no bit.  `row_flags` marks the
// rows touched so far in this call (mark_rows_kernel, from the kNN records of every training iteration).
__global__ __launch_bounds__(256) void mark_rows_kernel(const float4* __restrict__ nbr, long total,
                                                        unsigned char* __restrict__ row_flags) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int raw = __float_as_int(nbr[i].w);
    if (raw >= 0) row_flags[raw & ~PIN_NBR_QUIRK_BIT] = 1;
}

__global__ __launch_bounds__(256) void adam_rows_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long n, int row_width,
                                                        const unsigned char* __restrict__ row_flags, float lr_over_bc1,
                                                        float inv_sqrt_bc2, float b1, float b2, float eps, int zero_grad) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (; i < n; i += stride) {
        if (!row_flags[i / row_width]) continue;
        float pi = p[i], mi = m[i], vi = v[i];
        adam_elem(pi, mi, vi, g[i], lr_over_bc1, inv_sqrt_bc2, b1, b2, eps);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (zero_grad) g[i] = 0.f;
    }
}

// Lazy exact Adam for the feature tables.  The dense step moves every row that has been touched since the
// optimiser reset (momentum), but a row's values only matter when a query reads it -- and its own step can wait
// until then as well.  Per iteration ONE launch visits the rows in the kNN records, before the forward pass
// (adam_lazy_prepare_kernel): a row that was last read at iteration a still owes step a (its gradient is sitting
// in `g`); it takes that step now, then replays the gradient-free steps a+1 .. t-1, and is marked as owing step t.
// Rows that were touched and then left alone settle their pending step and the replay up to the final step once,
// at the end (adam_lazy_flush_kernel).  Every element goes through the same sequence of adam_elem calls as in the
// dense schedule, hence bit-identical tables (tests/test_gpu_parity.py); first-touch rows start from m = v = 0
// without the state arrays ever being cleared.  One owner per row and launch via a compare-and-swap on pend[row].
// pend[row]: 0 = untouched since the reset; -a = owes step a, moment arrays not valid yet (first step); +a = owes step a.
__device__ __forceinline__ void lazy_settle(float& pi, float& mi, float& vi, float gi, int pend, int upto,
                                            const float* __restrict__ coef, int t_max, float b1, float b2, float eps) {
    const int a = pend < 0 ? -pend : pend;
    adam_elem(pi, mi, vi, gi, coef[a], coef[t_max + 1 + a], b1, b2, eps);
    for (int s = a + 1; s <= upto; ++s) adam_elem(pi, mi, vi, 0.f, coef[s], coef[t_max + 1 + s], b1, b2, eps);
}

// the decoder image (mlp_h2.h) follows the decoder's step: parameter e just became x
__device__ __forceinline__ void dense_image_entry(const pin_adam_dense& d, int e, float x) {
    if (d.image == nullptr) return;
    unsigned char* w = reinterpret_cast<unsigned char*>(d.image);
    if (d.hidden == 64) QuadDecoderH<64>::stage_param(e, x, d.levels, d.out_dim, w);
    else QuadDecoderH<32>::stage_param(e, x, d.levels, d.out_dim, w);
}

// The dense tensor that rides along with the lazy launches (the decoder), step `dense_step`, element e.  grad_partial: the weight
// gradient of the step as the weight-gradient launch left it (pin_train_params.defer_dec_reduce) -- summed here exactly as
// train_finalize_kernel sums it (same order, same scale), so that the decoder's step is the bits of the two-launch form
__device__ __forceinline__ void dense_rider_step(const pin_adam_dense& dense, long e, const float* __restrict__ coef, int t_max, int dense_step,
                                                 float b1, float b2, float eps) {
    float gi = dense.grad[e];
    if (dense.grad_partial != nullptr) {
        float v[DW_SLOTS];
#pragma unroll
        for (int c = 0; c < DW_SLOTS; ++c) v[c] = dense.grad_partial[(size_t)c * dense.n + e];
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < DW_SLOTS; ++c) t += v[c];
        gi += t * dense.partial_scale;
    }
    float pi = dense.param[e], mi = dense.exp_avg[e], vi = dense.exp_avg_sq[e];
    adam_elem(pi, mi, vi, gi, coef[dense_step], coef[t_max + 1 + dense_step], b1, b2, eps);
    dense.param[e] = pi; dense.exp_avg[e] = mi; dense.exp_avg_sq[e] = vi;
    dense.grad[e] = 0.f;
    dense_image_entry(dense, (int)e, pi);
}

// The decoder's blocks ride in the lazy launches behind the `work` blocks of the rows in the launch's numbering; in the order the
// device starts them they come FIRST (with the slot copies to sum they are the longest blocks of the launch: started last, they
// were its tail -- 4 us per training iteration under the trace)
__device__ __forceinline__ int lazy_block(int work_blocks) {
    const int dense_blocks = (int)gridDim.x - work_blocks, b = (int)blockIdx.x;
    return b < dense_blocks ? work_blocks + b : b - dense_blocks;
}

__global__ __launch_bounds__(256) void adam_lazy_prepare_kernel(const float4* __restrict__ nbr, long n_records,
                                                                float* __restrict__ p, float* __restrict__ g,
                                                                float* __restrict__ m, float* __restrict__ v,
                                                                int* __restrict__ pend, int step,
                                                                const float* __restrict__ coef, int t_max,
                                                                float b1, float b2, float eps, int rec_blocks,
                                                                pin_adam_dense dense, int dense_step) {
    const int bid = lazy_block(rec_blocks);
    if (bid >= rec_blocks) {  // the dense tensor that rides along (the decoder), step `dense_step`
        const long e = (long)(bid - rec_blocks) * 256 + threadIdx.x;
        if (e < dense.n) dense_rider_step(dense, e, coef, t_max, dense_step, b1, b2, eps);
        return;
    }
    // the step coefficients go through LDS: lazy_settle indexes them per lane inside its replay loop, and a global
    // load there is a dependent ~1 us round trip per replayed step
    extern __shared__ float lazy_coef[];
    for (int i = threadIdx.x; i < 2 * (t_max + 1); i += 256) lazy_coef[i] = coef[i];
    __syncthreads();
    const long tid = (long)bid * 256 + threadIdx.x;
    const long rec = tid >> 3;
    const int j = (int)(tid & 7), lane = threadIdx.x & 63;
    int row = -1, own = 0, n = 0;
    if (rec < n_records) {
        const int raw = __float_as_int(nbr[rec].w);
        if (raw >= 0) row = raw & ~PIN_NBR_QUIRK_BIT;
    }
    // One owner per row and launch: the first record of a row to swap its `pend` entry to "owes step `step`" settles it;
    // the other records of the row find +-step there (or lose the compare-and-swap) and leave.  No separate claim array:
    // the election rides on the word the owner has to update anyway (r02a: an atomicMax stamp on a second int per row --
    // 210k more random line updates per iteration, which the tile kernel that follows could feel).
    // The row's values are requested together with its pending word, before anybody knows whether this record will own the row
    // (nobody but the owner writes a row in this launch, so what an owner-to-be reads here is what it would read later): one
    // dependent memory round trip less in a kernel that is a chain of four.  Records that lose the election drop the values.
    const size_t i = (size_t)(row >= 0 ? row : 0) * PIN_FEATURE_DIM + j;
    float pi = 0.f, gi = 0.f, mi = 0.f, vi = 0.f;
    if (row >= 0) { pi = p[i]; gi = g[i]; mi = m[i]; vi = v[i]; }
    if (row >= 0 && j == 0) {
        const int cur = pend[row];  // (a stale value only makes the compare-and-swap fail)
        // (PIN_ADAM_ROW_EXCLUDED: a row some other step owns -- the halo rows of the spatially sharded mapper, dp.hip)
        if (cur != step && cur != -step && cur != PIN_ADAM_ROW_EXCLUDED) {
            const int want = cur == 0 ? -step : step;
            if (atomicCAS(pend + row, cur, want) == cur) { own = 1; n = cur; }
        }
    }
    own = __shfl(own, lane & ~7, 64);  // the 8 lanes of a record follow their leader
    n = __shfl(n, lane & ~7, 64);
    if (!own || n == 0) return;  // (a first touch has nothing to settle: its moments start from zero at its first step)
    if (n < 0) { mi = 0.f; vi = 0.f; }  // (first step of the row: the moment arrays are not valid yet)
    lazy_settle(pi, mi, vi, gi, n, step - 1, lazy_coef, t_max, b1, b2, eps);
    p[i] = pi; m[i] = mi; v[i] = vi;
    g[i] = 0.f;
}

// The same step for LARGE batches (many more records than rows: a 2^20 batch holds 13 M records over 2.2 M rows, six
// visits per row of which five lose the election): the records only FLAG their rows (mark_rows_kernel), then one pass
// over the rows settles the flagged ones -- eight lanes per row, no compare-and-swap, the same lazy_settle arithmetic, so
// the tables are the bits of the record-parallel form.  Clears the flags it consumes.
__global__ __launch_bounds__(256) void adam_lazy_prepare_rows_kernel(float* __restrict__ p, float* __restrict__ g,
                                                                     float* __restrict__ m, float* __restrict__ v,
                                                                     int* __restrict__ pend, unsigned char* __restrict__ flags,
                                                                     long n_rows, int step, const float* __restrict__ coef, int t_max,
                                                                     float b1, float b2, float eps, int row_blocks,
                                                                     pin_adam_dense dense, int dense_step, int all_rows) {
    const int bid = lazy_block(row_blocks);
    if (bid >= row_blocks) {  // the dense tensor that rides along (the decoder), step `dense_step`
        const long e = (long)(bid - row_blocks) * 256 + threadIdx.x;
        if (e < dense.n) dense_rider_step(dense, e, coef, t_max, dense_step, b1, b2, eps);
        return;
    }
    extern __shared__ float lazy_coef[];  // (see adam_lazy_prepare_kernel)
    for (int i = threadIdx.x; i < 2 * (t_max + 1); i += 256) lazy_coef[i] = coef[i];
    __syncthreads();
    const long stride = (long)row_blocks * 32;
    for (long row = (long)bid * 32 + (threadIdx.x >> 3); row < n_rows; row += stride) {
        // (all_rows: a batch with several records per row touches practically every row -- no marking pass, every row counts as
        // read: a row no query reads settles a zero gradient on zero moments, which changes no bit of it)
        if (!all_rows && !flags[row]) continue;
        const int j = threadIdx.x & 7;
        const int n = pend[row];
        // (all eight lanes of the row have read `n` before lane 0 rewrites it: the eight are in one wave, the store below is
        // program-ordered behind the load above)
        if (j == 0) { if (!all_rows) flags[row] = 0; if (n != PIN_ADAM_ROW_EXCLUDED) pend[row] = n == 0 ? -step : step; }
        if (n == 0 || n == PIN_ADAM_ROW_EXCLUDED || n == step || n == -step) continue;  // first touch: nothing to settle yet
        const size_t i = (size_t)row * PIN_FEATURE_DIM + j;
        float pi = p[i], mi = 0.f, vi = 0.f;
        if (n > 0) { mi = m[i]; vi = v[i]; }
        lazy_settle(pi, mi, vi, g[i], n, step - 1, lazy_coef, t_max, b1, b2, eps);
        p[i] = pi; m[i] = mi; v[i] = vi;
        g[i] = 0.f;
    }
}

__global__ __launch_bounds__(256) void adam_lazy_flush_kernel(float* __restrict__ p, float* __restrict__ g,
                                                              float* __restrict__ m_, float* __restrict__ v,
                                                              const int* __restrict__ pend, long n, int t_final,
                                                              const float* __restrict__ coef, int t_max, float b1, float b2,
                                                              float eps, int row_blocks, pin_adam_dense dense) {
    const int bid = lazy_block(row_blocks);
    if (bid >= row_blocks) {  // the dense tensor's last step
        const long e = (long)(bid - row_blocks) * 256 + threadIdx.x;
        if (e < dense.n) dense_rider_step(dense, e, coef, t_max, t_final, b1, b2, eps);
        return;
    }
    extern __shared__ float lazy_coef[];  // (see adam_lazy_prepare_kernel)
    for (int i = threadIdx.x; i < 2 * (t_max + 1); i += 256) lazy_coef[i] = coef[i];
    __syncthreads();
    // A lane per ROW looks at its pending word (most rows of a map are untouched in a call: 2.2 M rows, a few hundred thousand
    // read); the touched rows of the wave's 64 are then settled eight at a time, eight lanes per row.  (r01-r05: a lane per
    // ELEMENT, eight looks at every word: 60-110 us per Mapper.mapping call on the bench map.)
    const long n_rows = n / PIN_FEATURE_DIM;
    const int lane = threadIdx.x & 63, j = lane & 7, sub = lane >> 3;
    const long wave0 = ((long)bid * 256 + threadIdx.x) >> 6, n_waves = (long)row_blocks * 4;
    for (long base = wave0 * 64; base < n_rows; base += n_waves * 64) {
        const long row = base + lane;
        const int nn = row < n_rows ? pend[row] : 0;
        unsigned long long todo = __builtin_amdgcn_ballot_w64(nn != 0 && nn != PIN_ADAM_ROW_EXCLUDED);
        while (todo != 0ull) {
            // this group's row: the (sub + 1)-th set bit of `todo`, if there is one
            unsigned long long m = todo;
            for (int s = 0; s < sub && m != 0ull; ++s) m &= m - 1ull;
            const int src = m != 0ull ? __builtin_ctzll(m) : -1;
            const int pn = __shfl(nn, src < 0 ? 0 : src, 64);
            if (src >= 0) {
                const size_t i = (size_t)(base + src) * PIN_FEATURE_DIM + j;
                float pi = p[i], mi = 0.f, vi = 0.f;
                if (pn > 0) { mi = m_[i]; vi = v[i]; }
                lazy_settle(pi, mi, vi, g[i], pn, t_final, lazy_coef, t_max, b1, b2, eps);
                p[i] = pi; m_[i] = mi; v[i] = vi;
                g[i] = 0.f;
            }
            for (int s = 0; s < 8 && todo != 0ull; ++s) todo &= todo - 1ull;  // (wave-uniform: the eight rows just settled)
        }
    }
}

}  // namespace pin

using namespace pin;

extern "C" int64_t pin_train_workspace_bytes(int32_t n_queries, int32_t hidden, int32_t levels, int32_t expand) {
    return (int64_t)train_ws_floats(n_queries, hidden, levels, expand < 1 ? 1 : expand) * 4;
}

extern "C" int pin_gather_batch(const float* pool_coord, const float* pool_label, const float* pool_weight,
                                const int32_t* pool_ts, const int32_t* index, int32_t n, float* coord_out,
                                float* label_out, float* weight_out, int32_t* ts_out, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n >= 0, "n < 0");
    if (n == 0) return 0;
    PIN_CHECK_ARG(pool_coord && pool_label && index && coord_out && label_out, "NULL pointer");
    hipLaunchKernelGGL(gather_batch_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), pool_coord, pool_label,
                       pool_weight, pool_ts, index, n, coord_out, label_out, weight_out, ts_out);
    PIN_CHECK_LAUNCH();
    return 0;
}

static int gather_batches(const float* pool_coord, const float* pool_label, const float* pool_weight, const int32_t* pool_ts,
                          const float* pool_color, int32_t color_channels, const int64_t* index_history, int32_t n_history,
                          const int64_t* index_new_batch, const int64_t* new_idx, int32_t n, float* coord_out, float* label_out,
                          float* weight_out, int32_t* ts_out, float* color_out, float* query_out, int32_t n_eik,
                          int32_t decimation, int32_t first, float eps, int32_t n_batches, int64_t hist_stride,
                          int64_t new_stride, void* stream) {
    PIN_CHECK_ARG(n >= 0 && n_history >= 0 && n_history <= n && color_channels >= 0 && n_batches >= 0 && n_batches <= 65535, "bad sizes");
    if (n == 0 || n_batches == 0) return 0;
    PIN_CHECK_ARG(pool_coord && pool_label && pool_weight && pool_ts && coord_out && label_out && weight_out && ts_out, "NULL pointer");
    PIN_CHECK_ARG(n_history == 0 || index_history, "index_history NULL");
    PIN_CHECK_ARG(n_history == n || (index_new_batch && new_idx), "index_new_batch / new_idx NULL");
    PIN_CHECK_ARG(color_channels == 0 || (pool_color && color_out), "colour pool / output NULL");
    PIN_CHECK_ARG(n_batches == 1 || (hist_stride >= n_history && new_stride >= n - n_history), "index strides shorter than a batch");
    PIN_CHECK_ARG(query_out == nullptr || (n_eik >= 0 && decimation >= 1 && first >= 0 &&
                                           (n_eik == 0 || first + (long)(n_eik - 1) * decimation < n)),
                  "query_out: n_eik / first too large for decimation");
    hipLaunchKernelGGL(gather_batch_drawn_kernel, dim3(cdiv(n, 256), n_batches), dim3(256), 0, as_stream(stream), pool_coord, pool_label,
                       pool_weight, pool_ts, pool_color, color_channels, reinterpret_cast<const long long*>(index_history),
                       n_history, reinterpret_cast<const long long*>(index_new_batch), reinterpret_cast<const long long*>(new_idx),
                       n, coord_out, label_out, weight_out, ts_out, color_out, query_out, n_eik, decimation < 1 ? 1 : decimation,
                       first, eps, (long)hist_stride, (long)new_stride);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_gather_batch_drawn(const float* pool_coord, const float* pool_label, const float* pool_weight,
                                      const int32_t* pool_ts, const float* pool_color, int32_t color_channels,
                                      const int64_t* index_history, int32_t n_history, const int64_t* index_new_batch,
                                      const int64_t* new_idx, int32_t n, float* coord_out, float* label_out,
                                      float* weight_out, int32_t* ts_out, float* color_out, float* query_out,
                                      int32_t n_eik, int32_t decimation, int32_t first, float eps, void* stream) {
    PIN_ENTER();
    return gather_batches(pool_coord, pool_label, pool_weight, pool_ts, pool_color, color_channels, index_history, n_history,
                          index_new_batch, new_idx, n, coord_out, label_out, weight_out, ts_out, color_out, query_out, n_eik,
                          decimation, first, eps, 1, 0, 0, stream);
}

extern "C" int pin_gather_batches_drawn(const float* pool_coord, const float* pool_label, const float* pool_weight,
                                        const int32_t* pool_ts, const float* pool_color, int32_t color_channels,
                                        const int64_t* index_history, int32_t n_history, const int64_t* index_new_batch,
                                        const int64_t* new_idx, int32_t n, float* coord_out, float* label_out,
                                        float* weight_out, int32_t* ts_out, float* color_out, float* query_out,
                                        int32_t n_eik, int32_t decimation, int32_t first, float eps, int32_t n_batches,
                                        int64_t hist_stride, int64_t new_stride, void* stream) {
    PIN_ENTER();
    return gather_batches(pool_coord, pool_label, pool_weight, pool_ts, pool_color, color_channels, index_history, n_history,
                          index_new_batch, new_idx, n, coord_out, label_out, weight_out, ts_out, color_out, query_out, n_eik,
                          decimation, first, eps, n_batches, hist_stride, new_stride, stream);
}

// (pin_gather_records_drawn) one thread per (batch, sample, neighbour): 16-byte records, k of them contiguous per sample
__global__ __launch_bounds__(256) void gather_records_kernel(const float4* __restrict__ pool_nbr, const int* __restrict__ pool_nn,
                                                             int k, const long long* __restrict__ index_hist, int n_hist,
                                                             const long long* __restrict__ index_new_batch,
                                                             const long long* __restrict__ new_idx, int n, long q_per_batch,
                                                             long hist_stride, long new_stride, float4* __restrict__ nbr_out,
                                                             int* __restrict__ nn_out) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    const int t = (int)(tid % k);
    const long i = tid / k;
    if (i >= n) return;
    const size_t b = blockIdx.y;
    const size_t s = (size_t)(i < n_hist ? index_hist[b * hist_stride + i] : new_idx[index_new_batch[b * new_stride + (i - n_hist)]]);
    const size_t o = b * (size_t)q_per_batch + (size_t)i;
    nbr_out[o * k + t] = pool_nbr[s * k + t];
    if (t == 0) nn_out[o] = pool_nn[s];
}

// ---- training-mode side effects of a whole call at once (pin_count_draws + pin_certainty_from_records) ----------------------
// query_feature(training_mode=True) adds every main query's IDW weights to the certainty of its neighbours and raises their
// ts_update (neural_points.py:685-710).  Neither feeds back into the loss, and when a call's neighbour records are per POOL SAMPLE
// (pin_gather_records_drawn: the neural points do not move while the map trains) a sample drawn m times adds m x the same weights:
// count the draws of every pool row, then ONE pass over the pool rows does the atomics -- 2 M x k instead of iterations x batch x k
// (12.6 M x 8 at a 2^20 batch: 0.45 of the tile kernel's 1.34 ms per iteration).  m x w instead of w added m times: the
// certainty is compared at 1e-4.
__global__ __launch_bounds__(256) void count_draws_kernel(const long long* __restrict__ hist, long n_hist, const long long* __restrict__ newb,
                                                          const long long* __restrict__ new_idx, long n_new, int* __restrict__ count) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n_hist) atomicAdd(count + hist[i], 1);
    else if (i < n_hist + n_new) atomicAdd(count + new_idx[newb[i - n_hist]], 1);
}
__global__ __launch_bounds__(256) void certainty_from_records_kernel(const float4* __restrict__ rec_nbr, const int* __restrict__ rec_nn, int k,
                                                                     const int* __restrict__ pool_to_rec, const int* __restrict__ count,
                                                                     const int* __restrict__ pool_ts, long n_pool, float* __restrict__ cert,
                                                                     int* __restrict__ ts_rw) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pool) return;
    const int c = count[i];
    if (c == 0) return;
    const int r = pool_to_rec != nullptr ? pool_to_rec[i] : (int)i;
    if (r < 0) return;  // (another rank's sample)
    NbrW nb;
    float vx[PIN_MAX_K], vy[PIN_MAX_K], vz[PIN_MAX_K];
    bool quirk[PIN_MAX_K];
    neighbor_weights(rec_nbr, rec_nn[r], r, k, nb, vx, vy, vz, quirk);
    const int my_ts = pool_ts != nullptr ? pool_ts[i] : 0;
    const float m = (float)c;
#pragma unroll
    for (int t = 0; t < PIN_MAX_K; ++t)
        if (nb.idx[t] >= 0) {
            atomicAdd(cert + nb.idx[t], m * nb.w[t]);
            if (ts_rw != nullptr && pool_ts != nullptr && ts_rw[nb.idx[t]] < my_ts) atomicMax(ts_rw + nb.idx[t], my_ts);
        }
}

extern "C" int pin_count_draws(const int64_t* index_history, int64_t n_history_total, const int64_t* index_new_batch,
                               const int64_t* new_idx, int64_t n_new_total, int32_t* count, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n_history_total >= 0 && n_new_total >= 0 && count, "bad arguments");
    const long n = (long)(n_history_total + n_new_total);
    if (n == 0) return 0;
    PIN_CHECK_ARG((n_history_total == 0 || index_history) && (n_new_total == 0 || (index_new_batch && new_idx)), "NULL index array");
    hipLaunchKernelGGL(count_draws_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), reinterpret_cast<const long long*>(index_history),
                       (long)n_history_total, reinterpret_cast<const long long*>(index_new_batch), reinterpret_cast<const long long*>(new_idx),
                       (long)n_new_total, count);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_certainty_from_records(const float* rec_nbr, const int32_t* rec_nn, int32_t k, const int32_t* pool_to_rec,
                                          const int32_t* count, const int32_t* pool_ts, int64_t n_pool, float* certainty_rw,
                                          int32_t* ts_update_rw, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n_pool >= 0 && k >= 1 && k <= PIN_MAX_K, "bad sizes");
    if (n_pool == 0) return 0;
    PIN_CHECK_ARG(rec_nbr && rec_nn && count && certainty_rw, "NULL pointer");
    hipLaunchKernelGGL(certainty_from_records_kernel, dim3(cdiv((long)n_pool, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(rec_nbr), rec_nn, k, pool_to_rec, count, pool_ts, (long)n_pool, certainty_rw, ts_update_rw);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_gather_records_drawn(const float* pool_nbr, const int32_t* pool_nn, int32_t k, const int64_t* index_history,
                                        int32_t n_history, const int64_t* index_new_batch, const int64_t* new_idx, int32_t n,
                                        int64_t q_per_batch, int32_t n_batches, int64_t hist_stride, int64_t new_stride,
                                        float* nbr_out, int32_t* nn_out, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n >= 0 && n_batches >= 1 && k >= 1 && k <= PIN_MAX_K && q_per_batch >= n, "bad sizes");
    if (n == 0) return 0;
    PIN_CHECK_ARG(pool_nbr && pool_nn && nbr_out && nn_out, "NULL pointer");
    PIN_CHECK_ARG(n_history >= 0 && n_history <= n && (n_history == 0 || index_history) &&
                      (n_history == n || (index_new_batch && new_idx)), "index arrays");  // (an all-new draw has no history rows)
    hipLaunchKernelGGL(gather_records_kernel, dim3(cdiv((long)n * k, 256), n_batches), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(pool_nbr), pool_nn, k, reinterpret_cast<const long long*>(index_history), n_history,
                       reinterpret_cast<const long long*>(index_new_batch), reinterpret_cast<const long long*>(new_idx), n,
                       (long)q_per_batch, (long)hist_stride, (long)new_stride, reinterpret_cast<float4*>(nbr_out), nn_out);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_train_make_queries(const float* coord, int32_t n_main, int32_t n_eik, int32_t decimation,
                                      int32_t first, float eps, float* query_out, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n_main >= 0 && n_eik >= 0 && decimation >= 1, "bad sizes");
    const int total = n_main + 6 * n_eik;
    if (total == 0) return 0;
    PIN_CHECK_ARG(coord && query_out, "NULL pointer");
    PIN_CHECK_ARG(first >= 0 && (n_eik == 0 || first + (long)(n_eik - 1) * decimation < n_main),
                  "n_eik / first too large for decimation");
    hipLaunchKernelGGL(make_queries_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), coord, n_main,
                       n_eik, decimation, first, eps, query_out);
    PIN_CHECK_LAUNCH();
    return 0;
}

// weighted_first: fused tile kernel + streamed weight gradient (train_fused.h); OD = 1 the SDF term, OD = 3 the colour term
template <int H, int L, int OD>
static int launch_fused_l(const pin_field* f, const pin_train_params* tp, const FusedColor& fcol, const float* query,
                          const float4* nb4, const int32_t* nn_count, const float* sdf_label, const float* sample_weight,
                          const int32_t* sample_ts, float* certainty_rw, int32_t* ts_update_rw, float* feat_grad, float* dec_grad,
                          double* loss_out, float* pred_out, void* workspace, int n_cu, hipStream_t s, int phase = 3) {
    // phase: bit 0 = the tile kernel (gather .. backward, operand stream out), bit 1 = weight gradient + finalize.  The two
    // halves take the same arguments; run apart (pin_train_params.defer_weight_grad, pin_train_weight_grad) the second can
    // share the device with the next iteration's optimiser launch on another stream.
    using G = DwGeom<H>;
    constexpr int lds_bytes = train_fused_lds_bytes<H>(L);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&train_fused_kernel<H, L, OD>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (attr != hipSuccess) return fail(-2, "training tile kernel: cannot reserve %d bytes of LDS: %s", lds_bytes, hipGetErrorString(attr));
    DwStream ws;
    ws.n_tiles = fused_tiles(tp->n_main, tp->n_eik);
    // (stream_at: a lane's byte offset inside one region of the operand stream is a 32-bit number)
    if ((long)ws.n_tiles * 128 * G::MT * 8 >= (1L << 32))
        return fail(-1, "training batch too large for one launch: %d tiles (limit %ld)", ws.n_tiles, (1L << 32) / (128 * G::MT * 8) - 1);
    ws.d = reinterpret_cast<uint2*>(workspace);
    ws.a = ws.d + G::total((size_t)ws.n_tiles, L);
    // power of two that maps a unit loss gradient to ~1 (see train_fused.h "Scaling"); the colour loss is normalised by
    // the number of surface samples, known on the device only: unit = its smallest value (every sample on the surface)
    float unit;
    if (OD == 1) {
        const float unit_main = tp->inv_n_main * f->sdf_scale / tp->sigma;
        const float unit_eik = tp->n_eik > 0 ? tp->weight_e * tp->inv_n_eik * f->sdf_scale / tp->eik_eps : 0.f;
        unit = fmaxf(unit_main, unit_eik);
    } else {
        // (a shard of a larger batch -- spatial data-parallel ranks -- is normalised by the surface count of the WHOLE batch: up
        // to world x n_main; inv_n_main carries the global batch then)
        const float n_glob = tp->inv_n_main > 0.f ? fmaxf(1.f / tp->inv_n_main, (float)tp->n_main) : (float)tp->n_main;
        unit = 8.f * fcol.weight_i / (3.f * n_glob);
    }
    const float dscale = exp2f(-ceilf(log2f(fmaxf(unit, 1e-30f))));
    const int want_dec = dec_grad != nullptr;
    const int n_dec = H * MLP_IN + H + (L - 1) * (H * H + H) + OD * H + OD;
    float* dw_partial = reinterpret_cast<float*>(ws.a + G::total((size_t)ws.n_tiles, L));
    double* loss_partial = reinterpret_cast<double*>(dw_partial + (size_t)DW_SLOTS * FUSED_NDEC_MAX);
    const unsigned char* image = reinterpret_cast<unsigned char*>(loss_partial + 1024);
    const int grid = min(train_grid_cus(n_cu, ws.n_tiles), ws.n_tiles);
    const bool image_kept = tp->dec_image_current && f->dec_image != nullptr && f->dec_image_bytes == QuadDecoderH<H>::bytes(L);
    if (image_kept) image = reinterpret_cast<const unsigned char*>(f->dec_image);  // kept current by the optimiser (pin_adam_dense.image)
    // large batches: the layers' inputs stay out of the operand stream, the weight-gradient launch runs the forward pass again
    const bool recompute = want_dec && dw_recompute(ws.n_tiles);
    const bool hi_only = recompute && dw_delta_hi_only();
    if (phase & 1) {
        if (!image_kept)
            hipLaunchKernelGGL((train_stage_kernel<H>), dim3(STAGE_BLOCKS), dim3(512), 0, s, *f, const_cast<unsigned char*>(image));
        hipLaunchKernelGGL((train_fused_kernel<H, L, OD>), dim3(grid), dim3(TFW_BLOCK), lds_bytes, s, *f, *tp, query, nb4, nn_count, sdf_label,
                           sample_weight, sample_ts, certainty_rw, ts_update_rw, feat_grad, pred_out, ws, want_dec, dscale, image, dw_partial,
                           n_dec, loss_partial, fcol, recompute ? (hi_only ? 2 : 0) : 1);
        PIN_CHECK_LAUNCH();
    }
    if (!(phase & 2)) return 0;
    if (want_dec && recompute && hi_only) {
        constexpr int dlds = train_dw_recompute_lds_bytes<H>(L);
        static const hipError_t dattr = hipFuncSetAttribute(reinterpret_cast<const void*>(&train_dw_recompute_kernel<H, true>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, train_dw_recompute_lds_bytes<H>(MLP_MAX_LEVELS));
        if (dattr != hipSuccess) return fail(-2, "weight-gradient kernel: cannot reserve LDS: %s", hipGetErrorString(dattr));
        const int chunk = dw_chunk(ws.n_tiles, dwr_groups(L), n_cu);
        hipLaunchKernelGGL((train_dw_recompute_kernel<H, true>), dim3(cdiv(ws.n_tiles, chunk), dwr_groups(L)), dim3(DWR_WAVES * 64), dlds, s, ws, L, OD,
                           n_dec, dw_partial, chunk, image);
        PIN_CHECK_LAUNCH();
    } else if (want_dec && recompute) {
        constexpr int dlds = train_dw_recompute_lds_bytes<H>(L);
        static const hipError_t dattr = hipFuncSetAttribute(reinterpret_cast<const void*>(&train_dw_recompute_kernel<H>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, train_dw_recompute_lds_bytes<H>(MLP_MAX_LEVELS));
        if (dattr != hipSuccess) return fail(-2, "weight-gradient kernel: cannot reserve LDS: %s", hipGetErrorString(dattr));
        const int chunk = dw_chunk(ws.n_tiles, dwr_groups(L), n_cu);
        hipLaunchKernelGGL((train_dw_recompute_kernel<H>), dim3(cdiv(ws.n_tiles, chunk), dwr_groups(L)), dim3(DWR_WAVES * 64), dlds, s, ws, L, OD,
                           n_dec, dw_partial, chunk, image);
        PIN_CHECK_LAUNCH();
    } else if (want_dec) {
        const int chunk = dw_chunk(ws.n_tiles, L + 1, n_cu);
        hipLaunchKernelGGL((train_dw_stream_kernel<H>), dim3(cdiv(ws.n_tiles, chunk), L + 1), dim3(DW_WAVES * 64), 0, s, ws, L, OD, n_dec,
                           dw_partial, 0, chunk);
        PIN_CHECK_LAUNCH();
    }
    if (tp->defer_dec_reduce && want_dec) {  // the optimiser's decoder step sums the slot copies itself (pin_adam_dense.grad_partial)
        tl_deferred = {dw_partial, (int64_t)n_dec, 1.0f / dscale};
        return 0;
    }
    hipLaunchKernelGGL(train_finalize_kernel, dim3(want_dec ? cdiv(n_dec, 256) : 1), dim3(256), 0, s, dw_partial, n_dec, 1.0f / dscale,
                       dec_grad, loss_partial, grid, loss_out, OD == 1 ? 2 : 1);
    PIN_CHECK_LAUNCH();
    return 0;
}

template <int H, int OD>
static int launch_fused(const pin_field* f, const pin_train_params* tp, const FusedColor& fcol, const float* query, const float4* nb4,
                        const int32_t* nn_count, const float* sdf_label, const float* sample_weight, const int32_t* sample_ts,
                        float* certainty_rw, int32_t* ts_update_rw, float* feat_grad, float* dec_grad, double* loss_out,
                        float* pred_out, void* workspace, int n_cu, hipStream_t s, int phase = 3) {
#define PIN_LF(LL) return launch_fused_l<H, LL, OD>(f, tp, fcol, query, nb4, nn_count, sdf_label, sample_weight, sample_ts, certainty_rw, \
                                                    ts_update_rw, feat_grad, dec_grad, loss_out, pred_out, workspace, n_cu, s, phase)
    switch (f->levels) {
        case 1: PIN_LF(1);
        case 2: PIN_LF(2);
        case 3: PIN_LF(3);
        default: PIN_LF(4);
    }
#undef PIN_LF
}

// weighted_first with the analytic Eikonal term (train_fused_an_kernel): two operand streams, any decoder depth
template <int H, int L>
static int launch_fused_an_l(const pin_field* f, const pin_train_params* tp, const float* query, const float4* nb4,
                             const int32_t* nn_count, const float* sdf_label, const float* sample_weight, const int32_t* sample_ts,
                             float* certainty_rw, int32_t* ts_update_rw, float* feat_grad, float* dec_grad, double* loss_out,
                             float* pred_out, void* workspace, int64_t workspace_bytes, int n_cu, hipStream_t s) {
    using G = DwGeom<H>;
    constexpr int lds_bytes = train_fused_an_lds_bytes<H>(L);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&train_fused_an_kernel<H, L>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (attr != hipSuccess) return fail(-2, "training tile kernel: cannot reserve %d bytes of LDS: %s", lds_bytes, hipGetErrorString(attr));
    DwStream ws, ws2;
    ws.n_tiles = ws2.n_tiles = cdiv(tp->n_main, 16);
    if ((long)ws.n_tiles * 128 * G::MT * 8 >= (1L << 32))  // (stream_at's 32-bit lane offsets)
        return fail(-1, "training batch too large for one launch: %d tiles (limit %ld)", ws.n_tiles, (1L << 32) / (128 * G::MT * 8) - 1);
    const size_t per_stream = 2 * G::total((size_t)ws.n_tiles, L);
    const size_t need = 2 * per_stream * sizeof(uint2) + ((size_t)DW_SLOTS * FUSED_NDEC_MAX + 2048 + 32768) * 4;
    PIN_CHECK_ARG((size_t)workspace_bytes >= need, "workspace too small");
    ws.d = reinterpret_cast<uint2*>(workspace);
    ws.a = ws.d + G::total((size_t)ws.n_tiles, L);
    ws2.d = ws.d + per_stream;
    ws2.a = ws.a + per_stream;
    const float unit_main = tp->inv_n_main * f->sdf_scale / tp->sigma;
    const float unit_eik = 2.f * tp->weight_e * tp->inv_n_eik * f->sdf_scale;
    const float dscale = exp2f(-ceilf(log2f(fmaxf(fmaxf(unit_main, unit_eik), 1e-30f))));
    const int want_dec = dec_grad != nullptr;
    const int n_dec = H * MLP_IN + H + (L - 1) * (H * H + H) + H + 1;
    float* dw_partial = reinterpret_cast<float*>(ws.d + 2 * per_stream);
    double* loss_partial = reinterpret_cast<double*>(dw_partial + (size_t)DW_SLOTS * FUSED_NDEC_MAX);
    const unsigned char* image = reinterpret_cast<unsigned char*>(loss_partial + 1024);
    const int grid = min(train_grid_cus(n_cu, ws.n_tiles), ws.n_tiles);
    if (tp->dec_image_current && f->dec_image != nullptr && f->dec_image_bytes == QuadDecoderH<H>::bytes(L))
        image = reinterpret_cast<const unsigned char*>(f->dec_image);  // kept current by the optimiser (pin_adam_dense.image)
    else
        hipLaunchKernelGGL((train_stage_kernel<H>), dim3(STAGE_BLOCKS), dim3(512), 0, s, *f, const_cast<unsigned char*>(image));
    hipLaunchKernelGGL((train_fused_an_kernel<H, L>), dim3(grid), dim3(tf_an_block<H, L>()), lds_bytes, s, *f, *tp, query, nb4, nn_count, sdf_label,
                       sample_weight, sample_ts, certainty_rw, ts_update_rw, feat_grad, pred_out, ws, ws2, want_dec, dscale, image,
                       dw_partial, n_dec, loss_partial);
    PIN_CHECK_LAUNCH();
    if (want_dec) {
        const int chunk = dw_chunk(ws.n_tiles, L + 1, n_cu);
        const dim3 dgrid(cdiv(ws.n_tiles, chunk), L + 1);
        hipLaunchKernelGGL((train_dw_stream_kernel<H>), dgrid, dim3(DW_WAVES * 64), 0, s, ws, L, 1, n_dec, dw_partial, 0, chunk);
        hipLaunchKernelGGL((train_dw_stream_kernel<H>), dgrid, dim3(DW_WAVES * 64), 0, s, ws2, L, 1, n_dec, dw_partial, 1, chunk);
        PIN_CHECK_LAUNCH();
    }
    if (tp->defer_dec_reduce && want_dec) {  // the optimiser's decoder step sums the slot copies itself (pin_adam_dense.grad_partial)
        tl_deferred = {dw_partial, (int64_t)n_dec, 1.0f / dscale};
        return 0;
    }
    hipLaunchKernelGGL(train_finalize_kernel, dim3(want_dec ? cdiv(n_dec, 256) : 1), dim3(256), 0, s, dw_partial, n_dec, 1.0f / dscale,
                       dec_grad, loss_partial, grid, loss_out, 2);
    PIN_CHECK_LAUNCH();
    return 0;
}

template <int H>
static int launch_fused_an(const pin_field* f, const pin_train_params* tp, const float* query, const float4* nb4, const int32_t* nn_count,
                           const float* sdf_label, const float* sample_weight, const int32_t* sample_ts, float* certainty_rw,
                           int32_t* ts_update_rw, float* feat_grad, float* dec_grad, double* loss_out, float* pred_out, void* workspace,
                           int64_t workspace_bytes, int n_cu, hipStream_t s) {
#define PIN_LA(LL) return launch_fused_an_l<H, LL>(f, tp, query, nb4, nn_count, sdf_label, sample_weight, sample_ts, certainty_rw, ts_update_rw, \
                                                   feat_grad, dec_grad, loss_out, pred_out, workspace, workspace_bytes, n_cu, s)
    switch (f->levels) {
        case 1: PIN_LA(1);
        case 2: PIN_LA(2);
        case 3: PIN_LA(3);
        default: PIN_LA(4);
    }
#undef PIN_LA
}

// weighted_first = False with a one-layer decoder: groups of three (query, neighbour)-column tiles (train_fused.h);
// AN = the Eikonal term on the analytic gradient of every sample (a second operand stream for the derivative network)
template <int H, bool AN>
static int launch_fused_nwf(const pin_field* f, const pin_train_params* tp, const float* query, const float4* nb4,
                            const int32_t* nn_count, const float* sdf_label, const float* sample_weight, const int32_t* sample_ts,
                            float* certainty_rw, int32_t* ts_update_rw, float* feat_grad, float* dec_grad, double* loss_out,
                            float* pred_out, void* workspace, int64_t workspace_bytes, int n_cu, hipStream_t s, int phase = 3) {
    using G = DwGeom<H>;
    constexpr int L = 1;
    constexpr int lds_bytes = train_fused_nwf_lds_bytes<H>();
    DwStream ws, ws2;
    const int n_groups = (tp->n_main + 5) / 6 + tp->n_eik;
    ws.n_tiles = ws2.n_tiles = 3 * n_groups;
    if ((long)ws.n_tiles * 128 * G::MT * 8 >= (1L << 32))  // (stream_at's 32-bit lane offsets)
        return fail(-1, "training batch too large for one launch: %d tiles (limit %ld)", ws.n_tiles, (1L << 32) / (128 * G::MT * 8) - 1);
    const size_t per_stream = 2 * G::total((size_t)ws.n_tiles, L);
    const size_t need = (AN ? 2 : 1) * per_stream * sizeof(uint2) + ((size_t)DW_SLOTS * FUSED_NDEC_MAX + 2048 + 32768) * 4;
    PIN_CHECK_ARG((size_t)workspace_bytes >= need, "workspace too small");
    ws.d = reinterpret_cast<uint2*>(workspace);
    ws.a = ws.d + G::total((size_t)ws.n_tiles, L);
    ws2.d = ws.d + (AN ? per_stream : 0);
    ws2.a = ws.a + (AN ? per_stream : 0);
    const float unit_main = tp->inv_n_main * f->sdf_scale / tp->sigma;
    const float unit_eik = AN ? 2.f * tp->weight_e * tp->inv_n_eik
                              : (tp->n_eik > 0 ? tp->weight_e * tp->inv_n_eik * f->sdf_scale / tp->eik_eps : 0.f);
    const float dscale = exp2f(-ceilf(log2f(fmaxf(fmaxf(unit_main, unit_eik), 1e-30f))));
    const int want_dec = dec_grad != nullptr;
    const int n_dec = H * MLP_IN + H + H + 1;
    float* dw_partial = reinterpret_cast<float*>(ws.d + (AN ? 2 : 1) * per_stream);
    double* loss_partial = reinterpret_cast<double*>(dw_partial + (size_t)DW_SLOTS * FUSED_NDEC_MAX);
    const unsigned char* image = reinterpret_cast<unsigned char*>(loss_partial + 1024);
    const int grid = min(n_cu, cdiv(n_groups, tf_nwf_block<H, AN>() / 64));
    if (phase & 1) {
        if (tp->dec_image_current && f->dec_image != nullptr && f->dec_image_bytes == QuadDecoderH<H>::bytes(L))
            image = reinterpret_cast<const unsigned char*>(f->dec_image);  // kept current by the optimiser (pin_adam_dense.image)
        else
            hipLaunchKernelGGL((train_stage_kernel<H>), dim3(STAGE_BLOCKS), dim3(512), 0, s, *f, const_cast<unsigned char*>(image));
        hipLaunchKernelGGL((train_fused_nwf_kernel<H, AN>), dim3(grid), dim3(tf_nwf_block<H, AN>()), lds_bytes, s, *f, *tp, query, nb4, nn_count, sdf_label,
                           sample_weight, sample_ts, certainty_rw, ts_update_rw, feat_grad, pred_out, ws, ws2, want_dec, dscale, image,
                           dw_partial, n_dec, loss_partial);
        PIN_CHECK_LAUNCH();
    }
    if (!(phase & 2)) return 0;
    if (want_dec) {
        const int chunk = dw_chunk(ws.n_tiles, L + 1, n_cu);
        const dim3 dgrid(cdiv(ws.n_tiles, chunk), L + 1);
        hipLaunchKernelGGL((train_dw_stream_kernel<H>), dgrid, dim3(DW_WAVES * 64), 0, s, ws, L, 1, n_dec, dw_partial, 0, chunk);
        if (AN) hipLaunchKernelGGL((train_dw_stream_kernel<H>), dgrid, dim3(DW_WAVES * 64), 0, s, ws2, L, 1, n_dec, dw_partial, 1, chunk);
        PIN_CHECK_LAUNCH();
    }
    if (tp->defer_dec_reduce && want_dec) {  // the optimiser's decoder step sums the slot copies itself (pin_adam_dense.grad_partial)
        tl_deferred = {dw_partial, (int64_t)n_dec, 1.0f / dscale};
        return 0;
    }
    hipLaunchKernelGGL(train_finalize_kernel, dim3(want_dec ? cdiv(n_dec, 256) : 1), dim3(256), 0, s, dw_partial, n_dec, 1.0f / dscale,
                       dec_grad, loss_partial, grid, loss_out, 2);
    PIN_CHECK_LAUNCH();
    return 0;
}

static int cu_count() {
    static const int n_cu = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n_cu;
}

// phase: 3 = the whole step; 1 = up to the operand stream (the weight gradient is left to pin_train_weight_grad); 2 = that
static int train_step_impl(const pin_field* f, const pin_train_params* tp, const float* query, const float* nbr,
                           const int32_t* nn_count, const float* sdf_label, const float* sample_weight,
                           const int32_t* sample_ts, float* certainty_rw, int32_t* ts_update_rw,
                           float* feat_grad, float* dec_grad, double* loss_out, float* pred_out,
                           void* workspace, int64_t workspace_bytes, void* stream, int phase) {
    PIN_CHECK_ARG(f && tp, "NULL params");
    PIN_CHECK_ARG(f->k >= 1 && f->k <= PIN_MAX_K, "k must be in [1, 8]");
    PIN_CHECK_ARG(f->hidden == 32 || f->hidden == 64, "hidden must be 32 or 64");
    PIN_CHECK_ARG(f->levels >= 1 && f->levels <= MLP_MAX_LEVELS, "levels must be in [1, 4]");
    const int expand = f->weighted_first ? 1 : f->k;
    PIN_CHECK_ARG(tp->n_main > 0 && tp->n_eik >= 0, "bad batch sizes");
    const bool analytic = tp->eik_analytic != 0;
    PIN_CHECK_ARG(!analytic || tp->n_eik == 0, "analytic Eikonal term: no probe samples (n_eik = 0)");
    PIN_CHECK_ARG(!analytic || f->weighted_first != 0 || f->levels == 1,
                  "analytic Eikonal term with per-neighbour decoding: one-layer decoders (run_livox.yaml)");
    PIN_CHECK_ARG(!analytic || use_split_decoder(), "analytic Eikonal term: split-fp16 decoder only (unset PIN_MLP)");
    const int Q = tp->n_main + 6 * tp->n_eik;
    const int H = f->hidden, L = f->levels;
    PIN_CHECK_ARG(workspace && workspace_bytes >= (int64_t)train_ws_floats(analytic ? 2 * Q : Q, H, L, expand) * 4, "workspace too small");
    PIN_CHECK_ARG(loss_out && f->dec, "NULL pointer");
    PIN_CHECK_ARG(!(phase & 1) || (query && nbr && nn_count && sdf_label && feat_grad && f->feats), "NULL pointer");
    PIN_CHECK_ARG(!(phase & 1) || !tp->loss_weight_on || sample_weight, "loss_weight_on needs sample_weight");
    const bool fused = f->weighted_first != 0 || (L == 1 && use_split_decoder());
    PIN_CHECK_ARG(phase == 3 || fused, "defer_weight_grad / pin_train_weight_grad: only the fused tile paths split in two");
    hipStream_t s = as_stream(stream);
    TrainWs ws;
    ws.Qs = ((Q + 63) / 64) * 64;
    ws.QsT = ws.Qs * expand;
    carve_ws(ws, reinterpret_cast<float*>(workspace), H, L);
    const float4* nb4 = reinterpret_cast<const float4*>(nbr);
    const dim3 mgrid(cdiv(ws.Qs, MF_BLOCK)), mblock(MF_BLOCK);
    // (per-neighbour decoding only: weighted_first takes the fused path below)
#define PIN_TRAIN_MFMA(KERNEL, ...)                                                                  \
    do {                                                                                             \
        if (H == 64) hipLaunchKernelGGL((KERNEL<64, false>), mgrid, mblock, 0, s, __VA_ARGS__);      \
        else hipLaunchKernelGGL((KERNEL<32, false>), mgrid, mblock, 0, s, __VA_ARGS__);              \
    } while (0)
    const bool quad = f->weighted_first != 0;
    const int want_dec = dec_grad != nullptr;
    if (!fused) PIN_CHECK_HIP(hipMemsetAsync(loss_out, 0, 2 * sizeof(double), s));  // (the fused paths sum per-block partials)
    if (quad && analytic) {  // weighted_first, Eikonal term on the autograd gradient of every sample: train_fused_an_kernel
        PIN_CHECK_ARG(phase == 3, "analytic Eikonal term: the step is not split in two");
        return H == 64 ? launch_fused_an<64>(f, tp, query, nb4, nn_count, sdf_label, sample_weight, sample_ts, certainty_rw, ts_update_rw,
                                             feat_grad, dec_grad, loss_out, pred_out, workspace, workspace_bytes, cu_count(), s)
                       : launch_fused_an<32>(f, tp, query, nb4, nn_count, sdf_label, sample_weight, sample_ts, certainty_rw, ts_update_rw,
                                             feat_grad, dec_grad, loss_out, pred_out, workspace, workspace_bytes, cu_count(), s);
    }
    if (quad) {  // weighted_first: the fused tile kernel + the streamed weight gradient (train_fused.h)
        FusedColor none;
        memset(&none, 0, sizeof(none));
        return H == 64 ? launch_fused<64, 1>(f, tp, none, query, nb4, nn_count, sdf_label, sample_weight, sample_ts, certainty_rw,
                                             ts_update_rw, feat_grad, dec_grad, loss_out, pred_out, workspace, cu_count(), s, phase)
                       : launch_fused<32, 1>(f, tp, none, query, nb4, nn_count, sdf_label, sample_weight, sample_ts, certainty_rw,
                                             ts_update_rw, feat_grad, dec_grad, loss_out, pred_out, workspace, cu_count(), s, phase);
    }
    if (L == 1 && use_split_decoder()) {  // per-neighbour decoding, one-layer decoder: (query, neighbour)-column tiles
#define PIN_NWF(HH, ANALYTIC)                                                                                                   \
    launch_fused_nwf<HH, ANALYTIC>(f, tp, query, nb4, nn_count, sdf_label, sample_weight, sample_ts, certainty_rw, ts_update_rw, \
                                   feat_grad, dec_grad, loss_out, pred_out, workspace, workspace_bytes, cu_count(), s, phase)
        if (analytic) return H == 64 ? PIN_NWF(64, true) : PIN_NWF(32, true);
        return H == 64 ? PIN_NWF(64, false) : PIN_NWF(32, false);
#undef PIN_NWF
    }
    PIN_CHECK_ARG(!analytic, "analytic Eikonal term: built for per-neighbour decoding with a one-layer decoder (run_livox.yaml)");
    // deeper per-neighbour decoders: 64 queries per wave, activations and deltas through the unit-major workspace
    PIN_TRAIN_MFMA(train_fwd_mfma_kernel, *f, query, nb4, nn_count, Q, tp->n_main, ws, certainty_rw, ts_update_rw, sample_ts);
    PIN_CHECK_LAUNCH();
    hipLaunchKernelGGL(train_loss_kernel, dim3(cdiv(tp->n_main + tp->n_eik, 256)), dim3(256), 0, s, *tp, sdf_label,
                       sample_weight, ws, loss_out);
    PIN_CHECK_LAUNCH();
    PIN_TRAIN_MFMA(train_bwd_mfma_kernel, *f, nb4, nn_count, Q, ws, feat_grad, want_dec);
    PIN_CHECK_LAUNCH();
    if (want_dec) {
        DwLayers dl;
        // sample columns: Q for weighted_first; k blocks of Qs otherwise (padding columns carry zero deltas)
        const int QT = expand == 1 ? Q : ws.QsT;
        dl.H = H; dl.L = L; dl.Q = QT; dl.Qs = ws.QsT; dl.OD = 1;
        int per_wave = 64;  // multiple of 16; cap the grid at ~256 blocks per layer
        while ((long)cdiv(QT, per_wave * 4) > 256) per_wave *= 2;
        dl.per_wave = per_wave;
        hipLaunchKernelGGL(train_dw_kernel, dim3(cdiv(QT, per_wave * 4), L + 1), dim3(256), 0, s, ws, dl, dec_grad);
        PIN_CHECK_LAUNCH();
    }
    if (pred_out) PIN_CHECK_HIP(hipMemcpyAsync(pred_out, ws.pred, sizeof(float) * tp->n_main, hipMemcpyDeviceToDevice, s));
    return 0;
}

extern "C" int pin_train_step(const pin_field* f, const pin_train_params* tp, const float* query, const float* nbr,
                              const int32_t* nn_count, const float* sdf_label, const float* sample_weight,
                              const int32_t* sample_ts, float* certainty_rw, int32_t* ts_update_rw,
                              float* feat_grad, float* dec_grad, double* loss_out, float* pred_out,
                              void* workspace, int64_t workspace_bytes, void* stream) {
    PIN_ENTER();
    tl_deferred = {nullptr, 0, 0.f};
    return train_step_impl(f, tp, query, nbr, nn_count, sdf_label, sample_weight, sample_ts, certainty_rw, ts_update_rw, feat_grad,
                           dec_grad, loss_out, pred_out, workspace, workspace_bytes, stream, (tp && tp->defer_weight_grad) ? 1 : 3);
}

extern "C" int pin_train_deferred_partial(const float** partial_out, int32_t* slots_out, int64_t* n_out, float* scale_out) {
    PIN_ENTER();
    PIN_CHECK_ARG(partial_out && slots_out && n_out && scale_out, "NULL pointer");
    if (tl_deferred.partial == nullptr) return -1;  // (not an error to report: the caller asks whether the call deferred)
    *partial_out = tl_deferred.partial; *slots_out = DW_SLOTS; *n_out = tl_deferred.n; *scale_out = tl_deferred.scale;
    return 0;
}

// The two-stream form of a group (engine.MapTrainer.step_batch's `overlap` path, the default with a colour decoder): per iteration
//   main:  lazy-Adam launch of the geometry table (no rider) -> [wait: the decoder's step of the iteration before] -> tile kernel
//   side:  [wait: the tile kernel] -> weight gradient + its reduction -> the decoder's step (dense rider alone, image written through)
//   main:  lazy-Adam launch of the colour table with the colour decoder riding along -> colour tile kernel + its weight gradient
// so that the weight gradient and the decoder's step of the SDF term run beside the colour term and the next iteration's lazy
// launch.  The two events are the library's own (per thread); the caller orders the streams in front of and behind the group.
// the colour term of iteration i of a group (mapper.py:668-671, 802-812): the colour table's lazy launch with the colour decoder
// riding along (unless frozen), then the colour tile kernel and its weight gradient on the iteration's queries / records
static int train_group_color(const pin_train_group* g, int i, int step, const float* query, const float* nbr, const int32_t* nn,
                             const float* label, const float* weight, void* stream) {
    float* const cfeats = const_cast<float*>(g->fc->feats);
    const pin_adam_dense* cd = g->c_dense.param != nullptr ? &g->c_dense : nullptr;
    int rc = g->rows_form
                 ? pin_adam_lazy_prepare_rows(nbr, g->n_records, cfeats, g->c_feat_grad, g->c_exp_avg, g->c_exp_avg_sq, g->c_pending,
                                              g->c_row_flags, g->n_rows, step, g->coef, g->t_max, g->beta1, g->beta2, g->eps, cd, stream)
                 : pin_adam_lazy_prepare(nbr, g->n_records, cfeats, g->c_feat_grad, g->c_exp_avg, g->c_exp_avg_sq, g->c_pending, step,
                                         g->coef, g->t_max, g->beta1, g->beta2, g->eps, cd, stream);
    if (rc) return rc;
    return pin_train_color_step(g->fc, g->cp, query, nbr, nn, label, g->color_label + (int64_t)i * g->color_stride, weight,
                                g->c_feat_grad, cd ? g->c_dec_grad : nullptr, g->c_loss_out, g->c_workspace, g->c_workspace_bytes, stream);
}

static int train_group_two_streams(const pin_field* f, pin_train_params t, pin_train_group* g, void* stream) {
    static thread_local hipEvent_t ev_main = nullptr, ev_side = nullptr;
    if (ev_main == nullptr) {
        if (hipEventCreateWithFlags(&ev_main, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev_side, hipEventDisableTiming) != hipSuccess) {
            ev_main = ev_side = nullptr;
            return fail(-2, "pin_train_group_steps: cannot create the events of the two-stream form");
        }
    }
    PIN_CHECK_ARG(g->partial == nullptr, "the two-stream form reduces every iteration's weight gradient itself");
    if (g->fc != nullptr)
        PIN_CHECK_ARG(g->cp && g->color_label && g->c_feat_grad && g->c_workspace && g->c_pending && g->c_loss_out, "colour branch: NULL pointer");
    const hipStream_t main = reinterpret_cast<hipStream_t>(stream), side = reinterpret_cast<hipStream_t>(g->side_stream);
    pin_adam_dense d = g->dense;
    d.grad_partial = nullptr; d.partial_slots = 0; d.partial_scale = 0.f;
    float* const feats = const_cast<float*>(f->feats);
    t.defer_weight_grad = 1;
    t.defer_dec_reduce = 0;
    for (int i = 0; i < g->n_iters; ++i) {
        const int step = g->first_step + i;
        const float* nbr = g->nbr + (int64_t)i * g->nbr_stride;
        const float* query = g->query + (int64_t)i * g->query_stride;
        const int32_t* nn = g->nn + (int64_t)i * g->nn_stride;
        const float* label = g->sdf_label + (int64_t)i * g->label_stride;
        const float* weight = g->sample_weight ? g->sample_weight + (int64_t)i * g->weight_stride : nullptr;
        int rc = g->rows_form
                     ? pin_adam_lazy_prepare_rows(nbr, g->n_records, feats, g->feat_grad, g->exp_avg, g->exp_avg_sq, g->pending, g->row_flags,
                                                  g->n_rows, step, g->coef, g->t_max, g->beta1, g->beta2, g->eps, nullptr, stream)
                     : pin_adam_lazy_prepare(nbr, g->n_records, feats, g->feat_grad, g->exp_avg, g->exp_avg_sq, g->pending, step, g->coef,
                                             g->t_max, g->beta1, g->beta2, g->eps, nullptr, stream);
        if (rc) return rc;
        if (i > 0 && hipStreamWaitEvent(main, ev_side, 0) != hipSuccess) return fail(-2, "pin_train_group_steps: hipStreamWaitEvent");
        rc = pin_train_step(f, &t, query, nbr, nn, label, weight, g->sample_ts ? g->sample_ts + (int64_t)i * g->ts_stride : nullptr,
                            g->certainty_rw, g->ts_update_rw, g->feat_grad, g->dec_grad, g->loss_out, nullptr, g->workspace,
                            g->workspace_bytes, stream);
        if (rc) return rc;
        if (hipEventRecord(ev_main, main) != hipSuccess || hipStreamWaitEvent(side, ev_main, 0) != hipSuccess)
            return fail(-2, "pin_train_group_steps: cannot order the side stream behind the tile kernel");
        rc = pin_train_weight_grad(f, &t, g->dec_grad, g->loss_out, g->workspace, g->workspace_bytes, g->side_stream);
        if (rc) return rc;
        rc = pin_adam_lazy_flush(nullptr, nullptr, nullptr, nullptr, nullptr, 0, step, g->coef, g->t_max, g->beta1, g->beta2, g->eps, &d,
                                 g->side_stream);
        if (rc) return rc;
        if (hipEventRecord(ev_side, side) != hipSuccess) return fail(-2, "pin_train_group_steps: hipEventRecord");
        if (g->fc != nullptr) {
            rc = train_group_color(g, i, step, query, nbr, nn, label, weight, stream);
            if (rc) return rc;
        }
    }
    return 0;
}

extern "C" int pin_train_group_steps(const pin_field* f, const pin_train_params* tp, pin_train_group* g, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(f && tp && g && g->n_iters >= 0 && g->first_step >= 1, "bad arguments");
    PIN_CHECK_ARG(g->query && g->nbr && g->nn && g->sdf_label && g->feat_grad && g->workspace && g->pending && g->coef, "NULL pointer");
    const bool rider = g->dense.param != nullptr;  // (NULL: a frozen decoder -- no rider, no weight gradient, utils/tools.py:263-292)
    PIN_CHECK_ARG(!rider || (g->dec_grad != nullptr && g->dense.grad == g->dec_grad), "the rider must be the decoder whose gradient the steps write");
    if (g->fc != nullptr)
        PIN_CHECK_ARG(g->cp && g->color_label && g->c_feat_grad && g->c_workspace && g->c_pending && g->c_loss_out, "colour branch: NULL pointer");
    pin_train_params t = *tp;
    if (g->side_stream != nullptr) {
        PIN_CHECK_ARG(rider, "the two-stream form is the form of a TRAINED decoder (its weight gradient and step go to the side stream)");
        return train_group_two_streams(f, t, g, stream);
    }
    pin_adam_dense d = g->dense;
    float* const feats = const_cast<float*>(f->feats);  // (the field's table is what the optimiser steps)
    t.defer_weight_grad = 0;
    for (int i = 0; i < g->n_iters; ++i) {
        const int step = g->first_step + i;
        const float* nbr = g->nbr + (int64_t)i * g->nbr_stride;
        const float* query = g->query + (int64_t)i * g->query_stride;
        const int32_t* nn = g->nn + (int64_t)i * g->nn_stride;
        const float* label = g->sdf_label + (int64_t)i * g->label_stride;
        const float* weight = g->sample_weight ? g->sample_weight + (int64_t)i * g->weight_stride : nullptr;
        d.grad_partial = g->partial; d.partial_slots = g->partial_slots; d.partial_scale = g->partial_scale;
        int rc = g->rows_form
                     ? pin_adam_lazy_prepare_rows(nbr, g->n_records, feats, g->feat_grad, g->exp_avg, g->exp_avg_sq, g->pending, g->row_flags,
                                                  g->n_rows, step, g->coef, g->t_max, g->beta1, g->beta2, g->eps, rider ? &d : nullptr, stream)
                     : pin_adam_lazy_prepare(nbr, g->n_records, feats, g->feat_grad, g->exp_avg, g->exp_avg_sq, g->pending, step, g->coef,
                                             g->t_max, g->beta1, g->beta2, g->eps, rider ? &d : nullptr, stream);
        if (rc) return rc;
        g->partial = nullptr;  // (taken by the launch above)
        // (the deferred reduction: a trained decoder, nothing else between the launches that reads its gradient -- no colour branch --
        // and not the call's last iteration: engine.MapTrainer.step_batch's `defer_reduce`)
        t.defer_dec_reduce = (rider && g->fc == nullptr && !(g->last_of_call && i == g->n_iters - 1)) ? 1 : 0;
        rc = pin_train_step(f, &t, query, nbr, nn, label, weight, g->sample_ts ? g->sample_ts + (int64_t)i * g->ts_stride : nullptr,
                            g->certainty_rw, g->ts_update_rw, g->feat_grad, rider ? g->dec_grad : nullptr, g->loss_out, nullptr, g->workspace,
                            g->workspace_bytes, stream);
        if (rc) return rc;
        if (t.defer_dec_reduce && tl_deferred.partial != nullptr) {
            PIN_CHECK_ARG(tl_deferred.n == d.n, "the deferred weight gradient belongs to a decoder of another size");
            g->partial = tl_deferred.partial; g->partial_slots = DW_SLOTS; g->partial_scale = tl_deferred.scale;
        }
        if (g->fc != nullptr) {  // the colour term in line (a frozen SDF decoder, or a caller that wants one stream)
            rc = train_group_color(g, i, step, query, nbr, nn, label, weight, stream);
            if (rc) return rc;
        }
    }
    return 0;
}

extern "C" int pin_train_weight_grad(const pin_field* f, const pin_train_params* tp, float* dec_grad, double* loss_out,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
    PIN_ENTER();
    tl_deferred = {nullptr, 0, 0.f};
    return train_step_impl(f, tp, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, dec_grad,
                           loss_out, nullptr, workspace, workspace_bytes, stream, 2);
}

extern "C" int pin_train_color_step(const pin_field* fc, const pin_train_color_params* tp, const float* query,
                                    const float* nbr, const int32_t* nn_count, const float* sdf_label,
                                    const float* color_label, const float* sample_weight, float* feat_grad,
                                    float* dec_grad, double* loss_out, void* workspace, int64_t workspace_bytes,
                                    void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(fc && tp, "NULL params");
    PIN_CHECK_ARG(fc->out_dim == 3, "colour field must have 3 output heads");
    PIN_CHECK_ARG(fc->k >= 1 && fc->k <= PIN_MAX_K && (fc->hidden == 32 || fc->hidden == 64) && fc->levels >= 1 &&
                      fc->levels <= MLP_MAX_LEVELS, "bad colour field");
    PIN_CHECK_ARG(tp->n_main > 0, "bad batch size");
    const int Q = tp->n_main, H = fc->hidden, L = fc->levels;
    const int expand = fc->weighted_first ? 1 : fc->k;
    PIN_CHECK_ARG(workspace && workspace_bytes >= (int64_t)train_ws_floats(Q, H, L, expand) * 4 + 256, "workspace too small");
    PIN_CHECK_ARG(query && nbr && nn_count && sdf_label && color_label && feat_grad && loss_out && fc->feats && fc->dec,
                  "NULL pointer");
    PIN_CHECK_ARG(!tp->loss_weight_on || sample_weight, "loss_weight_on needs sample_weight");
    hipStream_t s = as_stream(stream);
    TrainWs ws;
    ws.Qs = ((Q + 63) / 64) * 64;
    ws.QsT = ws.Qs * expand;
    carve_ws(ws, reinterpret_cast<float*>(workspace), H, L);
    int* count = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + (size_t)train_ws_floats(Q, H, L, expand) * 4);
    const float4* nb4 = reinterpret_cast<const float4*>(nbr);
    const dim3 mgrid(cdiv(ws.Qs, MF_BLOCK)), mblock(MF_BLOCK);
    const bool given = tp->surface_count != nullptr;  // (a shard of a larger batch: the caller knows the batch's count)
    if (given) count = const_cast<int*>(tp->surface_count);
    else PIN_CHECK_HIP(hipMemsetAsync(count, 0, sizeof(int), s));
    if (fc->weighted_first) {  // the fused tile kernel with three heads + the streamed weight gradient (train_fused.h)
        if (!given) hipLaunchKernelGGL(color_count_kernel, dim3(cdiv(Q, 256)), dim3(256), 0, s, sdf_label, Q, tp->surface_range, count);
        pin_train_params sp;
        memset(&sp, 0, sizeof(sp));
        sp.n_main = Q;  // plain tiles of 16 samples, no probes
        sp.dec_image_current = tp->dec_image_current;
        sp.inv_n_main = (given && tp->n_main_global > Q) ? 1.f / (float)tp->n_main_global : 0.f;  // (only the gradient scale's choice reads it in the colour kernel)
        FusedColor fcol;
        fcol.color = color_label; fcol.count = count; fcol.surface_range = tp->surface_range; fcol.weight_i = tp->weight_i;
        fcol.loss_weight_on = tp->loss_weight_on;
        return H == 64 ? launch_fused<64, 3>(fc, &sp, fcol, query, nb4, nn_count, sdf_label, sample_weight, nullptr, nullptr, nullptr,
                                             feat_grad, dec_grad, loss_out, nullptr, workspace, cu_count(), s)
                       : launch_fused<32, 3>(fc, &sp, fcol, query, nb4, nn_count, sdf_label, sample_weight, nullptr, nullptr, nullptr,
                                             feat_grad, dec_grad, loss_out, nullptr, workspace, cu_count(), s);
    }
    PIN_CHECK_HIP(hipMemsetAsync(loss_out, 0, sizeof(double), s));
#define PIN_TRAIN_C(KERNEL, ...)                                                                    \
    do {                                                                                            \
        if (H == 64) hipLaunchKernelGGL((KERNEL<64, false, 3>), mgrid, mblock, 0, s, __VA_ARGS__);  \
        else hipLaunchKernelGGL((KERNEL<32, false, 3>), mgrid, mblock, 0, s, __VA_ARGS__);          \
    } while (0)
    PIN_TRAIN_C(train_fwd_mfma_kernel, *fc, query, nb4, nn_count, Q, Q, ws, (float*)nullptr, (int*)nullptr, (const int*)nullptr);
    if (!given) hipLaunchKernelGGL(color_count_kernel, dim3(cdiv(Q, 256)), dim3(256), 0, s, sdf_label, Q, tp->surface_range, count);
    hipLaunchKernelGGL(train_color_loss_kernel, dim3(cdiv(Q, 256)), dim3(256), 0, s, *tp, sdf_label, color_label,
                       sample_weight, ws, count, loss_out);
    const int want_dec = dec_grad != nullptr;
    PIN_TRAIN_C(train_bwd_mfma_kernel, *fc, nb4, nn_count, Q, ws, feat_grad, want_dec);
#undef PIN_TRAIN_C
    if (want_dec) {
        DwLayers dl;
        const int QT = expand == 1 ? Q : ws.QsT;
        dl.H = H; dl.L = L; dl.Q = QT; dl.Qs = ws.QsT; dl.OD = 3;
        int per_wave = 64;
        while ((long)cdiv(QT, per_wave * 4) > 256) per_wave *= 2;
        dl.per_wave = per_wave;
        hipLaunchKernelGGL(train_dw_kernel, dim3(cdiv(QT, per_wave * 4), L + 1), dim3(256), 0, s, ws, dl, dec_grad);
    }
    PIN_CHECK_LAUNCH();
    return 0;
}

// ---- semantic head (sem.h) ----------------------------------------------------------------------------------------------------
static size_t sem_ws_floats(int Q, int H, int L, int expand) {
    const size_t Qs = (size_t)((Q + 63) / 64) * 64, QsT = Qs * (size_t)(expand < 1 ? 1 : expand);
    return QsT * (12 + (size_t)L * H + (size_t)L * H + SEM_MAX_HEADS);
}

extern "C" int64_t pin_sem_workspace_bytes(int32_t n_queries, int32_t hidden, int32_t levels, int32_t expand) {
    return (int64_t)sem_ws_floats(n_queries, hidden, levels, expand) * 4 + 256;
}

extern "C" int pin_sem_select(const int32_t* labels, int32_t n, int32_t freespace_label_on, int32_t decimation, uint8_t* selected_out,
                              int32_t* count_out, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n >= 0 && decimation >= 1 && count_out, "bad arguments");
    hipStream_t s = as_stream(stream);
    PIN_CHECK_HIP(hipMemsetAsync(count_out, 0, sizeof(int), s));
    if (n == 0) return 0;
    PIN_CHECK_ARG(labels && selected_out, "NULL pointer");
    hipLaunchKernelGGL(sem_select_kernel, dim3(1), dim3(1024), 0, s, labels, n, freespace_label_on, decimation, selected_out, count_out);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_gather_labels_drawn(const int32_t* pool_sem, const int64_t* index_history, int32_t n_history,
                                       const int64_t* index_new_batch, const int64_t* new_idx, int32_t n, int32_t n_batches,
                                       int64_t hist_stride, int64_t new_stride, int32_t* out, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n >= 0 && n_batches >= 0 && n_batches <= 65535 && n_history >= 0 && n_history <= n, "bad sizes");
    if (n == 0 || n_batches == 0) return 0;
    PIN_CHECK_ARG(pool_sem && out && (n_history == 0 || index_history) && (n_history == n || (index_new_batch && new_idx)), "NULL pointer");
    hipLaunchKernelGGL(gather_labels_drawn_kernel, dim3(cdiv(n, 256), n_batches), dim3(256), 0, as_stream(stream), pool_sem,
                       reinterpret_cast<const long long*>(index_history), n_history, reinterpret_cast<const long long*>(index_new_batch),
                       reinterpret_cast<const long long*>(new_idx), n, (long)hist_stride, (long)new_stride, out);
    PIN_CHECK_LAUNCH();
    return 0;
}

static int check_sem_field(const pin_field* f, int heads) {
    PIN_CHECK_ARG(f != nullptr && f->dec != nullptr, "field / decoder NULL");
    PIN_CHECK_ARG(f->k >= 1 && f->k <= PIN_MAX_K && (f->hidden == 32 || f->hidden == 64) && f->levels >= 1 && f->levels <= MLP_MAX_LEVELS,
                  "bad semantic field (k, hidden in {32, 64}, 1..4 layers)");
    PIN_CHECK_ARG(heads >= 2 && heads <= SEM_MAX_HEADS, "semantic heads must be in [2, 32] (sem_class_count + 1)");
    return 0;
}

extern "C" int pin_train_sem_step(const pin_field* f, const pin_sem_params* sp, const float* query, const float* nbr,
                                  const int32_t* nn_count, float* feat_grad, float* dec_grad, double* loss_out, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(sp != nullptr, "params NULL");
    if (int e = check_sem_field(f, sp->heads)) return e;
    const int Q = sp->n_main, H = f->hidden, L = f->levels;
    if (Q == 0) return 0;
    PIN_CHECK_ARG(Q > 0, "bad batch size");
    const int expand = f->weighted_first ? 1 : f->k;
    PIN_CHECK_ARG(workspace && workspace_bytes >= pin_sem_workspace_bytes(Q, H, L, expand), "workspace too small");
    PIN_CHECK_ARG(query && nbr && nn_count && feat_grad && loss_out && f->feats && sp->labels && sp->selected && sp->count, "NULL pointer");
    hipStream_t s = as_stream(stream);
    TrainWs ws;
    memset(&ws, 0, sizeof(ws));
    ws.Qs = ((Q + 63) / 64) * 64;
    ws.QsT = ws.Qs * expand;
    float* w = reinterpret_cast<float*>(workspace);
    ws.z = w; w += (size_t)12 * ws.QsT;
    ws.h = w; w += (size_t)L * H * ws.QsT;
    ws.d = w;
    SemTrain st;
    st.labels = sp->labels; st.sel = sp->selected; st.count = sp->count; st.weight_s = sp->weight_s; st.S = sp->heads;
    const int want_dec = dec_grad != nullptr;
    const float4* nb4 = reinterpret_cast<const float4*>(nbr);
    const dim3 grid(cdiv(ws.Qs, MF_BLOCK)), block(MF_BLOCK);
#define PIN_SEM_T(HH, WW) hipLaunchKernelGGL((sem_train_kernel<HH, WW>), grid, block, 0, s, *f, query, nb4, nn_count, Q, ws, st, feat_grad, want_dec, loss_out)
    if (H == 64) { if (f->weighted_first) PIN_SEM_T(64, true); else PIN_SEM_T(64, false); }
    else { if (f->weighted_first) PIN_SEM_T(32, true); else PIN_SEM_T(32, false); }
#undef PIN_SEM_T
    PIN_CHECK_LAUNCH();
    if (want_dec) {  // dW_l = sum_q delta_l[q] (x) input_l[q] over the unit-major rows (train_dw_kernel; the head layer has `heads` rows)
        DwLayers dl;
        const int QT = expand == 1 ? Q : ws.QsT;
        dl.H = H; dl.L = L; dl.Q = QT; dl.Qs = ws.QsT; dl.OD = sp->heads;
        int per_wave = 64;
        while ((long)cdiv(QT, per_wave * 4) > 256) per_wave *= 2;
        dl.per_wave = per_wave;
        hipLaunchKernelGGL(train_dw_kernel, dim3(cdiv(QT, per_wave * 4), L + 1), dim3(256), 0, s, ws, dl, dec_grad);
        PIN_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int pin_sem_query(const pin_field* f, const float* query, const float* nbr, const int32_t* nn_count, int32_t n, int32_t heads,
                             int32_t* label_out, float* logprob_out, void* stream) {
    PIN_ENTER();
    if (int e = check_sem_field(f, heads)) return e;
    PIN_CHECK_ARG(n >= 0, "n < 0");
    if (n == 0) return 0;
    PIN_CHECK_ARG(query && nbr && nn_count && f->feats && (label_out || logprob_out), "NULL pointer");
    const float4* nb4 = reinterpret_cast<const float4*>(nbr);
    const dim3 grid(cdiv(n, MF_BLOCK)), block(MF_BLOCK);
    hipStream_t s = as_stream(stream);
#define PIN_SEM_Q(HH, WW) hipLaunchKernelGGL((sem_query_kernel<HH, WW, 0>), grid, block, 0, s, *f, query, nb4, nn_count, n, heads, 0, label_out, logprob_out)
    if (f->hidden == 64) { if (f->weighted_first) PIN_SEM_Q(64, true); else PIN_SEM_Q(64, false); }
    else { if (f->weighted_first) PIN_SEM_Q(32, true); else PIN_SEM_Q(32, false); }
#undef PIN_SEM_Q
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_decoder_sem(const pin_field* f, const float* feat_in, int32_t n, int32_t heads, int32_t raw, float* out, void* stream) {
    PIN_ENTER();
    if (int e = check_sem_field(f, heads)) return e;
    PIN_CHECK_ARG(n >= 0, "n < 0");
    if (n == 0) return 0;
    PIN_CHECK_ARG(feat_in && out, "NULL pointer");
    const dim3 grid(cdiv(n, MF_BLOCK)), block(MF_BLOCK);
    hipStream_t s = as_stream(stream);
    if (f->hidden == 64) hipLaunchKernelGGL((sem_query_kernel<64, true, 1>), grid, block, 0, s, *f, feat_in, (const float4*)nullptr, (const int*)nullptr, n, heads, raw, (int*)nullptr, out);
    else hipLaunchKernelGGL((sem_query_kernel<32, true, 1>), grid, block, 0, s, *f, feat_in, (const float4*)nullptr, (const int*)nullptr, n, heads, raw, (int*)nullptr, out);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t step,
                             float lr, float beta1, float beta2, float eps, int32_t zero_grad, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n >= 0 && step >= 1, "bad n / step");
    if (n == 0) return 0;
    PIN_CHECK_ARG(param && grad && exp_avg && exp_avg_sq, "NULL pointer");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq,
                       (long)n, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, zero_grad);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_mark_rows(const float* nbr, int64_t n_records, uint8_t* row_flags, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n_records >= 0, "n_records < 0");
    if (n_records == 0) return 0;
    PIN_CHECK_ARG(nbr && row_flags, "NULL pointer");
    hipLaunchKernelGGL(mark_rows_kernel, dim3(cdiv(n_records, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(nbr), (long)n_records, row_flags);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_adam_step_rows(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n_rows,
                                  int32_t row_width, const uint8_t* row_flags, int32_t step, float lr, float beta1,
                                  float beta2, float eps, int32_t zero_grad, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n_rows >= 0 && row_width >= 1 && step >= 1, "bad sizes");
    if (n_rows == 0) return 0;
    PIN_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && row_flags, "NULL pointer");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const long n = (long)n_rows * row_width;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(adam_rows_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, n,
                       row_width, row_flags, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, zero_grad);
    PIN_CHECK_LAUNCH();
    return 0;
}

static int lazy_dense(const pin_adam_dense* dense, const float* coef, pin_adam_dense& d) {
    memset(&d, 0, sizeof(d));
    if (dense != nullptr && dense->n > 0) {
        PIN_CHECK_ARG(dense->param && dense->grad && dense->exp_avg && dense->exp_avg_sq && coef, "dense tensor: NULL pointer");
        if (dense->image != nullptr) {
            const int H = dense->hidden, L = dense->levels, OD = dense->out_dim;
            PIN_CHECK_ARG((H == 32 || H == 64) && L >= 1 && L <= MLP_MAX_LEVELS && (OD == 1 || OD == 3) &&
                              dense->n == (int64_t)H * MLP_IN + H + (int64_t)(L - 1) * (H * H + H) + OD * H + OD,
                          "dense tensor: the image's decoder shape does not match the parameter count");
        }
        PIN_CHECK_ARG(dense->grad_partial == nullptr || dense->partial_slots == DW_SLOTS,
                      "dense tensor: grad_partial must hold the slot copies pin_train_deferred_partial reports");
        d = *dense;
    }
    return 0;
}

extern "C" int pin_adam_lazy_prepare(const float* nbr, int64_t n_records, float* param, float* grad, float* exp_avg,
                                     float* exp_avg_sq, int32_t* pending, int32_t step, const float* coef, int32_t t_max,
                                     float beta1, float beta2, float eps, const pin_adam_dense* dense, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n_records >= 0 && step >= 1 && step <= t_max && t_max < 4096, "bad step (t_max < 4096: the coefficient table is staged in LDS)");
    pin_adam_dense d;
    if (int e = lazy_dense(step > 1 ? dense : nullptr, coef, d)) return e;  // (nothing to step before the first iteration)
    if (n_records == 0 && d.n == 0) return 0;
    PIN_CHECK_ARG(n_records == 0 || (nbr && param && grad && exp_avg && exp_avg_sq && pending && coef), "NULL pointer");
    const int rec_blocks = (int)cdiv(n_records * 8, 256), dense_blocks = (int)cdiv(d.n, 256);
    hipLaunchKernelGGL(adam_lazy_prepare_kernel, dim3(rec_blocks + dense_blocks), dim3(256), 2 * (t_max + 1) * sizeof(float), as_stream(stream),
                       reinterpret_cast<const float4*>(nbr), (long)n_records, param, grad, exp_avg, exp_avg_sq, pending, step, coef,
                       t_max, beta1, beta2, eps, rec_blocks, d, step - 1);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_adam_lazy_prepare_rows(const float* nbr, int64_t n_records, float* param, float* grad, float* exp_avg,
                                          float* exp_avg_sq, int32_t* pending, uint8_t* row_flags, int64_t n_rows, int32_t step,
                                          const float* coef, int32_t t_max, float beta1, float beta2, float eps,
                                          const pin_adam_dense* dense, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n_records >= 0 && n_rows >= 0 && step >= 1 && step <= t_max && t_max < 4096, "bad sizes / step (t_max < 4096)");
    pin_adam_dense d;
    if (int e = lazy_dense(step > 1 ? dense : nullptr, coef, d)) return e;  // (nothing to step before the first iteration)
    if (n_records == 0 && d.n == 0) return 0;
    PIN_CHECK_ARG(n_records == 0 || (nbr && param && grad && exp_avg && exp_avg_sq && pending && row_flags && coef), "NULL pointer");
    hipStream_t s = as_stream(stream);
    // from three records per row on (a 2^20 batch: six) the marking pass is skipped and every row counts as read (PIN_LAZY_ALL_ROWS=0:
    // always mark): 71 us of a 2.17 ms iteration at C4
    static const bool all_ok = [] { const char* e = getenv("PIN_LAZY_ALL_ROWS"); return !(e && e[0] == '0'); }();
    const int all_rows = all_ok && n_records >= 3 * n_rows ? 1 : 0;
    if (n_records > 0 && !all_rows)
        hipLaunchKernelGGL(mark_rows_kernel, dim3(cdiv(n_records, 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(nbr), (long)n_records,
                           row_flags);
    const int row_blocks = n_records > 0 ? (int)min((long)cdiv(n_rows, 32), 8192L) : 0, dense_blocks = (int)cdiv(d.n, 256);
    hipLaunchKernelGGL(adam_lazy_prepare_rows_kernel, dim3(row_blocks + dense_blocks), dim3(256), 2 * (t_max + 1) * sizeof(float), s, param,
                       grad, exp_avg, exp_avg_sq, pending, row_flags, (long)n_rows, step, coef, t_max, beta1, beta2, eps, row_blocks, d, step - 1,
                       all_rows);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_adam_lazy_flush(float* param, float* grad, float* exp_avg, float* exp_avg_sq, const int32_t* pending,
                                   int64_t n_rows, int32_t t_final, const float* coef, int32_t t_max, float beta1, float beta2,
                                   float eps, const pin_adam_dense* dense, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n_rows >= 0 && t_final >= 0 && t_final <= t_max && t_max < 4096, "bad sizes (t_max < 4096)");
    if (t_final == 0) return 0;
    pin_adam_dense d;
    if (int e = lazy_dense(dense, coef, d)) return e;
    if (n_rows == 0 && d.n == 0) return 0;
    PIN_CHECK_ARG(n_rows == 0 || (param && grad && exp_avg && exp_avg_sq && pending && coef), "NULL pointer");
    const long n = (long)n_rows * PIN_FEATURE_DIM;
    const long row_waves = (n_rows + 63) / 64;  // a wave per 64 rows, at most 4096 blocks of four waves
    const int blocks = n_rows == 0 ? 0 : (int)((row_waves + 3) / 4 < 4096 ? (row_waves + 3) / 4 : 4096), dense_blocks = (int)cdiv(d.n, 256);
    hipLaunchKernelGGL(adam_lazy_flush_kernel, dim3(blocks + dense_blocks), dim3(256), 2 * (t_max + 1) * sizeof(float), as_stream(stream), param, grad, exp_avg,
                       exp_avg_sq, pending, n, t_final, coef, t_max, beta1, beta2, eps, blocks, d);
    PIN_CHECK_LAUNCH();
    return 0;
}

// pin_warmup (common.hip): asking for a kernel's attributes makes the runtime load this translation unit's code object now
// instead of inside the first frame that launches one of its kernels
namespace pin {
int pin_warm_train() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&make_queries_kernel)) == hipSuccess ? 0 : -2;
}
}  // namespace pin

#ifdef PIN_TF_STAMPS
extern "C" int pin_debug_tf_stamps(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pin::g_tf_stamps), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif
