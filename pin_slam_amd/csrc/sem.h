// Semantic head (config.semantic_on; config/lidar_slam/run_demo_sem.yaml): a second decoder over the SAME interpolated geometry
// feature with S = sem_class_count + 1 output heads, read through a log-softmax.
//   Decoder.sem_label_prob      model/decoder.py:100-103      F.log_softmax(mlp(features), dim=-1)
//   Mapper.mapping              utils/mapper.py:664-667       sem_pred = sem_label_prob(geo_feature) [ x weight_knn, summed over k ]
//                               utils/mapper.py:782-800       NLLLoss(mean) over the labelled samples ([::sem_label_decimation]
//                                                             of them), cur_loss += weight_s * sem_loss
//   Tracker.query_source_points utils/tracker.py:336-341      argmax of the (weighted) log-probabilities
//   Mesher.query_points         utils/mesher.py:137-145       the same
//
// Kernels: 64 queries per wave on the fp32 matrix cores, the building blocks of mlp_mfma.h for the H-wide layers and a head
// layer of up to 32 outputs laid out like a hidden layer with two output tiles (SemHead): logits land in the D layout --
// lane (n, g), register r of tile (mo, nt) holds head 16 mo + 4 g + r of query 16 nt + n -- which is also the B operand of
// the transposed head product, so the log-softmax, the NLL gradient and the seed of the backward sweep need no data movement
// beyond two 4-lane reductions per query.  Forward, loss and backward of a sample are ONE kernel (the loss of a sample does
// not depend on other samples once the number of selected samples is known: sem_select_kernel); activations and deltas go to
// the unit-major workspace of train.hip's generic path and train_dw_kernel turns them into the decoder's weight gradient.
// Not a benchmark path: no shipped BASELINE configuration has semantic_on.
#pragma once

namespace pin {

constexpr int SEM_MAX_HEADS = 32;

template <int H>
struct SemHead {
    static constexpr int MT = H / 16, MO = SEM_MAX_HEADS / 16;
    static constexpr int F_FLOATS = MO * MT * 256, TOTAL = F_FLOATS + SEM_MAX_HEADS;

    // lout.weight [S][H] then lout.bias [S] (state_dict order) -> the A-operand image of MfmaDecoder::hidden with MO output
    // tiles, rows >= S zero; bias at F_FLOATS + c
    __device__ static void stage(const float* __restrict__ head, int S, float* __restrict__ dst, int tid, int nthreads) {
        for (int e = tid; e < F_FLOATS; e += nthreads) {
            const int r = e & 3, lane = (e >> 2) & 63, kt = (e >> 8) % MT, mo = e / (256 * MT);
            const int c = 16 * mo + (lane & 15), u = 16 * kt + 4 * (lane >> 4) + r;
            dst[e] = c < S ? head[c * H + u] : 0.f;
        }
        for (int c = tid; c < SEM_MAX_HEADS; c += nthreads) dst[F_FLOATS + c] = c < S ? head[S * H + c] : 0.f;
    }

    // logits[mo][nt][r] = bias + sum_u Wo[16 mo + 4 g + r][u] h[u][query 16 nt + n]
    template <int NT>
    __device__ __forceinline__ static void forward(const float* __restrict__ F, const v4f_t (&h)[MT][NT], v4f_t (&lg)[MO][NT]) {
        const int lane = threadIdx.x & 63, g = lane >> 4;
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) {
            const v4f_t b4 = *reinterpret_cast<const v4f_t*>(F + F_FLOATS + 16 * mo + 4 * g);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) lg[mo][nt] = b4;
#pragma unroll
            for (int kt = 0; kt < MT; ++kt) {
                const v4f_t a4 = *reinterpret_cast<const v4f_t*>(F + ((mo * MT + kt) * 64 + lane) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        lg[mo][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[r], h[kt][nt][r], lg[mo][nt], 0, 0, 0);
            }
        }
    }

    // transposed head product: h[mj][nt][r] = mask ? sum_c Wo[c][16 mj + 4 g + r] dl[c][query] : 0   (MfmaDecoder::back_hidden
    // with the MO head tiles as the reduction dimension)
    template <int NT>
    __device__ __forceinline__ static void seed(const float* __restrict__ F, unsigned long long mm, const v4f_t (&dl)[MO][NT],
                                                v4f_t (&h)[MT][NT]) {
        const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
#pragma unroll
        for (int mj = 0; mj < MT; ++mj) {
            v4f_t acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = (v4f_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ko = 0; ko < MO; ++ko)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float at = F[((ko * MT + mj) * 64 + 16 * (n >> 2) + 4 * g + r) * 4 + (n & 3)];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, dl[ko][nt][r], acc[nt], 0, 0, 0);
                }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    h[mj][nt][r] = ((mm >> ((mj * NT + nt) * 4 + r)) & 1ull) ? acc[nt][r] : 0.f;
        }
    }

    // max / sum over the heads of a query: over this lane's 2 x 4 registers, then over the query's four lanes (n, g = 0..3)
    __device__ __forceinline__ static float rows_max(float v) {
        v = fmaxf(v, __shfl_xor(v, 16, 64));
        return fmaxf(v, __shfl_xor(v, 32, 64));
    }

    // logits -> log-softmax over the S valid heads, in place (F.log_softmax: x - max - log(sum exp(x - max)))
    template <int NT>
    __device__ __forceinline__ static void log_softmax(v4f_t (&lg)[MO][NT], int S) {
        const int g = (threadIdx.x & 63) >> 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float m = -INFINITY;
#pragma unroll
            for (int mo = 0; mo < MO; ++mo)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * mo + 4 * g + r < S) m = fmaxf(m, lg[mo][nt][r]);
            m = rows_max(m);
            float se = 0.f;
#pragma unroll
            for (int mo = 0; mo < MO; ++mo)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * mo + 4 * g + r < S) se += expf(lg[mo][nt][r] - m);
            se = rows_sum_lds(se);
            const float lse = m + logf(se);
#pragma unroll
            for (int mo = 0; mo < MO; ++mo)
#pragma unroll
                for (int r = 0; r < 4; ++r) lg[mo][nt][r] = (16 * mo + 4 * g + r < S) ? lg[mo][nt][r] - lse : -INFINITY;
        }
    }
};

// mask word of layer i out of the (statically indexed) array: a dynamic index would put the array into scratch memory
__device__ __forceinline__ unsigned long long sem_mask_of(const unsigned long long (&mk)[MLP_MAX_LEVELS], int i) {
    return i == 0 ? mk[0] : (i == 1 ? mk[1] : (i == 2 ? mk[2] : mk[3]));
}

template <int H>
struct SemLds {
    static constexpr int HEAD = MfmaLds<H>::TOTAL;            // the head image sits behind MfmaDecoder's weights + wave scratch
    static constexpr int TOTAL = HEAD + SemHead<H>::TOTAL;
};

__host__ __device__ inline size_t sem_head_offset(int H, int L) { return (size_t)H * MLP_IN + H + (size_t)(L - 1) * ((size_t)H * H + H); }

// hidden layers of the 64 queries of this wave (z of the own query per lane): h = last layer's activations, masks of every layer;
// STORE: activations to the unit-major workspace (train_dw_kernel's X operand)
template <int H, bool STORE>
__device__ __forceinline__ void sem_hidden_forward(const float* __restrict__ w, int L, float* __restrict__ xb, const float (&z)[MLP_IN],
                                                   v4f_t (&h)[H / 16][4], unsigned long long (&mk)[MLP_MAX_LEVELS],
                                                   float* __restrict__ hws, size_t Qs, size_t col0) {
    using D = MfmaDecoder<H>;
    constexpr int MT = H / 16, NT = 4;
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    v4f_t acc[MT][NT];
    D::put_z(xb, z);
    auto store = [&](int l) {
        if constexpr (STORE) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        hws[((size_t)l * H + 16 * mt + 4 * g + r) * Qs + col0 + 16 * nt + n] = h[mt][nt][r];
        }
    };
#pragma unroll
    for (int l = 0; l < MLP_MAX_LEVELS; ++l) mk[l] = 0ull;
    D::template layer0<NT>(w, xb, 0, acc);
    mk[0] = D::template relu<NT>(acc, h);
    store(0);
    for (int l = 1; l < L; ++l) {
        D::template hidden<NT>(w + D::OFF_HID + (l - 1) * D::HID_SZ, h, acc);
        const unsigned long long mm = D::template relu<NT>(acc, h);
        mk[1] = l == 1 ? mm : mk[1]; mk[2] = l == 2 ? mm : mk[2]; mk[3] = l == 3 ? mm : mk[3];
        store(l);
    }
    wave_lds_sync();
}

// ---- selection of the samples the loss runs over (utils/mapper.py:786-799) --------------------------------------------------
// label_mask = label > 0 (>= 0 with freespace_label_on); of the masked samples, IN ORDER, every `decimation`-th one
// (sem_pred[label_mask][::dec]).  One block walks the batch (an ordered rank is a prefix count): sel[i] in {0, 1},
// count_out = number selected.  n = 16 384: 16 trips; 2^20: 1 024 trips of ~0.1 us.
__global__ __launch_bounds__(1024) void sem_select_kernel(const int* __restrict__ labels, int n, int freespace_on, int decimation,
                                                          unsigned char* __restrict__ sel, int* __restrict__ count_out) {
    __shared__ int wave_cnt[16];
    __shared__ int base_s;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int selected = 0;
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        const bool m = i < n && (freespace_on ? labels[i] >= 0 : labels[i] > 0);
        const unsigned long long b = __ballot(m);
        const int before = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(b);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        const int rank = off + before;
        const bool s = m && (rank % decimation) == 0;
        if (i < n) sel[i] = s ? 1 : 0;
        selected += s ? 1 : 0;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += wave_cnt[w];
            base_s += t;
        }
        __syncthreads();
    }
    // number selected = ceil(masked / decimation); summed from the flags (one atomic per wave on a zeroed word)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) selected += __shfl_xor(selected, o, 64);
    if (lane == 0 && selected) atomicAdd(count_out, selected);
}

// Mapper.get_batch's sem_label gather (utils/mapper.py:490-491): rows index_history[i] for i < n_hist, rows
// new_idx[index_new_batch[i - n_hist]] after that -- the index arrays of pin_gather_batches_drawn
__global__ __launch_bounds__(256) void gather_labels_drawn_kernel(const int* __restrict__ pool_sem, const long long* __restrict__ index_hist,
                                                                  int n_hist, const long long* __restrict__ index_new_batch,
                                                                  const long long* __restrict__ new_idx, int n, long hist_stride,
                                                                  long new_stride, int* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t b = blockIdx.y;
    const size_t s = (size_t)(i < n_hist ? index_hist[b * hist_stride + i] : new_idx[index_new_batch[b * new_stride + (i - n_hist)]]);
    out[b * (size_t)n + i] = pool_sem[s];
}

struct SemTrain {
    const int* labels;            // [Q] semantic label of every batch sample
    const unsigned char* sel;     // [Q] 1 = the sample is in the loss (sem_select_kernel)
    const int* count;             // number of selected samples (the mean's denominator)
    float weight_s;               // config.weight_s
    int S;                        // heads
};

// One training iteration's semantic term: forward, NLL loss, backward into the geometry features (atomic scatter, as
// train_bwd_mfma_kernel) and -- through the workspace -- into the semantic decoder.  f.feats = geometry features, f.dec = the
// semantic decoder's flat parameters.
template <int H, bool WF>
__global__ __launch_bounds__(MF_BLOCK) void sem_train_kernel(pin_field f, const float* __restrict__ query, const float4* __restrict__ nbr,
                                                             const int* __restrict__ nn_count, int Q, TrainWs ws, SemTrain st,
                                                             float* __restrict__ feat_grad, int want_dec, double* __restrict__ loss_out) {
    using D = MfmaDecoder<H>;
    using Hd = SemHead<H>;
    constexpr int MT = H / 16, NT = 4, MO = Hd::MO;
    __shared__ __attribute__((aligned(16))) float lds[SemLds<H>::TOTAL];
    float* xb = lds + MfmaLds<H>::W + (threadIdx.x >> 6) * D::scratch_floats();
    float* Fo = lds + SemLds<H>::HEAD;
    D::stage(f.dec, f.levels, lds, threadIdx.x, MF_BLOCK, 0);
    Hd::stage(f.dec + sem_head_offset(H, f.levels), st.S, Fo, threadIdx.x, MF_BLOCK);
    __syncthreads();
    const int q0 = (blockIdx.x * (MF_BLOCK / 64) + (threadIdx.x >> 6)) * 64;
    if (q0 >= ws.Qs) return;  // whole wave
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int qi = q0 + lane;
    const bool active = qi < Q;
    const int qq = active ? qi : Q - 1;
    NbrW nb;
    float vx[PIN_MAX_K], vy[PIN_MAX_K], vz[PIN_MAX_K];
    bool quirk[PIN_MAX_K];
    neighbor_weights(nbr, nn_count[qq], qq, f.k, nb, vx, vy, vz, quirk);
    const float qx = query[3 * qq], qy = query[3 * qq + 1], qz = query[3 * qq + 2];
    const size_t QsT = ws.QsT;
    // per query tile (the D layout's view): label, selection, the loss gradient's scale
    int lab[NT];
    float coef[NT];
    const float inv_cnt = st.weight_s / fmaxf((float)(*st.count), 1.f);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int q = q0 + 16 * nt + n;
        const bool on = q < Q && st.sel[q] != 0;
        lab[nt] = q < Q ? st.labels[q] : -1;
        coef[nt] = on ? inv_cnt : 0.f;
        if (lab[nt] < 0 || lab[nt] >= st.S) coef[nt] = 0.f;  // (a label outside the heads cannot be a target; NLLLoss would raise)
    }
    double loss = 0.0;
    float* sdz = xb;                 // [32][8]
    float* sw = xb + 256;            // [32][8]
    int* sidx = reinterpret_cast<int*>(xb + 512);  // [32][8]

    // forward / loss / backward of one column block (col0: its first column in the unit-major workspace; wq[nt] = the factor
    // on the block's log-probabilities in sem_pred: 1 with interpolate-first decoding, w_t of neighbour t otherwise)
    auto column_block = [&](const float (&z)[MLP_IN], const float (&wq)[NT], size_t col0, float (&dz)[MLP_IN]) {
        v4f_t h[MT][NT], acc[MT][NT];
        unsigned long long mk[MLP_MAX_LEVELS];
        if (want_dec) {
#pragma unroll
            for (int j = 0; j < MLP_IN; ++j) ws.z[(size_t)j * QsT + col0 + lane] = z[j];
            ws.z[(size_t)11 * QsT + col0 + lane] = 0.f;
            sem_hidden_forward<H, true>(lds, f.levels, xb, z, h, mk, ws.h, QsT, col0);
        } else {
            sem_hidden_forward<H, false>(lds, f.levels, xb, z, h, mk, nullptr, QsT, col0);
        }
        v4f_t lg[MO][NT];
        Hd::template forward<NT>(Fo, h, lg);
        Hd::template log_softmax<NT>(lg, st.S);
        // NLL: loss -= wq * logp[label];  d loss / d logit_c = coef * wq * (softmax_c - [c == label])
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float pick = 0.f;
#pragma unroll
            for (int mo = 0; mo < MO; ++mo)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * mo + 4 * g + r;
                    const float lp = lg[mo][nt][r];
                    const bool hit = c == lab[nt];
                    pick += (hit && coef[nt] != 0.f) ? lp : 0.f;
                    const float p = c < st.S ? expf(lp) : 0.f;
                    lg[mo][nt][r] = coef[nt] * wq[nt] * (p - (hit ? 1.f : 0.f));
                }
            pick = rows_sum_lds(pick);
            if (g == 0 && coef[nt] != 0.f) loss -= (double)(wq[nt] * pick);
        }
        if (want_dec) {
#pragma unroll
            for (int mo = 0; mo < MO; ++mo)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ws.d[((size_t)f.levels * H + 16 * mo + 4 * g + r) * QsT + col0 + 16 * nt + n] = lg[mo][nt][r];
        }
        auto put = [&](int l) {
            if (!want_dec) return;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ws.d[((size_t)l * H + 16 * mt + 4 * g + r) * QsT + col0 + 16 * nt + n] = h[mt][nt][r];
        };
        const int L = f.levels;
        Hd::template seed<NT>(Fo, sem_mask_of(mk, L - 1), lg, h);
        put(L - 1);
        for (int l = L - 1; l >= 1; --l) {
            D::template back_hidden<NT>(lds + D::OFF_HID + (l - 1) * D::HID_SZ, sem_mask_of(mk, l - 1), h, acc);
            put(l - 1);
        }
        D::template back_input<NT>(lds, h, xb, 0);
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < MLP_IN; ++j) dz[j] = xb[lane * D::XSTRIDE + j];
        wave_lds_sync();
    };

    const bool mine = active && st.sel[qq] != 0;  // this lane's own query contributes gradients
    if constexpr (WF) {
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
        const float one[NT] = {1.f, 1.f, 1.f, 1.f};
        float dz[MLP_IN];
        column_block(z, one, (size_t)q0, dz);
        // feature-gradient scatter: one atomic instruction per query = its 8 neighbours x 8 feature dims (train_bwd_mfma_kernel)
        for (int half = 0; half < 2; ++half) {
            if ((lane >> 5) == half) {
                const int ql = lane & 31;
#pragma unroll
                for (int j = 0; j < PIN_FEATURE_DIM; ++j) sdz[ql * 8 + j] = dz[j];
#pragma unroll
                for (int t = 0; t < PIN_MAX_K; ++t) {
                    sw[ql * 8 + t] = nb.w[t];
                    sidx[ql * 8 + t] = mine ? nb.idx[t] : -1;
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
            // the neighbour's weight per query TILE: lane (n, g) needs w_t of queries 16 nt + n
            xb[lane] = idx >= 0 ? wt : 0.f;
            wave_lds_sync();
            float wq[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wq[nt] = xb[16 * nt + n];
            wave_lds_sync();
            float dz[MLP_IN];
            column_block(z, wq, (size_t)t * ws.Qs + q0, dz);
            // one neighbour per query here: 8 queries x 8 feature dims per atomic instruction
#pragma unroll
            for (int j = 0; j < PIN_FEATURE_DIM; ++j) sdz[lane * 8 + j] = dz[j];
            sidx[lane] = (mine && idx >= 0) ? idx : -1;
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
    loss = wave_sum(loss);
    if (lane == 0 && loss != 0.0) atomicAdd(loss_out, loss);
}

// ---- inference: labels (argmax) and / or the log-probabilities ----------------------------------------------------------------
// MODE 0: neighbours from the kNN records (Tracker.query_source_points / Mesher.query_points); MODE 1: `query` holds [n][11]
// decoder inputs (Decoder.sem_label_prob on given features; raw = 1: the plain mlp() outputs instead of the log-softmax)
template <int H, bool WF, int MODE>
__global__ __launch_bounds__(MF_BLOCK) void sem_query_kernel(pin_field f, const float* __restrict__ query, const float4* __restrict__ nbr,
                                                             const int* __restrict__ nn_count, int n_q, int S, int raw,
                                                             int* __restrict__ label_out, float* __restrict__ prob_out) {
    using D = MfmaDecoder<H>;
    using Hd = SemHead<H>;
    constexpr int MT = H / 16, NT = 4, MO = Hd::MO;
    __shared__ __attribute__((aligned(16))) float lds[SemLds<H>::TOTAL];
    float* xb = lds + MfmaLds<H>::W + (threadIdx.x >> 6) * D::scratch_floats();
    float* Fo = lds + SemLds<H>::HEAD;
    D::stage(f.dec, f.levels, lds, threadIdx.x, MF_BLOCK, 0);
    Hd::stage(f.dec + sem_head_offset(H, f.levels), S, Fo, threadIdx.x, MF_BLOCK);
    __syncthreads();
    const int q0 = (blockIdx.x * (MF_BLOCK / 64) + (threadIdx.x >> 6)) * 64;
    if (q0 >= n_q) return;  // whole wave
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int qi = q0 + lane;
    const int qq = qi < n_q ? qi : n_q - 1;
    v4f_t pred[MO][NT];
#pragma unroll
    for (int mo = 0; mo < MO; ++mo)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) pred[mo][nt] = (v4f_t){0.f, 0.f, 0.f, 0.f};
    v4f_t h[MT][NT];
    unsigned long long mk[MLP_MAX_LEVELS];
    if constexpr (MODE == 1) {
        float z[MLP_IN];
#pragma unroll
        for (int j = 0; j < MLP_IN; ++j) z[j] = query[(size_t)qq * MLP_IN + j];
        sem_hidden_forward<H, false>(lds, f.levels, xb, z, h, mk, nullptr, 0, 0);
        Hd::template forward<NT>(Fo, h, pred);
        if (!raw) Hd::template log_softmax<NT>(pred, S);
    } else {
        NbrW nb;
        float vx[PIN_MAX_K], vy[PIN_MAX_K], vz[PIN_MAX_K];
        bool quirk[PIN_MAX_K];
        neighbor_weights(nbr, nn_count[qq], qq, f.k, nb, vx, vy, vz, quirk);
        const float qx = query[3 * qq], qy = query[3 * qq + 1], qz = query[3 * qq + 2];
        if constexpr (WF) {
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
            sem_hidden_forward<H, false>(lds, f.levels, xb, z, h, mk, nullptr, 0, 0);
            Hd::template forward<NT>(Fo, h, pred);
            Hd::template log_softmax<NT>(pred, S);
        } else {
#pragma unroll 1
            for (int t = 0; t < f.k; ++t) {  // sem_pred = sum_t w_t log_softmax(mlp([f_t; v_t]))  (tracker.py:337-339)
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
                xb[lane] = idx >= 0 ? wt : 0.f;
                wave_lds_sync();
                float wq[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wq[nt] = xb[16 * nt + n];
                wave_lds_sync();
                sem_hidden_forward<H, false>(lds, f.levels, xb, z, h, mk, nullptr, 0, 0);
                v4f_t lg[MO][NT];
                Hd::template forward<NT>(Fo, h, lg);
                Hd::template log_softmax<NT>(lg, S);
#pragma unroll
                for (int mo = 0; mo < MO; ++mo)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)  // (a padded head stays out of the argmax: 0 * -inf would be NaN)
                            pred[mo][nt][r] = (16 * mo + 4 * g + r < S) ? fmaf(wq[nt], lg[mo][nt][r], pred[mo][nt][r]) : -INFINITY;
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int q = q0 + 16 * nt + n;
        if (prob_out != nullptr && q < n_q) {
#pragma unroll
            for (int mo = 0; mo < MO; ++mo)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * mo + 4 * g + r;
                    if (c < S) prob_out[(size_t)q * S + c] = pred[mo][nt][r];
                }
        }
        if (label_out != nullptr) {  // torch.argmax: the FIRST maximum
            float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
            for (int mo = 0; mo < MO; ++mo)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * mo + 4 * g + r;
                    const float v = pred[mo][nt][r];
                    if (c < S && (v > bv || (v == bv && c < bi))) { bv = v; bi = c; }
                }
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (g == 0 && q < n_q) label_out[q] = bi == 0x7fffffff ? 0 : bi;
        }
    }
}

}  // namespace pin
