// One training iteration of Mapper.mapping (utils/mapper.py:645-818), weighted_first, behind pin_train_step /
// pin_train_color_step:
//
//   train_stage_kernel<H>      the decoder as the tile kernel reads it from LDS (split fp16 pieces, both directions,
//                              MFMA operand order), once per call; every block of the tile kernel copies it linearly
//   train_fused_kernel<H,L,OD> gather + IDW interpolation (neural_points.py:590-746), decoder forward, BCE-with-logits +
//                              Eikonal loss and its gradient (utils/loss.py:45-63, mapper.py:732-780), decoder backward,
//                              feature-gradient scatter, training-mode side effects -- per 16-query tile, four lanes per
//                              query as in gn_quad.h, decoder on the split-fp16 matrix cores (mlp_h2.h).  Nothing of a
//                              tile goes through memory between forward and backward; what leaves the kernel for the
//                              decoder's weight gradient is the OPERAND STREAM of the next launch:
//   train_dw_stream_kernel<H>  dW_l = sum_q delta_{l+1}[q] (x) a_l[q]: a 16x16x16 MFMA per (16 out units, 16 in units,
//                              16 queries of a tile) and piece product, operands read exactly as the instruction wants them
//   train_finalize_kernel      sums the slot copies of the weight gradient into dec_grad and the per-block loss sums
//
// r01-r02a had three launches (forward with activation stores, backward with delta stores, a K = 4 fp32 MFMA GEMM over
// unit-major fp32 rows) of ~30 us each at the reference's batch of 16k samples: 2.1 kB per query written with 64-byte
// granularity, read back twice, the decoder staged twice, the neighbours gathered twice.
//
// Tile -> query map.  The six +-eps probes of an Eikonal sample must sit in one tile (the loss gradient of a probe
// needs the other five predictions): a MIXED tile carries 2 samples x 6 probes in columns 0..11 and four main samples
// in columns 12..15; the remaining main samples fill plain tiles of 16.  n_tiles = ceil(Q / 16) up to rounding: no
// padded columns to decode.
//
// Transposing without data movement.  The quad layout holds a 16-query x 16-unit block as "query in the lane, four
// units in the registers" -- which IS the A operand of v_mfma_f32_16x16x16_f16 (M = query, K = unit).  Multiplying
// it by the identity returns the block in the D layout, "unit in the lane, four queries in the registers": the operand
// layout with K = query that the weight gradient needs.  fp16 pieces pass through the fp32 result exactly.
//
// Scaling.  Loss gradients are ~1 / batch (1e-5): far below the fp16 normal range.  The backward sweep is linear in
// d loss / d prediction, so it runs on gradients multiplied by a power of two (`dscale`, chosen by the host from the
// loss normalisation so that a unit loss gradient maps to ~1); feature gradients and weight gradients are multiplied
// by 1 / dscale on the way out -- exact.
//
// OD = 3: the colour term of the iteration (regress_color + L1 on the surface samples, mapper.py:668-675, 804-812;
// color_diff_loss, utils/loss.py:31-42) through the same kernels: plain tiles of 16 samples over the colour feature
// table, three sigmoid heads, one backward sweep seeded with sum_c dL/dh_c wo_c.
#pragma once
#include "mlp_h2.h"

namespace pin {

struct FusedColor {          // OD = 3 only
    const float* color;      // [n_main][3] measured colours
    const int* count;        // [1] surface samples in the batch (color_count_kernel)
    float surface_range, weight_i;
    int loss_weight_on;
};

constexpr int TF_BLOCK = 512;  // 8 waves per CU, 2 per SIMD (<= 256 VGPRs): the per-neighbour and analytic-Eikonal kernels
// the two shapes of those kernels whose tile state exceeds 256 registers (64-wide per-neighbour decoding with the analytic
// Eikonal term: three tiles' pieces + the derivative network; the analytic term at three or four 64-wide layers) run ONE wave per SIMD
// with the other half of the register file instead of spilling to scratch memory
template <int H, bool AN> constexpr int tf_nwf_block() { return (H == 64 && AN) ? 256 : TF_BLOCK; }
template <int H, int L> constexpr int tf_an_block() { return (H == 64 && L >= 3) ? 256 : TF_BLOCK; }
constexpr int TFW_BLOCK = 768; // train_fused_kernel: 12 waves per CU, 3 per SIMD (<= 168 VGPRs) -- a tile is a latency chain
                               // (~80 k cycles for ~11 k cycles of vector issue): throughput at large batches is waves in flight
// (DW_SLOTS partial weight gradients, train.hip: chunk c of the streamed product adds into slot c % DW_SLOTS)

// operand stream of the weight-gradient launch.  Layer index lam = 0..L: delta_{lam+1} (x) a_lam, with a_0 = z (one
// 16-input block), a_l = post-ReLU activations (MT blocks), delta_{L+1} = d loss / d head (one block, unit 0).
// One block and piece = 64 lanes x 8 bytes (four fp16 of consecutive queries for the lane's unit).
struct DwStream {
    uint2* d;     // [lam][tile][block][piece][lane]
    uint2* a;
    int n_tiles;
};
template <int H>
struct DwGeom {
    static constexpr int MT = H / 16;
    __host__ __device__ static constexpr int d_blocks(int L, int lam) { return lam < L ? MT : 1; }
    __host__ __device__ static constexpr int a_blocks(int lam) { return lam == 0 ? 1 : MT; }
    // offsets in uint2 units (128 per block: 2 pieces x 64 lanes)
    __host__ __device__ static size_t d_off(size_t n_tiles, int lam) { return n_tiles * 128 * (size_t)(lam * MT); }
    __host__ __device__ static size_t a_off(size_t n_tiles, int lam) { return lam == 0 ? 0 : n_tiles * 128 * (size_t)(1 + (lam - 1) * MT); }
    __host__ __device__ static size_t total(size_t n_tiles, int L) { return n_tiles * 128 * (size_t)(L * MT + 1); }  // of each stream
};

// Element `elem` of a stream region: a UNIFORM 64-bit base (scalar registers) plus a 32-bit byte offset (one vector register
// per lane) -- the store then takes its base from the scalar file instead of keeping one 64-bit lane address per region in
// vector registers across the tile loop (10 regions at four layers).  8 * elem < 2^32: n_tiles < 2^20 at H = 64
// (launch_fused_l checks it).
__device__ __forceinline__ uint2* stream_at(uint2* region, unsigned int elem) {
#ifdef PIN_AB_STREAM_64  // (A/B builds only: one 64-bit lane address per region)
    return region + (size_t)elem;
#else
    return reinterpret_cast<uint2*>(reinterpret_cast<char*>(region) + (size_t)(elem * 8u));
#endif
}

__host__ __device__ inline int fused_mixed_tiles(int n_eik) { return (n_eik + 1) >> 1; }
__host__ __device__ inline int fused_tiles(int n_main, int n_eik) {
    const int nm = fused_mixed_tiles(n_eik);
    const int rest = n_main - 4 * nm;
    return nm + (rest > 0 ? (rest + 15) >> 4 : 0);
}

typedef _Float16 v4h_raw __attribute__((ext_vector_type(4)));

// a 16 x 16 block of packed fp16 (quad layout, words w0 w1 = units 4g..4g+3 of the lane's query) -> D layout
__device__ __forceinline__ uint2 transpose_block(unsigned int w0, unsigned int w1, v4h_t ident) {
    const v2u_t a = {w0, w1};
    const v4f_t t = __builtin_amdgcn_mfma_f32_16x16x16f16(as_h4(a), ident, (v4f_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    const v2h_t lo = {(_Float16)t[0], (_Float16)t[1]}, hi = {(_Float16)t[2], (_Float16)t[3]};  // exact: t holds fp16 values
    return make_uint2(__builtin_bit_cast(unsigned int, lo), __builtin_bit_cast(unsigned int, hi));
}

#ifdef PIN_TF_STAMPS
// debug builds (scripts/exp/tf_stamps.py): wall-clock stamps (100 MHz) of the phases of a wave's first tile, one row per block
__device__ unsigned long long g_tf_stamps[1024 * 16];
#ifndef PIN_TF_STAMP_TILE
#define PIN_TF_STAMP_TILE 0  // which of the wave's tiles is stamped (0 = its first)
#endif
#define TF_STAMP(i) do { if (lane == 0 && wave == (PIN_TF_STAMPS) && stamped == (PIN_TF_STAMP_TILE)) g_tf_stamps[(blockIdx.x & 1023) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define TF_STAMP(i) do { } while (0)
#endif
template <int H, int L, int OD = 1>
__global__ __launch_bounds__(TFW_BLOCK, 1) void train_fused_kernel(pin_field f, pin_train_params tp,
                                                                  const float* __restrict__ query,
                                                                  const float4* __restrict__ nbr,
                                                                  const int* __restrict__ nn_count,
                                                                  const float* __restrict__ label,
                                                                  const float* __restrict__ weight,
                                                                  const int* __restrict__ sample_ts,
                                                                  float* __restrict__ cert_rw, int* __restrict__ ts_rw,
                                                                  float* __restrict__ feat_grad, float* __restrict__ pred_out,
                                                                  DwStream ws, int want_dec, float dscale,
                                                                  const unsigned char* __restrict__ dec_image,
                                                                  float* __restrict__ dw_partial, int n_dec,
                                                                  double* __restrict__ loss_partial, FusedColor fcol, int acts_out) {
    // acts_out = 0: only the deltas and the decoder input z leave for the weight gradient; train_dw_recompute_kernel
    // runs the forward pass again from z (large batches: half the operand stream).  acts_out = 2: the same with the HIGH fp16
    // pieces of the deltas only (train.hip, dw_delta_hi_only)
    const bool delta_lo = acts_out != 2;
    using Q = QuadDecoderH<H>;
    using G = DwGeom<H>;
    constexpr int MT = Q::MT, NJ = Q::NJ;
    extern __shared__ __attribute__((aligned(16))) unsigned char tf_smem[];  // decoder image, scatter patches, loss sums
    unsigned char* const lds = tf_smem;
    constexpr int IMG = (Q::bytes(L) + 15) & ~15;
    float* const xch = reinterpret_cast<float*>(lds + IMG) + (threadIdx.x >> 6) * (3 * 16 * 8);  // per wave: dz, w, idx [16][8]
    double (*lred)[2] = reinterpret_cast<double (*)[2]>(lds + IMG + (TFW_BLOCK / 64) * 3 * 16 * 8 * 4);
    const int lane = threadIdx.x & 63, nq = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const int n_main = tp.n_main, n_eik = tp.n_eik;
    const int n_mixed = fused_mixed_tiles(n_eik);
    const int n_tiles = ws.n_tiles;
    const int n_waves = gridDim.x * (TFW_BLOCK / 64);
    // dscale is a power of two (launch_fused_l): its reciprocal by exponent arithmetic, exact, on the scalar unit
    const float inv_dscale = __uint_as_float(0x7f000000u - __float_as_uint(dscale));
    // identity operand of the transposing MFMA: B[k = 4 (lane >> 4) + r][n = lane & 15]
    v4h_t ident;
#pragma unroll
    for (int r = 0; r < 4; ++r) ident[r] = (nq == 4 * g + r) ? (_Float16)1.0f : (_Float16)0.0f;
    float* sdz = xch;
    float* sw = sdz + 16 * 8;
    int* sidx = reinterpret_cast<int*>(sdz + 2 * 16 * 8);
    double acc_bce = 0.0, acc_eik = 0.0;
#ifdef PIN_TF_STAMPS
    int stamped = 0;  // tiles this wave has finished
#endif
#if !defined(PIN_TF_STAMPS) || PIN_TF_STAMP_TILE == 0
    TF_STAMP(0);
#endif
    if (want_dec) {  // the slot partials of the weight-gradient launch start from zero (it runs after this kernel)
        const int n = DW_SLOTS * n_dec;
        for (int i = blockIdx.x * TFW_BLOCK + threadIdx.x; i < n; i += gridDim.x * TFW_BLOCK) dw_partial[i] = 0.f;
    }
    {   // the decoder image, split and permuted once per call by train_stage_kernel: a linear copy per block, in
        // flight behind the first tile's gather loads
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(dec_image);
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(lds);
        constexpr int N16 = Q::bytes(L) >> 4, TRIPS = (N16 + TFW_BLOCK - 1) / TFW_BLOCK;
        uint4 v[TRIPS];  // (all loads of the copy in flight at once)
#pragma unroll
        for (int it = 0; it < TRIPS; ++it) {
            const int i = it * TFW_BLOCK + threadIdx.x;
            v[it] = src[i < N16 ? i : 0];
        }
#pragma unroll
        for (int it = 0; it < TRIPS; ++it) {
            const int i = it * TFW_BLOCK + threadIdx.x;
            if (i < N16) dst[i] = v[it];
        }
    }
    bool staged = false;
    for (int tile = blockIdx.x + gridDim.x * wave;; tile += n_waves) {
        const bool work = tile < n_tiles;
        if (!work && staged) break;
#if defined(PIN_TF_STAMPS) && PIN_TF_STAMP_TILE > 0
        TF_STAMP(0);
#endif
        // ---- which query this column is
        const int tl = work ? tile : 0;
        bool is_probe = false;
        int qi, probe_a = 0;
        bool valid;
        {
            // (opaque copy of the lane index, as at the scatter below: the column arithmetic is redone per tile, its
            // loop-invariant pieces would otherwise be spilled)
            int lane_t = lane;
            asm volatile("" : "+v"(lane_t));
            const int nq_t = lane_t & 15;
            if (tl < n_mixed) {
                if (nq_t < 12) {
                    const int s = 2 * tl + (nq_t >= 6 ? 1 : 0);
                    probe_a = nq_t >= 6 ? nq_t - 6 : nq_t;
                    is_probe = true;
                    valid = s < n_eik;
                    qi = n_main + 6 * s + probe_a;
                } else {
                    qi = 4 * tl + (nq_t - 12);
                    valid = qi < n_main;
                }
            } else {
                qi = 4 * n_mixed + 16 * (tl - n_mixed) + nq_t;
                valid = qi < n_main;
            }
        }
        const bool active = work && valid;
        const int qq = valid ? qi : 0;
        const float qx = query[3 * qq], qy = query[3 * qq + 1], qz = query[3 * qq + 2];
        NbrW nb;
        float vx[PIN_MAX_K], vy[PIN_MAX_K], vz[PIN_MAX_K];
        bool quirk[PIN_MAX_K];
        neighbor_weights(nbr, nn_count[qq], qq, f.k, nb, vx, vy, vz, quirk);
        float4 row[PIN_MAX_K];
#pragma unroll
        for (int t = 0; t < PIN_MAX_K; ++t) {
            const size_t id = nb.idx[t] >= 0 ? (size_t)nb.idx[t] : 0;
            row[t] = reinterpret_cast<const float4*>(f.feats)[id * (PIN_FEATURE_DIM / 4) + (g & 1)];
        }
        TF_STAMP(1);  // records + rows requested
        if (!staged) {
            __syncthreads();
            staged = true;
            if (!work) break;
        }
        TF_STAMP(2);  // image in LDS
        // training-mode side effects (neural_points.py:685-710).  (r05, measured and not kept -- phase stamps of a tile at the
        // reference's batch, scripts/exp/tf_stamps.py: 27 us = 5.2 until the rows are requested + 1.6 image barrier + 8.2 until
        // the decoder input is there + 2.9 forward + 1.8 head and loss + 3.5 backward + 3.5 scatter.  The 8.2 contain this
        // block: a wave that wants its rows waits for everything it has issued, these atomics and the look at the time
        // stamps included.  Moved to the end of the tile the same microseconds are spent waiting there (backward 6.9, scatter
        // 6.8: 26.5 us a tile); without the look (an atomic maximum per neighbour) the interpolation waits for 16 atomics
        // instead of ~8: 30.6 us.  The tile is a chain of memory operations at one wave per SIMD wherever they sit.)
        if (g == 3 && active && !is_probe && cert_rw != nullptr) {
            const int my_ts = sample_ts != nullptr ? sample_ts[qi] : 0;
#pragma unroll
            for (int t = 0; t < PIN_MAX_K; ++t)
                if (nb.idx[t] >= 0) {
                    atomicAdd(cert_rw + nb.idx[t], nb.w[t]);
                    // (ts_update only grows: a row that already shows a later frame needs no atomic -- a stale read
                    // can only be smaller than the truth, in which case the atomic is issued anyway)
                    if (ts_rw != nullptr && sample_ts != nullptr && ts_rw[nb.idx[t]] < my_ts) atomicMax(ts_rw + nb.idx[t], my_ts);
                }
        }
        float z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < PIN_MAX_K; ++t) {
            const bool val = nb.idx[t] >= 0;  // invalid neighbours: weight 0 and a zeroed row add exact zeros
            float y[4] = {0.f, 0.f, 0.f, 0.f};
            if (g < 2) {
                if (val) { y[0] = row[t].x; y[1] = row[t].y; y[2] = row[t].z; y[3] = row[t].w; }
            } else if (g == 2 && val) {
                float v[3];
                neighbor_vector_only(f, nb.idx[t], quirk[t], vx[t], vy[t], vz[t], qx, qy, qz, v);
                y[0] = v[0]; y[1] = v[1]; y[2] = v[2];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = fmaf(nb.w[t], y[r], z[r]);
        }
        // the scatter at the end needs the weights and indices: park them in the wave's LDS patch now (lane g == 2)
        // (the previous tile's scatter has finished reading the patch: wave_lds_sync at the end of the loop body)
        if (g == 2) {
#pragma unroll
            for (int t = 0; t < PIN_MAX_K; ++t) { sw[nq * 8 + t] = nb.w[t]; sidx[nq * 8 + t] = nb.idx[t]; }
        }
        // ---- forward.  The pieces of a layer's input are the B operand of its product AND the A operand of its weight
        // gradient: they go out to the operand stream right here and only the layer's ReLU pattern (16 bits per lane)
        // stays for the backward sweep -- r03a kept all pieces in registers (64 at 4 x 64) and ran 2 waves per SIMD
        const unsigned int tbase = (unsigned int)tile * 128u + (unsigned int)lane;
        const unsigned int tbig = (unsigned int)tile * (128u * MT) + (unsigned int)lane;
        auto stream_acts = [&](const v4u_t (&ph)[NJ], const v4u_t (&pl)[NJ], int l) {  // a_l, l = 1 .. L
            uint2* __restrict__ A = stream_at(ws.a + G::a_off(n_tiles, l), tbig);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                A[mt * 128] = transpose_block(ph[mt >> 1][2 * (mt & 1)], ph[mt >> 1][2 * (mt & 1) + 1], ident);
                A[mt * 128 + 64] = transpose_block(pl[mt >> 1][2 * (mt & 1)], pl[mt >> 1][2 * (mt & 1) + 1], ident);
            }
        };
        auto pattern16 = [](const v4f_t (&h)[MT]) {  // bit 4 mt + r: unit (mt, r) of this lane is active (h = relu(.) >= +0)
            unsigned int m = 0u;
#pragma unroll
            for (int mt = MT - 1; mt >= 0; --mt)
#pragma unroll
                for (int r = 3; r >= 0; --r) m = (m << 1) | ((__float_as_uint(h[mt][r]) + 0x7fffffffu) >> 31);
            return m;
        };
        v2u_t zh, zl;
        Q::split_input(z, zh, zl);
        TF_STAMP(3);  // gather arithmetic done (the rows have arrived)
        if (want_dec && work) {
            uint2* __restrict__ A = stream_at(ws.a + G::a_off(n_tiles, 0), tbase);
            A[0] = transpose_block(zh[0], zh[1], ident);
            A[64] = transpose_block(zl[0], zl[1], ident);
        }
        v4f_t h[MT], acc[MT];
        unsigned int pat[L];  // ReLU patterns of a_1 .. a_L
        Q::layer0(lds, L, zh, zl, acc);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[mt][r] = relu1(acc[mt][r]);
#pragma unroll
        for (int l = 1; l <= L; ++l) {
            v4u_t ph[NJ], pl[NJ];
            pat[l - 1] = pattern16(h);
            const bool out = want_dec && acts_out;
            if (l < L || out) Q::split_acts(h, ph, pl);
            if (out) stream_acts(ph, pl, l);
            if (l < L) {
                Q::load_bias(lds, L, l, acc);
                Q::matmul(lds + Q::off_hidf(L, l), ph, pl, acc);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[mt][r] = relu1(acc[mt][r]);
            }
        }
        TF_STAMP(4);  // forward layers
        const float* __restrict__ O = reinterpret_cast<const float*>(lds + Q::off_out(L));
        float dxc[OD];  // d loss / d head c of this column, times dscale
        if constexpr (OD == 1) {
            float x = 0.f;
#pragma unroll
            for (int kt = 0; kt < MT; ++kt) {
                const v4f_t wo = *reinterpret_cast<const v4f_t*>(O + 16 * kt + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) x = fmaf(wo[r], h[kt][r], x);
            }
            x = rows_sum(x);
            x += O[MF_OD_MAX * H];
            const float pred = f.sdf_scale * x;
            if (active && !is_probe && g == 0 && pred_out != nullptr) pred_out[qi] = pred;
            // ---- loss and d loss / d prediction (train_loss_kernel's arithmetic)
            float dp = 0.f;
            const int c0 = (lane & 48) + (nq >= 6 ? 6 : 0);
            float P[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) P[j] = __shfl(pred, c0 + j, 64);
            if (is_probe) {
                const float two_eps = 2.f * tp.eik_eps;
                const float gx = (P[0] - P[1]) / two_eps, gy = (P[2] - P[3]) / two_eps, gz = (P[4] - P[5]) / two_eps;
                const float n = sqrtf(gx * gx + gy * gy + gz * gz);
                const float r = n - 1.f;
                if (active && probe_a == 0 && g == 0) acc_eik += (double)(r * r);
                const float c = n > 0.f ? tp.weight_e * 2.f * r * tp.inv_n_eik / (n * two_eps) : 0.f;
                const float ga = probe_a < 2 ? gx : (probe_a < 4 ? gy : gz);
                dp = (probe_a & 1) ? -c * ga : c * ga;
            } else {
                const float xl = pred / tp.sigma;
                const float y = 1.f / (1.f + expf(-label[qq] / tp.sigma));
                float l = fmaxf(xl, 0.f) - xl * y + log1pf(expf(-fabsf(xl)));
                float gg = 1.f / (1.f + expf(-xl)) - y;
                if (tp.loss_weight_on) { const float w = fabsf(weight[qq]); l *= w; gg *= w; }
                if (active && g == 0) acc_bce += (double)l;
                dp = gg * tp.inv_n_main / tp.sigma;
            }
            dxc[0] = active ? dp * f.sdf_scale * dscale : 0.f;  // the prediction is sdf_scale * head
        } else {
            // colour heads: p_c = sigmoid(head_c); L1 against the measured colour on the surface samples
            // (train_color_loss_kernel's arithmetic), then through the sigmoid
            const bool on = fabsf(label[qq]) < fcol.surface_range;
            const float wt = fcol.loss_weight_on ? fabsf(weight[qq]) : 1.f;
            const float scale = fcol.weight_i * wt / fmaxf((float)(*fcol.count) * 3.f, 1.f);
#pragma unroll
            for (int c = 0; c < OD; ++c) {
                float o = 0.f;
#pragma unroll
                for (int kt = 0; kt < MT; ++kt) {
                    const v4f_t wo = *reinterpret_cast<const v4f_t*>(O + c * H + 16 * kt + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o = fmaf(wo[r], h[kt][r], o);
                }
                o = rows_sum(o);
                o += O[MF_OD_MAX * H + c];
                const float pc = sigmoidf_(o);
                const float diff = pc - fcol.color[3 * (size_t)qq + c];
                const float dpred = on ? (diff > 0.f ? scale : (diff < 0.f ? -scale : 0.f)) : 0.f;
                if (active && on && g == 0) acc_bce += (double)(wt * fabsf(diff));
                dxc[c] = active ? dpred * pc * (1.f - pc) * dscale : 0.f;
            }
        }
        TF_STAMP(5);  // head + loss
        // ---- backward
        if (want_dec) {  // out layer: delta = d loss / d heads in units 0 .. OD - 1 of a 16-unit block (its input a_L is out already)
            unsigned int dh0, dl0, dh1 = 0u, dl1 = 0u;
            h2_split2((g == 0) ? dxc[0] : 0.f, (g == 0 && OD > 1) ? dxc[OD > 1 ? 1 : 0] : 0.f, dh0, dl0);
            if constexpr (OD > 2) h2_split2((g == 0) ? dxc[2] : 0.f, 0.f, dh1, dl1);
            uint2* __restrict__ D = stream_at(ws.d + G::d_off(n_tiles, L), tbase);
            D[0] = transpose_block(dh0, dh1, ident);
            if (delta_lo) D[64] = transpose_block(dl0, dl1, ident);
        }
        bool any_dx = false;
#pragma unroll
        for (int c = 0; c < OD; ++c) any_dx = any_dx || dxc[c] != 0.f;
#pragma unroll
        for (int kt = 0; kt < MT; ++kt) {
            v4f_t sd = (v4f_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < OD; ++c) {
                const v4f_t wo = *reinterpret_cast<const v4f_t*>(O + c * H + 16 * kt + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) sd[r] = fmaf(dxc[c], wo[r], sd[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) h[kt][r] = ((pat[L - 1] >> (4 * kt + r)) & 1u) ? sd[r] : 0.f;
        }
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            // h = delta_{l+1} (H units); its pieces feed the transposed product and the weight-gradient stream
            v4u_t bh[NJ], bl[NJ];
            Q::split_acts(h, bh, bl);
            if (want_dec) {
                uint2* __restrict__ D = stream_at(ws.d + G::d_off(n_tiles, l), tbig);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    D[mt * 128] = transpose_block(bh[mt >> 1][2 * (mt & 1)], bh[mt >> 1][2 * (mt & 1) + 1], ident);
                    if (delta_lo) D[mt * 128 + 64] = transpose_block(bl[mt >> 1][2 * (mt & 1)], bl[mt >> 1][2 * (mt & 1) + 1], ident);
                }
            }
            if (l > 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4f_t){0.f, 0.f, 0.f, 0.f};
                Q::matmul(lds + Q::off_hidb(L, l), bh, bl, acc);
                // ReLU pattern of a_l (exact: taken from the fp32 activations of the forward pass)
#pragma unroll
                for (int mj = 0; mj < MT; ++mj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[mj][r] = ((pat[l - 1] >> (4 * mj + r)) & 1u) ? acc[mj][r] : 0.f;
            } else {
                float dz[4];
                Q::input_backward(lds, L, bh, bl, dz);
                // ---- feature-gradient scatter, one atomic instruction per QUERY (64 lanes = 8 neighbours x 8 feature
                // dims, whole 32-byte rows per instruction): exchange through the wave's LDS patch
                const bool live = active && any_dx;
                if (g < 2) {
                    float ids = inv_dscale;  // (a local copy, not a vector pair held across the loop body)
                    asm volatile("" : "+v"(ids));
#pragma unroll
                    for (int r = 0; r < 4; ++r) sdz[nq * 8 + 4 * g + r] = dz[r] * ids;
                } else if (g == 3) {
                    if (!live) {
#pragma unroll
                        for (int t = 0; t < PIN_MAX_K; ++t) sidx[nq * 8 + t] = -1;
                    }
                }
            }
        }
        TF_STAMP(6);  // backward sweep
        wave_lds_sync();
        {
            // (the lane index goes through an opaque move: what derives from it here is computed here, per tile, instead of
            // sitting in registers across the whole loop body -- the tile kernel runs at its 168-register limit)
            int lane_s = lane;
            asm volatile("" : "+v"(lane_s));
            const int t = lane_s >> 3, j = lane_s & 7;
            for (int i = 0; i < 16; ++i) {
                const int idx = sidx[i * 8 + t];
                if (idx >= 0) atomicAdd(feat_grad + (size_t)idx * PIN_FEATURE_DIM + j, sw[i * 8 + t] * sdz[i * 8 + j]);
            }
        }
        wave_lds_sync();
        TF_STAMP(7);  // scatter issued
#ifdef PIN_TF_STAMPS
        ++stamped;
#endif
    }
    // loss values: one pair per block, summed by train_finalize_kernel (no atomics, no clearing launch)
    acc_bce = wave_sum(acc_bce);
    acc_eik = wave_sum(acc_eik);
    if (lane == 0) { lred[wave][0] = acc_bce; lred[wave][1] = acc_eik; }
    __syncthreads();
    if (threadIdx.x < 2) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < TFW_BLOCK / 64; ++w) t += lred[w][threadIdx.x];
        loss_partial[2 * blockIdx.x + threadIdx.x] = t;
    }
}

// ---- per-neighbour decoding (weighted_first = False; run_kitti.yaml and eight more shipped configs), one H-wide layer
// The decoder runs once per NEIGHBOUR and the predictions are weighted afterwards (mapper.py:658-662).  As in
// gn_accumulate_quad_nwf_kernel a decoder column is a (query, neighbour) pair: a tile = 2 queries x 8 neighbour slots,
// everything that mixes the neighbours of a query is a sum over 8 consecutive lanes of a DPP row.  The six probes of an
// Eikonal sample are THREE tiles (tile j = the +- pair of axis j): a wave works on groups of three tiles -- forward x 3
// (a one-layer decoder leaves 20 piece words per tile to keep), loss, backward x 3 -- and main samples go through the
// same code six at a time.  Feature gradients: a column owns one row, 8 columns x 8 dims per atomic instruction.
//
// AN: the Eikonal term on the ANALYTIC gradient of every main sample (numerical_grad_on False: run_livox.yaml:27;
// mapper.py:642-643, 677-678, 760-782 -- autograd differentiates the loss through get_gradient a second time).  No
// probes.  The forward pass also takes g = d pred / d q (the Jacobian of gn_accumulate_quad_nwf_kernel: through the
// neighbour vectors and through the IDW weights).  With c = d loss / d g, the scalar c . g is
//     sum_t  s x_t (c . d w_t / d q)  +  s w_t (W0^T (pattern_t .* wo)) . chat_t,      chat_t = d (input of column t) / d q  c,
// linear in the predictions x_t -- so the first term only adds (c . d w_t / d q) to the upstream of column t -- and in
// the DERIVATIVE network of column t along chat_t (same weights, no biases, the ReLU pattern of the forward pass as a
// fixed mask; the pattern itself contributes nothing almost everywhere).  Its weight gradient is a second operand
// stream (ws2) of the same geometry: d W0 += (s w_t pattern_t .* wo) (x) chat_t, d wo += s w_t (pattern_t .* W0 chat_t);
// no bias and no feature gradient comes from it.
__device__ __forceinline__ void neighbor_vector_rot(const pin_field& f, int idx, bool quirk, float vgx, float vgy, float vgz,
                                                    float qx, float qy, float qz, float (&v)[3], float (&Rm)[9]) {
    v[0] = vgx; v[1] = vgy; v[2] = vgz;
    if (quirk) {  // the reference gathers local point #1 for non-local neighbours
        const float* p = f.pos + 3 * (size_t)idx;
        v[0] = qx - p[0]; v[1] = qy - p[1]; v[2] = qz - p[2];
    }
    Rm[0] = 1.f; Rm[1] = 0.f; Rm[2] = 0.f; Rm[3] = 0.f; Rm[4] = 1.f; Rm[5] = 0.f; Rm[6] = 0.f; Rm[7] = 0.f; Rm[8] = 1.f;
    if (f.orient != nullptr) {  // apply_quaternion_rotation (utils/tools.py:428-437): v_i = sum_j Rm[3 i + j] x_j
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

template <int H, bool AN>
__global__ __launch_bounds__((tf_nwf_block<H, AN>()), 1) void train_fused_nwf_kernel(pin_field f, pin_train_params tp,
                                                                      const float* __restrict__ query,
                                                                      const float4* __restrict__ nbr,
                                                                      const int* __restrict__ nn_count,
                                                                      const float* __restrict__ label,
                                                                      const float* __restrict__ weight,
                                                                      const int* __restrict__ sample_ts,
                                                                      float* __restrict__ cert_rw, int* __restrict__ ts_rw,
                                                                      float* __restrict__ feat_grad, float* __restrict__ pred_out,
                                                                      DwStream ws, DwStream ws2, int want_dec, float dscale,
                                                                      const unsigned char* __restrict__ dec_image,
                                                                      float* __restrict__ dw_partial, int n_dec,
                                                                      double* __restrict__ loss_partial) {
    constexpr int BLK = tf_nwf_block<H, AN>();  // (the LDS patches stay laid out for TF_BLOCK / 64 waves)
    using Q = QuadDecoderH<H>;
    using G = DwGeom<H>;
    constexpr int MT = Q::MT, NJ = Q::NJ, L = 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char tf_smem[];
    unsigned char* const lds = tf_smem;
    constexpr int IMG = (Q::bytes(L) + 15) & ~15;
    float* const sdz = reinterpret_cast<float*>(lds + IMG) + (threadIdx.x >> 6) * (16 * 8 + 16);  // per wave: dz [16][8], idx [16]
    int* const sidx = reinterpret_cast<int*>(sdz + 16 * 8);
    double (*lred)[2] = reinterpret_cast<double (*)[2]>(lds + IMG + (TF_BLOCK / 64) * (16 * 8 + 16) * 4);
    const int lane = threadIdx.x & 63, nq = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const int q2 = nq >> 3, t = nq & 7;
    const int n_main = tp.n_main, n_eik = tp.n_eik, kk = f.k;
    const int n_main_groups = (n_main + 5) / 6, n_groups = n_main_groups + n_eik;
    const int n_tiles = ws.n_tiles;  // 3 * n_groups
    const int n_waves = gridDim.x * (BLK / 64);
    const float inv_dscale = 1.0f / dscale, s = f.sdf_scale;
    v4h_t ident;
#pragma unroll
    for (int r = 0; r < 4; ++r) ident[r] = (nq == 4 * g + r) ? (_Float16)1.0f : (_Float16)0.0f;
    double acc_bce = 0.0, acc_eik = 0.0;
    if (want_dec) {
        const int n = DW_SLOTS * n_dec;
        for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) dw_partial[i] = 0.f;
    }
    {
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(dec_image);
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(lds);
        constexpr int n16 = Q::bytes(L) >> 4;
        for (int i = threadIdx.x; i < n16; i += BLK) dst[i] = src[i];
    }
    __syncthreads();
    const float* __restrict__ O = reinterpret_cast<const float*>(lds + Q::off_out(L));
    v4f_t wo[MT];
#pragma unroll
    for (int kt = 0; kt < MT; ++kt) wo[kt] = *reinterpret_cast<const v4f_t*>(O + 16 * kt + 4 * g);
    const float bo = O[MF_OD_MAX * H];
    const float4* __restrict__ rows = reinterpret_cast<const float4*>(f.feats) + (g & 1);
    for (int grp = blockIdx.x + gridDim.x * wave; grp < n_groups; grp += n_waves) {
        const bool eik = grp >= n_main_groups;
        // ---- forward of the three tiles
        v2u_t zh[3], zl[3];
        v4u_t ph[3][NJ], pl[3][NJ];
        float wt[3], pred[3];
        int id[3], qidx[3];
        bool act[3], valn[3];
        // (AN) per column: q - P_t, d u_t / d q = cg (q - P_t); per query: 1 / S, sum_t cg (q - P_t), d pred / d q
        float ex[3], ey[3], ez[3], cgn[3], invS[3], Gx[3], Gy[3], Gz[3], gx[3], gy[3], gz[3];
        bool rot[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int a = 2 * j + q2;  // query of the group: main sample 6 grp + a / probe a of Eikonal sample grp - n_main_groups
            const int qi = eik ? n_main + 6 * (grp - n_main_groups) + a : 6 * grp + a;
            const bool valid = eik || qi < n_main;
            const int qq = valid ? qi : 0;
            qidx[j] = qq; act[j] = valid;
            const float4 e = nbr[(size_t)qq * kk + (t < kk ? t : 0)];
            const int nn = nn_count[qq];
            const int raw = __float_as_int(e.w);
            const bool val = t < kk && raw >= 0;
            const int idn = val ? (raw & ~PIN_NBR_QUIRK_BIT) : 0;
            const float4 ft = rows[2 * (size_t)(unsigned int)idn];
            float ut = val ? 1.0f / (dist2_exact(e.x, e.y, e.z) + IDW_EPS) : 0.f;  // (neighbor_weights' arithmetic)
            if (nn == 0 && t < kk) ut = IDW_EPS;
            const float S = octet_sum(ut);
            const float w = val ? ut / S : 0.f;
            float v[3] = {e.x, e.y, e.z};
            float Rm[9];
            const bool quirk = val && (raw & PIN_NBR_QUIRK_BIT) != 0;
            const bool special = f.orient != nullptr || __builtin_amdgcn_ballot_w64(quirk) != 0ull;  // after PGO / a flagged neighbour (rare)
            if (special) {
                const float qx = query[3 * qq], qy = query[3 * qq + 1], qz = query[3 * qq + 2];
                if constexpr (AN) {
                    if (val) neighbor_vector_rot(f, idn, quirk, e.x, e.y, e.z, qx, qy, qz, v, Rm);
                } else {
                    if (val) neighbor_vector_only(f, idn, quirk, e.x, e.y, e.z, qx, qy, qz, v);
                }
            }
            float z[4];
            z[0] = !val ? 0.f : (g < 2 ? ft.x : (g == 2 ? v[0] : 0.f));
            z[1] = !val ? 0.f : (g < 2 ? ft.y : (g == 2 ? v[1] : 0.f));
            z[2] = !val ? 0.f : (g < 2 ? ft.z : (g == 2 ? v[2] : 0.f));
            z[3] = !val ? 0.f : (g < 2 ? ft.w : 0.f);
            Q::split_input(z, zh[j], zl[j]);
            v4f_t h[MT], acc[MT];
            Q::layer0(lds, L, zh[j], zl[j], acc);
            float x = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { h[mt][r] = relu1(acc[mt][r]); x = fmaf(wo[mt][r], h[mt][r], x); }
            x = rows_sum(x);
            x += bo;
            Q::split_acts(h, ph[j], pl[j]);
            wt[j] = w; id[j] = idn; valn[j] = val;
            pred[j] = octet_sum(w * (s * x));  // mapper.py:658-662
            if constexpr (AN) {  // d pred / d q (tools.py:247-260 through Decoder.mlp, the neighbour vectors and the weights)
                v4f_t hb[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) hb[mt][r] = h[mt][r] > 0.f ? wo[mt][r] : 0.f;
                float a4[4];
                Q::input_backward(lds, L, hb, a4);  // d x_t / d (input of column t); the position part sits in the g == 2 lanes
                float d0 = 0.f, d1 = 0.f, d2 = 0.f;
                if (g == 2 && val) {
                    if (special) {
                        d0 = w * (Rm[0] * a4[0] + Rm[3] * a4[1] + Rm[6] * a4[2]);
                        d1 = w * (Rm[1] * a4[0] + Rm[4] * a4[1] + Rm[7] * a4[2]);
                        d2 = w * (Rm[2] * a4[0] + Rm[5] * a4[1] + Rm[8] * a4[2]);
                    } else { d0 = w * a4[0]; d1 = w * a4[1]; d2 = w * a4[2]; }
                }
                d0 = quad_lanes_sum(octet_sum(d0)); d1 = quad_lanes_sum(octet_sum(d1)); d2 = quad_lanes_sum(octet_sum(d2));
                const float uv = val ? ut : 0.f;
                const float cg = -2.f * uv * uv, cs = cg * (s * x), iS = 1.0f / S;
                const float ax = octet_sum(cs * e.x), ay = octet_sum(cs * e.y), az = octet_sum(cs * e.z);
                Gx[j] = octet_sum(cg * e.x); Gy[j] = octet_sum(cg * e.y); Gz[j] = octet_sum(cg * e.z);
                gx[j] = s * d0 + (ax - pred[j] * Gx[j]) * iS;
                gy[j] = s * d1 + (ay - pred[j] * Gy[j]) * iS;
                gz[j] = s * d2 + (az - pred[j] * Gz[j]) * iS;
                ex[j] = e.x; ey[j] = e.y; ez[j] = e.z; cgn[j] = cg; invS[j] = iS; rot[j] = special && val;
            }
            if (!eik && valid && t == 0 && g == 0 && pred_out != nullptr) pred_out[qi] = pred[j];
            // training-mode side effects (neural_points.py:685-710), main samples only
            if (!eik && valid && val && g == 3 && cert_rw != nullptr) {
                atomicAdd(cert_rw + idn, w);
                if (ts_rw != nullptr && sample_ts != nullptr) {
                    const int my_ts = sample_ts[qi];
                    if (ts_rw[idn] < my_ts) atomicMax(ts_rw + idn, my_ts);
                }
            }
        }
        // ---- loss and d loss / d prediction of this lane's query in each tile (train_loss_kernel's arithmetic)
        float dp[3];
        if (eik) {
            float ga[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {  // tile j holds the + (q2 = 0) and - (q2 = 1) probe of axis j
                const float other = __shfl_xor(pred[j], 8, 64);
                const float pp = q2 == 0 ? pred[j] : other, pm = q2 == 0 ? other : pred[j];
                ga[j] = (pp - pm) / (2.f * tp.eik_eps);
            }
            const float n = sqrtf(ga[0] * ga[0] + ga[1] * ga[1] + ga[2] * ga[2]);
            const float r = n - 1.f;
            if (lane == 0) acc_eik += (double)(r * r);
            const float c = n > 0.f ? tp.weight_e * 2.f * r * tp.inv_n_eik / (n * 2.f * tp.eik_eps) : 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) dp[j] = q2 == 0 ? c * ga[j] : -c * ga[j];
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int qq = qidx[j];
                const float xl = pred[j] / tp.sigma;
                const float y = 1.f / (1.f + expf(-label[qq] / tp.sigma));
                float l = fmaxf(xl, 0.f) - xl * y + log1pf(expf(-fabsf(xl)));
                float gg = 1.f / (1.f + expf(-xl)) - y;
                if (tp.loss_weight_on) { const float w = fabsf(weight[qq]); l *= w; gg *= w; }
                if (act[j] && t == 0 && g == 0) acc_bce += (double)l;
                dp[j] = act[j] ? gg * tp.inv_n_main / tp.sigma : 0.f;
            }
        }
        float cx[3], cy[3], cz[3];  // (AN) d loss / d g of this lane's query
        if constexpr (AN) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float n = sqrtf(gx[j] * gx[j] + gy[j] * gy[j] + gz[j] * gz[j]);
                const float r = n - 1.f;  // mapper.py:778-781, every sample (gradient_decimation = 1, config.py:438-439)
                if (act[j] && t == 0 && g == 0) acc_eik += (double)(r * r);
                const float cf = (act[j] && n > 0.f) ? tp.weight_e * 2.f * r * tp.inv_n_eik / n : 0.f;
                cx[j] = cf * gx[j]; cy[j] = cf * gy[j]; cz[j] = cf * gz[j];
            }
        }
        // ---- backward of the three tiles
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned int tile = 3u * (unsigned int)grp + (unsigned int)j;
            const unsigned int tbase = tile * 128u + (unsigned int)lane, tbig = tile * (128u * MT) + (unsigned int)lane;
            float up = dp[j] * wt[j];  // pred = sum_t w_t * sdf_scale * head_t
            if constexpr (AN)  // c . d w_t / d q,  d w_t / d q = (d u_t / d q - w_t sum_t' d u_t' / d q) / S
                up += (cgn[j] * (cx[j] * ex[j] + cy[j] * ey[j] + cz[j] * ez[j]) - wt[j] * (cx[j] * Gx[j] + cy[j] * Gy[j] + cz[j] * Gz[j])) * invS[j];
            const float dx = (act[j] && valn[j]) ? up * s * dscale : 0.f;
            v4f_t h[MT];
#pragma unroll
            for (int mj = 0; mj < MT; ++mj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned int word = ph[j][mj >> 1][2 * (mj & 1) + (r >> 1)] | pl[j][mj >> 1][2 * (mj & 1) + (r >> 1)];
                    const bool on = (r & 1) ? ((word & 0x7fff0000u) != 0u) : ((word & 0x7fffu) != 0u);
                    h[mj][r] = on ? dx * wo[mj][r] : 0.f;
                }
            v4u_t bh[NJ], bl[NJ];
            Q::split_acts(h, bh, bl);
            if (want_dec) {
                unsigned int dh0, dl0;
                h2_split2((g == 0) ? dx : 0.f, 0.f, dh0, dl0);
                uint2* __restrict__ D1 = stream_at(ws.d + G::d_off(n_tiles, 1), tbase);
                D1[0] = transpose_block(dh0, 0u, ident);
                D1[64] = transpose_block(dl0, 0u, ident);
                uint2* __restrict__ A1 = stream_at(ws.a + G::a_off(n_tiles, 1), tbig);
                uint2* __restrict__ D0 = stream_at(ws.d + G::d_off(n_tiles, 0), tbig);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    A1[mt * 128] = transpose_block(ph[j][mt >> 1][2 * (mt & 1)], ph[j][mt >> 1][2 * (mt & 1) + 1], ident);
                    A1[mt * 128 + 64] = transpose_block(pl[j][mt >> 1][2 * (mt & 1)], pl[j][mt >> 1][2 * (mt & 1) + 1], ident);
                    D0[mt * 128] = transpose_block(bh[mt >> 1][2 * (mt & 1)], bh[mt >> 1][2 * (mt & 1) + 1], ident);
                    D0[mt * 128 + 64] = transpose_block(bl[mt >> 1][2 * (mt & 1)], bl[mt >> 1][2 * (mt & 1) + 1], ident);
                }
                uint2* __restrict__ A0 = stream_at(ws.a + G::a_off(n_tiles, 0), tbase);
                A0[0] = transpose_block(zh[j][0], zh[j][1], ident);
                A0[64] = transpose_block(zl[j][0], zl[j][1], ident);
                if constexpr (AN) {  // the derivative network along chat_t (x dscale), upstream tau_t = s w_t
                    float ch[4] = {0.f, 0.f, 0.f, 0.f};
                    const bool on_col = act[j] && valn[j];
                    if (g == 2 && on_col) {
                        float c0 = cx[j], c1 = cy[j], c2 = cz[j];
                        if (rot[j]) {
                            const int qq = qidx[j];
                            float v[3], Rm[9];
                            neighbor_vector_rot(f, id[j], false, ex[j], ey[j], ez[j], query[3 * qq], query[3 * qq + 1], query[3 * qq + 2], v, Rm);
                            c0 = Rm[0] * cx[j] + Rm[1] * cy[j] + Rm[2] * cz[j];
                            c1 = Rm[3] * cx[j] + Rm[4] * cy[j] + Rm[5] * cz[j];
                            c2 = Rm[6] * cx[j] + Rm[7] * cy[j] + Rm[8] * cz[j];
                        }
                        ch[0] = c0 * dscale; ch[1] = c1 * dscale; ch[2] = c2 * dscale;
                    }
                    v2u_t th, tl;
                    Q::split_input(ch, th, tl);
                    v4f_t tacc[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) tacc[mt] = (v4f_t){0.f, 0.f, 0.f, 0.f};
                    Q::layer0_add(lds, L, th, tl, tacc);
                    const float tau = on_col ? s * wt[j] : 0.f;
                    v4f_t tm[MT], tb[MT];
#pragma unroll
                    for (int mj = 0; mj < MT; ++mj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const unsigned int word = ph[j][mj >> 1][2 * (mj & 1) + (r >> 1)] | pl[j][mj >> 1][2 * (mj & 1) + (r >> 1)];
                            const bool on = (r & 1) ? ((word & 0x7fff0000u) != 0u) : ((word & 0x7fffu) != 0u);
                            tm[mj][r] = on ? tacc[mj][r] : 0.f;
                            tb[mj][r] = on ? tau * wo[mj][r] : 0.f;
                        }
                    v4u_t mh[NJ], ml[NJ], th2[NJ], tl2[NJ];
                    Q::split_acts(tm, mh, ml);
                    Q::split_acts(tb, th2, tl2);
                    unsigned int eh0, el0;
                    h2_split2((g == 0) ? tau : 0.f, 0.f, eh0, el0);
                    uint2* __restrict__ E1 = stream_at(ws2.d + G::d_off(n_tiles, 1), tbase);
                    E1[0] = transpose_block(eh0, 0u, ident);
                    E1[64] = transpose_block(el0, 0u, ident);
                    uint2* __restrict__ B1 = stream_at(ws2.a + G::a_off(n_tiles, 1), tbig);
                    uint2* __restrict__ E0 = stream_at(ws2.d + G::d_off(n_tiles, 0), tbig);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        B1[mt * 128] = transpose_block(mh[mt >> 1][2 * (mt & 1)], mh[mt >> 1][2 * (mt & 1) + 1], ident);
                        B1[mt * 128 + 64] = transpose_block(ml[mt >> 1][2 * (mt & 1)], ml[mt >> 1][2 * (mt & 1) + 1], ident);
                        E0[mt * 128] = transpose_block(th2[mt >> 1][2 * (mt & 1)], th2[mt >> 1][2 * (mt & 1) + 1], ident);
                        E0[mt * 128 + 64] = transpose_block(tl2[mt >> 1][2 * (mt & 1)], tl2[mt >> 1][2 * (mt & 1) + 1], ident);
                    }
                    uint2* __restrict__ B0 = stream_at(ws2.a + G::a_off(n_tiles, 0), tbase);
                    B0[0] = transpose_block(th[0], th[1], ident);
                    B0[64] = transpose_block(tl[0], tl[1], ident);
                }
            }
            float dz[4];
            Q::input_backward(lds, L, bh, bl, dz);
            // feature-gradient scatter: this column's row gets d loss / d (its feature row) = dz[0..7]
            if (g < 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sdz[nq * 8 + 4 * g + r] = dz[r] * inv_dscale;
            } else if (g == 2) {
                sidx[nq] = dx != 0.f ? id[j] : -1;
            }
            wave_lds_sync();
            {
                const int cc = lane >> 3, jd = lane & 7;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int col = half * 8 + cc;
                    const int idx = sidx[col];
                    if (idx >= 0) atomicAdd(feat_grad + (size_t)idx * PIN_FEATURE_DIM + jd, sdz[col * 8 + jd]);
                }
            }
            wave_lds_sync();
        }
    }
    acc_bce = wave_sum(acc_bce);
    acc_eik = wave_sum(acc_eik);
    if (lane == 0) { lred[wave][0] = acc_bce; lred[wave][1] = acc_eik; }
    __syncthreads();
    if (threadIdx.x < 2) {
        double tt = 0.0;
#pragma unroll
        for (int w = 0; w < BLK / 64; ++w) tt += lred[w][threadIdx.x];
        loss_partial[2 * blockIdx.x + threadIdx.x] = tt;
    }
}

template <int H>
constexpr int train_fused_nwf_lds_bytes() {
    return ((QuadDecoderH<H>::bytes(1) + 15) & ~15) + (TF_BLOCK / 64) * (16 * 8 + 16) * 4 + (TF_BLOCK / 64) * 2 * 8;
}

// ---- weighted_first with the analytic Eikonal term (numerical_grad_on False with weighted_first True; any depth) ----
// mapper.py:642-643, 677-678, 760-782 with the decoder applied to the interpolated input z = sum_t w_t [f_t; v_t].
// Per sample, with a = d head / d z (the UNIT backward sweep through the decoder):
//     g = d pred / d q = s (M^T a_pos + (sum_t (a . y_t) d u_t/dq  -  (a . z) G) / S),   d u_t / d q = -2 u_t^2 (q - P_t),
// G = sum_t d u_t / d q, M = sum_t w_t R_t (gn_quad.h's Jacobian).  With c = d loss / d g the scalar c . g equals
// s a . zdot, zdot = (d z / d q) c -- linear in the derivative network along zdot (same weights, no biases, the forward
// ReLU patterns as fixed masks):  c . g = s wo . t_L,  t_1 = D_1 W_0 zdot,  t_{l+1} = D_{l+1} W_l t_l.  So
//     d (c . g) / d W_l = s deltau_{l+1} (x) t_l   (deltau = the unit sweep's deltas, which the BCE sweep also is,
//                                                   times d loss / d head: ONE backward sweep serves both terms),
//     d (c . g) / d f_t = s a_feat (c . d w_t / d q),
// a second operand stream (ws2: D = deltau pieces, A = t_l pieces, no bias gradient) next to the BCE term's.
template <int H, int L>
__global__ __launch_bounds__((tf_an_block<H, L>()), 1) void train_fused_an_kernel(pin_field f, pin_train_params tp,
                                                                     const float* __restrict__ query,
                                                                     const float4* __restrict__ nbr,
                                                                     const int* __restrict__ nn_count,
                                                                     const float* __restrict__ label,
                                                                     const float* __restrict__ weight,
                                                                     const int* __restrict__ sample_ts,
                                                                     float* __restrict__ cert_rw, int* __restrict__ ts_rw,
                                                                     float* __restrict__ feat_grad, float* __restrict__ pred_out,
                                                                     DwStream ws, DwStream ws2, int want_dec, float dscale,
                                                                     const unsigned char* __restrict__ dec_image,
                                                                     float* __restrict__ dw_partial, int n_dec,
                                                                     double* __restrict__ loss_partial) {
    constexpr int BLK = tf_an_block<H, L>();  // (the LDS patches stay laid out for TF_BLOCK / 64 waves)
    using Q = QuadDecoderH<H>;
    using G = DwGeom<H>;
    constexpr int MT = Q::MT, NJ = Q::NJ;
    constexpr int PATCH = 6 * 16 * 8;  // per wave: dz, w, idx [16][8], d u_t / d q [16][8][3]
    extern __shared__ __attribute__((aligned(16))) unsigned char tf_smem[];
    unsigned char* const lds = tf_smem;
    constexpr int IMG = (Q::bytes(L) + 15) & ~15;
    float* const sdz = reinterpret_cast<float*>(lds + IMG) + (threadIdx.x >> 6) * PATCH;
    float* const sw = sdz + 16 * 8;
    int* const sidx = reinterpret_cast<int*>(sdz + 2 * 16 * 8);
    float* const sgu = sdz + 3 * 16 * 8;
    double (*lred)[2] = reinterpret_cast<double (*)[2]>(lds + IMG + (TF_BLOCK / 64) * PATCH * 4);
    const int lane = threadIdx.x & 63, nq = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const int n_main = tp.n_main;
    const int n_tiles = ws.n_tiles;
    const int n_waves = gridDim.x * (BLK / 64);
    const float inv_dscale = 1.0f / dscale, s = f.sdf_scale;
    const bool orient = f.orient != nullptr;
    v4h_t ident;
#pragma unroll
    for (int r = 0; r < 4; ++r) ident[r] = (nq == 4 * g + r) ? (_Float16)1.0f : (_Float16)0.0f;
    double acc_bce = 0.0, acc_eik = 0.0;
    if (want_dec) {
        const int n = DW_SLOTS * n_dec;
        for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) dw_partial[i] = 0.f;
    }
    {
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(dec_image);
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(lds);
        constexpr int n16 = Q::bytes(L) >> 4;
        for (int i = threadIdx.x; i < n16; i += BLK) dst[i] = src[i];
    }
    __syncthreads();
    const float* __restrict__ O = reinterpret_cast<const float*>(lds + Q::off_out(L));
    // ReLU pattern of a layer out of the pieces of its activations (train_fused_kernel's backward sweep)
    auto piece_on = [](const v4u_t (&ph)[NJ], const v4u_t (&pl)[NJ], int mj, int r) {
        const unsigned int word = ph[mj >> 1][2 * (mj & 1) + (r >> 1)] | pl[mj >> 1][2 * (mj & 1) + (r >> 1)];
        return (r & 1) ? ((word & 0x7fff0000u) != 0u) : ((word & 0x7fffu) != 0u);
    };
    for (int tile = blockIdx.x + gridDim.x * wave; tile < n_tiles; tile += n_waves) {
        const int qi = 16 * tile + nq;
        const bool active = qi < n_main;
        const int qq = active ? qi : 0;
        const float qx = query[3 * qq], qy = query[3 * qq + 1], qz = query[3 * qq + 2];
        NbrW nb;
        float vx[PIN_MAX_K], vy[PIN_MAX_K], vz[PIN_MAX_K], u[PIN_MAX_K + 1];
        bool quirk[PIN_MAX_K];
        neighbor_weights(nbr, nn_count[qq], qq, f.k, nb, vx, vy, vz, quirk, u);
        const float invS = 1.0f / u[PIN_MAX_K];
        float4 row[PIN_MAX_K];
#pragma unroll
        for (int t = 0; t < PIN_MAX_K; ++t) {
            const size_t id = nb.idx[t] >= 0 ? (size_t)nb.idx[t] : 0;
            row[t] = reinterpret_cast<const float4*>(f.feats)[id * (PIN_FEATURE_DIM / 4) + (g & 1)];
        }
        if (g == 3 && active && cert_rw != nullptr) {  // training-mode side effects (neural_points.py:685-710)
            const int my_ts = sample_ts != nullptr ? sample_ts[qi] : 0;
#pragma unroll
            for (int t = 0; t < PIN_MAX_K; ++t)
                if (nb.idx[t] >= 0) {
                    atomicAdd(cert_rw + nb.idx[t], nb.w[t]);
                    if (ts_rw != nullptr && sample_ts != nullptr && ts_rw[nb.idx[t]] < my_ts) atomicMax(ts_rw + nb.idx[t], my_ts);
                }
        }
        // ---- gather: z, and what the derivative of z with respect to q needs (gn_quad.h's QuadIn)
        float z[4] = {0.f, 0.f, 0.f, 0.f}, Y[3][4], M[9];
        float Gx = 0.f, Gy = 0.f, Gz = 0.f, wsum = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) Y[c][r] = 0.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) M[i] = 0.f;
#pragma unroll
        for (int t = 0; t < PIN_MAX_K; ++t) {
            const bool val = nb.idx[t] >= 0;
            const float cg = -2.f * u[t] * u[t];
            const float g0 = cg * vx[t], g1 = cg * vy[t], g2 = cg * vz[t];  // d u_t / d q (0 for an invalid neighbour)
            Gx += g0; Gy += g1; Gz += g2; wsum += nb.w[t];
            float y[4] = {0.f, 0.f, 0.f, 0.f};
            if (g < 2) {
                if (val) { y[0] = row[t].x; y[1] = row[t].y; y[2] = row[t].z; y[3] = row[t].w; }
            } else if (g == 2 && val) {
                float v[3] = {vx[t], vy[t], vz[t]};
                if (orient || quirk[t]) {
                    float Rm[9];
                    neighbor_vector_rot(f, nb.idx[t], quirk[t], vx[t], vy[t], vz[t], qx, qy, qz, v, Rm);
#pragma unroll
                    for (int i = 0; i < 9; ++i) M[i] = fmaf(nb.w[t], Rm[i], M[i]);
                } else {
                    M[0] += nb.w[t]; M[4] += nb.w[t]; M[8] += nb.w[t];
                }
                y[0] = v[0]; y[1] = v[1]; y[2] = v[2];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                z[r] = fmaf(nb.w[t], y[r], z[r]);
                Y[0][r] = fmaf(g0, y[r], Y[0][r]);
                Y[1][r] = fmaf(g1, y[r], Y[1][r]);
                Y[2][r] = fmaf(g2, y[r], Y[2][r]);
            }
            if (g == 2) {
                sw[nq * 8 + t] = nb.w[t]; sidx[nq * 8 + t] = active ? nb.idx[t] : -1;
                sgu[(nq * 8 + t) * 3] = g0; sgu[(nq * 8 + t) * 3 + 1] = g1; sgu[(nq * 8 + t) * 3 + 2] = g2;
            }
        }
        (void)wsum;
        // ---- forward
        v2u_t zh, zl;
        Q::split_input(z, zh, zl);
        v4f_t h[MT], acc[MT];
        v4u_t ph[L][NJ], pl[L][NJ];
        Q::layer0(lds, L, zh, zl, acc);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[mt][r] = relu1(acc[mt][r]);
#pragma unroll
        for (int l = 1; l < L; ++l) {
            Q::split_acts(h, ph[l - 1], pl[l - 1]);
            Q::load_bias(lds, L, l, acc);
            Q::matmul(lds + Q::off_hidf(L, l), ph[l - 1], pl[l - 1], acc);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) h[mt][r] = relu1(acc[mt][r]);
        }
        Q::split_acts(h, ph[L - 1], pl[L - 1]);
        float x = 0.f;
#pragma unroll
        for (int kt = 0; kt < MT; ++kt) {
            const v4f_t wo = *reinterpret_cast<const v4f_t*>(O + 16 * kt + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) x = fmaf(wo[r], h[kt][r], x);
        }
        x = rows_sum(x);
        x += O[MF_OD_MAX * H];
        const float pred = s * x;
        if (active && g == 0 && pred_out != nullptr) pred_out[qi] = pred;
        float dxm;  // d BCE / d head, times dscale (train_loss_kernel's arithmetic)
        {
            const float xl = pred / tp.sigma;
            const float yl = 1.f / (1.f + expf(-label[qq] / tp.sigma));
            float l = fmaxf(xl, 0.f) - xl * yl + log1pf(expf(-fabsf(xl)));
            float gg = 1.f / (1.f + expf(-xl)) - yl;
            if (tp.loss_weight_on) { const float w = fabsf(weight[qq]); l *= w; gg *= w; }
            if (active && g == 0) acc_bce += (double)l;
            dxm = active ? gg * tp.inv_n_main / tp.sigma * s * dscale : 0.f;
        }
        // ---- the unit backward sweep: deltau_l; the BCE term's deltas are dxm * deltau_l
        const unsigned int tbase = (unsigned int)tile * 128u + (unsigned int)lane;
        const unsigned int tbig = (unsigned int)tile * (128u * MT) + (unsigned int)lane;
        if (want_dec) {
            unsigned int dh0, dl0, eh0, el0;
            h2_split2((g == 0) ? dxm : 0.f, 0.f, dh0, dl0);
            h2_split2((g == 0 && active) ? 1.f : 0.f, 0.f, eh0, el0);
            uint2* __restrict__ D = stream_at(ws.d + G::d_off(n_tiles, L), tbase);
            D[0] = transpose_block(dh0, 0u, ident);
            D[64] = transpose_block(dl0, 0u, ident);
            uint2* __restrict__ E = stream_at(ws2.d + G::d_off(n_tiles, L), tbase);
            E[0] = transpose_block(eh0, 0u, ident);
            E[64] = transpose_block(el0, 0u, ident);
            uint2* __restrict__ A = stream_at(ws.a + G::a_off(n_tiles, L), tbig);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                A[mt * 128] = transpose_block(ph[L - 1][mt >> 1][2 * (mt & 1)], ph[L - 1][mt >> 1][2 * (mt & 1) + 1], ident);
                A[mt * 128 + 64] = transpose_block(pl[L - 1][mt >> 1][2 * (mt & 1)], pl[L - 1][mt >> 1][2 * (mt & 1) + 1], ident);
            }
        }
#pragma unroll
        for (int kt = 0; kt < MT; ++kt) {
            const v4f_t wo = *reinterpret_cast<const v4f_t*>(O + 16 * kt + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) h[kt][r] = (active && h[kt][r] > 0.f) ? wo[r] : 0.f;
        }
        float a[4];
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            v4u_t uh[NJ], ul[NJ];
            Q::split_acts(h, uh, ul);
            if (want_dec) {
                v4f_t hm[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) hm[mt][r] = dxm * h[mt][r];
                v4u_t bh[NJ], bl[NJ];
                Q::split_acts(hm, bh, bl);
                uint2* __restrict__ D = stream_at(ws.d + G::d_off(n_tiles, l), tbig);
                uint2* __restrict__ E = stream_at(ws2.d + G::d_off(n_tiles, l), tbig);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    D[mt * 128] = transpose_block(bh[mt >> 1][2 * (mt & 1)], bh[mt >> 1][2 * (mt & 1) + 1], ident);
                    D[mt * 128 + 64] = transpose_block(bl[mt >> 1][2 * (mt & 1)], bl[mt >> 1][2 * (mt & 1) + 1], ident);
                    E[mt * 128] = transpose_block(uh[mt >> 1][2 * (mt & 1)], uh[mt >> 1][2 * (mt & 1) + 1], ident);
                    E[mt * 128 + 64] = transpose_block(ul[mt >> 1][2 * (mt & 1)], ul[mt >> 1][2 * (mt & 1) + 1], ident);
                }
                if (l > 0) {
                    uint2* __restrict__ A = stream_at(ws.a + G::a_off(n_tiles, l), tbig);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        A[mt * 128] = transpose_block(ph[l - 1][mt >> 1][2 * (mt & 1)], ph[l - 1][mt >> 1][2 * (mt & 1) + 1], ident);
                        A[mt * 128 + 64] = transpose_block(pl[l - 1][mt >> 1][2 * (mt & 1)], pl[l - 1][mt >> 1][2 * (mt & 1) + 1], ident);
                    }
                } else {
                    uint2* __restrict__ A = stream_at(ws.a + G::a_off(n_tiles, 0), tbase);
                    A[0] = transpose_block(zh[0], zh[1], ident);
                    A[64] = transpose_block(zl[0], zl[1], ident);
                }
            }
            if (l > 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4f_t){0.f, 0.f, 0.f, 0.f};
                Q::matmul(lds + Q::off_hidb(L, l), uh, ul, acc);
#pragma unroll
                for (int mj = 0; mj < MT; ++mj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[mj][r] = piece_on(ph[l - 1], pl[l - 1], mj, r) ? acc[mj][r] : 0.f;
            } else {
                Q::input_backward(lds, L, uh, ul, a);
            }
        }
        // ---- d pred / d q, the Eikonal term and c = d loss / d g (times s dscale: what multiplies the derivative network)
        float c0, c1, c2, cG;
        {
            float ax = 0.f, ay = 0.f, az = 0.f, cbar = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ax = fmaf(a[r], Y[0][r], ax); ay = fmaf(a[r], Y[1][r], ay); az = fmaf(a[r], Y[2][r], az);
                cbar = fmaf(a[r], z[r], cbar);
            }
            if (g == 2) {  // M^T a_pos
                d0 = M[0] * a[0] + M[3] * a[1] + M[6] * a[2];
                d1 = M[1] * a[0] + M[4] * a[1] + M[7] * a[2];
                d2 = M[2] * a[0] + M[5] * a[1] + M[8] * a[2];
            }
            ax = quad_lanes_sum(ax); ay = quad_lanes_sum(ay); az = quad_lanes_sum(az); cbar = quad_lanes_sum(cbar);
            d0 = quad_lanes_sum(d0); d1 = quad_lanes_sum(d1); d2 = quad_lanes_sum(d2);
            const float gx = s * (d0 + (ax - cbar * Gx) * invS);
            const float gy = s * (d1 + (ay - cbar * Gy) * invS);
            const float gz = s * (d2 + (az - cbar * Gz) * invS);
            const float n = sqrtf(gx * gx + gy * gy + gz * gz);
            const float r = n - 1.f;  // mapper.py:778-781, every sample (gradient_decimation = 1, config.py:438-439)
            if (active && g == 0) acc_eik += (double)(r * r);
            const float cf = (active && n > 0.f) ? tp.weight_e * 2.f * r * tp.inv_n_eik / n * s * dscale : 0.f;
            c0 = cf * gx; c1 = cf * gy; c2 = cf * gz;
            cG = c0 * Gx + c1 * Gy + c2 * Gz;
        }
        if (want_dec) {  // ---- the derivative network along zdot = (d z / d q) c: its activations are stream 2's A operands
            float zd[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) zd[r] = (c0 * Y[0][r] + c1 * Y[1][r] + c2 * Y[2][r] - cG * z[r]) * invS;
            if (g == 2) {
                zd[0] += M[0] * c0 + M[1] * c1 + M[2] * c2;
                zd[1] += M[3] * c0 + M[4] * c1 + M[5] * c2;
                zd[2] += M[6] * c0 + M[7] * c1 + M[8] * c2;
            }
            v2u_t th, tl;
            Q::split_input(zd, th, tl);
            uint2* __restrict__ B0 = stream_at(ws2.a + G::a_off(n_tiles, 0), tbase);
            B0[0] = transpose_block(th[0], th[1], ident);
            B0[64] = transpose_block(tl[0], tl[1], ident);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4f_t){0.f, 0.f, 0.f, 0.f};
            Q::layer0_add(lds, L, th, tl, acc);
#pragma unroll
            for (int l = 1; l <= L; ++l) {
#pragma unroll
                for (int mj = 0; mj < MT; ++mj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[mj][r] = piece_on(ph[l - 1], pl[l - 1], mj, r) ? acc[mj][r] : 0.f;
                v4u_t mh[NJ], ml[NJ];
                Q::split_acts(h, mh, ml);
                uint2* __restrict__ B = stream_at(ws2.a + G::a_off(n_tiles, l), tbig);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    B[mt * 128] = transpose_block(mh[mt >> 1][2 * (mt & 1)], mh[mt >> 1][2 * (mt & 1) + 1], ident);
                    B[mt * 128 + 64] = transpose_block(ml[mt >> 1][2 * (mt & 1)], ml[mt >> 1][2 * (mt & 1) + 1], ident);
                }
                if (l < L) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4f_t){0.f, 0.f, 0.f, 0.f};
                    Q::matmul(lds + Q::off_hidf(L, l), mh, ml, acc);
                }
            }
        }
        // ---- feature-gradient scatter: row t gets a_feat (w_t d loss / d head + c . d w_t / d q)
        if (g < 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sdz[nq * 8 + 4 * g + r] = a[r] * inv_dscale;
        } else if (g == 2) {
            const float base = dxm - cG * invS;
#pragma unroll
            for (int t = 0; t < PIN_MAX_K; ++t) {
                const float* gu = sgu + (nq * 8 + t) * 3;
                sw[nq * 8 + t] = fmaf(sw[nq * 8 + t], base, (c0 * gu[0] + c1 * gu[1] + c2 * gu[2]) * invS);
            }
        }
        wave_lds_sync();
        {
            const int t = lane >> 3, j = lane & 7;
            for (int i = 0; i < 16; ++i) {
                const int idx = sidx[i * 8 + t];
                if (idx >= 0) atomicAdd(feat_grad + (size_t)idx * PIN_FEATURE_DIM + j, sw[i * 8 + t] * sdz[i * 8 + j]);
            }
        }
        wave_lds_sync();
    }
    acc_bce = wave_sum(acc_bce);
    acc_eik = wave_sum(acc_eik);
    if (lane == 0) { lred[wave][0] = acc_bce; lred[wave][1] = acc_eik; }
    __syncthreads();
    if (threadIdx.x < 2) {
        double tt = 0.0;
#pragma unroll
        for (int w = 0; w < BLK / 64; ++w) tt += lred[w][threadIdx.x];
        loss_partial[2 * blockIdx.x + threadIdx.x] = tt;
    }
}

template <int H>
constexpr int train_fused_an_lds_bytes(int L) {
    return ((QuadDecoderH<H>::bytes(L) + 15) & ~15) + (TF_BLOCK / 64) * 6 * 16 * 8 * 4 + (TF_BLOCK / 64) * 2 * 8;
}

// the decoder image of train_fused_kernel, once per call: the hidden layers over blocks 0 .. STAGE_BLOCKS - 4 (the split
// is a chain of memory round trips, one trip per thread here), the three small parts on a block each
constexpr int STAGE_BLOCKS = 15;
template <int H>
__global__ __launch_bounds__(512) void train_stage_kernel(pin_field f, unsigned char* __restrict__ out) {
    constexpr int NB0 = STAGE_BLOCKS - 3;
    const int OD = f.out_dim > 1 ? f.out_dim : 1;
    if ((int)blockIdx.x < NB0) QuadDecoderH<H>::stage(f.dec, f.levels, out, blockIdx.x * 512 + threadIdx.x, NB0 * 512, OD, 0);
    else QuadDecoderH<H>::stage(f.dec, f.levels, out, threadIdx.x, 512, OD, (int)blockIdx.x - NB0 + 1);
}

template <int H>
constexpr int train_fused_lds_bytes(int L) {
    return ((QuadDecoderH<H>::bytes(L) + 15) & ~15) + (TFW_BLOCK / 64) * 3 * 16 * 8 * 4 + (TFW_BLOCK / 64) * 2 * 8;
}

// ---- weight gradient over the operand stream ----------------------------------------------------------------------
// grid (chunks of DW_CHUNK tiles, L + 1 layers), 16 waves.  A wave owns one 16-unit block of OUTPUT units of its layer
// and a PHASE of the chunk's tiles (layers with fewer output blocks have more phases), and all input blocks: per tile
// 2 + 2 * AB operand loads of 8 bytes per lane (512 contiguous bytes per wave and load), issued four tiles ahead of
// the 3 * AB + 2 MFMAs that consume them; hi*hi into the main accumulators, the two cross products into a second set
// folded in with 2^-11; the bias gradient is the product with a block of ones.  The phases of a block are added up
// through LDS (two halving steps), and one wave per output block adds the result into the partial gradient of the
// chunk's SLOT (chunk % DW_SLOTS, train_finalize_kernel sums the slots): at the reference's batch of 16k samples a
// version with one wave per 8 tiles adding straight into the 13k decoder gradients spent 33 of its 47 us on atomics.
constexpr int DW_CHUNK = 32;   // tiles per block
constexpr int DW_WAVES = 16;
constexpr int DW_VALS = 20;    // per lane: 4 input blocks x 4 + 4 bias sums

template <int H>
__global__ __launch_bounds__(DW_WAVES * 64) void train_dw_stream_kernel(DwStream ws, int L, int OD, int n_dec,
                                                                        float* __restrict__ partial, int no_bias, int chunk) {
    using G = DwGeom<H>;
    constexpr int MT = G::MT;
    __shared__ float red[DW_WAVES / 2][DW_VALS][64];
    const int lam = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int DB = G::d_blocks(L, lam), AB = G::a_blocks(lam);  // DB in {1, 2, 4}
    const int ob = wave % DB, phase = wave / DB, phases = DW_WAVES / DB;
    const size_t n_tiles = (size_t)ws.n_tiles;
    const uint2* __restrict__ D = ws.d + G::d_off(n_tiles, lam) + (size_t)ob * 128 + lane;
    const uint2* __restrict__ A = ws.a + G::a_off(n_tiles, lam) + lane;
    const int t0 = blockIdx.x * chunk, t1 = min(t0 + chunk, ws.n_tiles);
    v4f_t mainv[MT], cross[MT], bmain = (v4f_t){0.f, 0.f, 0.f, 0.f}, bcross = (v4f_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ib = 0; ib < MT; ++ib) { mainv[ib] = (v4f_t){0.f, 0.f, 0.f, 0.f}; cross[ib] = (v4f_t){0.f, 0.f, 0.f, 0.f}; }
    const v4h_t ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
    auto as4 = [](uint2 v) { const v2u_t u = {v.x, v.y}; return as_h4(u); };
    constexpr int TPT = MT >= 4 ? 3 : 4;  // tiles per trip (all their operand loads are issued before the first MFMA; 128 VGPRs at 16 waves)
    for (int t = t0 + phase; t < t1; t += TPT * phases) {
        uint2 dh[TPT], dl[TPT], ah[TPT][MT], al[TPT][MT];
#pragma unroll
        for (int u = 0; u < TPT; ++u) {
            const int tu = t + u * phases;
            const size_t tt = (size_t)(tu < t1 ? tu : t);
            dh[u] = D[tt * 128 * DB];
            dl[u] = D[tt * 128 * DB + 64];
#pragma unroll
            for (int ib = 0; ib < MT; ++ib) {
                const int bb = ib < AB ? ib : 0;
                ah[u][ib] = A[(tt * AB + bb) * 128];
                al[u][ib] = A[(tt * AB + bb) * 128 + 64];
            }
        }
#pragma unroll
        for (int u = 0; u < TPT; ++u) {
            if (t + u * phases >= t1) break;
            const v4h_t d_h = as4(dh[u]), d_l = as4(dl[u]);
            bmain = __builtin_amdgcn_mfma_f32_16x16x16f16(d_h, ones, bmain, 0, 0, 0);
            bcross = __builtin_amdgcn_mfma_f32_16x16x16f16(d_l, ones, bcross, 0, 0, 0);
#pragma unroll
            for (int ib = 0; ib < MT; ++ib) {
                if (ib >= AB) continue;
                const v4h_t a_h = as4(ah[u][ib]), a_l = as4(al[u][ib]);
                mainv[ib] = __builtin_amdgcn_mfma_f32_16x16x16f16(d_h, a_h, mainv[ib], 0, 0, 0);
                cross[ib] = __builtin_amdgcn_mfma_f32_16x16x16f16(d_h, a_l, cross[ib], 0, 0, 0);
                cross[ib] = __builtin_amdgcn_mfma_f32_16x16x16f16(d_l, a_h, cross[ib], 0, 0, 0);
            }
        }
    }
    // fold the cross sums in; add the phases up: waves [half, 2 half) park, waves [0, half) add, until DB waves are left
    float val[DW_VALS];
#pragma unroll
    for (int ib = 0; ib < MT; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) val[4 * ib + r] = fmaf(cross[ib][r], H2_DOWN, mainv[ib][r]);
#pragma unroll
    for (int ib = MT; ib < 4; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) val[4 * ib + r] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) val[16 + r] = fmaf(bcross[r], H2_DOWN, bmain[r]);
    for (int half = DW_WAVES / 2; half >= DB; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
#pragma unroll
            for (int c = 0; c < DW_VALS; ++c) red[wave - half][c][lane] = val[c];
        }
        __syncthreads();
        if (wave < half) {
#pragma unroll
            for (int c = 0; c < DW_VALS; ++c) val[c] += red[wave][c][lane];
        }
        __syncthreads();
    }
    if (wave >= DB) return;
    // D layout: element [o = 4 g + r][i = n] of block (ob, ib).  state_dict order: W0 [H][11], b0, (W [H][H], b)*, lout
    const int rows = lam < L ? H : OD;
    const int cols_out = lam == 0 ? MLP_IN : H;
    size_t off = 0;
    for (int u = 0; u < lam; ++u) off += (size_t)H * (u == 0 ? MLP_IN : H) + H;
    float* __restrict__ gW = partial + (size_t)(blockIdx.x % DW_SLOTS) * n_dec + off;
    float* __restrict__ gb = gW + (size_t)rows * cols_out;
#pragma unroll
    for (int ib = 0; ib < MT; ++ib) {
        if (ib >= AB) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 16 * ob + 4 * g + r, i = 16 * ib + n;
            const float v = val[4 * ib + r];
            if (o < rows && i < cols_out && v != 0.f) atomicAdd(gW + (size_t)o * cols_out + i, v);
        }
    }
    if (n == 0 && !no_bias) {  // (the derivative-network stream of the analytic Eikonal term has no bias gradient)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 16 * ob + 4 * g + r;
            const float v = val[16 + r];
            if (o < rows && v != 0.f) atomicAdd(gb + o, v);
        }
    }
}

// ---- weight gradient with the forward pass run again (large batches) ----------------------------------------------------
// At 2^20 samples the operand stream is 7.3 of the 11.9 GB an iteration moves (profiles/r03_pmc_c4.json), and half of it is
// the layers' INPUTS -- which are a function of the decoder input z (one block per tile) and the weights.  Here the tile
// kernel streams only z and the deltas; a wave takes z back to the query-in-the-lane layout (the identity-MFMA transpose
// applied twice is the identity), runs layers 0 .. lam - 1 with the tile kernel's own calls -- same instructions, same
// bits -- and transposes a_lam into the A operand of dW_lam = delta_{lam+1} (x) a_lam.  The matrix cores and the vector
// ALU of train_dw_stream_kernel are idle four fifths of the time (it waits for HBM); this spends them instead of bytes.
// grid (chunks, L + 1 layers), 8 waves: a wave = a phase of the chunk's tiles, ALL output blocks of the layer (64 + 16
// accumulators); the cross products of a tile are folded into the main accumulators tile by tile.
constexpr int DWR_WAVES = 8;
constexpr int DWR_VALS = 120;       // per lane: 4 x 4 blocks x 4 + 4 x 4 bias sums of the primary layer, 16 + 16 of the secondary, padding
constexpr int DWR_RED_VALS = 40;    // reduced through LDS in three parts
// Layers are handled in GROUPS that share one recomputation: {0, 1} (layer 0 reads z itself), the middle layers singly,
// {L - 1, L} (the out layer needs one more layer on top of L - 1 and has a single row block): 7 instead of 10 layer passes
// per tile at L = 4.
__host__ __device__ inline int dwr_groups(int L) { return L == 1 ? 1 : (L == 2 ? 2 : L - 1); }
__host__ __device__ inline void dwr_group(int L, int grp, int& pri, int& sec) {
    if (grp == 0) { pri = 1; sec = 0; return; }
    const int lam = grp + 1;
    if (lam == L - 1 && lam >= 2) { pri = lam; sec = L; } else { pri = lam; sec = -1; }
}
template <int H>
constexpr int train_dw_recompute_lds_bytes(int L) {
    return ((QuadDecoderH<H>::bytes(L) + 15) & ~15) + (DWR_WAVES / 2) * DWR_RED_VALS * 64 * 4;
}
// HI_ONLY: the deltas were streamed as their high fp16 pieces only (train_fused_kernel, acts_out = 2): no low piece is read
// and the product d_lo (x) a_hi is not formed -- two matrix instructions per pair of blocks instead of three
template <int H, bool HI_ONLY = false>
__global__ __launch_bounds__(DWR_WAVES * 64, 1) void train_dw_recompute_kernel(DwStream ws, int L, int OD, int n_dec,
                                                                              float* __restrict__ partial, int chunk,
                                                                              const unsigned char* __restrict__ dec_image) {
    using Q = QuadDecoderH<H>;
    using G = DwGeom<H>;
    constexpr int MT = Q::MT, NJ = Q::NJ;
    extern __shared__ __attribute__((aligned(16))) unsigned char dwr_smem[];
    unsigned char* const lds = dwr_smem;
    const int IMG = (Q::bytes(L) + 15) & ~15;
    float (*red)[DWR_RED_VALS][64] = reinterpret_cast<float (*)[DWR_RED_VALS][64]>(lds + IMG);
    int pri, sec;
    dwr_group(L, (int)blockIdx.y, pri, sec);
    const int depth = sec > pri ? sec : pri;  // layers to run: a_depth
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int DBp = G::d_blocks(L, pri);      // (the primary layer has MT input blocks: pri >= 1)
    {   // the forward images of layers 0 .. depth - 1 and the biases, at the offsets the decoder calls expect
        auto copy = [&](int from, int to) {
            const uint4* __restrict__ src = reinterpret_cast<const uint4*>(dec_image + from);
            uint4* __restrict__ dst = reinterpret_cast<uint4*>(lds + from);
            for (int i = threadIdx.x; i < ((to - from) >> 4); i += DWR_WAVES * 64) dst[i] = src[i];
        };
        copy(Q::off_hidf(L, 1), Q::off_hidf(L, depth));
        copy(Q::off_l0f(L), Q::off_l0b(L));
        copy(Q::off_bias(L), Q::off_out(L));
        __syncthreads();
    }
    v4h_t ident;
#pragma unroll
    for (int r = 0; r < 4; ++r) ident[r] = (n == 4 * g + r) ? (_Float16)1.0f : (_Float16)0.0f;
    const v4h_t ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
    auto as4 = [](uint2 v) { const v2u_t u = {v.x, v.y}; return as_h4(u); };
    // acc += d (x) a for one pair of blocks: hi * hi into the accumulator, the two cross products folded in tile by tile
    auto outer = [&](v4f_t& acc, uint2 dh, uint2 dl, v4h_t a_h, v4h_t a_l) {
        const v4h_t d_h = as4(dh), d_l = as4(dl);
        v4f_t c = (v4f_t){0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_16x16x16f16(d_h, a_l, c, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(d_h, a_h, acc, 0, 0, 0);
        if constexpr (!HI_ONLY) c = __builtin_amdgcn_mfma_f32_16x16x16f16(d_l, a_h, c, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = fmaf(c[r], H2_DOWN, acc[r]);
    };
    auto rowsum = [&](v4f_t& acc, uint2 dh, uint2 dl) {  // bias gradient: the product with a block of ones
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(as4(dh), ones, acc, 0, 0, 0);
        if constexpr (!HI_ONLY) {
            v4f_t c = (v4f_t){0.f, 0.f, 0.f, 0.f};
            c = __builtin_amdgcn_mfma_f32_16x16x16f16(as4(dl), ones, c, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fmaf(c[r], H2_DOWN, acc[r]);
        }
    };
    const size_t n_tiles = (size_t)ws.n_tiles;
    const uint2* __restrict__ Dp = ws.d + G::d_off(n_tiles, pri) + lane;
    const uint2* __restrict__ Ds = ws.d + G::d_off(n_tiles, sec < 0 ? 0 : sec) + lane;
    const uint2* __restrict__ A0 = ws.a + G::a_off(n_tiles, 0) + lane;
    const int t0 = blockIdx.x * chunk, t1 = min(t0 + chunk, ws.n_tiles);
    v4f_t mainv[MT][MT], bmain[MT], secv[MT], sbias[MT];
#pragma unroll
    for (int ob = 0; ob < MT; ++ob) {
        bmain[ob] = secv[ob] = sbias[ob] = (v4f_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ib = 0; ib < MT; ++ib) mainv[ob][ib] = (v4f_t){0.f, 0.f, 0.f, 0.f};
    }
    uint2 z_h = make_uint2(0u, 0u), z_l = make_uint2(0u, 0u);
    if (t0 + wave < t1) { z_h = A0[(size_t)(t0 + wave) * 128]; z_l = A0[(size_t)(t0 + wave) * 128 + 64]; }
    for (int t = t0 + wave; t < t1; t += DWR_WAVES) {
        uint2 dh[MT], dl[MT], sh[MT], sl[MT];
#pragma unroll
        for (int ob = 0; ob < MT; ++ob) {
            const int bb = ob < DBp ? ob : 0;
            dh[ob] = Dp[((size_t)t * DBp + bb) * 128];
            dl[ob] = HI_ONLY ? make_uint2(0u, 0u) : Dp[((size_t)t * DBp + bb) * 128 + 64];
            sh[ob] = sl[ob] = make_uint2(0u, 0u);
        }
        if (sec == 0) {  // delta_1: MT blocks
#pragma unroll
            for (int ob = 0; ob < MT; ++ob) {
                sh[ob] = Ds[((size_t)t * MT + ob) * 128];
                if constexpr (!HI_ONLY) sl[ob] = Ds[((size_t)t * MT + ob) * 128 + 64];
            }
        } else if (sec > 0) {  // d loss / d heads: one block
            sh[0] = Ds[(size_t)t * 128];
            if constexpr (!HI_ONLY) sl[0] = Ds[(size_t)t * 128 + 64];
        }
        const uint2 zc_h = z_h, zc_l = z_l;
        if (t + DWR_WAVES < t1) { z_h = A0[(size_t)(t + DWR_WAVES) * 128]; z_l = A0[(size_t)(t + DWR_WAVES) * 128 + 64]; }
        if (sec == 0) {  // layer 0: its input is z as streamed
            const v4h_t a_h = as4(zc_h), a_l = as4(zc_l);
#pragma unroll
            for (int ob = 0; ob < MT; ++ob) { outer(secv[ob], sh[ob], sl[ob], a_h, a_l); rowsum(sbias[ob], sh[ob], sl[ob]); }
        }
        // z back in the query-in-the-lane layout, then the tile kernel's forward pass
        const uint2 qh = transpose_block(zc_h.x, zc_h.y, ident), ql = transpose_block(zc_l.x, zc_l.y, ident);
        const v2u_t zh = {qh.x, qh.y}, zl = {ql.x, ql.y};
        v4f_t h[MT], acc[MT];
        Q::layer0(lds, L, zh, zl, acc);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[mt][r] = relu1(acc[mt][r]);
        v4u_t ph[NJ], pl[NJ];
        for (int l = 1; l < pri; ++l) {
            Q::split_acts(h, ph, pl);
            Q::load_bias(lds, L, l, acc);
            Q::matmul(lds + Q::off_hidf(L, l), ph, pl, acc);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) h[mt][r] = relu1(acc[mt][r]);
        }
        Q::split_acts(h, ph, pl);  // a_pri
#pragma unroll
        for (int ib = 0; ib < MT; ++ib) {
            const v4h_t a_h = as4(transpose_block(ph[ib >> 1][2 * (ib & 1)], ph[ib >> 1][2 * (ib & 1) + 1], ident));
            const v4h_t a_l = as4(transpose_block(pl[ib >> 1][2 * (ib & 1)], pl[ib >> 1][2 * (ib & 1) + 1], ident));
#pragma unroll
            for (int ob = 0; ob < MT; ++ob) {
                if (ob >= DBp) continue;
                outer(mainv[ob][ib], dh[ob], dl[ob], a_h, a_l);
            }
        }
#pragma unroll
        for (int ob = 0; ob < MT; ++ob) {
            if (ob >= DBp) continue;
            rowsum(bmain[ob], dh[ob], dl[ob]);
        }
        if (sec > pri) {  // the out layer: one more layer on top, a single block of rows
            Q::load_bias(lds, L, pri, acc);
            Q::matmul(lds + Q::off_hidf(L, pri), ph, pl, acc);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) h[mt][r] = relu1(acc[mt][r]);
            Q::split_acts(h, ph, pl);
#pragma unroll
            for (int ib = 0; ib < MT; ++ib) {
                const v4h_t a_h = as4(transpose_block(ph[ib >> 1][2 * (ib & 1)], ph[ib >> 1][2 * (ib & 1) + 1], ident));
                const v4h_t a_l = as4(transpose_block(pl[ib >> 1][2 * (ib & 1)], pl[ib >> 1][2 * (ib & 1) + 1], ident));
                outer(secv[ib], sh[0], sl[0], a_h, a_l);
            }
            rowsum(sbias[0], sh[0], sl[0]);
        }
    }
    // add the eight phases up: three parts of the values through the patch behind the image
    float val[DWR_VALS];
#pragma unroll
    for (int c = 0; c < DWR_VALS; ++c) val[c] = 0.f;
#pragma unroll
    for (int ob = 0; ob < MT; ++ob) {
#pragma unroll
        for (int ib = 0; ib < MT; ++ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) val[16 * ob + 4 * ib + r] = mainv[ob][ib][r];
#pragma unroll
        for (int r = 0; r < 4; ++r) { val[64 + 4 * ob + r] = bmain[ob][r]; val[80 + 4 * ob + r] = secv[ob][r]; val[96 + 4 * ob + r] = sbias[ob][r]; }
    }
    __syncthreads();
#pragma unroll
    for (int part = 0; part < DWR_VALS / DWR_RED_VALS; ++part) {
        if (part == 2 && sec < 0) break;  // (uniform: no secondary layer in this block)
        for (int half = DWR_WAVES / 2; half >= 1; half >>= 1) {
            if (wave >= half && wave < 2 * half) {
#pragma unroll
                for (int c = 0; c < DWR_RED_VALS; ++c) red[wave - half][c][lane] = val[part * DWR_RED_VALS + c];
            }
            __syncthreads();
            if (wave < half) {
#pragma unroll
                for (int c = 0; c < DWR_RED_VALS; ++c) val[part * DWR_RED_VALS + c] += red[wave][c][lane];
            }
            __syncthreads();
        }
    }
    if (wave != 0) return;
    float* __restrict__ slot = partial + (size_t)(blockIdx.x % DW_SLOTS) * n_dec;
    // D layout: element [o = 4 g + r][i = n] of block (ob, ib) of layer lam.  state_dict order: W0 [H][11], b0, (W [H][H], b)*, lout
    auto emit = [&](int lam, int ob, int ib, int r, float v, bool bias) {
        const int rows = lam < L ? H : OD, cols = lam == 0 ? MLP_IN : H;
        size_t off = 0;
        for (int u = 0; u < lam; ++u) off += (size_t)H * (u == 0 ? MLP_IN : H) + H;
        const int o = 16 * ob + 4 * g + r, i = 16 * ib + n;
        if (v == 0.f || o >= rows) return;
        if (bias) { if (n == 0) atomicAdd(slot + off + (size_t)rows * cols + o, v); }
        else if (i < cols) atomicAdd(slot + off + (size_t)o * cols + i, v);
    };
#pragma unroll
    for (int ob = 0; ob < MT; ++ob) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (ob < DBp) {
#pragma unroll
                for (int ib = 0; ib < MT; ++ib) emit(pri, ob, ib, r, val[16 * ob + 4 * ib + r], false);
                emit(pri, ob, 0, r, val[64 + 4 * ob + r], true);
            }
            if (sec == 0) {
                emit(0, ob, 0, r, val[80 + 4 * ob + r], false);
                emit(0, ob, 0, r, val[96 + 4 * ob + r], true);
            } else if (sec > 0) {
                emit(sec, 0, ob, r, val[80 + 4 * ob + r], false);   // (secv is indexed by the INPUT block there)
                if (ob == 0) emit(sec, 0, 0, r, val[96 + r], true);
            }
        }
    }
}

// dec_grad += (sum of the slot partials) / dscale; block 0 adds up the per-block loss sums of the tile kernel
__global__ __launch_bounds__(256) void train_finalize_kernel(const float* __restrict__ partial, int n_dec, float inv_dscale,
                                                             float* __restrict__ dec_grad,
                                                             const double* __restrict__ loss_partial, int n_loss,
                                                             double* __restrict__ loss_out, int n_loss_out) {
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        double b = 0.0, e = 0.0;
        for (int i = threadIdx.x; i < n_loss; i += 64) { b += loss_partial[2 * i]; e += loss_partial[2 * i + 1]; }
        b = wave_sum(b); e = wave_sum(e);
        if (threadIdx.x == 0) { loss_out[0] = b; if (n_loss_out > 1) loss_out[1] = e; }
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (dec_grad == nullptr || i >= n_dec) return;
    float v[DW_SLOTS];
#pragma unroll
    for (int c = 0; c < DW_SLOTS; ++c) v[c] = partial[(size_t)c * n_dec + i];
    float t = 0.f;
#pragma unroll
    for (int c = 0; c < DW_SLOTS; ++c) t += v[c];
    dec_grad[i] += t * inv_dscale;
}

}  // namespace pin
