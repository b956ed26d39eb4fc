// Shallow-MLP decoder on the fp32 matrix cores (v_mfma_f32_16x16x4_f32) for gfx950.
//
// Decoder.mlp (model/decoder.py:61-80), its input Jacobian and its backward pass for 64 queries
// per wave.  Matrix roles:  D[unit][query] += W[unit][k] * H[k][query]   (M = 16 output units per
// tile, N = 16 queries per tile, K = 4 inputs per instruction).  Why the matrix cores although
// the fp32 MFMA rate equals the vector rate: operand delivery.  On the vector path every FMA
// needs its own wave-uniform weight (scalar-load bound for 53 KB of 4x64 weights) or a
// broadcast LDS read (LDS bound); an MFMA consumes ONE weight register per lane for
// 16x16x4 = 1024 FMAs, so all weights stream from LDS at a few % of its bandwidth.
//
// Layout trick (no data movement between layers): the 16x16x4 result tile puts
//   D[unit = 16*mt + 4*g + r][query = n]  in lane (n = lane & 15, g = lane >> 4), register r,
// and the B operand of the next layer wants  H[k][query = n]  in lane (n, g) for k-slot g.
// The k index of an MFMA is a free permutation as long as A and B agree, so K-step (mt, r) of
// the next layer is declared to cover units {16*mt + 4*g + r, g = 0..3}: the activation
// registers ARE the B operands, and the weights are stored in LDS pre-permuted to match
// (A operand of step (kt, r), lane (i, g):  W[16*mt_out + i][16*kt + 4*g + r]).
// The same array read with a different index serves the transposed products of the Jacobian
// and of the backward pass.  Output heads: 1 (SDF) or 3 (colour, Decoder.regress_color).
#pragma once
#include "mlp.h"

namespace pin {

typedef float v4f_t __attribute__((ext_vector_type(4)));

// LDS hand-off between lanes of ONE wave: LDS ops of a wave execute in order, so only the
// compiler has to be kept from moving them across this point.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int MF_BLOCK = 256;  // 4 waves share one LDS image of the weights
constexpr int MF_OD_MAX = 3;   // output heads: 1 (sdf) or 3 (colour)


__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

template <int H>
struct MfmaDecoder {
    static constexpr int MT = H / 16;          // unit tiles
    static constexpr int XSTRIDE = 13;         // per-query stride of the z / a exchange buffer
    // LDS image (floats)
    static constexpr int OFF_A0 = 0;                       // [MT][3][64]      layer-0 forward operand
    static constexpr int OFF_A0T = OFF_A0 + MT * 3 * 64;   // [MT][4][64]      layer-0 transposed operand
    static constexpr int OFF_B0 = OFF_A0T + MT * 4 * 64;   // [H]
    static constexpr int OFF_HID = OFF_B0 + H;             // per hidden layer: F [MT][MT][64][4], bias [H]
    static constexpr int HID_SZ = H * H + H;
    __host__ __device__ static constexpr int off_out(int L) { return OFF_HID + (L - 1) * HID_SZ; }  // Wo [3][H], bo [3], pad
    __host__ __device__ static constexpr int weight_floats(int L) { return off_out(L) + MF_OD_MAX * H + 4; }
    __host__ __device__ static constexpr int scratch_floats() { return 64 * XSTRIDE; }  // per wave

    // ------------------------------------------------------------------------------------ staging
    // Block-cooperative: permute the flat state_dict-ordered parameters into the LDS image; loads
    // are issued in batches of 8 independent requests per thread.
    template <typename SrcIndex>
    __device__ __forceinline__ static void copy_permuted(const float* __restrict__ src, float* __restrict__ dst, int count,
                                                         int tid, int nthreads, SrcIndex idx) {
        for (int e0 = tid; e0 < count; e0 += 8 * nthreads) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * nthreads;
                const int si = e < count ? idx(e) : -1;
                v[u] = si >= 0 ? src[si] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * nthreads;
                if (e < count) dst[e] = v[u];
            }
        }
    }

    __device__ static void stage(const float* __restrict__ dec, int L, float* __restrict__ w, int tid, int nthreads,
                                 int OD = 1) {
        const float* W0 = dec;
        copy_permuted(W0, w + OFF_A0, MT * 3 * 64, tid, nthreads, [](int e) {
            const int lane = e & 63, s = (e >> 6) % 3, mt = e / (3 * 64);
            const int c = 4 * s + (lane >> 4);
            return c < MLP_IN ? (16 * mt + (lane & 15)) * MLP_IN + c : -1;
        });
        copy_permuted(W0, w + OFF_A0T, MT * 4 * 64, tid, nthreads, [](int e) {
            const int lane = e & 63, r = (e >> 6) & 3, kt = e >> 8;
            const int c = lane & 15;
            return c < MLP_IN ? (16 * kt + 4 * (lane >> 4) + r) * MLP_IN + c : -1;
        });
        copy_permuted(dec + H * MLP_IN, w + OFF_B0, H, tid, nthreads, [](int e) { return e; });
        const float* P = dec + H * MLP_IN + H;
        for (int l = 1; l < L; ++l) {
            float* F = w + OFF_HID + (l - 1) * HID_SZ;
            copy_permuted(P, F, H * H, tid, nthreads, [](int e) {
                const int r = e & 3, lane = (e >> 2) & 63, kt = (e >> 8) % MT, mt = e / (256 * MT);
                return (16 * mt + (lane & 15)) * H + 16 * kt + 4 * (lane >> 4) + r;
            });
            copy_permuted(P + H * H, F + H * H, H, tid, nthreads, [](int e) { return e; });
            P += H * H + H;
        }
        // lout.weight [OD][H] then lout.bias [OD]  ->  Wo at O[c*H + u], bias at O[3H + c]
        copy_permuted(P, w + off_out(L), OD * H, tid, nthreads, [](int e) { return e; });
        copy_permuted(P + OD * H, w + off_out(L) + MF_OD_MAX * H, OD, tid, nthreads, [](int e) { return e; });
    }

    // ----------------------------------------------------------------------------- building blocks
    // all blocks work on NT query tiles (16 queries each) starting at row qb of the exchange buffer
    template <int NT>
    __device__ __forceinline__ static void layer0(const float* __restrict__ w, const float* __restrict__ xb, int qb,
                                                  v4f_t (&acc)[MT][NT]) {
        const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
        float zb[NT][3];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s = 0; s < 3; ++s) zb[nt][s] = xb[(qb + 16 * nt + n) * XSTRIDE + 4 * s + g];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const v4f_t b4 = *reinterpret_cast<const v4f_t*>(w + OFF_B0 + 16 * mt + 4 * g);
            float a0[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) a0[s] = w[OFF_A0 + (mt * 3 + s) * 64 + lane];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                v4f_t c = b4;
#pragma unroll
                for (int s = 0; s < 3; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], zb[nt][s], c, 0, 0, 0);
                acc[mt][nt] = c;
            }
        }
    }

    template <int NT>
    __device__ __forceinline__ static void hidden(const float* __restrict__ F, const v4f_t (&h)[MT][NT], v4f_t (&acc)[MT][NT]) {
        const int lane = threadIdx.x & 63, g = lane >> 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const v4f_t b4 = *reinterpret_cast<const v4f_t*>(F + H * H + 16 * mt + 4 * g);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = b4;
#pragma unroll
            for (int kt = 0; kt < MT; ++kt) {
                const v4f_t a4 = *reinterpret_cast<const v4f_t*>(F + ((mt * MT + kt) * 64 + lane) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[r], h[kt][nt][r], acc[mt][nt], 0, 0, 0);
            }
        }
    }

    // h = relu(acc); returns the mask word, bit (mt*NT+nt)*4+r
    template <int NT>
    __device__ __forceinline__ static unsigned long long relu(const v4f_t (&acc)[MT][NT], v4f_t (&h)[MT][NT]) {
        unsigned long long mm = 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool on = acc[mt][nt][r] > 0.f;
                    mm |= (unsigned long long)on << ((mt * NT + nt) * 4 + r);
                    h[mt][nt][r] = on ? acc[mt][nt][r] : 0.f;
                }
        return mm;
    }

    // output heads: xo[c][nt] = bias_c + sum_u Wo[c][u] h[u][query], complete in all 4 k-groups
    template <int NT, int OD>
    __device__ __forceinline__ static void out_layer(const float* __restrict__ O, const v4f_t (&h)[MT][NT], float (&xo)[OD][NT]) {
        const int g = (threadIdx.x & 63) >> 4;
#pragma unroll
        for (int c = 0; c < OD; ++c)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) xo[c][nt] = 0.f;
#pragma unroll
        for (int c = 0; c < OD; ++c)
#pragma unroll
            for (int kt = 0; kt < MT; ++kt) {
                const v4f_t wo = *reinterpret_cast<const v4f_t*>(O + c * H + 16 * kt + 4 * g);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xo[c][nt] = fmaf(wo[r], h[kt][nt][r], xo[c][nt]);
            }
#pragma unroll
        for (int c = 0; c < OD; ++c)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                xo[c][nt] = rows_sum(xo[c][nt]);
                xo[c][nt] += O[MF_OD_MAX * H + c];
            }
    }

    // seed of a transposed sweep: h[u][q] = mask ? sum_c coef[c][q] * Wo[c][u] : 0
    template <int NT, int OD>
    __device__ __forceinline__ static void seed(const float* __restrict__ O, unsigned long long mm, const float (&coef)[OD][NT],
                                                v4f_t (&h)[MT][NT]) {
        const int g = (threadIdx.x & 63) >> 4;
#pragma unroll
        for (int kt = 0; kt < MT; ++kt) {
            v4f_t wo[OD];
#pragma unroll
            for (int c = 0; c < OD; ++c) wo[c] = *reinterpret_cast<const v4f_t*>(O + c * H + 16 * kt + 4 * g);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = 0.f;
#pragma unroll
                    for (int c = 0; c < OD; ++c) v = fmaf(coef[c][nt], wo[c][r], v);
                    h[kt][nt][r] = ((mm >> ((kt * NT + nt) * 4 + r)) & 1ull) ? v : 0.f;
                }
        }
    }

    // transposed hidden layer: h <- mask_prev .* (W_l^T h), W_l read out of the forward image
    template <int NT>
    __device__ __forceinline__ static void back_hidden(const float* __restrict__ F, unsigned long long mm, v4f_t (&h)[MT][NT],
                                                       v4f_t (&acc)[MT][NT]) {
        const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
#pragma unroll
        for (int mj = 0; mj < MT; ++mj) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mj][nt] = (v4f_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ki = 0; ki < MT; ++ki)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // W_l[16*ki + 4*g + r][16*mj + n] out of the forward image (see header)
                    const float at = F[((ki * MT + mj) * 64 + 16 * (n >> 2) + 4 * g + r) * 4 + (n & 3)];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mj][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, h[ki][nt][r], acc[mj][nt], 0, 0, 0);
                }
        }
#pragma unroll
        for (int mj = 0; mj < MT; ++mj)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    h[mj][nt][r] = ((mm >> ((mj * NT + nt) * 4 + r)) & 1ull) ? acc[mj][nt][r] : 0.f;
    }

    // transposed layer 0, results to rows qb.. of the exchange buffer (components 4g + r)
    template <int NT>
    __device__ __forceinline__ static void back_input(const float* __restrict__ w, const v4f_t (&h)[MT][NT],
                                                      float* __restrict__ xb, int qb) {
        const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
        v4f_t ai[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) ai[nt] = (v4f_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < MT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float at = w[OFF_A0T + (kt * 4 + r) * 64 + lane];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    ai[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, h[kt][nt][r], ai[nt], 0, 0, 0);
            }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * g + r < 12) xb[(qb + 16 * nt + n) * XSTRIDE + 4 * g + r] = ai[nt][r];
    }

    __device__ __forceinline__ static void put_z(float* __restrict__ xb, const float (&z)[MLP_IN]) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int j = 0; j < MLP_IN; ++j) xb[lane * XSTRIDE + j] = z[j];
        xb[lane * XSTRIDE + 11] = 0.f;
        wave_lds_sync();
    }

    // value of this lane's own query (row `lane`) out of per-tile values
    template <int NT>
    __device__ __forceinline__ static float own(const float (&v)[NT], int qb) {
        const int lane = threadIdx.x & 63, n = lane & 15;
        float o = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) o += (lane - qb == 16 * nt + n) ? v[nt] : 0.f;
        return o;
    }

    // ------------------------------------------------------------------ inference: value (+ Jacobian)
    // Forward (+ input Jacobian) for the 64 queries of this wave in passes of NT query tiles
    // (fewer live accumulators -> 2 waves per SIMD).  OD = 1: returns the raw MLP output, a_in =
    // d out / d z.  OD = 3 (colour): pout[c] = sigmoid(out_c) and the returned value / a_in refer
    // to  sum_c kappa[c] * sigmoid(out_c)  (kappa = intensity weights or a one-hot channel).
    template <bool GRAD, int NT, int OD>
    __device__ __forceinline__ static float run_heads(const float* __restrict__ w, int L, float* __restrict__ xb,
                                                      const float (&z)[MLP_IN], const float (&kappa)[OD], float (&pout)[OD],
                                                      float (&a_in)[MLP_IN]) {
        const int lane = threadIdx.x & 63;
        put_z(xb, z);
        float val = 0.f;
#pragma unroll
        for (int c = 0; c < OD; ++c) pout[c] = 0.f;
#pragma unroll 1
        for (int p = 0; p < 4 / NT; ++p) {
            const int qb = 16 * NT * p;
            v4f_t h[MT][NT], acc[MT][NT];
            unsigned long long m0, m1 = 0, m2 = 0, m3 = 0;
            layer0<NT>(w, xb, qb, acc);
            m0 = relu<NT>(acc, h);
            for (int l = 1; l < L; ++l) {
                hidden<NT>(w + OFF_HID + (l - 1) * HID_SZ, h, acc);
                const unsigned long long mm = relu<NT>(acc, h);
                m1 = l == 1 ? mm : m1; m2 = l == 2 ? mm : m2; m3 = l == 3 ? mm : m3;
            }
            const float* __restrict__ O = w + off_out(L);
            float xo[OD][NT], coef[OD][NT];
            out_layer<NT, OD>(O, h, xo);
            const bool mine = (lane >> 4) / NT == p;
#pragma unroll
            for (int c = 0; c < OD; ++c) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (OD == 1) { coef[c][nt] = 1.f; }
                    else { const float s = sigmoidf_(xo[c][nt]); xo[c][nt] = s; coef[c][nt] = kappa[c] * s * (1.f - s); }
                }
                const float o = own<NT>(xo[c], qb);
                if (mine) { pout[c] = o; val += (OD == 1 ? 1.f : kappa[c]) * o; }
            }
            if (GRAD) {
                const unsigned long long mlast = L == 1 ? m0 : L == 2 ? m1 : L == 3 ? m2 : m3;
                seed<NT, OD>(O, mlast, coef, h);
                for (int l = L - 1; l >= 1; --l) {
                    const unsigned long long mm = l == 1 ? m0 : l == 2 ? m1 : m2;
                    back_hidden<NT>(w + OFF_HID + (l - 1) * HID_SZ, mm, h, acc);
                }
                back_input<NT>(w, h, xb, qb);
            }
        }
        if (GRAD) {
            wave_lds_sync();
#pragma unroll
            for (int j = 0; j < MLP_IN; ++j) a_in[j] = xb[lane * XSTRIDE + j];
            wave_lds_sync();
        }
        return val;
    }

    template <bool GRAD, int NT = 2>
    __device__ __forceinline__ static float run(const float* __restrict__ w, int L, float* __restrict__ xb,
                                                const float (&z)[MLP_IN], float (&a_in)[MLP_IN]) {
        const float kappa[1] = {1.f};
        float p[1];
        return run_heads<GRAD, NT, 1>(w, L, xb, z, kappa, p, a_in);
    }

    // ---------------------------------------------------------------- training: forward with stores
    // Leaves activations (unit-major rows, for the weight-gradient GEMM) and ReLU masks (one word
    // per lane and layer) in the workspace; out[c] = raw head outputs of this lane's query.
    template <int OD>
    __device__ __forceinline__ static void forward_store(const float* __restrict__ w, int L, float* __restrict__ xb,
                                                         const float (&z)[MLP_IN], float* __restrict__ hws, size_t Qs,
                                                         size_t q0, unsigned long long* __restrict__ mws, size_t mask_stride,
                                                         float (&out)[OD]) {
        constexpr int NT = 4;
        const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
        put_z(xb, z);
        v4f_t h[MT][NT], acc[MT][NT];
        auto store = [&](int l, unsigned long long mm) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        hws[((size_t)l * H + 16 * mt + 4 * g + r) * Qs + q0 + 16 * nt + n] = h[mt][nt][r];
            mws[(size_t)l * mask_stride + lane] = mm;
        };
        layer0<NT>(w, xb, 0, acc);
        store(0, relu<NT>(acc, h));
        for (int l = 1; l < L; ++l) {
            hidden<NT>(w + OFF_HID + (l - 1) * HID_SZ, h, acc);
            store(l, relu<NT>(acc, h));
        }
        float xo[OD][NT];
        out_layer<NT, OD>(w + off_out(L), h, xo);
#pragma unroll
        for (int c = 0; c < OD; ++c) out[c] = own<NT>(xo[c], 0);
        wave_lds_sync();
    }

    // ------------------------------------------------------------------- training: backward with stores
    // dx[c] = d loss / d head output c of this lane's query.  Layer deltas are written unit-major
    // for the weight-gradient GEMM; dz = d loss / d z of this lane's query.
    template <int OD>
    __device__ __forceinline__ static void backward_store(const float* __restrict__ w, int L, float* __restrict__ xb,
                                                          const float (&dx)[OD], const unsigned long long* __restrict__ mws,
                                                          size_t mask_stride, float* __restrict__ dws, size_t Qs, size_t q0,
                                                          bool store, float (&dz)[MLP_IN]) {
        constexpr int NT = 4;
        const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
#pragma unroll
        for (int c = 0; c < OD; ++c) xb[c * 64 + lane] = dx[c];
        wave_lds_sync();
        float coef[OD][NT];
#pragma unroll
        for (int c = 0; c < OD; ++c)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) coef[c][nt] = xb[c * 64 + 16 * nt + n];
        wave_lds_sync();
        v4f_t h[MT][NT], acc[MT][NT];
        auto put = [&](int l) {
            if (!store) return;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dws[((size_t)l * H + 16 * mt + 4 * g + r) * Qs + q0 + 16 * nt + n] = h[mt][nt][r];
        };
        seed<NT, OD>(w + off_out(L), mws[(size_t)(L - 1) * mask_stride + lane], coef, h);
        put(L - 1);
        for (int l = L - 1; l >= 1; --l) {
            back_hidden<NT>(w + OFF_HID + (l - 1) * HID_SZ, mws[(size_t)(l - 1) * mask_stride + lane], h, acc);
            put(l - 1);
        }
        back_input<NT>(w, h, xb, 0);
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < MLP_IN; ++j) dz[j] = xb[lane * XSTRIDE + j];
        wave_lds_sync();
    }
};

template <int H>
struct MfmaLds {
    static constexpr int W = MfmaDecoder<H>::weight_floats(MLP_MAX_LEVELS);
    static constexpr int TOTAL = W + (MF_BLOCK / 64) * MfmaDecoder<H>::scratch_floats();
};

}  // namespace pin
