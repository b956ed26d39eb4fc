// Shallow-MLP decoder on the fp32 matrix cores (v_mfma_f32_16x16x4_f32) for gfx950.
//
// Decoder.mlp (model/decoder.py:61-80) and its input Jacobian for 64 queries per wave.
// Matrix roles:  D[unit][query] += W[unit][k] * H[k][query]   (M = 16 output units per tile,
// N = 16 queries per tile, K = 4 inputs per instruction).  Why the matrix cores although the
// fp32 MFMA rate equals the vector rate: operand delivery.  On the vector path every FMA needs
// its own wave-uniform weight (scalar-load bound for 53 KB of 4x64 weights) or a broadcast LDS
// read (LDS bound); an MFMA consumes ONE weight register per lane for 16x16x4 = 1024 FMAs,
// so all weights stream from LDS at a few % of its bandwidth.
//
// Layout trick (no data movement between layers): the 16x16x4 result tile puts
//   D[unit = 16*mt + 4*g + r][query = n]  in lane (n = lane & 15, g = lane >> 4), register r,
// and the B operand of the next layer wants  H[k][query = n]  in lane (n, g) for k-slot g.
// The k index of an MFMA is a free permutation as long as A and B agree, so K-step (mt, r) of
// the next layer is declared to cover units {16*mt + 4*g + r, g = 0..3}: the activation
// registers ARE the B operands, and the weights are stored in LDS pre-permuted to match
// (A operand of step (kt, r), lane (i, g):  W[16*mt_out + i][16*kt + 4*g + r]).
// The same array read with a different index serves the transposed product of the Jacobian.
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

// PIN_DECODER=valu selects the thread-per-query vector decoder (A/B runs); default: matrix cores
static inline bool use_mfma_decoder() {
    static const int on = [] {
        const char* e = getenv("PIN_DECODER");
        return (e != nullptr && strcmp(e, "valu") == 0) ? 0 : 1;
    }();
    return on != 0;
}

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
    __host__ __device__ static constexpr int off_out(int L) { return OFF_HID + (L - 1) * HID_SZ; }  // Wo [H], bo, pad
    __host__ __device__ static constexpr int weight_floats(int L) { return off_out(L) + H + 4; }
    __host__ __device__ static constexpr int scratch_floats() { return 64 * XSTRIDE; }  // per wave

    // Block-cooperative: permute the flat state_dict-ordered parameters into the LDS image.
    // Loads are issued in batches of 8 independent requests per thread: a plain load->store loop
    // is one L2 round trip per element (measured: ~35k of a wave's ~50k wait cycles).
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

    __device__ static void stage(const float* __restrict__ dec, int L, float* __restrict__ w, int tid, int nthreads) {
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
        copy_permuted(P, w + off_out(L), H + 1, tid, nthreads, [](int e) { return e; });
    }

    // Forward (+ input Jacobian) for the 64 queries of this wave, NT query tiles (16 queries
    // each) per pass: fewer live accumulators -> 2 waves per SIMD, whose memory phases then
    // overlap the other wave's MFMAs.  z: this lane's query input.  Returns the raw MLP output
    // of this lane's query; a_in = d out / d z if GRAD.
    template <bool GRAD, int NT = 2>
    __device__ __forceinline__ static float run(const float* __restrict__ w, int L, float* __restrict__ xb,
                                                const float (&z)[MLP_IN], float (&a_in)[MLP_IN]) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int j = 0; j < MLP_IN; ++j) xb[lane * XSTRIDE + j] = z[j];
        xb[lane * XSTRIDE + 11] = 0.f;
        wave_lds_sync();
        float out = 0.f;
#pragma unroll 1
        for (int p = 0; p < 4 / NT; ++p) {
            const float o = pass<GRAD, NT>(w, L, xb, 16 * NT * p);
            if ((lane >> 4) / NT == p) out = o;
        }
        if (GRAD) {
            wave_lds_sync();
#pragma unroll
            for (int j = 0; j < MLP_IN; ++j) a_in[j] = xb[lane * XSTRIDE + j];
            wave_lds_sync();
        }
        return out;
    }

    // one pass: queries qb .. qb + 16*NT - 1 of the wave (rows of xb); z rows are consumed before
    // the Jacobian rows of the same queries are written back
    template <bool GRAD, int NT>
    __device__ __forceinline__ static float pass(const float* __restrict__ w, int L, float* __restrict__ xb, int qb) {
        const int lane = threadIdx.x & 63;
        const int n = lane & 15, g = lane >> 4;
        v4f_t h[MT][NT];   // activations: [unit tile][query tile], register r = unit 4g + r
        v4f_t acc[MT][NT];
        unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0;  // ReLU masks per layer, bit (mt*NT+nt)*4+r
        {
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
        auto relu_mask = [&](unsigned long long& m) {
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
            m = mm;
        };
        relu_mask(m0);
        for (int l = 1; l < L; ++l) {
            const float* __restrict__ F = w + OFF_HID + (l - 1) * HID_SZ;
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
            unsigned long long mm;
            relu_mask(mm);
            m1 = l == 1 ? mm : m1; m2 = l == 2 ? mm : m2; m3 = l == 3 ? mm : m3;
        }
        // ---- output layer: partial dot over this lane's units, reduced over the 4 k-groups
        const float* __restrict__ O = w + off_out(L);
        float xo[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) xo[nt] = 0.f;
#pragma unroll
        for (int kt = 0; kt < MT; ++kt) {
            const v4f_t wo = *reinterpret_cast<const v4f_t*>(O + 16 * kt + 4 * g);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) xo[nt] = fmaf(wo[r], h[kt][nt][r], xo[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            xo[nt] += __shfl_xor(xo[nt], 16, 64);
            xo[nt] += __shfl_xor(xo[nt], 32, 64);
        }
        // this lane's own query is row `lane`: inside this pass it is query tile (lane - qb) / 16
        float out = O[H];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) out += (lane - qb == 16 * nt + n) ? xo[nt] : 0.f;
        if (!GRAD) return out;

        // ---- input Jacobian: a = W_out masked, then a <- mask .* (W_l^T a) down the layers
        auto layer_mask = [&](int l) { return l == 0 ? m0 : l == 1 ? m1 : l == 2 ? m2 : m3; };
        {
            const unsigned long long mm = layer_mask(L - 1);
#pragma unroll
            for (int kt = 0; kt < MT; ++kt) {
                const v4f_t wo = *reinterpret_cast<const v4f_t*>(O + 16 * kt + 4 * g);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        h[kt][nt][r] = ((mm >> ((kt * NT + nt) * 4 + r)) & 1ull) ? wo[r] : 0.f;
            }
        }
        for (int l = L - 1; l >= 1; --l) {
            const float* __restrict__ F = w + OFF_HID + (l - 1) * HID_SZ;
            const unsigned long long mm = layer_mask(l - 1);
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
        // layer 0 transposed: a_in[c][q] = sum_i W0[i][c] a0[i][q]
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
        // ---- exchange back: lane (n, g) holds components 4g + r of query qb + 16*nt + n
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * g + r < 12) xb[(qb + 16 * nt + n) * XSTRIDE + 4 * g + r] = ai[nt][r];
        return out;
    }

    // ---- training: forward that leaves activations (unit-major rows, for the weight-gradient
    // GEMM) and ReLU masks (one 64-bit word per lane and layer) in the workspace ----------------
    __device__ __forceinline__ static float forward_store(const float* __restrict__ w, int L, float* __restrict__ xb,
                                                          const float (&z)[MLP_IN], float* __restrict__ hws, size_t Qs,
                                                          size_t q0, unsigned long long* __restrict__ mws,
                                                          size_t mask_stride) {
        const int lane = threadIdx.x & 63;
        const int n = lane & 15, g = lane >> 4;
#pragma unroll
        for (int j = 0; j < MLP_IN; ++j) xb[lane * XSTRIDE + j] = z[j];
        xb[lane * XSTRIDE + 11] = 0.f;
        wave_lds_sync();
        v4f_t h[MT][4];
        v4f_t acc[MT][4];
        {
            float zb[4][3];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int s = 0; s < 3; ++s) zb[nt][s] = xb[(16 * nt + n) * XSTRIDE + 4 * s + g];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const v4f_t b4 = *reinterpret_cast<const v4f_t*>(w + OFF_B0 + 16 * mt + 4 * g);
                float a0[3];
#pragma unroll
                for (int s = 0; s < 3; ++s) a0[s] = w[OFF_A0 + (mt * 3 + s) * 64 + lane];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    v4f_t c = b4;
#pragma unroll
                    for (int s = 0; s < 3; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], zb[nt][s], c, 0, 0, 0);
                    acc[mt][nt] = c;
                }
            }
        }
        auto relu_store = [&](int l) {
            unsigned long long mm = 0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool on = acc[mt][nt][r] > 0.f;
                        mm |= (unsigned long long)on << ((mt * 4 + nt) * 4 + r);
                        const float v = on ? acc[mt][nt][r] : 0.f;
                        h[mt][nt][r] = v;
                        hws[((size_t)l * H + 16 * mt + 4 * g + r) * Qs + q0 + 16 * nt + n] = v;
                    }
            mws[(size_t)l * mask_stride + lane] = mm;
        };
        relu_store(0);
        for (int l = 1; l < L; ++l) {
            const float* __restrict__ F = w + OFF_HID + (l - 1) * HID_SZ;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const v4f_t b4 = *reinterpret_cast<const v4f_t*>(F + H * H + 16 * mt + 4 * g);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = b4;
#pragma unroll
                for (int kt = 0; kt < MT; ++kt) {
                    const v4f_t a4 = *reinterpret_cast<const v4f_t*>(F + ((mt * MT + kt) * 64 + lane) * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[r], h[kt][nt][r], acc[mt][nt], 0, 0, 0);
                }
            }
            relu_store(l);
        }
        const float* __restrict__ O = w + off_out(L);
        float xo[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < MT; ++kt) {
            const v4f_t wo = *reinterpret_cast<const v4f_t*>(O + 16 * kt + 4 * g);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) xo[nt] = fmaf(wo[r], h[kt][nt][r], xo[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            xo[nt] += __shfl_xor(xo[nt], 16, 64);
            xo[nt] += __shfl_xor(xo[nt], 32, 64);
        }
        wave_lds_sync();
        return O[H] + (g == 0 ? xo[0] : g == 1 ? xo[1] : g == 2 ? xo[2] : xo[3]);
    }

    // ---- training: layer deltas from d loss / d out (per lane's own query), written unit-major
    // for the weight-gradient GEMM; returns d loss / d z of this lane's query --------------------
    __device__ __forceinline__ static void backward_store(const float* __restrict__ w, int L, float* __restrict__ xb,
                                                          float dx, const unsigned long long* __restrict__ mws,
                                                          size_t mask_stride, float* __restrict__ dws, size_t Qs,
                                                          size_t q0, bool store, float (&dz)[MLP_IN]) {
        const int lane = threadIdx.x & 63;
        const int n = lane & 15, g = lane >> 4;
        xb[lane] = dx;
        wave_lds_sync();
        float dxq[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) dxq[nt] = xb[16 * nt + n];
        wave_lds_sync();
        const float* __restrict__ O = w + off_out(L);
        v4f_t h[MT][4];
        v4f_t acc[MT][4];
        auto put = [&](int l) {
            if (!store) return;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dws[((size_t)l * H + 16 * mt + 4 * g + r) * Qs + q0 + 16 * nt + n] = h[mt][nt][r];
        };
        {
            const unsigned long long mm = mws[(size_t)(L - 1) * mask_stride + lane];
#pragma unroll
            for (int kt = 0; kt < MT; ++kt) {
                const v4f_t wo = *reinterpret_cast<const v4f_t*>(O + 16 * kt + 4 * g);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        h[kt][nt][r] = ((mm >> ((kt * 4 + nt) * 4 + r)) & 1ull) ? wo[r] * dxq[nt] : 0.f;
            }
            put(L - 1);
        }
        for (int l = L - 1; l >= 1; --l) {
            const float* __restrict__ F = w + OFF_HID + (l - 1) * HID_SZ;
            const unsigned long long mm = mws[(size_t)(l - 1) * mask_stride + lane];
#pragma unroll
            for (int mj = 0; mj < MT; ++mj) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mj][nt] = (v4f_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ki = 0; ki < MT; ++ki)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float at = F[((ki * MT + mj) * 64 + 16 * (n >> 2) + 4 * g + r) * 4 + (n & 3)];
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[mj][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, h[ki][nt][r], acc[mj][nt], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int mj = 0; mj < MT; ++mj)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        h[mj][nt][r] = ((mm >> ((mj * 4 + nt) * 4 + r)) & 1ull) ? acc[mj][nt][r] : 0.f;
            put(l - 1);
        }
        v4f_t ai[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) ai[nt] = (v4f_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < MT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float at = w[OFF_A0T + (kt * 4 + r) * 64 + lane];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    ai[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, h[kt][nt][r], ai[nt], 0, 0, 0);
            }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * g + r < 12) xb[(16 * nt + n) * XSTRIDE + 4 * g + r] = ai[nt][r];
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
