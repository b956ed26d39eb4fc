// The quad-layout decoder (mlp_quad.h) on the fp16 matrix cores with fp32-equivalent products.
//
// Why: on gfx950 the fp32 MFMA (16x16x4, 32 cycles) runs at the vector rate AND blocks the vector pipe
// of its SIMD while it executes (scripts/mfma_valu_overlap.hip: MFMA time + VALU time add up, even across
// waves), so the decoder pays 2048 cycles per 64x64 layer and tile plus all its vector work.  The 16-bit
// MFMA (16x16x32, 16 cycles) is 16x faster per flop.  Every fp32 factor is split into TWO fp16 pieces,
//     x = hi + 2^-11 lo',   hi = rne16(x),   lo' = rne16((x - hi) * 2^11),
// x - hi is exact in fp32, |x - hi| <= 2^-12 |x|, so the representation error is <= 2^-24 |x|: that of fp32
// itself.  The residual is carried scaled by 2^11 so that it is a normal fp16 number whenever hi is.  A product
// a*b is taken as hi*hi (main accumulator) + 2^-11 (hi*lo' + lo'*hi) (cross accumulator, folded in with one FMA
// per output); the dropped lo*lo term is <= 2^-24 |a b|.  Products of fp16 pieces are exact and accumulation is
// fp32 inside the MFMA.  3 MFMAs and ~3 vector instructions per split value replace the 6 MFMAs and ~5.5
// instructions of the three-piece bf16 scheme this file supersedes (round 1-2a), at the same accuracy
// (tests/test_gpu_variants.py compares with the fp32 MFMA image; scripts/decoder_bench.hip with a double reference).
//
// Range: fp16 holds |x| < 65504; the weights and activations of the SDF decoder are O(1..100) (a larger value
// would become inf and show up as NaN outputs, never as a silently wrong number).  Values below the fp16 normal
// range keep an ABSOLUTE error <= 2^-36 (hi subnormal, residual still scaled), and the 16-bit MFMA does not flush
// subnormal inputs (scripts/decoder_bench.hip runs the decoder with activations ~1e-5 and ~1e-7).
// The ReLU pattern of a layer for the transposed sweep is exact: it is read from the upper halves of the fp32
// activations (any positive normal float has a non-zero upper half), not from the rounded fp16 pieces.
//
// Layout: results keep the 16x16 D layout of mlp_quad.h (lane (n, g), register r of tile mt holds unit
// 16 mt + 4 g + r of query n).  The 16x16x32 B operand wants 8 consecutive k of column n in lane (n, g):
// K-block j takes the lane's units of tiles 2j and 2j+1, i.e. k slot (g, i) <-> unit
// u(j, g, i) = 16 (2j + i/4) + 4g + i%4, and the weights are staged pre-permuted to match, once per
// block and per direction (forward image W[out][u], transposed image W[u][in]) as packed fp16 pieces.
#pragma once
#include "mlp_quad.h"

namespace pin {

typedef _Float16 v8h_t __attribute__((ext_vector_type(8)));
typedef _Float16 v4h_t __attribute__((ext_vector_type(4)));
typedef _Float16 v2h_t __attribute__((ext_vector_type(2)));
typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
typedef unsigned int v2u_t __attribute__((ext_vector_type(2)));

constexpr float H2_UP = 2048.f, H2_DOWN = 1.f / 2048.f;

// the two fp16 pieces of two floats, packed (first value in the low half): v_cvt_pk_f16_f32, 2 x v_cvt_f32_f16,
// v_pk_add_f32, v_pk_mul_f32, v_cvt_pk_f16_f32
__device__ __forceinline__ void h2_split2(float x0, float x1, unsigned int& h, unsigned int& l) {
    const v2h_t hh = {(_Float16)x0, (_Float16)x1};
    const float r0 = (x0 - (float)hh[0]) * H2_UP, r1 = (x1 - (float)hh[1]) * H2_UP;
    const v2h_t ll = {(_Float16)r0, (_Float16)r1};
    h = __builtin_bit_cast(unsigned int, hh);
    l = __builtin_bit_cast(unsigned int, ll);
}
// upper halves of two floats, packed (first value in the low half): zero iff the (non-negative) value is zero
__device__ __forceinline__ unsigned int top_pack(float lo_half, float hi_half) {
    return __builtin_amdgcn_perm(__float_as_uint(hi_half), __float_as_uint(lo_half), 0x07060302u);
}
// max(x, 0) as ONE instruction: fmaxf() first canonicalises its operand (a second v_max_f32) because the compiler
// cannot know that an MFMA result is never a signalling NaN; the signed-integer maximum of the bit pattern with 0 is
// the same function on floats (a negative float is a negative integer) and needs no such step.  (Inline assembly is
// not an option: the hazard recogniser does not see an MFMA -> VALU dependency through it.)
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
__device__ __forceinline__ v8h_t as_h8(v4u_t v) { return __builtin_bit_cast(v8h_t, v); }
__device__ __forceinline__ v4h_t as_h4(v2u_t v) { return __builtin_bit_cast(v4h_t, v); }

template <int H>
struct QuadDecoderH {
    static_assert(H % 32 == 0, "the 16-bit K-block is 32 units");
    static constexpr int MT = H / 16, NJ = H / 32, NP = 2;
    // LDS image, byte offsets (L = number of H-wide layers, 1 <= L <= MLP_MAX_LEVELS)
    static constexpr int HID_PIECE = MT * NJ * 64 * 16;  // one piece of one hidden layer, one direction
    static constexpr int HID_DIR = NP * HID_PIECE;
    __host__ __device__ static constexpr int off_hidf(int, int l) { return (l - 1) * HID_DIR; }  // [p][mt][j][lane] 16 B
    __host__ __device__ static constexpr int off_hidb(int L, int l) { return (L - 1 + l - 1) * HID_DIR; }
    __host__ __device__ static constexpr int off_l0f(int L) { return 2 * (L - 1) * HID_DIR; }      // [p][mt][lane] 8 B
    __host__ __device__ static constexpr int off_l0b(int L) { return off_l0f(L) + NP * MT * 64 * 8; }  // [p][j][lane] 16 B
    __host__ __device__ static constexpr int off_bias(int L) { return off_l0b(L) + NP * NJ * 64 * 16; }  // [L][H] f32
    __host__ __device__ static constexpr int off_out(int L) { return off_bias(L) + L * H * 4; }  // Wo [3][H], bo [3], pad
    __host__ __device__ static constexpr int bytes(int L) { return off_out(L) + (MF_OD_MAX * H + 4) * 4; }

    __device__ __forceinline__ static int unit_of(int j, int g, int i) { return 16 * (2 * j + (i >> 2)) + 4 * g + (i & 3); }

    // ------------------------------------------------------------------------------------ staging
    // dec: flat state_dict order (W0 [H][11], b0, hidden (W [H][H], b)*, lout.weight [OD][H], lout.bias [OD])
    // `part` selects one of the four independent pieces of work (0 hidden layers, 1 layer 0 forward, 2 layer 0 transposed,
    // 3 biases and output head) so that a multi-block launch can run them side by side; -1 = everything
    __device__ static void stage(const float* __restrict__ dec, int L, unsigned char* __restrict__ w, int tid, int nthreads,
                                 int OD = 1, int part = -1) {
        // hidden layers, both directions: one 16-byte slot = 8 k-values of one lane, two pieces.  One flat
        // loop over (layer, direction, tile, K-block, lane), four slots per thread and trip so that 32 loads are
        // in flight per thread: the staging is a chain of memory round trips, not arithmetic.
        constexpr int SLOTS = 2 * MT * NJ * 64;  // per layer
        const float* const P1 = dec + H * MLP_IN + H;
        const int n_slots = (L - 1) * SLOTS;
        auto fetch = [&](int e, float (&x)[8]) {
            const int l1 = e / SLOTS, s = e % SLOTS;
            const int lane = s & 63, j = (s >> 6) % NJ, mt = (s / (64 * NJ)) % MT, dir = s / (64 * NJ * MT);
            const int row = 16 * mt + (lane & 15), g = lane >> 4;
            const float* __restrict__ P = P1 + (size_t)l1 * (H * H + H);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int u = unit_of(j, g, i);
                x[i] = dir == 0 ? P[row * H + u] : P[u * H + row];
            }
        };
        auto emit = [&](int e, const float (&x)[8]) {
            const int l1 = e / SLOTS, s = e % SLOTS;
            const int dir = s / (64 * NJ * MT), rest = s % (64 * NJ * MT);
            v4u_t ph, pl;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                unsigned int a, b;
                h2_split2(x[2 * p], x[2 * p + 1], a, b);
                ph[p] = a; pl[p] = b;
            }
            unsigned char* base = w + (dir == 0 ? off_hidf(L, l1 + 1) : off_hidb(L, l1 + 1)) + rest * 16;
            *reinterpret_cast<v4u_t*>(base) = ph;
            *reinterpret_cast<v4u_t*>(base + HID_PIECE) = pl;
        };
        for (int e0 = tid; e0 < ((part < 0 || part == 0) ? n_slots : 0); e0 += 4 * nthreads) {
            float x[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) fetch(e0 + u * nthreads < n_slots ? e0 + u * nthreads : e0, x[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e0 + u * nthreads < n_slots) emit(e0 + u * nthreads, x[u]);
        }
        for (int e = tid; e < ((part < 0 || part == 3) ? (L - 1) * H : 0); e += nthreads)
            reinterpret_cast<float*>(w + off_bias(L))[H + e] = P1[(size_t)(e / H) * (H * H + H) + H * H + e % H];
        const float* P = P1 + (size_t)(L - 1) * (H * H + H);
        // layer 0 forward: lane (m, g) holds W0[16 mt + m][4g + i], i = 0..3 (zero beyond the 11 inputs)
        for (int e = tid; e < ((part < 0 || part == 1) ? MT * 64 : 0); e += nthreads) {
            const int lane = e & 63, mt = e >> 6, m = lane & 15, g = lane >> 4;
            float x[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = 4 * g + i < MLP_IN ? dec[(16 * mt + m) * MLP_IN + 4 * g + i] : 0.f;
            v2u_t ph, pl;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                unsigned int a, b;
                h2_split2(x[2 * p], x[2 * p + 1], a, b);
                ph[p] = a; pl[p] = b;
            }
            unsigned char* base = w + off_l0f(L) + e * 8;
            *reinterpret_cast<v2u_t*>(base) = ph;
            *reinterpret_cast<v2u_t*>(base + MT * 64 * 8) = pl;
        }
        // layer 0 transposed: lane (c, g) holds W0[u(j, g, i)][c]
        for (int e = tid; e < ((part < 0 || part == 2) ? NJ * 64 : 0); e += nthreads) {
            const int lane = e & 63, j = e >> 6, c = lane & 15, g = lane >> 4;
            v4u_t ph, pl;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float x0 = c < MLP_IN ? dec[unit_of(j, g, 2 * p) * MLP_IN + c] : 0.f;
                const float x1 = c < MLP_IN ? dec[unit_of(j, g, 2 * p + 1) * MLP_IN + c] : 0.f;
                unsigned int a, b;
                h2_split2(x0, x1, a, b);
                ph[p] = a; pl[p] = b;
            }
            unsigned char* base = w + off_l0b(L) + e * 16;
            *reinterpret_cast<v4u_t*>(base) = ph;
            *reinterpret_cast<v4u_t*>(base + NJ * 64 * 16) = pl;
        }
        if (part >= 0 && part != 3) return;
        for (int e = tid; e < H; e += nthreads) reinterpret_cast<float*>(w + off_bias(L))[e] = dec[H * MLP_IN + e];
        // lout.weight [OD][H] then lout.bias [OD]  ->  Wo at O[c*H + u], bias at O[3H + c]
        float* O = reinterpret_cast<float*>(w + off_out(L));
        for (int e = tid; e < OD * H; e += nthreads) O[e] = P[e];
        for (int e = tid; e < OD; e += nthreads) O[MF_OD_MAX * H + e] = P[OD * H + e];
    }

    // The image entries of ONE parameter (flat state_dict index e, new value x): what stage() writes for it, parameter-major.
    // The optimiser's decoder step calls this for every parameter it updates (train.hip: pin_adam_dense.image), so an image
    // staged once stays current without a staging launch per training iteration.  Padding entries never change.
    __device__ __forceinline__ static void stage_param(int e, float x, int L, int OD, unsigned char* __restrict__ w) {
        unsigned int ph, pl;
        h2_split2(x, 0.f, ph, pl);
        const unsigned short hi = (unsigned short)(ph & 0xffffu), lo = (unsigned short)(pl & 0xffffu);
        auto put = [&](unsigned char* at, int piece_stride) {
            *reinterpret_cast<unsigned short*>(at) = hi;
            *reinterpret_cast<unsigned short*>(at + piece_stride) = lo;
        };
        if (e < H * MLP_IN) {  // W0[r][c]: forward image lane (r % 16, c / 4) of tile r / 16, transposed image lane (c, g(r))
            const int r = e / MLP_IN, c = e % MLP_IN;
            put(w + off_l0f(L) + ((r >> 4) * 64 + (r & 15) + 16 * (c >> 2)) * 8 + 2 * (c & 3), MT * 64 * 8);
            const int t = r >> 4;
            put(w + off_l0b(L) + ((t >> 1) * 64 + c + 16 * ((r >> 2) & 3)) * 16 + 2 * (4 * (t & 1) + (r & 3)), NJ * 64 * 16);
            return;
        }
        e -= H * MLP_IN;
        if (e < H) { reinterpret_cast<float*>(w + off_bias(L))[e] = x; return; }
        e -= H;
        for (int l1 = 0; l1 < L - 1; ++l1) {
            if (e < H * H) {  // hidden W[r][c]: forward image (row r, k slot of unit c), transposed image (row c, k slot of unit r)
                const int r = e / H, c = e % H;
                const int tc = c >> 4, tr = r >> 4;
                const int sf = (((r >> 4) * NJ) + (tc >> 1)) * 64 + (r & 15) + 16 * ((c >> 2) & 3);
                put(w + off_hidf(L, l1 + 1) + sf * 16 + 2 * (4 * (tc & 1) + (c & 3)), HID_PIECE);
                const int sb = (((c >> 4) * NJ) + (tr >> 1)) * 64 + (c & 15) + 16 * ((r >> 2) & 3);
                put(w + off_hidb(L, l1 + 1) + sb * 16 + 2 * (4 * (tr & 1) + (r & 3)), HID_PIECE);
                return;
            }
            e -= H * H;
            if (e < H) { reinterpret_cast<float*>(w + off_bias(L))[H * (l1 + 1) + e] = x; return; }
            e -= H;
        }
        float* O = reinterpret_cast<float*>(w + off_out(L));
        if (e < OD * H) O[e] = x;
        else if (e < OD * H + OD) O[MF_OD_MAX * H + (e - OD * H)] = x;
    }

    // ------------------------------------------------------------------------------ building blocks
    // ReLU pattern words of a layer's post-ReLU activations, in the word order of split_acts
    __device__ __forceinline__ static void pattern_of(const v4f_t (&h)[MT], v4u_t (&sg)[NJ]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int mt = 2 * j + (p >> 1), r = 2 * (p & 1);
                sg[j][p] = top_pack(h[mt][r], h[mt][r + 1]);
            }
    }

    // B operands of a layer: the lane's 16 activations as NJ x 2 pieces of 8 packed fp16
    __device__ __forceinline__ static void split_acts(const v4f_t (&h)[MT], v4u_t (&bh)[NJ], v4u_t (&bl)[NJ]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int mt = 2 * j + (p >> 1), r = 2 * (p & 1);
                unsigned int a, b;
                h2_split2(h[mt][r], h[mt][r + 1], a, b);
                bh[j][p] = a; bl[j][p] = b;
            }
    }

    // acc[mt] += W h over one staged direction image: hi*hi into acc, the two cross products into a second
    // accumulator that is folded in with its 2^-11 at the end (four independent MFMA chains per tile pair)
    __device__ __forceinline__ static void matmul(const unsigned char* __restrict__ img, const v4u_t (&bh)[NJ],
                                                  const v4u_t (&bl)[NJ], v4f_t (&acc)[MT]) {
        const int lane = threadIdx.x & 63;
        const v4u_t* __restrict__ A = reinterpret_cast<const v4u_t*>(img) + lane;
        v4f_t cr[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) cr[mt] = (v4f_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int m0 = 0; m0 < MT; m0 += 2) {
                v8h_t ah[2], al[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int slot = ((m0 + q) * NJ + j) * 64;
                    ah[q] = as_h8(A[slot]);
                    al[q] = as_h8(A[slot + HID_PIECE / 16]);
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) cr[m0 + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[q], as_h8(bh[j]), cr[m0 + q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[m0 + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q], as_h8(bh[j]), acc[m0 + q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 2; ++q) cr[m0 + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q], as_h8(bl[j]), cr[m0 + q], 0, 0, 0);
            }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][r] = fmaf(cr[mt][r], H2_DOWN, acc[mt][r]);
    }

    __device__ __forceinline__ static void load_bias(const unsigned char* __restrict__ w, int L, int l, v4f_t (&acc)[MT]) {
        const int g = (threadIdx.x & 63) >> 4;
        const float* __restrict__ B = reinterpret_cast<const float*>(w + off_bias(L)) + l * H;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = *reinterpret_cast<const v4f_t*>(B + 16 * mt + 4 * g);
    }

    // layer 0: acc = b0 + W0 z, z[r] = input component 4g + r of this lane's query (K = 16 instruction); the
    // second form takes the two packed pieces of z (the training kernel keeps them for the weight gradient)
    __device__ __forceinline__ static void split_input(const float (&z)[4], v2u_t& zh, v2u_t& zl) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            unsigned int a, b;
            h2_split2(z[2 * p], z[2 * p + 1], a, b);
            zh[p] = a; zl[p] = b;
        }
    }
    __device__ __forceinline__ static void layer0(const unsigned char* __restrict__ w, int L, v2u_t zh, v2u_t zl, v4f_t (&acc)[MT]) {
        load_bias(w, L, 0, acc);
        layer0_add(w, L, zh, zl, acc);
    }
    // acc += W0 z (no bias: the derivative network of the analytic Eikonal term, train_fused.h)
    __device__ __forceinline__ static void layer0_add(const unsigned char* __restrict__ w, int L, v2u_t zh, v2u_t zl, v4f_t (&acc)[MT]) {
        const int lane = threadIdx.x & 63;
        const v2u_t* __restrict__ A = reinterpret_cast<const v2u_t*>(w + off_l0f(L)) + lane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const v4h_t ah = as_h4(A[mt * 64]), al = as_h4(A[(MT + mt) * 64]);
            v4f_t c = (v4f_t){0.f, 0.f, 0.f, 0.f};
            c = __builtin_amdgcn_mfma_f32_16x16x16f16(al, as_h4(zh), c, 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, as_h4(zh), acc[mt], 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, as_h4(zl), c, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][r] = fmaf(c[r], H2_DOWN, acc[mt][r]);
        }
    }
    __device__ __forceinline__ static void layer0(const unsigned char* __restrict__ w, int L, const float (&z)[4], v4f_t (&acc)[MT]) {
        v2u_t zh, zl;
        split_input(z, zh, zl);
        layer0(w, L, zh, zl, acc);
    }

    // transposed layer 0: a[r] = sum_u W0[u][4g + r] h[u]; the first form takes the pieces of h
    __device__ __forceinline__ static void input_backward(const unsigned char* __restrict__ w, int L, const v4u_t (&bh)[NJ],
                                                          const v4u_t (&bl)[NJ], float (&a)[4]) {
        const int lane = threadIdx.x & 63;
        const v4u_t* __restrict__ A = reinterpret_cast<const v4u_t*>(w + off_l0b(L)) + lane;
        v4f_t c[NJ], x[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const v8h_t ah = as_h8(A[j * 64]), al = as_h8(A[(NJ + j) * 64]);
            v4f_t t = (v4f_t){0.f, 0.f, 0.f, 0.f}, u = (v4f_t){0.f, 0.f, 0.f, 0.f};
            u = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, as_h8(bh[j]), u, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, as_h8(bh[j]), t, 0, 0, 0);
            u = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, as_h8(bl[j]), u, 0, 0, 0);
            c[j] = t; x[j] = u;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = c[0][r], sx = x[0][r];
#pragma unroll
            for (int j = 1; j < NJ; ++j) { s += c[j][r]; sx += x[j][r]; }
            a[r] = fmaf(sx, H2_DOWN, s);
        }
    }
    __device__ __forceinline__ static void input_backward(const unsigned char* __restrict__ w, int L, const v4f_t (&h)[MT],
                                                          float (&a)[4]) {
        v4u_t bh[NJ], bl[NJ];
        split_acts(h, bh, bl);
        input_backward(w, L, bh, bl, a);
    }

    // ReLU pattern of a layer, read back from the pattern words of its (post-ReLU, hence >= 0) activations:
    // element (mj, r) sits in word [mj / 2][2 (mj % 2) + r / 2], half r % 2 (pattern_of).  h <- pattern .* acc
    __device__ __forceinline__ static void mask_by_pieces(const v4u_t (&sg)[NJ], const v4f_t (&acc)[MT], v4f_t (&h)[MT]) {
#pragma unroll
        for (int mj = 0; mj < MT; ++mj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned int word = sg[mj >> 1][2 * (mj & 1) + (r >> 1)];
                const bool on = (r & 1) ? (word > 0xffffu) : ((word & 0xffffu) != 0u);
                h[mj][r] = on ? acc[mj][r] : 0.f;
            }
    }

    // forward + input Jacobian of one 16-query tile (same contract as QuadDecoder<H>::run) with the number of layers
    // known at compile time: both sweeps fully unrolled; the ReLU patterns are 8 words per layer and lane (pattern_of)
    template <int L>
    __device__ __forceinline__ static float run(const unsigned char* __restrict__ w, const float (&z)[4], float (&a)[4]) {
        static_assert(L >= 1 && L <= MLP_MAX_LEVELS, "1..4 layers");
        const int g = (threadIdx.x & 63) >> 4;
        v4f_t h[MT], acc[MT];
        v4u_t sg[L > 1 ? L - 1 : 1][NJ];
        layer0(w, L, z, acc);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[mt][r] = relu1(acc[mt][r]);
#pragma unroll
        for (int l = 1; l < L; ++l) {
            v4u_t bh[NJ], bl[NJ];
            pattern_of(h, sg[l - 1]);
            split_acts(h, bh, bl);
            load_bias(w, L, l, acc);
            matmul(w + off_hidf(L, l), bh, bl, acc);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) h[mt][r] = relu1(acc[mt][r]);
        }
        // output head, and the seed of the transposed sweep: the output weights under the last ReLU pattern
        const float* __restrict__ O = reinterpret_cast<const float*>(w + off_out(L));
        float x = 0.f;
#pragma unroll
        for (int kt = 0; kt < MT; ++kt) {
            const v4f_t wo = *reinterpret_cast<const v4f_t*>(O + 16 * kt + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x = fmaf(wo[r], h[kt][r], x);
                h[kt][r] = h[kt][r] > 0.f ? wo[r] : 0.f;
            }
        }
        x = rows_sum_lds(x);
        x += O[MF_OD_MAX * H];
#pragma unroll
        for (int l = L - 1; l >= 1; --l) {
            v4u_t bh[NJ], bl[NJ];
            split_acts(h, bh, bl);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4f_t){0.f, 0.f, 0.f, 0.f};
            matmul(w + off_hidb(L, l), bh, bl, acc);
            mask_by_pieces(sg[l - 1], acc, h);
        }
        input_backward(w, L, h, a);
        return x;
    }

    // forward pass only (inference: Mesher.query_points, Tracker.query_source_points without the gradient): the value of
    // run<L>, bit for bit -- the same products in the same order -- without the ReLU patterns and the transposed sweep.
    // OD heads (1: sdf; 3: the colour heads before their sigmoid), complete in the four lanes of the query.
    template <int L, int OD = 1>
    __device__ __forceinline__ static void forward(const unsigned char* __restrict__ w, const float (&z)[4], float (&x)[OD]) {
        static_assert(L >= 1 && L <= MLP_MAX_LEVELS, "1..4 layers");
        const int g = (threadIdx.x & 63) >> 4;
        v4f_t h[MT], acc[MT];
        layer0(w, L, z, acc);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[mt][r] = relu1(acc[mt][r]);
#pragma unroll
        for (int l = 1; l < L; ++l) {
            v4u_t bh[NJ], bl[NJ];
            split_acts(h, bh, bl);
            load_bias(w, L, l, acc);
            matmul(w + off_hidf(L, l), bh, bl, acc);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) h[mt][r] = relu1(acc[mt][r]);
        }
        const float* __restrict__ O = reinterpret_cast<const float*>(w + off_out(L));
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
            x[c] = o + O[MF_OD_MAX * H + c];
        }
    }

    // The colour decoder (3 sigmoid heads, Decoder.regress_color, model/decoder.py:112) on the same tile layout:
    // returns value = sum_c kappa[c] sigmoid(head_c) (complete in the four lanes of the query) and, when `grad`,
    // a[r] = d value / d z[4g + r]: ONE transposed sweep seeded with sum_c kappa[c] s_c (1 - s_c) wo_c under the last
    // ReLU pattern.  The image is staged with OD = 3 (stage(dec, L, w, tid, nthreads, 3)).
    template <int L>
    __device__ __forceinline__ static float run_color(const unsigned char* __restrict__ w, const float (&z)[4],
                                                      const float (&kappa)[3], bool grad, float (&a)[4],
                                                      float (*heads)[3] = nullptr) {  // heads: the three sigmoid outputs, if wanted
        static_assert(L >= 1 && L <= MLP_MAX_LEVELS, "1..4 layers");
        const int g = (threadIdx.x & 63) >> 4;
        v4f_t h[MT], acc[MT];
        v4u_t sg[L > 1 ? L - 1 : 1][NJ];
        layer0(w, L, z, acc);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[mt][r] = relu1(acc[mt][r]);
#pragma unroll
        for (int l = 1; l < L; ++l) {
            v4u_t bh[NJ], bl[NJ];
            pattern_of(h, sg[l - 1]);
            split_acts(h, bh, bl);
            load_bias(w, L, l, acc);
            matmul(w + off_hidf(L, l), bh, bl, acc);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) h[mt][r] = relu1(acc[mt][r]);
        }
        const float* __restrict__ O = reinterpret_cast<const float*>(w + off_out(L));
        float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int kt = 0; kt < MT; ++kt) {
                const v4f_t wo = *reinterpret_cast<const v4f_t*>(O + c * H + 16 * kt + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[c] = fmaf(wo[r], h[kt][r], o[c]);
            }
            o[c] = rows_sum(o[c]);
            o[c] += O[MF_OD_MAX * H + c];
        }
        float value = 0.f, coef[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float sgm = 1.f / (1.f + expf(-o[c]));
            value = fmaf(kappa[c], sgm, value);
            coef[c] = kappa[c] * sgm * (1.f - sgm);
            if (heads != nullptr) (*heads)[c] = sgm;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = 0.f;
        if (!grad) return value;  // (wave-uniform)
#pragma unroll
        for (int kt = 0; kt < MT; ++kt) {
            v4f_t sd = (v4f_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const v4f_t wo = *reinterpret_cast<const v4f_t*>(O + c * H + 16 * kt + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) sd[r] = fmaf(coef[c], wo[r], sd[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) h[kt][r] = h[kt][r] > 0.f ? sd[r] : 0.f;
        }
#pragma unroll
        for (int l = L - 1; l >= 1; --l) {
            v4u_t bh[NJ], bl[NJ];
            split_acts(h, bh, bl);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4f_t){0.f, 0.f, 0.f, 0.f};
            matmul(w + off_hidb(L, l), bh, bl, acc);
            mask_by_pieces(sg[l - 1], acc, h);
        }
        input_backward(w, L, h, a);
        return value;
    }
};

// One interface over the two decoder images, for kernels templated on the arithmetic: SPLIT = false is the fp32
// MFMA image of mlp_quad.h, SPLIT = true the split-fp16 one.  `smem` is the block's dynamic LDS.
template <int H, bool SPLIT>
struct QuadDec;
template <int H>
struct QuadDec<H, false> {
    using Q = QuadDecoder<H>;
    __host__ __device__ static constexpr int bytes(int) { return Q::TOTAL * (int)sizeof(float); }
    __device__ __forceinline__ static void stage(const float* __restrict__ dec, int L, unsigned char* smem, int tid, int nthreads) {
        Q::stage(dec, L, reinterpret_cast<float*>(smem), tid, nthreads);
    }
    template <int LC>  // (LC unused: the fp32 image takes the layer count at run time)
    __device__ __forceinline__ static float run(const unsigned char* smem, int L, const float (&z)[4], float (&a)[4]) {
        return Q::run(reinterpret_cast<const float*>(smem), L, z, a);
    }
};
template <int H>
struct QuadDec<H, true> {
    using Q = QuadDecoderH<H>;
    __host__ __device__ static constexpr int bytes(int L) { return Q::bytes(L); }
    __device__ __forceinline__ static void stage(const float* __restrict__ dec, int L, unsigned char* smem, int tid, int nthreads) {
        Q::stage(dec, L, smem, tid, nthreads);
    }
    template <int LC>
    __device__ __forceinline__ static float run(const unsigned char* smem, int, const float (&z)[4], float (&a)[4]) {
        return Q::template run<LC>(smem, z, a);
    }
};

// PIN_MLP=f32 keeps the fp32 MFMA decoder (A/B runs); default: split fp16
static inline bool use_split_decoder() {
    static const int on = [] {
        const char* e = getenv("PIN_MLP");
        return (e != nullptr && strcmp(e, "f32") == 0) ? 0 : 1;
    }();
    return on != 0;
}

}  // namespace pin
