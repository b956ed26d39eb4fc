// The quad-layout decoder (mlp_quad.h) on the bf16 matrix cores with fp32-equivalent products.
//
// Why: on gfx950 the fp32 MFMA (16x16x4, 32 cycles) runs at the vector rate AND blocks the vector pipe
// of its SIMD while it executes (scripts/mfma_valu_overlap.hip: MFMA time + VALU time add up, even across
// waves), so the decoder pays 2048 cycles per 64x64 layer and tile plus all its vector work.  The bf16
// MFMA (16x16x32, 16 cycles) is 16x faster per flop.  Every fp32 factor is split EXACTLY into three
// bf16 pieces (x = hi + mid + lo, 8 + 8 + 8 mantissa bits, by truncation), and a product a*b is taken as
// the six piece products of weight >= 2^-16 relative (hh, hm, mh, hl, lh, mm); the dropped ones are
// below 2^-24, i.e. below fp32 rounding, and accumulation stays fp32 inside the MFMA.  Measured against
// a double reference the 64x64 layer is MORE accurate than the fp32 FMA chain (4e-7 vs 1.1e-6 abs at
// |y| ~ 3).  48 bf16 MFMAs (~820 cycles) + ~90 vector instructions for the split replace 64 fp32 MFMAs
// (2048 cycles) per layer and tile.
//
// Layout: results keep the 16x16 D layout of mlp_quad.h (lane (n, g), register r of tile mt holds unit
// 16 mt + 4 g + r of query n).  The 16x16x32 B operand wants 8 consecutive k of column n in lane (n, g):
// K-block j takes the lane's units of tiles 2j and 2j+1, i.e. k slot (g, i) <-> unit
// u(j, g, i) = 16 (2j + i/4) + 4g + i%4, and the weights are staged pre-permuted to match, once per
// block and per direction (forward image W[out][u], transposed image W[u][in]) as packed bf16 pieces.
#pragma once
#include "mlp_quad.h"

namespace pin {

typedef __bf16 v8bf_t __attribute__((ext_vector_type(8)));
typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
typedef unsigned int v2u_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf_top(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
__device__ __forceinline__ unsigned int bf_pack(float lo_half, float hi_half) {  // upper halves of two floats
    return __builtin_amdgcn_perm(__float_as_uint(hi_half), __float_as_uint(lo_half), 0x07060302u);
}
// three exact bf16 pieces of two floats, packed (first value in the low half)
__device__ __forceinline__ void bf_split2(float x0, float x1, unsigned int& h, unsigned int& m, unsigned int& l) {
    h = bf_pack(x0, x1);
    const float r0 = x0 - bf_top(x0), r1 = x1 - bf_top(x1);
    m = bf_pack(r0, r1);
    const float s0 = r0 - bf_top(r0), s1 = r1 - bf_top(r1);
    l = bf_pack(s0, s1);
}
// max(x, 0) as ONE instruction: fmaxf() first canonicalises its operand (a second v_max_f32) because the compiler
// cannot know that an MFMA result is never a signalling NaN; the signed-integer maximum of the bit pattern with 0 is
// the same function on floats (a negative float is a negative integer) and needs no such step.  (Inline assembly is
// not an option: the hazard recogniser does not see an MFMA -> VALU dependency through it.)
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
__device__ __forceinline__ v8bf_t as_bf8(v4u_t v) { return __builtin_bit_cast(v8bf_t, v); }
__device__ __forceinline__ v4s_t as_s4(v2u_t v) { return __builtin_bit_cast(v4s_t, v); }

template <int H>
struct QuadDecoderB {
    static_assert(H % 32 == 0, "the bf16 K-block is 32 units");
    static constexpr int MT = H / 16, NJ = H / 32;
    // LDS image, byte offsets (L = number of H-wide layers, 1 <= L <= MLP_MAX_LEVELS)
    static constexpr int HID_PIECE = MT * NJ * 64 * 16;  // one piece of one hidden layer, one direction
    static constexpr int HID_DIR = 3 * HID_PIECE;
    __host__ __device__ static constexpr int off_hidf(int, int l) { return (l - 1) * HID_DIR; }  // [p][mt][j][lane] 16 B
    __host__ __device__ static constexpr int off_hidb(int L, int l) { return (L - 1 + l - 1) * HID_DIR; }
    __host__ __device__ static constexpr int off_l0f(int L) { return 2 * (L - 1) * HID_DIR; }      // [p][mt][lane] 8 B
    __host__ __device__ static constexpr int off_l0b(int L) { return off_l0f(L) + 3 * MT * 64 * 8; }  // [p][j][lane] 16 B
    __host__ __device__ static constexpr int off_bias(int L) { return off_l0b(L) + 3 * NJ * 64 * 16; }  // [L][H] f32
    __host__ __device__ static constexpr int off_out(int L) { return off_bias(L) + L * H * 4; }  // Wo [3][H], bo [3], pad
    __host__ __device__ static constexpr int bytes(int L) { return off_out(L) + (MF_OD_MAX * H + 4) * 4; }

    __device__ __forceinline__ static int unit_of(int j, int g, int i) { return 16 * (2 * j + (i >> 2)) + 4 * g + (i & 3); }

    // ------------------------------------------------------------------------------------ staging
    // dec: flat state_dict order (W0 [H][11], b0, hidden (W [H][H], b)*, lout.weight [OD][H], lout.bias [OD])
    __device__ static void stage(const float* __restrict__ dec, int L, unsigned char* __restrict__ w, int tid, int nthreads,
                                 int OD = 1) {
        // hidden layers, both directions: one 16-byte slot = 8 k-values of one lane, three pieces.  One flat
        // loop over (layer, direction, tile, K-block, lane), two slots per thread and trip so that 16 loads are
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
            v4u_t ph, pm, pl;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                unsigned int a, b, c;
                bf_split2(x[2 * p], x[2 * p + 1], a, b, c);
                ph[p] = a; pm[p] = b; pl[p] = c;
            }
            unsigned char* base = w + (dir == 0 ? off_hidf(L, l1 + 1) : off_hidb(L, l1 + 1)) + rest * 16;
            *reinterpret_cast<v4u_t*>(base) = ph;
            *reinterpret_cast<v4u_t*>(base + HID_PIECE) = pm;
            *reinterpret_cast<v4u_t*>(base + 2 * HID_PIECE) = pl;
        };
        for (int e0 = tid; e0 < n_slots; e0 += 2 * nthreads) {
            const int e1 = e0 + nthreads;
            float x0[8], x1[8];
            fetch(e0, x0);
            fetch(e1 < n_slots ? e1 : e0, x1);
            emit(e0, x0);
            if (e1 < n_slots) emit(e1, x1);
        }
        for (int e = tid; e < (L - 1) * H; e += nthreads)
            reinterpret_cast<float*>(w + off_bias(L))[H + e] = P1[(size_t)(e / H) * (H * H + H) + H * H + e % H];
        const float* P = P1 + (size_t)(L - 1) * (H * H + H);
        // layer 0 forward: lane (m, g) holds W0[16 mt + m][4g + i], i = 0..3 (zero beyond the 11 inputs)
        for (int e = tid; e < MT * 64; e += nthreads) {
            const int lane = e & 63, mt = e >> 6, m = lane & 15, g = lane >> 4;
            float x[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = 4 * g + i < MLP_IN ? dec[(16 * mt + m) * MLP_IN + 4 * g + i] : 0.f;
            v2u_t ph, pm, pl;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                unsigned int a, b, c;
                bf_split2(x[2 * p], x[2 * p + 1], a, b, c);
                ph[p] = a; pm[p] = b; pl[p] = c;
            }
            unsigned char* base = w + off_l0f(L) + e * 8;
            *reinterpret_cast<v2u_t*>(base) = ph;
            *reinterpret_cast<v2u_t*>(base + MT * 64 * 8) = pm;
            *reinterpret_cast<v2u_t*>(base + 2 * MT * 64 * 8) = pl;
        }
        // layer 0 transposed: lane (c, g) holds W0[u(j, g, i)][c]
        for (int e = tid; e < NJ * 64; e += nthreads) {
            const int lane = e & 63, j = e >> 6, c = lane & 15, g = lane >> 4;
            v4u_t ph, pm, pl;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float x0 = c < MLP_IN ? dec[unit_of(j, g, 2 * p) * MLP_IN + c] : 0.f;
                const float x1 = c < MLP_IN ? dec[unit_of(j, g, 2 * p + 1) * MLP_IN + c] : 0.f;
                unsigned int a, b, cc;
                bf_split2(x0, x1, a, b, cc);
                ph[p] = a; pm[p] = b; pl[p] = cc;
            }
            unsigned char* base = w + off_l0b(L) + e * 16;
            *reinterpret_cast<v4u_t*>(base) = ph;
            *reinterpret_cast<v4u_t*>(base + NJ * 64 * 16) = pm;
            *reinterpret_cast<v4u_t*>(base + 2 * NJ * 64 * 16) = pl;
        }
        for (int e = tid; e < H; e += nthreads) reinterpret_cast<float*>(w + off_bias(L))[e] = dec[H * MLP_IN + e];
        // lout.weight [OD][H] then lout.bias [OD]  ->  Wo at O[c*H + u], bias at O[3H + c]
        float* O = reinterpret_cast<float*>(w + off_out(L));
        for (int e = tid; e < OD * H; e += nthreads) O[e] = P[e];
        for (int e = tid; e < OD; e += nthreads) O[MF_OD_MAX * H + e] = P[OD * H + e];
    }

    // ------------------------------------------------------------------------------ building blocks
    // B operands of a layer: the lane's 16 activations as NJ x 3 pieces of 8 packed bf16
    __device__ __forceinline__ static void split_acts(const v4f_t (&h)[MT], v4u_t (&bh)[NJ], v4u_t (&bm)[NJ], v4u_t (&bl)[NJ]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int mt = 2 * j + (p >> 1), r = 2 * (p & 1);
                unsigned int a, b, c;
                bf_split2(h[mt][r], h[mt][r + 1], a, b, c);
                bh[j][p] = a; bm[j][p] = b; bl[j][p] = c;
            }
    }

    // acc[mt] += W h over one staged direction image (smallest products first)
    __device__ __forceinline__ static void matmul(const unsigned char* __restrict__ img, const v4u_t (&bh)[NJ],
                                                  const v4u_t (&bm)[NJ], const v4u_t (&bl)[NJ], v4f_t (&acc)[MT]) {
        const int lane = threadIdx.x & 63;
        const v4u_t* __restrict__ A = reinterpret_cast<const v4u_t*>(img) + lane;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int m0 = 0; m0 < MT; m0 += 2) {
                v8bf_t ah[2], am[2], al[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int slot = ((m0 + q) * NJ + j) * 64;
                    ah[q] = as_bf8(A[slot]);
                    am[q] = as_bf8(A[slot + HID_PIECE / 16]);
                    al[q] = as_bf8(A[slot + 2 * HID_PIECE / 16]);
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[m0 + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[q], as_bf8(bh[j]), acc[m0 + q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[m0 + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[q], as_bf8(bl[j]), acc[m0 + q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[m0 + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[q], as_bf8(bm[j]), acc[m0 + q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[m0 + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[q], as_bf8(bh[j]), acc[m0 + q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[m0 + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[q], as_bf8(bm[j]), acc[m0 + q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[m0 + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[q], as_bf8(bh[j]), acc[m0 + q], 0, 0, 0);
            }
    }

    __device__ __forceinline__ static void load_bias(const unsigned char* __restrict__ w, int L, int l, v4f_t (&acc)[MT]) {
        const int g = (threadIdx.x & 63) >> 4;
        const float* __restrict__ B = reinterpret_cast<const float*>(w + off_bias(L)) + l * H;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = *reinterpret_cast<const v4f_t*>(B + 16 * mt + 4 * g);
    }

    // layer 0: acc = b0 + W0 z, z[r] = input component 4g + r of this lane's query (K = 16 instruction)
    __device__ __forceinline__ static void layer0(const unsigned char* __restrict__ w, int L, const float (&z)[4], v4f_t (&acc)[MT]) {
        const int lane = threadIdx.x & 63;
        load_bias(w, L, 0, acc);
        v2u_t zh, zm, zl;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            unsigned int a, b, c;
            bf_split2(z[2 * p], z[2 * p + 1], a, b, c);
            zh[p] = a; zm[p] = b; zl[p] = c;
        }
        const v2u_t* __restrict__ A = reinterpret_cast<const v2u_t*>(w + off_l0f(L)) + lane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const v4s_t ah = as_s4(A[mt * 64]), am = as_s4(A[(MT + mt) * 64]), al = as_s4(A[(2 * MT + mt) * 64]);
            v4f_t c = acc[mt];
            c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, as_s4(zh), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, as_s4(zl), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(am, as_s4(zm), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(am, as_s4(zh), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, as_s4(zm), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, as_s4(zh), c, 0, 0, 0);
            acc[mt] = c;
        }
    }

    // transposed layer 0: a[r] = sum_u W0[u][4g + r] h[u]
    __device__ __forceinline__ static void input_backward(const unsigned char* __restrict__ w, int L, const v4f_t (&h)[MT],
                                                          float (&a)[4]) {
        const int lane = threadIdx.x & 63;
        v4u_t bh[NJ], bm[NJ], bl[NJ];
        split_acts(h, bh, bm, bl);
        const v4u_t* __restrict__ A = reinterpret_cast<const v4u_t*>(w + off_l0b(L)) + lane;
        v4f_t c[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const v8bf_t ah = as_bf8(A[j * 64]), am = as_bf8(A[(NJ + j) * 64]), al = as_bf8(A[(2 * NJ + j) * 64]);
            v4f_t t = (v4f_t){0.f, 0.f, 0.f, 0.f};
            t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, as_bf8(bh[j]), t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, as_bf8(bl[j]), t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, as_bf8(bm[j]), t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, as_bf8(bh[j]), t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, as_bf8(bm[j]), t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, as_bf8(bh[j]), t, 0, 0, 0);
            c[j] = t;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = c[0][r];
#pragma unroll
            for (int j = 1; j < NJ; ++j) s += c[j][r];
            a[r] = s;
        }
    }

    // ReLU pattern of a layer, read back from the packed bf16 hi pieces of its (post-ReLU, hence >= 0) activations:
    // element (mj, r) sits in word [mj / 2][2 (mj % 2) + r / 2], half r % 2 (split_acts); a positive fp32 value has a
    // non-zero upper half.  h <- pattern .* acc
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
    // known at compile time: both sweeps fully unrolled, and no mask words -- the hi pieces of every layer's
    // activations (the B operand of the next layer anyway) stay in registers until the transposed sweep has used them
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
            v4u_t bm[NJ], bl[NJ];
            split_acts(h, sg[l - 1], bm, bl);
            load_bias(w, L, l, acc);
            matmul(w + off_hidf(L, l), sg[l - 1], bm, bl, acc);
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
        x += __shfl_xor(x, 16, 64);
        x += __shfl_xor(x, 32, 64);
        x += O[MF_OD_MAX * H];
#pragma unroll
        for (int l = L - 1; l >= 1; --l) {
            v4u_t bh[NJ], bm[NJ], bl[NJ];
            split_acts(h, bh, bm, bl);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4f_t){0.f, 0.f, 0.f, 0.f};
            matmul(w + off_hidb(L, l), bh, bm, bl, acc);
            mask_by_pieces(sg[l - 1], acc, h);
        }
        input_backward(w, L, h, a);
        return x;
    }
};

// One interface over the two decoder images, for kernels templated on the arithmetic: BF = false is the fp32
// MFMA image of mlp_quad.h, BF = true the split-bf16 one.  `smem` is the block's dynamic LDS.
template <int H, bool BF>
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
    using Q = QuadDecoderB<H>;
    __host__ __device__ static constexpr int bytes(int L) { return Q::bytes(L); }
    __device__ __forceinline__ static void stage(const float* __restrict__ dec, int L, unsigned char* smem, int tid, int nthreads) {
        Q::stage(dec, L, smem, tid, nthreads);
    }
    template <int LC>
    __device__ __forceinline__ static float run(const unsigned char* smem, int, const float (&z)[4], float (&a)[4]) {
        return Q::template run<LC>(smem, z, a);
    }
};

// PIN_MLP=f32 keeps the fp32 MFMA decoder (A/B runs); default: split bf16
static inline bool use_bf3_decoder() {
    static const int on = [] {
        const char* e = getenv("PIN_MLP");
        return (e != nullptr && strcmp(e, "f32") == 0) ? 0 : 1;
    }();
    return on != 0;
}

}  // namespace pin
