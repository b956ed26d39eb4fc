// Device helpers of the brick cache shared by the search kernels (brick.hip) and the fused
// Gauss-Newton tile kernel (gn_fused.h): directory lookup, brick keys, the exact per-cell probe.
#pragma once
#include "pin_common.h"

namespace pin {

constexpr unsigned long long BRICK_EMPTY = ~0ull;
constexpr int BRICK_GROUP = 16;
constexpr int BRICK_BLOCK = 256;

// 16-lane (one DPP row) all-reductions: quad xor 1, quad xor 2, row_half_mirror, row_mirror
template <int CTRL>
__device__ __forceinline__ unsigned int dpp_u32(unsigned int v) {
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ unsigned int row_min_u32(unsigned int v) {
    v = min(v, dpp_u32<0xB1>(v));
    v = min(v, dpp_u32<0x4E>(v));
    v = min(v, dpp_u32<0x141>(v));
    v = min(v, dpp_u32<0x140>(v));
    return v;
}
__device__ __forceinline__ unsigned int row_sum_u32(unsigned int v) {
    v += dpp_u32<0xB1>(v);
    v += dpp_u32<0x4E>(v);
    v += dpp_u32<0x141>(v);
    v += dpp_u32<0x140>(v);
    return v;
}
// the same over aligned groups of G = 8 or 16 lanes
template <int G>
__device__ __forceinline__ unsigned int group_min_u32(unsigned int v) {
    v = min(v, dpp_u32<0xB1>(v));
    v = min(v, dpp_u32<0x4E>(v));
    v = min(v, dpp_u32<0x141>(v));
    if (G == 16) v = min(v, dpp_u32<0x140>(v));
    return v;
}
template <int G>
__device__ __forceinline__ unsigned int group_sum_u32(unsigned int v) {
    v += dpp_u32<0xB1>(v);
    v += dpp_u32<0x4E>(v);
    v += dpp_u32<0x141>(v);
    if (G == 16) v += dpp_u32<0x140>(v);
    return v;
}

// XCD-aware block order (MI355X: 8 XCDs with their own 4 MB L2; block b is dispatched to XCD b % 8).  The source points of
// a registration are Morton-ordered, so a CONTIGUOUS range of queries is a compact piece of space: logical block
// (b & 7) * per + (b >> 3), per = ceil(blocks / 8), gives XCD x the x-th eighth of the queries -- its L2 then holds one
// eighth of the bricks' entries / the candidate lists / the feature rows instead of a slice of everything, and the records
// this launch writes stay in the L2 of the XCD whose tile-kernel blocks read them next (gn_quad.h uses the same split).
// Launch 8 * per blocks; blocks whose logical index is past the end leave at once.
// (PIN_XCD=0 in the environment turns the mapping off for A/B runs: every translation unit that uses it keeps its own copy of
// the switch, set once by xcd_mode_init() from its launchers)
static __device__ int g_xcd_on = 1;
static inline int xcd_mode_init(const char* var = "PIN_XCD", int dflt = 1) {
    static const int done = [var, dflt] {
        const char* e = getenv(var);
        const int on = e ? (e[0] != '0') : dflt;
        return hipMemcpyToSymbol(HIP_SYMBOL(g_xcd_on), &on, sizeof(int)) == hipSuccess ? 1 : -1;
    }();
    return done;
}
__device__ __forceinline__ int xcd_logical_block(int b, int n_blocks) {
    if (!g_xcd_on) return b;
    const int per = (n_blocks + 7) >> 3;
    return (b & 7) * per + (b >> 3);
}
static inline int xcd_grid(int n_blocks) { return 8 * ((n_blocks + 7) / 8); }

__device__ __forceinline__ unsigned long long brick_key(int bx, int by, int bz) {
    return ((unsigned long long)(unsigned)(bx + (1 << 20)) << 42) | ((unsigned long long)(unsigned)(by + (1 << 20)) << 21) |
           (unsigned long long)(unsigned)(bz + (1 << 20));
}
__device__ __forceinline__ unsigned int mix64(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return (unsigned int)k;
}

__device__ __forceinline__ int dir_find(const pin_brick_cache& bc, unsigned long long key) {
    unsigned int h = mix64(key) & bc.dir_mask;
    for (int probe = 0; probe < 64; ++probe) {
        const unsigned long long k = bc.dir_keys[h];
        if (k == key) return bc.dir_vals[h];
        if (k == BRICK_EMPTY) return -1;
        h = (h + 1) & bc.dir_mask;
    }
    return -1;
}

// What a query needs of a brick, from ONE 32-byte directory slot per probe (dir_pack, written by
// brick_publish_kernel): base < 0 means "not cached" (absent, or dropped for lack of room) and sends
// the cell to the exact probe.
struct BrickInfo {
    int base;
    unsigned int lo, hi;
};
__device__ __forceinline__ BrickInfo dir_lookup(const pin_brick_cache& bc, unsigned long long key) {
    BrickInfo r;
    r.base = -1; r.lo = 0; r.hi = 0;
    unsigned int h = mix64(key) & bc.dir_mask;
    const ulonglong2* __restrict__ pack = reinterpret_cast<const ulonglong2*>(bc.dir_pack);
    for (int probe = 0; probe < 64; ++probe) {
        const ulonglong2 a = pack[2 * (size_t)h], b = pack[2 * (size_t)h + 1];  // key, mask | base, pad
        if (a.x == key) {
            r.base = (int)(unsigned int)b.x;
            r.lo = (unsigned int)a.y; r.hi = (unsigned int)(a.y >> 32);
            return r;
        }
        if (a.x == BRICK_EMPTY) return r;
        h = (h + 1) & bc.dir_mask;
    }
    return r;
}

// the reference's lookup chain for one cell: table -> time filter -> index space
__device__ __forceinline__ bool lookup_cell(const pin_search_params& sp, long long cx, long long cy, long long cz,
                                            float d_cur, float4& P, int& l) {
    const long long h = cx * PRIME0 + cy * PRIME1 + cz * PRIME2;
    const long long m = mod_nonneg(h, sp.buffer_size);
    const int j = sp.table[m];
    if (j < 0) return false;
    P = reinterpret_cast<const float4*>(sp.pos4)[j];
    if (sp.travel_dist != nullptr) {
        const float dts = sp.travel_dist[__float_as_int(P.w)];
        if (!(fabsf(d_cur - dts) < sp.diff_travel_dist_local)) return false;
    }
    l = j;
    if (sp.global2local != nullptr) {
        l = sp.global2local[j];
        if (l == PIN_NONLOCAL) l = 1 | PIN_NBR_QUIRK_BIT;
    }
    return l >= 0;
}


}  // namespace pin
