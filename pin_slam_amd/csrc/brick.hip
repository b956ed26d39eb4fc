// Brick cache: a cell-coherent view of the voxel hash for the per-frame hot queries (gfx950).
//
// The reference probes  table[hash(cell)]  for Kc candidate cells per query
// (model/neural_points.py:965-978); its hash scatters neighbouring cells over a 200 MB
// table, so every probe is its own memory sector and is followed by two more dependent random
// gathers (position/timestamp, global2local).  Measured on MI355X (profiles/r01_knn_pmc.json):
// 2.7x the algorithmic bytes, 60 % L2 hit rate.
//
// Once per frame (after reset_local_map) we evaluate the SAME lookups for every cell of every
// 4x4x4-cell "brick" near a local neural point and store the answers contiguously:
//   directory : open-addressing hash  brick coordinate -> brick id   (ours; small, L2 resident)
//   header    : 64-bit occupancy mask + base offset per brick
//   entries   : (x, y, z, local index bits) per occupied cell, contiguous per brick
// An entry is exactly what the reference chain  table -> travel-distance filter -> global2local
// would yield for that cell (hash collisions included: a colliding far point is stored and is
// rejected by the distance test at query time, as in the reference).  Entries that can never
// pass the distance test for any query that probes the cell are pruned.  Cells of bricks that
// are not in the directory fall back to the exact slow path, so results are identical to
// pin_knn_query in all cases (tests/test_gpu_bricks.py checks bit equality).
#include "brick.h"
#include "compact.h"

namespace pin {

// ---- build ----------------------------------------------------------------------------------
// A: every local point registers the (up to 8) bricks that cover its +-n cell neighbourhood.
// Brick ids are allocated wave-aggregated (one atomic per wave and slot, not per brick).
// (n_vb, here and in the fill / clear / publish kernels: the number of 256-thread units of work; the grid may be narrower --
// pin_brick_cache.build_grid, pin_brick_build -- and then walks them, so that the build leaves room for the launches of another stream)
__global__ __launch_bounds__(256) void brick_mark_kernel(pin_brick_cache bc, pin_search_params sp, int n_dilate,
                                                         int* __restrict__ counters, int n_vb) {
  for (int vb = blockIdx.x; vb < n_vb; vb += gridDim.x) {
    const int j = vb * 256 + threadIdx.x;
    bool work = j < sp.n_points && !(sp.global2local != nullptr && sp.global2local[j] < 0);
    int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    if (work) {
        const float4 P = reinterpret_cast<const float4*>(sp.pos4)[j];
        const int g[3] = {(int)voxel_coord(P.x, sp.resolution), (int)voxel_coord(P.y, sp.resolution),
                          (int)voxel_coord(P.z, sp.resolution)};
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = (g[a] - n_dilate) >> 2; hi[a] = (g[a] + n_dilate) >> 2; }
        // points are appended in voxel order per frame, so neighbours in memory mostly cover the
        // same bricks: skip the directory traffic when the previous local point already did it
        if (j > 0 && (sp.global2local == nullptr || sp.global2local[j - 1] >= 0)) {
            const float4 Q = reinterpret_cast<const float4*>(sp.pos4)[j - 1];
            const int q[3] = {(int)voxel_coord(Q.x, sp.resolution), (int)voxel_coord(Q.y, sp.resolution),
                              (int)voxel_coord(Q.z, sp.resolution)};
            bool same = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) same = same && ((q[a] - n_dilate) >> 2) == lo[a] && ((q[a] + n_dilate) >> 2) == hi[a];
            if (same) work = false;
        }
    }
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(bc.dir_keys);
    const int lane = threadIdx.x & 63;
    for (int s = 0; s < 8; ++s) {  // wave-uniform loop over the 2x2x2 brick slots
        const int bx = lo[0] + (s >> 2), by = lo[1] + ((s >> 1) & 1), bz = lo[2] + (s & 1);
        const bool has = work && bx <= hi[0] && by <= hi[1] && bz <= hi[2];
        bool won = false;
        unsigned int h = 0;
        unsigned long long key = 0;
        if (has) {
            key = brick_key(bx, by, bz);
            h = mix64(key) & bc.dir_mask;
            for (int probe = 0; probe < 64; ++probe) {
                unsigned long long prev = __hip_atomic_load(keys + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (prev == BRICK_EMPTY) prev = atomicCAS(keys + h, BRICK_EMPTY, key);
                if (prev == BRICK_EMPTY) { won = true; break; }
                if (prev == key) break;
                h = (h + 1) & bc.dir_mask;
                if (probe == 63) atomicOr(counters + 2, 2);
            }
        }
        const unsigned long long bal = __ballot(won);
        if (bal != 0ull) {
            int base = 0;
            const int leader = __ffsll((long long)bal) - 1;
            if (lane == leader) base = atomicAdd(counters + 0, __popcll(bal));
            base = __shfl(base, leader, 64);
            if (won) {
                const int id = base + __popcll(bal & ((1ull << lane) - 1ull));
                if (id < bc.max_bricks) { bc.brick_keys[id] = key; bc.dir_vals[h] = id; }
                else { bc.dir_vals[h] = -1; atomicOr(counters + 2, 1); }
            }
        }
    }
  }
}

// B: one lane per cell, one wave per brick at a time; a block (4 waves) takes 32 bricks and
// reserves their entries with ONE atomic (a same-address atomic per brick serialises in L2:
// that alone was 0.7 ms at 74k bricks).
constexpr int FILL_PER_WAVE = 8;
__global__ __launch_bounds__(256) void brick_fill_kernel(pin_brick_cache bc, pin_search_params sp, float prune_dist2,
                                                        int* __restrict__ counters, int n_vb) {
    __shared__ int wave_tot[4];
    __shared__ int block_base;
    const int nb = min(counters[0], bc.max_bricks);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float d_cur = sp.travel_dist ? sp.travel_dist[sp.cur_ts] : 0.f;
  for (int vb = blockIdx.x; vb < n_vb && vb * 4 * FILL_PER_WAVE < nb; vb += gridDim.x) {
    const int b0 = (vb * 4 + wave) * FILL_PER_WAVE;
    float4 P[FILL_PER_WAVE];
    int l[FILL_PER_WAVE];
    unsigned long long mask[FILL_PER_WAVE];
    int tot = 0;
#pragma unroll
    for (int u = 0; u < FILL_PER_WAVE; ++u) {
        const int b = b0 + u;
        bool ok = false;
        l[u] = -1;
        if (b < nb) {
            const unsigned long long key = bc.brick_keys[b];
            const int bx = (int)((key >> 42) & 0x1fffff) - (1 << 20), by = (int)((key >> 21) & 0x1fffff) - (1 << 20),
                      bz = (int)(key & 0x1fffff) - (1 << 20);
            const int cx = bx * 4 + (lane >> 4), cy = by * 4 + ((lane >> 2) & 3), cz = bz * 4 + (lane & 3);
            ok = lookup_cell(sp, cx, cy, cz, d_cur, P[u], l[u]);
            if (ok) {  // prune what no probing query can accept (exactness preserved: see header)
                const float r = sp.resolution;
                const float ex = P[u].x - (cx + 0.5f) * r, ey = P[u].y - (cy + 0.5f) * r, ez = P[u].z - (cz + 0.5f) * r;
                ok = (ex * ex + ey * ey + ez * ez) <= prune_dist2;
            }
        }
        if (!ok) l[u] = -1;
        mask[u] = __ballot(ok);
        tot += __popcll(mask[u]);
    }
    if (lane == 0) wave_tot[wave] = tot;
    __syncthreads();
    if (threadIdx.x == 0) block_base = atomicAdd(counters + 1, wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3]);
    __syncthreads();
    int base = block_base;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
#pragma unroll
    for (int u = 0; u < FILL_PER_WAVE; ++u) {
        const int b = b0 + u;
        if (b < nb) {
            // a brick whose entries do not fit is published as "not cached": its cells take the exact probe,
            // so an overflow costs time, never correctness, and the host may learn about it a frame later
            const bool fits = base + __popcll(mask[u]) <= bc.max_entries;
            if (lane == 0) { bc.brick_mask[b] = mask[u]; bc.brick_base[b] = fits ? base : -1; }
            if (l[u] >= 0) {
                const int e = base + __popcll(mask[u] & ((1ull << lane) - 1ull));
                if (e < bc.max_entries)
                    reinterpret_cast<float4*>(bc.entries)[e] = make_float4(P[u].x, P[u].y, P[u].z, __int_as_float(l[u]));
                else atomicOr(counters + 2, 4);
            }
        }
        base += __popcll(mask[u]);
    }
    __syncthreads();  // (wave_tot / block_base are rewritten by the next unit)
  }
}

// ---- point-driven build (r04; pin_brick_cache.build_ws != NULL) ---------------------------------------------------------------
// The cell-driven build above asks "what does the table hold for this cell?" for all 64 cells of every brick -- 4.7 M random table
// probes for 74 k bricks at 2.2 M points, 60 % of them for empty cells (0.22 ms), after a marking pass in which every local point
// probes the directory for up to 8 bricks (0.15 ms).  The same cache can be built from the POINTS:
//   * an entry of cell c is non-empty only if table[hash(c)] is a point near c; with no second cell within the pruning reach that
//     shares c's hash (checked on the host for the table size at hand: brick_alias_free) that point lies IN c.  So point j
//     contributes exactly to its own cell, and only if it owns its slot (table[hash(cell(j))] == j: the last writer of a
//     collision) and passes the time filter -- the same chain lookup_cell evaluates, entered from the other end: 2.2 M probes.
//   * the set of bricks -- those that intersect [g - n, g + n]^3 for the cell g of some local point -- is the set of the local
//     points' own bricks, dilated: neighbour d in {-1, 0, 1}^3 of an own brick is needed iff one of its occupied cells lies within
//     n of that face / edge / corner, which the brick's 64-bit mask of local-point cells answers with one AND per direction.
// Same directory contents (in another order of ids), same masks, same entries per cell: tests/test_gpu_bricks.py builds both ways.
struct BrickDirMasks { unsigned long long m[27]; };

// every lane of the wave calls this (has = this lane brings a key): directory probe + wave-aggregated id allocation
__device__ __forceinline__ void brick_insert(const pin_brick_cache& bc, unsigned long long key, bool has, int* __restrict__ counters) {
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(bc.dir_keys);
    const int lane = threadIdx.x & 63;
    bool won = false;
    unsigned int h = 0;
    if (has) {
        h = mix64(key) & bc.dir_mask;
        for (int probe = 0; probe < 64; ++probe) {
            unsigned long long prev = __hip_atomic_load(keys + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == BRICK_EMPTY) prev = atomicCAS(keys + h, BRICK_EMPTY, key);
            if (prev == BRICK_EMPTY) { won = true; break; }
            if (prev == key) break;
            h = (h + 1) & bc.dir_mask;
            if (probe == 63) atomicOr(counters + 2, 2);
        }
    }
    const unsigned long long bal = __ballot(won);
    if (bal != 0ull) {
        int base = 0;
        const int leader = __ffsll((long long)bal) - 1;
        if (lane == leader) base = atomicAdd(counters + 0, __popcll(bal));
        base = __shfl(base, leader, 64);
        if (won) {
            const int id = base + __popcll(bal & ((1ull << lane) - 1ull));
            if (id < bc.max_bricks) { bc.brick_keys[id] = key; bc.dir_vals[h] = id; }
            else { bc.dir_vals[h] = -1; atomicOr(counters + 2, 1); }
        }
    }
}

__device__ __forceinline__ bool brick_point_cell(const pin_search_params& sp, int j, float4& P, int (&g)[3]) {
    P = reinterpret_cast<const float4*>(sp.pos4)[j];
    g[0] = (int)voxel_coord(P.x, sp.resolution); g[1] = (int)voxel_coord(P.y, sp.resolution); g[2] = (int)voxel_coord(P.z, sp.resolution);
    return true;
}

// M1: every local point's OWN brick
__global__ __launch_bounds__(256) void brick_own_kernel(pin_brick_cache bc, pin_search_params sp, int* __restrict__ counters) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    bool work = j < sp.n_points && !(sp.global2local != nullptr && sp.global2local[j] < 0);
    unsigned long long key = 0;
    if (work) {
        float4 P; int g[3];
        brick_point_cell(sp, j, P, g);
        key = brick_key(g[0] >> 2, g[1] >> 2, g[2] >> 2);
        if (j > 0 && (sp.global2local == nullptr || sp.global2local[j - 1] >= 0)) {  // (neighbours in memory are often neighbours in space)
            float4 Q; int q[3];
            brick_point_cell(sp, j - 1, Q, q);
            if (brick_key(q[0] >> 2, q[1] >> 2, q[2] >> 2) == key) work = false;
        }
    }
    brick_insert(bc, key, work, counters);
}

// M1b: the cells of the local points, per own brick (what the dilation needs); thread 0 freezes the number of own bricks
__global__ __launch_bounds__(256) void brick_pmask_kernel(pin_brick_cache bc, pin_search_params sp, unsigned long long* __restrict__ pmask,
                                                          int* __restrict__ counters) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j == 0) counters[3] = min(counters[0], bc.max_bricks);
    if (j >= sp.n_points || (sp.global2local != nullptr && sp.global2local[j] < 0)) return;
    float4 P; int g[3];
    brick_point_cell(sp, j, P, g);
    const int id = dir_find(bc, brick_key(g[0] >> 2, g[1] >> 2, g[2] >> 2));
    if (id < 0 || id >= bc.max_bricks) return;
    const unsigned long long bit = 1ull << (((g[0] & 3) << 4) | ((g[1] & 3) << 2) | (g[2] & 3));
    if ((__hip_atomic_load(pmask + id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) == 0ull) atomicOr(pmask + id, bit);
}

// M2: one thread per (own brick, direction): the neighbour brick is needed iff a local point's cell lies within n of that side
__global__ __launch_bounds__(256) void brick_dilate_kernel(pin_brick_cache bc, const unsigned long long* __restrict__ pmask,
                                                           BrickDirMasks dm, int* __restrict__ counters) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int b = (int)(t / 27), d = (int)(t - (long)b * 27);
    bool has = b < counters[3] && d != 13;
    unsigned long long key = 0;
    if (has) {
        has = (pmask[b] & dm.m[d]) != 0ull;
        if (has) {
            const unsigned long long k0 = bc.brick_keys[b];
            const int bx = (int)((k0 >> 42) & 0x1fffff) - (1 << 20), by = (int)((k0 >> 21) & 0x1fffff) - (1 << 20),
                      bz = (int)(k0 & 0x1fffff) - (1 << 20);
            key = brick_key(bx + d / 9 - 1, by + (d / 3) % 3 - 1, bz + d % 3 - 1);
        }
    }
    brick_insert(bc, key, has, counters);
}

// A: every point that owns its table slot, passes the time filter and lies in a marked brick sets its cell's bit
__global__ __launch_bounds__(256) void brick_point_mask_kernel(pin_brick_cache bc, pin_search_params sp, float prune_dist2,
                                                               int2* __restrict__ tmp) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= sp.n_points) return;
    int2 out = make_int2(-1, -1);
    float4 P; int g[3];
    brick_point_cell(sp, j, P, g);
    const long long h = (long long)g[0] * PRIME0 + (long long)g[1] * PRIME1 + (long long)g[2] * PRIME2;
    bool ok = sp.table[mod_nonneg(h, sp.buffer_size)] == j;
    if (ok && sp.travel_dist != nullptr) {
        const float d_cur = sp.travel_dist[sp.cur_ts];
        const float dts = sp.travel_dist[__float_as_int(P.w)];
        ok = fabsf(d_cur - dts) < sp.diff_travel_dist_local;
    }
    int l = j;
    if (ok && sp.global2local != nullptr) {
        l = sp.global2local[j];
        if (l == PIN_NONLOCAL) l = 1 | PIN_NBR_QUIRK_BIT;
        ok = l >= 0;
    }
    if (ok) {  // (the pruning test of the cell-driven fill; a point is always within reach of its own cell)
        const float r = sp.resolution;
        const float ex = P.x - (g[0] + 0.5f) * r, ey = P.y - (g[1] + 0.5f) * r, ez = P.z - (g[2] + 0.5f) * r;
        ok = (ex * ex + ey * ey + ez * ez) <= prune_dist2;
    }
    if (ok) {
        const int id = dir_find(bc, brick_key(g[0] >> 2, g[1] >> 2, g[2] >> 2));
        if (id >= 0 && id < bc.max_bricks) {
            const int bit = ((g[0] & 3) << 4) | ((g[1] & 3) << 2) | (g[2] & 3);
            atomicOr(reinterpret_cast<unsigned long long*>(bc.brick_mask) + id, 1ull << bit);
            out = make_int2(id | (bit << 24), l);
        }
    }
    tmp[j] = out;
}

// B: entry ranges, one atomic per block of bricks
__global__ __launch_bounds__(256) void brick_bases_kernel(pin_brick_cache bc, int* __restrict__ counters) {
    __shared__ int wave_tot[4];
    __shared__ int block_base;
    const int nb = min(counters[0], bc.max_bricks);
    const int b = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cnt = b < nb ? __popcll(bc.brick_mask[b]) : 0;
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) block_base = atomicAdd(counters + 1, wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3]);
    __syncthreads();
    int base = block_base + incl - cnt;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    if (b < nb) {
        const bool fits = base + cnt <= bc.max_entries;
        bc.brick_base[b] = fits ? base : -1;  // (a brick whose entries do not fit is "not cached": its cells take the exact probe)
        if (!fits && cnt > 0) atomicOr(counters + 2, 4);
    }
}

// C: the entries, in cell-bit order inside their brick
__global__ __launch_bounds__(256) void brick_point_entries_kernel(pin_brick_cache bc, pin_search_params sp, const int2* __restrict__ tmp) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= sp.n_points) return;
    const int2 t = tmp[j];
    if (t.x < 0) return;
    const int id = t.x & 0xffffff, bit = t.x >> 24;
    const int base = bc.brick_base[id];
    if (base < 0) return;
    const unsigned long long m = bc.brick_mask[id];
    const int e = base + __popcll(m & ((1ull << bit) - 1ull));
    const float4 P = reinterpret_cast<const float4*>(sp.pos4)[j];
    reinterpret_cast<float4*>(bc.entries)[e] = make_float4(P.x, P.y, P.z, __int_as_float(t.y));
}

__global__ void brick_zero_masks_kernel(unsigned long long* a, unsigned long long* b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = 0ull; b[i] = 0ull; }
}

__global__ void brick_clear_kernel(unsigned long long* keys, int n, int* counters, float4* entries, int max_entries) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) keys[k] = BRICK_EMPTY;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4) counters[i] = 0;
    // the sentinel entry behind the last real one: what an empty cell reads in the query kernel (rejected by distance)
    if (i == 0) entries[max_entries] = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __int_as_float(-1));
}

// ---- query ----------------------------------------------------------------------------------
struct PoseB {
    float m[12];
    int on;
};

// C: one 32-byte slot per directory entry with everything a query reads (see dir_lookup)
__global__ __launch_bounds__(256) void brick_publish_kernel(pin_brick_cache bc) {
  for (unsigned int h = blockIdx.x * 256 + threadIdx.x; h <= bc.dir_mask; h += gridDim.x * 256) {
    const unsigned long long key = bc.dir_keys[h];
    unsigned long long mask = 0, base = 0xffffffffull;
    if (key != BRICK_EMPTY) {
        const int id = bc.dir_vals[h];
        if (id >= 0 && id < bc.max_bricks && bc.brick_base[id] >= 0) {
            mask = bc.brick_mask[id];
            base = (unsigned long long)(unsigned int)bc.brick_base[id];
        }
    }
    ulonglong2* pack = reinterpret_cast<ulonglong2*>(bc.dir_pack);
    pack[2 * (size_t)h] = make_ulonglong2(key, mask);
    pack[2 * (size_t)h + 1] = make_ulonglong2(base, 0ull);
  }
}

// T = tx | ty << 3 | tz << 6 (t = position of a candidate cell inside the query's 2x2x2-brick window, 0..7 per axis)
//   -> byte offset of the window brick's header row entry (select * 16) << 6 | bit of the cell in the brick's mask
struct KnnLut {
    unsigned short v[512];
    constexpr KnnLut() : v() {
        for (int i = 0; i < 512; ++i) {
            const int tx = i & 7, ty = (i >> 3) & 7, tz = i >> 6;
            const int sel = ((tx >> 2) << 2) | ((ty >> 2) << 1) | (tz >> 2);
            v[i] = (unsigned short)(((sel * 16) << 6) | ((tx & 3) << 4) | ((ty & 3) << 2) | (tz & 3));
        }
    }
};
__device__ const KnnLut KNN_LUT;

// one candidate cell of the cached path: header row of its brick -> offset of its entry, or of the sentinel entry
// (at index max_entries: +inf coordinates, rejected by the distance test) when the cell holds nothing
__device__ __forceinline__ unsigned int knn_cell_offset(const unsigned short* __restrict__ lut, const char* __restrict__ hrow,
                                                       unsigned int T, unsigned int sentinel) {
    const unsigned int v = lut[T];
    const uint4 h = *reinterpret_cast<const uint4*>(hrow + (v >> 6));  // (base, mask lo, mask hi, base + popc(lo))
    const unsigned int bit = v & 63u;
    const bool upper = bit >= 32u;
    const unsigned int word = upper ? h.z : h.y;
    const unsigned int below = word & __builtin_amdgcn_ubfe(0xffffffffu, 0u, bit);  // (v_bfe_u32 takes the width mod 32)
    const unsigned int off = (upper ? h.w : h.x) + (unsigned int)__popc(below);
    return __builtin_amdgcn_ubfe(word, bit, 1u) != 0u ? off : sentinel;
}

// EIGHT lanes per query, R = ceil(n_cand / 8) candidate cells per lane.
//
// The kernel is bound by its vector-instruction count (PMC r01 / r02: 17 M per 98.7k-query launch = 28 of 32 us of every
// SIMD), so the candidate pass is written for few instructions per cell:
//   * the cell arithmetic is ONE add per candidate: with p = (cell(q) - n_dilate) & 3 per axis (per query) and the
//     candidate's offset + n_dilate (0..4), t = p + d in 0..7 gives the window brick (t >> 2) and the position inside it
//     (t & 3) for all three axes from one packed word T = P + D (3-bit fields, no carries);
//   * (header offset, mask bit) of T come from a 512-entry table, the candidate's packed offset D from another, the 8
//     window bricks of the query (base, 64-bit occupancy, base + popcount(low word)) from a per-query LDS row the 8
//     lanes fill once: one 16-byte LDS read per candidate instead of three cross-lane shuffles;
//   * an empty cell reads a sentinel entry with +inf coordinates: no occupancy bits to carry to the distance pass;
//   * cells of window bricks that are not cached (absent from the directory, or dropped) need the reference's exact
//     probe: that pass runs only in waves that hold such a brick.
// Results are the same bits as before (tests: bricks == direct probe == reference on the fixtures and at 2.2 M points).
// (Measured and not kept: per-lane pre-extraction of the three smallest distances so that a selection round only
// compares heads -- 26.2 vs 26.8 us, not worth the refill logic; FOUR lanes per query with 21 candidates each -- 35 %
// fewer instructions per query on paper, 30.6 vs 26.3 us measured: at 104 registers the kernel keeps too few waves
// in flight for its entry gathers.)
// COHERENT SEARCH ACROSS GAUSS-NEWTON ITERATIONS (pin_gn_knn_coherent; knn_brick_kernel<R, true>).  Tracker.tracking searches the same source
// points 50 times under a pose that soon moves by micrometres (utils/tracker.py:114-184).  A full search leaves per query
// (i) where it stood, (ii) the entry offsets + candidate numbers of its k winners and (iii) a MARGIN: how far the query may
// move before anything about its result can change -- its voxel (distance to the nearest cell face), the set of accepted
// candidates (every candidate's distance to the acceptance radius) and the set of winners (gap between the k-th and the
// (k+1)-th accepted distance); a query's move of delta changes every distance by at most delta.  In a later iteration a
// query (8 lanes) that stands within its margin re-measures ONLY its k winners and re-ranks them on (distance bits, candidate
// number) -- the same record, bit for bit, as the full search would write (same entries, same distance arithmetic, same tie
// rule); a wave whose eight queries all do so is done after that, any other wave takes its remaining queries through the
// full search, which renews their state.
// (Measured and not kept: a pre-pass kernel that sends the queries outside their margins to a worklist for a compacted
// full search -- with ~1 % of the queries on the list the second launch still costs the ~12 us latency chain of one wave,
// and the pre-pass ~10 us of its own: no gain over the plain search.)
constexpr unsigned int COH_NONE = 0xffffffffu;
// The full search of the eight-lanes-per-query scheme for the groups with `active` set (the tables and header rows are this
// wave's; every lane of the wave takes part in their set-up).  knn_brick_kernel runs it for every query, the listed kernel below
// only for the groups it cannot serve from a candidate list.
template <int R, bool COH>
__device__ __forceinline__ void knn_full_search(const pin_search_params& sp, const pin_brick_cache& bc, const float qx, const float qy,
                                                const float qz, const bool active, const int qi, const int qq, const int k,
                                                unsigned short* const lut, unsigned int* const cpack, uint4* const hdr_grp,
                                                float4* __restrict__ nbr, int* __restrict__ nn_count,
                                                float4* __restrict__ coh_state, unsigned int* __restrict__ coh_win, const int coh_mode) {
    constexpr int G = 8;
    const int nd = bc.n_dilate;
    const int sub = threadIdx.x & (G - 1);
    const int wlane = threadIdx.x & 63;
    const float4* __restrict__ entries = reinterpret_cast<const float4*>(bc.entries);
    {   // this wave's copies of the two tables
        reinterpret_cast<uint4*>(lut)[wlane] = reinterpret_cast<const uint4*>(KNN_LUT.v)[wlane];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ci = wlane + 64 * h, c = ci < sp.n_cand ? ci : 0;
            cpack[ci] = (unsigned int)((bc.cand_dx[3 * c] + nd) | ((bc.cand_dx[3 * c + 1] + nd) << 3) | ((bc.cand_dx[3 * c + 2] + nd) << 6));
        }
    }
    // floor(q / res) as in voxel_coord (IEEE division); the cached path works on 32-bit cell coordinates, anything
    // farther than 2^29 cells from the origin (never the case for a metric map) takes the exact 64-bit probe
    float fx, fy, fz;
    {
#pragma clang fp contract(off)
        fx = floorf(__fdiv_rn(qx, sp.resolution)); fy = floorf(__fdiv_rn(qy, sp.resolution)); fz = floorf(__fdiv_rn(qz, sp.resolution));
    }
    const float lim = 536870912.f;  // 2^29
    const bool far = !(fabsf(fx) < lim && fabsf(fy) < lim && fabsf(fz) < lim);
    const int ix = far ? 0 : (int)fx, iy = far ? 0 : (int)fy, iz = far ? 0 : (int)fz;
    const int b0x = (ix - nd) >> 2, b0y = (iy - nd) >> 2, b0z = (iz - nd) >> 2;
    const unsigned int P = (unsigned int)(((ix - nd) & 3) | (((iy - nd) & 3) << 3) | (((iz - nd) & 3) << 6));
    bool uncached;
    {   // lane `sub` resolves window brick (sub >> 2, (sub >> 1) & 1, sub & 1) with one 32-byte directory load
        BrickInfo bi;
        bi.base = -1; bi.lo = 0; bi.hi = 0;
        if (!far) bi = dir_lookup(bc, brick_key(b0x + (sub >> 2), b0y + ((sub >> 1) & 1), b0z + (sub & 1)));
        uncached = bi.base < 0;
        if (uncached) { bi.lo = 0; bi.hi = 0; }  // its cells miss in the cached pass and are probed exactly below
        hdr_grp[sub] = make_uint4((unsigned int)bi.base, bi.lo, bi.hi, (unsigned int)bi.base + (unsigned int)__popc(bi.lo));
    }
    const bool general = __builtin_amdgcn_ballot_w64(uncached) != 0ull;  // some window brick of the wave is not cached
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the tables and header rows were written by other lanes of this wave
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const char* const hrow = reinterpret_cast<const char*>(hdr_grp);
    const unsigned int sentinel = (unsigned int)bc.max_entries;

    // accepted candidates: only the distance bits stay (registers); the k winners' entries are fetched
    // again at the end (cache hits), one winner per lane.  The candidate pass is straight-line: every lane
    // issues its R entry loads back to back, so a wave pays one memory round trip for all of them.
    unsigned int d2b[R];
    float4 E[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int c = r * G + sub;
        unsigned int off = knn_cell_offset(lut, hrow, P + cpack[c & 127], sentinel);
        if (r == R - 1) off = c < sp.n_cand ? off : sentinel;  // (only the last round can run past the candidate list)
        E[r] = entries[off];
    }
    int cnt = 0;
    float m_acc = __builtin_inff();  // COH: min over the candidates of |d2 - R^2| (how close anything is to being accepted / rejected)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float dx = E[r].x - qx, dy = E[r].y - qy, dz = E[r].z - qz;
        const float d2 = dist2_exact(dx, dy, dz);
        const bool acc = !(d2 > sp.max_valid_dist2);  // (the sentinel gives +inf)
        d2b[r] = acc ? __float_as_uint(d2) : 0xffffffffu;  // d2 >= 0: the bit pattern orders like the value
        cnt += acc ? 1 : 0;
        if (COH) m_acc = fminf(m_acc, fabsf(d2 - sp.max_valid_dist2));
    }
    const float d_cur = sp.travel_dist ? sp.travel_dist[sp.cur_ts] : 0.f;
    const long long gx = (long long)fx, gy = (long long)fy, gz = (long long)fz;  // (used by the exact probe only)
    if (general) {  // exact probe for the cells of uncached window bricks (rare: map border, a cache that overflowed)
#pragma unroll 1
        for (int r = 0; r < R; ++r) {
            const int c = r * G + sub;
            if (c >= sp.n_cand) continue;
            if (!far && (int)hdr_grp[lut[P + cpack[c]] >> 10].x >= 0) continue;  // cached brick: done above
            const int dxc = bc.cand_dx[3 * c], dyc = bc.cand_dx[3 * c + 1], dzc = bc.cand_dx[3 * c + 2];
            float4 Pp;
            int l = -1;
            if (!lookup_cell(sp, gx + dxc, gy + dyc, gz + dzc, d_cur, Pp, l)) continue;
            const float dx = Pp.x - qx, dy = Pp.y - qy, dz = Pp.z - qz;
            const float d2 = dist2_exact(dx, dy, dz);
            if (d2 > sp.max_valid_dist2) continue;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) d2b[rr] = rr == r ? __float_as_uint(d2) : d2b[rr];
            ++cnt;
        }
    }
    cnt = (int)group_sum_u32<G>((unsigned int)cnt);
    if (active && sub == 0) nn_count[qi] = cnt;

    // k rounds of an 8-lane tournament on (d2 bits, candidate order): two 32-bit group reductions
    // on the DPP path per round (no LDS crossbar, no 64-bit keys); lane t remembers winner t
    int mine = -1;
    unsigned int d_last = 0u;  // COH: distance bits of the last winner found
    int n_win = 0;
    for (int t = 0; t < k; ++t) {
        unsigned int bd = d2b[0];
        int br = 0;
#pragma unroll
        for (int r = 1; r < R; ++r)
            if (d2b[r] < bd) { bd = d2b[r]; br = r; }  // strict: the lowest r (= lowest candidate) wins ties
        const unsigned int wd = group_min_u32<G>(bd);
        if (wd == 0xffffffffu) break;
        d_last = wd;
        ++n_win;
        const unsigned int myc = bd == wd ? (unsigned int)(br * G + sub) : 0xffffffffu;
        const unsigned int wc = group_min_u32<G>(myc);
        if (myc == wc) {  // exactly one lane of the group
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (r == br) d2b[r] = 0xffffffffu;
        }
        if (sub == t) mine = (int)wc;
    }
    // lane t < k publishes record t: the winner's entry again (same bits as in the candidate pass)
    {
        const int cc = mine >= 0 ? mine : 0;
        const unsigned int T = P + cpack[cc];
        float4 rec = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
        if (mine >= 0) {
            float4 Ew = make_float4(0.f, 0.f, 0.f, 0.f);
            int l = -1;
            if (!far && (int)hdr_grp[lut[T] >> 10].x >= 0) {
                Ew = entries[knn_cell_offset(lut, hrow, T, sentinel)];
                l = __float_as_int(Ew.w);
            } else {
                lookup_cell(sp, gx + bc.cand_dx[3 * cc], gy + bc.cand_dx[3 * cc + 1], gz + bc.cand_dx[3 * cc + 2], d_cur, Ew, l);
            }
            const float dx = Ew.x - qx, dy = Ew.y - qy, dz = Ew.z - qz;
            rec = make_float4(-dx, -dy, -dz, __int_as_float(l));
        }
        if (active && sub < k) nbr[(size_t)qq * k + sub] = rec;
        if (COH && coh_mode != 0) {
            // what the coherent path of a later iteration needs: lane t's winner (entry offset | candidate << 24) ...
            const bool cached = mine >= 0 && !far && (int)hdr_grp[lut[T] >> 10].x >= 0;
            const unsigned int off = cached ? knn_cell_offset(lut, hrow, T, sentinel) : 0u;
            const bool bad = (mine >= 0 && !cached) || off >= 0x1000000u;  // a winner from the exact probe: no entry to come back to
            // ... and the margin.  The winners stay the winners while the query moves less than half the gap between the k-th
            // and the (k+1)-th accepted distance; a candidate at distance s crosses the acceptance radius Rv after
            // |s - Rv| = |d - Rv^2| / (s + Rv) >= |d - Rv^2| / (3 Rv) while s <= 2 Rv, and is more than Rv away otherwise.
            const float Rv = sqrtf(sp.max_valid_dist2);
            unsigned int d_next = 0xffffffffu;  // the (k+1)-th accepted distance, if the list was full
            if (n_win == k) {
                unsigned int bd = d2b[0];
#pragma unroll
                for (int r = 1; r < R; ++r) bd = min(bd, d2b[r]);
                d_next = group_min_u32<G>(bd);
            }
            float m_sel = __builtin_inff();
            if (d_next != 0xffffffffu) m_sel = 0.5f * (sqrtf(__uint_as_float(d_next)) - sqrtf(__uint_as_float(d_last)));  // half the gap
            float m_a = __uint_as_float(group_min_u32<G>(__float_as_uint(m_acc)));  // (non-negative floats order like their bits)
            m_a = fminf(m_a / (3.f * Rv), Rv);
            const float r = sp.resolution;
            const float ux = fmaf(-fx, r, qx), uy = fmaf(-fy, r, qy), uz = fmaf(-fz, r, qz);  // position inside the voxel
            const float m_cell = fminf(fminf(fminf(ux, r - ux), fminf(uy, r - uy)), fminf(uz, r - uz));
            // rounding of floor(q / res) and of the position inside the voxel: a few ulp of the coordinate
            const float safety = fmaxf(fmaxf(fabsf(qx), fabsf(qy)), fmaxf(fabsf(qz), 1.f)) * 4.8e-7f + 1e-6f;
            float m = fminf(fminf(m_sel, m_a), m_cell) - safety;
            const bool any_bad = group_sum_u32<G>((bad || uncached) ? 1u : 0u) != 0u;
            if (any_bad || far || !(m > 0.f)) m = 0.f;
            if (active) {
                coh_win[(size_t)qq * G + sub] = mine >= 0 && !bad ? (off | ((unsigned int)cc << 24)) : COH_NONE;
                if (sub == 0) coh_state[qq] = make_float4(qx, qy, qz, m * m);
            }
        }
    }
}

template <int R, bool COH>
__global__ __launch_bounds__(BRICK_BLOCK) void knn_brick_kernel(pin_search_params sp, pin_brick_cache bc,
                                                                const float* __restrict__ query, int n, int k, PoseB pose,
                                                                float* __restrict__ query_out, float4* __restrict__ nbr,
                                                                int* __restrict__ nn_count, const double* __restrict__ state,
                                                                float4* __restrict__ coh_state, unsigned int* __restrict__ coh_win,
                                                                int coh_mode) {
    constexpr int G = 8;
    // the tables are per WAVE: nothing in this kernel crosses a wave, so there is no block barrier (and a wave on the
    // coherent path leaves without keeping anybody waiting)
    __shared__ unsigned short lut_all[BRICK_BLOCK / 64][512];
    __shared__ unsigned int cpack_all[BRICK_BLOCK / 64][128];  // candidate -> (dx + nd) | (dy + nd) << 3 | (dz + nd) << 6
    __shared__ uint4 hdr[BRICK_BLOCK / G][8];                  // per query: its 2x2x2 window bricks
    if (state != nullptr) {
        if (state[PIN_GN_STATE_DONE] != 0.0) return;  // (block-uniform)
#pragma unroll
        for (int i = 0; i < 12; ++i) pose.m[i] = (float)state[i];
        pose.on = 1;
    }
    const int nd = bc.n_dilate;
    const int sub = threadIdx.x & (G - 1), grp = threadIdx.x / G;
    const int wlane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned short* const lut = lut_all[wv];
    unsigned int* const cpack = cpack_all[wv];
    const int lb = xcd_logical_block(blockIdx.x, (n * G + BRICK_BLOCK - 1) / BRICK_BLOCK);
    if ((long)lb * (BRICK_BLOCK / G) >= n) return;  // (the grid is rounded up to a multiple of 8 blocks)
    const int qi = (lb * BRICK_BLOCK + threadIdx.x) / G;
    bool active = qi < n;
    const int qq = active ? qi : n - 1;
    float qx = query[3 * qq + 0], qy = query[3 * qq + 1], qz = query[3 * qq + 2];
    if (pose.on) {
        const float* m = pose.m;
        const float tx = fmaf(qz, m[2], fmaf(qy, m[1], qx * m[0])) + m[3];
        const float ty = fmaf(qz, m[6], fmaf(qy, m[5], qx * m[4])) + m[7];
        const float tz = fmaf(qz, m[10], fmaf(qy, m[9], qx * m[8])) + m[11];
        qx = tx; qy = ty; qz = tz;
        if (query_out != nullptr && active && sub == 0) {
            query_out[3 * qi + 0] = qx; query_out[3 * qi + 1] = qy; query_out[3 * qi + 2] = qz;
        }
    }
    const float4* __restrict__ entries = reinterpret_cast<const float4*>(bc.entries);
    if (COH && coh_mode == 2) {
        const float4 st = coh_state[qq];  // (where the last full search of this query stood, its margin squared)
        const float ex = qx - st.x, ey = qy - st.y, ez = qz - st.z;
        const bool near = active && (ex * ex + ey * ey + ez * ez) < st.w;
        if (near) {  // lane t re-measures winner t and finds its rank among the eight
            const unsigned int w = coh_win[(size_t)qq * G + sub];
            const bool has = w != COH_NONE;
            const float4 Ew = entries[has ? (w & 0xffffffu) : (unsigned int)bc.max_entries];
            const float dx = Ew.x - qx, dy = Ew.y - qy, dz = Ew.z - qz;
            const unsigned int db = has ? __float_as_uint(dist2_exact(dx, dy, dz)) : 0xffffffffu;
            const unsigned int cb = has ? (w >> 24) : (0x100u + (unsigned int)sub);  // (empty slots rank behind, in lane order)
            int rank = 0;
#pragma unroll
            for (int j = 1; j < G; ++j) {  // (xor patterns below 8 stay inside the query's eight lanes, all of them here)
                const unsigned int od = (unsigned int)__shfl_xor((int)db, j, 64), oc = (unsigned int)__shfl_xor((int)cb, j, 64);
                rank += (od < db || (od == db && oc < cb)) ? 1 : 0;
            }
            if (rank < k) nbr[(size_t)qq * k + rank] = has ? make_float4(-dx, -dy, -dz, Ew.w) : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            // (nn_count stands: the set of accepted candidates has not changed)
        }
        active = active && !near;
        if (__builtin_amdgcn_ballot_w64(active) == 0ull) return;  // nobody left for the full search
    }
    knn_full_search<R, COH>(sp, bc, qx, qy, qz, active, qi, qq, k, lut_all[wv], cpack_all[wv], hdr[grp], nbr, nn_count, coh_state,
                            coh_win, coh_mode);
}

// The listed search proper: RR registers per lane, straight-line.  Slot r * 8 + sub of the query's list (candidate order) is
// lane `sub`'s register r; the tournament is the full search's -- ties go to the lowest register, then to the lowest slot, i.e.
// to the lowest candidate.
template <int RR, int SPEC, bool FROM_LDS>
__device__ __forceinline__ void listed_select(const pin_search_params& sp, const float4* __restrict__ entries, const unsigned int sentinel,
                                              const float qx, const float qy, const float qz, const bool go, const int cnt_list,
                                              const unsigned int* const ll, const unsigned int* __restrict__ const gl,
                                              const unsigned int (&spec)[SPEC > 0 ? SPEC : 1], const int k,
                                              const int qi, const int qq, float4* __restrict__ nbr, int* __restrict__ nn_count) {
    constexpr int G = 8;
    const int sub = threadIdx.x & (G - 1);
    unsigned int offs[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
        const int slot = r * G + sub;
        const bool has = go && slot < cnt_list;
        const int sl = has ? slot : 0;
        // (memory lists: the first SPEC registers were requested at the top of the kernel)
        const unsigned int o = FROM_LDS ? ll[sl] : (r < SPEC ? spec[r] : gl[sl]);
        offs[r] = has ? o : sentinel;
    }
    float4 E[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) E[r] = entries[offs[r]];
    unsigned int d2b[RR];
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < RR; ++r) {
        const float dx = E[r].x - qx, dy = E[r].y - qy, dz = E[r].z - qz;
        const float d2 = dist2_exact(dx, dy, dz);
        const bool acc = !(d2 > sp.max_valid_dist2);  // (the sentinel gives +inf)
        d2b[r] = acc ? __float_as_uint(d2) : 0xffffffffu;
        cnt += acc ? 1 : 0;
    }
    cnt = (int)group_sum_u32<G>((unsigned int)cnt);
    if (go && sub == 0) nn_count[qi] = cnt;
    int mine = -1;
    for (int t = 0; t < k; ++t) {
        unsigned int bd = d2b[0];
        int br = 0;
#pragma unroll
        for (int r = 1; r < RR; ++r)
            if (d2b[r] < bd) { bd = d2b[r]; br = r; }
        const unsigned int wd = group_min_u32<G>(bd);
        if (wd == 0xffffffffu) break;
        const unsigned int myc = bd == wd ? (unsigned int)(br * G + sub) : 0xffffffffu;
        const unsigned int wc = group_min_u32<G>(myc);
        if (myc == wc) {
#pragma unroll
            for (int r = 0; r < RR; ++r)
                if (r == br) d2b[r] = 0xffffffffu;
        }
        if (sub == t) mine = (int)wc;
    }
    float4 rec = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
    if (go && mine >= 0) {
        const float4 Ew = entries[FROM_LDS ? ll[mine] : gl[mine]];
        rec = make_float4(-(Ew.x - qx), -(Ew.y - qy), -(Ew.z - qz), Ew.w);
    }
    if (go && sub < k) nbr[(size_t)qq * k + sub] = rec;
}

// CANDIDATE LISTS ACROSS THE ITERATIONS OF ONE REGISTRATION (pin_gn_knn_listed; knn_brick_listed_kernel<R>).
// Tracker.tracking searches the same source points reg_iter_n times under a pose that moves by centimetres
// (utils/tracker.py:114-184), and two thirds of the full search's instructions do not depend on where inside its voxel a
// query stands: cell arithmetic, window-brick headers, occupancy bits and prefix counts turn the query's VOXEL into the entry
// offsets of its occupied candidate cells, and 44 % of the candidate cells of the bench map are empty.  So a full search
// leaves, per query, its voxel and the COMPACT list of those offsets in candidate order (<= n_cand, ~45 of 81 on the bench
// map); while the query stays in that voxel -- nearly always: a voxel is 40 cm -- a later iteration only loads the list,
// gathers the entries, measures them and runs the tournament over ceil(count / 8) registers per lane instead of R.  Same
// entries, same distance arithmetic, same acceptance test, and the slot order IS the candidate order, so the tie rule picks
// the same winner: the record is the full search's, bit for bit (tests/test_gpu_bricks.py).  A query that has changed voxel
// rebuilds its list (the set-up and candidate pass of the full search, then the listed search); queries near uncached bricks
// (map border, overflowed cache) or absurdly far out keep taking the full search with its exact probe.
template <int R, int SPEC_MAX>
__global__ __launch_bounds__(BRICK_BLOCK) void knn_brick_listed_kernel(pin_search_params sp, pin_brick_cache bc,
                                                                       const float* __restrict__ query, int n, int k,
                                                                       float* __restrict__ query_out, float4* __restrict__ nbr,
                                                                       int* __restrict__ nn_count, const double* __restrict__ state,
                                                                       int4* __restrict__ cell_state, unsigned int* __restrict__ cell_list,
                                                                       int rebuild) {
    constexpr int G = 8, CAP = R * G;
    __shared__ unsigned short lut_all[BRICK_BLOCK / 64][512];
    __shared__ unsigned int cpack_all[BRICK_BLOCK / 64][128];
    __shared__ uint4 hdr[BRICK_BLOCK / G][8];
    __shared__ unsigned int slist[BRICK_BLOCK / G][CAP];  // the list a group has just built (its lanes read each other's slots)
    if (state[PIN_GN_STATE_DONE] != 0.0) return;  // (block-uniform)
    float m[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) m[i] = (float)state[i];
    const int nd = bc.n_dilate;
    const int sub = threadIdx.x & (G - 1), grp = threadIdx.x / G;
    const int wlane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int lb = xcd_logical_block(blockIdx.x, (n * G + BRICK_BLOCK - 1) / BRICK_BLOCK);
    if ((long)lb * (BRICK_BLOCK / G) >= n) return;  // (the grid is rounded up to a multiple of 8 blocks)
    const int qi = (lb * BRICK_BLOCK + threadIdx.x) / G;
    const bool active = qi < n;
    const int qq = active ? qi : n - 1;
    float qx = query[3 * qq + 0], qy = query[3 * qq + 1], qz = query[3 * qq + 2];
    {
        const float tx = fmaf(qz, m[2], fmaf(qy, m[1], qx * m[0])) + m[3];
        const float ty = fmaf(qz, m[6], fmaf(qy, m[5], qx * m[4])) + m[7];
        const float tz = fmaf(qz, m[10], fmaf(qy, m[9], qx * m[8])) + m[11];
        qx = tx; qy = ty; qz = tz;
        if (active && sub == 0) { query_out[3 * qi + 0] = qx; query_out[3 * qi + 1] = qy; query_out[3 * qi + 2] = qz; }
    }
    float fx, fy, fz;
    {
#pragma clang fp contract(off)
        fx = floorf(__fdiv_rn(qx, sp.resolution)); fy = floorf(__fdiv_rn(qy, sp.resolution)); fz = floorf(__fdiv_rn(qz, sp.resolution));
    }
    const float lim = 536870912.f;  // 2^29, as in the full search
    const bool far = !(fabsf(fx) < lim && fabsf(fy) < lim && fabsf(fz) < lim);
    const int ix = far ? 0 : (int)fx, iy = far ? 0 : (int)fy, iz = far ? 0 : (int)fz;
    const int4 cs = rebuild ? make_int4(0, 0, 0, -2) : cell_state[qq];
    // The wave's dependent memory round trips, not its instructions, set the pace of this kernel (a wave holds 8 queries and a
    // SIMD 7 waves): the first SPEC registers' worth of the list is requested together with the voxel it belongs to, before
    // anybody knows whether the list is still valid or how long it is (the rows are allocated in full: any slot may be read)
    constexpr int SPEC = R < SPEC_MAX ? R : SPEC_MAX;
    unsigned int spec[SPEC > 0 ? SPEC : 1];
    if (!rebuild) {
#pragma unroll
        for (int r = 0; r < SPEC; ++r) spec[r] = cell_list[(size_t)qq * CAP + r * G + sub];
    } else {
#pragma unroll
        for (int r = 0; r < SPEC; ++r) spec[r] = 0u;
    }
    // cs.w: >= 0 = length of the list of voxel (x, y, z); -1 = this voxel's search needs the exact probe; -2 = nothing yet
    const bool same = !far && cs.x == ix && cs.y == iy && cs.z == iz;
    bool listed = same && cs.w >= 0;       // serve from the stored list
    bool unclean = far || (same && cs.w == -1);  // full search with the exact probe
    int cnt_list = listed ? cs.w : 0;
    bool built = false;
    const float4* __restrict__ entries = reinterpret_cast<const float4*>(bc.entries);
    const unsigned int sentinel = (unsigned int)bc.max_entries;
    unsigned int* const gl = cell_list + (size_t)qq * CAP;
    if (__builtin_amdgcn_ballot_w64(active && !listed) != 0ull) {
        // some query of this wave is new in its voxel: the full search's tables and header rows, for the whole wave
        unsigned short* const lut = lut_all[wv];
        unsigned int* const cpack = cpack_all[wv];
        reinterpret_cast<uint4*>(lut)[wlane] = reinterpret_cast<const uint4*>(KNN_LUT.v)[wlane];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ci = wlane + 64 * h, c = ci < sp.n_cand ? ci : 0;
            cpack[ci] = (unsigned int)((bc.cand_dx[3 * c] + nd) | ((bc.cand_dx[3 * c + 1] + nd) << 3) | ((bc.cand_dx[3 * c + 2] + nd) << 6));
        }
        const int b0x = (ix - nd) >> 2, b0y = (iy - nd) >> 2, b0z = (iz - nd) >> 2;
        const unsigned int P = (unsigned int)(((ix - nd) & 3) | (((iy - nd) & 3) << 3) | (((iz - nd) & 3) << 6));
        BrickInfo bi;
        bi.base = -1; bi.lo = 0; bi.hi = 0;
        if (!far && !listed) bi = dir_lookup(bc, brick_key(b0x + (sub >> 2), b0y + ((sub >> 1) & 1), b0z + (sub & 1)));
        const bool uncached = bi.base < 0;
        hdr[grp][sub] = make_uint4((unsigned int)bi.base, bi.lo, bi.hi, (unsigned int)bi.base + (unsigned int)__popc(bi.lo));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (!listed && !unclean) {  // (the eight lanes of a query agree on both)
            unclean = group_sum_u32<G>(uncached ? 1u : 0u) != 0u;
            if (!unclean) {
                const char* const hrow = reinterpret_cast<const char*>(&hdr[grp][0]);
                const unsigned int shift = (unsigned int)(wlane & 56);
                int base = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) {  // the occupied candidate cells' entry offsets, compacted in candidate order
                    const int c = r * G + sub;
                    unsigned int off = knn_cell_offset(lut, hrow, P + cpack[c & 127], sentinel);
                    if (r == R - 1) off = c < sp.n_cand ? off : sentinel;
                    const bool occ = off != sentinel;
                    const unsigned int gb = (unsigned int)(__builtin_amdgcn_ballot_w64(occ) >> shift) & 0xffu;
                    const int pos = base + __popc(gb & ((1u << sub) - 1u));
                    if (occ) { slist[grp][pos] = off; if (active) gl[pos] = off; }
                    base += __popc(gb);
                }
                cnt_list = base;
                built = true;
            }
            if (active && sub == 0) cell_state[qi] = make_int4(ix, iy, iz, unclean ? -1 : cnt_list);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // slist: written and read by different lanes of the group
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (__builtin_amdgcn_ballot_w64(active && unclean) != 0ull)
            knn_full_search<R, false>(sp, bc, qx, qy, qz, active && unclean, qi, qq, k, lut, cpack, hdr[grp], nbr, nn_count, nullptr, nullptr, 0);
    }
    const bool go = active && !unclean;
    // a wave in which some query has just built its list reads every list from LDS (the others copy theirs in): one code
    // path per wave; in the steady state nobody builds and the lists come straight from memory
    const bool from_lds = __builtin_amdgcn_ballot_w64(built) != 0ull;
    if (from_lds) {
        if (go && !built)
            for (int slot = sub; slot < cnt_list; slot += G) slist[grp][slot] = gl[slot];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // registers per lane this wave needs: the longest list of its eight queries
    int rounds;
    {
        const unsigned int rl = go ? (unsigned int)((cnt_list + G - 1) / G) : 0u;
        const unsigned int inv = row_min_u32(~rl);
        const unsigned int a = (unsigned int)__builtin_amdgcn_readlane((int)inv, 0), b = (unsigned int)__builtin_amdgcn_readlane((int)inv, 16),
                           c = (unsigned int)__builtin_amdgcn_readlane((int)inv, 32), d = (unsigned int)__builtin_amdgcn_readlane((int)inv, 48);
        rounds = (int)~min(min(a, b), min(c, d));
    }
    const unsigned int* const ll = slist[grp];
#define PIN_LISTED(RR)                                                                                                                   \
    do {                                                                                                                                 \
        if (from_lds) listed_select<RR, SPEC, true>(sp, entries, sentinel, qx, qy, qz, go, cnt_list, ll, gl, spec, k, qi, qq, nbr, nn_count);  \
        else listed_select<RR, SPEC, false>(sp, entries, sentinel, qx, qy, qz, go, cnt_list, ll, gl, spec, k, qi, qq, nbr, nn_count);          \
    } while (0)
    if (R > 4 && rounds > 8) PIN_LISTED(R);
    else if (R > 7 && rounds > 7) PIN_LISTED((R > 8 ? 8 : R));
    else if (R > 6 && rounds > 6) PIN_LISTED((R > 7 ? 7 : R));
    else if (R > 5 && rounds > 5) PIN_LISTED((R > 6 ? 6 : R));
    else if (R > 4 && rounds > 4) PIN_LISTED((R > 5 ? 5 : R));
    else PIN_LISTED((R > 4 ? 4 : R));
#undef PIN_LISTED
}

}  // namespace pin

using namespace pin;

extern "C" int64_t pin_brick_build_workspace_bytes(int32_t n_points, int32_t max_bricks) {
    return 1024 + 8 * (int64_t)(max_bricks < 0 ? 0 : max_bricks) + 8 * (int64_t)(n_points < 0 ? 0 : n_points);
}

// No second cell within `reach` cells of a cell shares its hash slot: d . (PRIME0, PRIME1, PRIME2) = 0 (mod buffer_size) has no
// solution 0 < |d|_inf <= reach.  (None for the table sizes in use -- 5e7, 1e7, 2e7, 1e5 up to reach 13 -- but the point-driven
// build is only exact when that holds, so it is checked, once per (size, reach).)
static bool brick_alias_free(long long B, int reach) {
    static long long seen_B = 0;
    static int seen_reach = 0;
    static bool seen_ok = false;
    if (B == seen_B && reach <= seen_reach && seen_ok) return true;
    if (B == seen_B && reach == seen_reach) return seen_ok;
    bool ok = B > 0;
    for (int x = -reach; x <= reach && ok; ++x)
        for (int y = -reach; y <= reach && ok; ++y)
            for (int z = -reach; z <= reach && ok; ++z) {
                if (x == 0 && y == 0 && z == 0) continue;
                const long long h = x * PRIME0 + y * PRIME1 + z * PRIME2;
                if (h % B == 0) ok = false;
            }
    seen_B = B; seen_reach = reach; seen_ok = ok;
    return ok;
}

extern "C" int pin_brick_build(const pin_search_params* sp, const pin_brick_cache* bc, int32_t* counters_out,
                               void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(sp && bc && counters_out, "NULL pointer");
    PIN_CHECK_ARG(sp->n_points > 0 && sp->table && sp->pos4, "empty map");
    PIN_CHECK_ARG(bc->dir_keys && bc->dir_vals && bc->brick_keys && bc->brick_mask && bc->brick_base && bc->entries && bc->dir_pack,
                  "brick cache buffers NULL");
    PIN_CHECK_ARG((bc->dir_mask & (bc->dir_mask + 1)) == 0 && bc->dir_mask > 0, "directory size must be a power of two");
    PIN_CHECK_ARG(bc->n_dilate >= 0 && bc->n_dilate <= 2, "n_dilate must be in [0, 2]");
    hipStream_t s = as_stream(stream);
    const int D = (int)bc->dir_mask + 1;
    // bc->build_grid = g > 0: at most g blocks per launch of the build, each walking its share of the work.  The build is
    // bound by random table / directory probes, not by arithmetic: a few waves per compute unit with eight probes in flight
    // per lane keep the memory system nearly as busy as 8 700 blocks do, and the dispatcher's slots stay free for the small
    // launches of the stream that runs beside it (pool filter, certainty query, the mapper's set-up).  0: one block per unit.
    // (ONE switch: the field.  The drop-in sets it from PIN_BRICK_BUILD_GRID, parsed and validated once in Python.)
    PIN_CHECK_ARG(bc->build_grid >= 0, "build_grid < 0");
    const int grid_cap = bc->build_grid;
    auto grid_of = [grid_cap](int units) { return dim3((unsigned)(grid_cap > 0 ? (units < grid_cap ? units : grid_cap) : units)); };
    hipLaunchKernelGGL(brick_clear_kernel, grid_of(cdiv(D, 256)), dim3(256), 0, s,
                       reinterpret_cast<unsigned long long*>(bc->dir_keys), D, counters_out,
                       reinterpret_cast<float4*>(bc->entries), bc->max_entries);
    // a probing query sits within (n+1) cells (per axis) of the cell centre and accepts points
    // within sqrt(max_valid_dist2): anything farther from the cell centre can never be accepted
    const float reach = (bc->n_dilate + 1.0f) * sp->resolution * 1.7320508f + sqrtf(sp->max_valid_dist2);
    const float prune = reach * reach * 1.02f;
    static const bool cells_forced = [] { const char* e = getenv("PIN_BRICK_BUILD"); return e && e[0] == 'c'; }();
    const bool by_points = bc->build_ws != nullptr && !cells_forced && bc->max_bricks < (1 << 24) &&
                           bc->build_ws_bytes >= pin_brick_build_workspace_bytes(sp->n_points, bc->max_bricks) &&
                           brick_alias_free(sp->buffer_size, (int)ceilf(reach * 1.01f / sp->resolution) + 2);
    if (by_points) {
        Carver cv{static_cast<char*>(bc->build_ws), static_cast<char*>(bc->build_ws) + bc->build_ws_bytes};
        unsigned long long* pmask = cv.take<unsigned long long>(bc->max_bricks);
        int2* tmp = cv.take<int2>(sp->n_points);
        const int n = bc->n_dilate;
        BrickDirMasks dm;
        for (int d = 0; d < 27; ++d) {  // cells of a brick within n of the side (d / 9 - 1, (d / 3) % 3 - 1, d % 3 - 1)
            const int dir[3] = {d / 9 - 1, (d / 3) % 3 - 1, d % 3 - 1};
            unsigned long long m = 0;
            for (int bit = 0; bit < 64; ++bit) {
                const int c[3] = {bit >> 4, (bit >> 2) & 3, bit & 3};
                bool in = true;
                for (int a = 0; a < 3; ++a) in = in && (dir[a] == 0 || (dir[a] < 0 ? c[a] < n : c[a] > 3 - n));
                if (in) m |= 1ull << bit;
            }
            dm.m[d] = m;
        }
        const int pb = cdiv(sp->n_points, 256);
        hipLaunchKernelGGL(brick_zero_masks_kernel, dim3(cdiv(bc->max_bricks, 256)), dim3(256), 0, s, pmask,
                           reinterpret_cast<unsigned long long*>(bc->brick_mask), bc->max_bricks);
        hipLaunchKernelGGL(brick_own_kernel, dim3(pb), dim3(256), 0, s, *bc, *sp, counters_out);
        hipLaunchKernelGGL(brick_pmask_kernel, dim3(pb), dim3(256), 0, s, *bc, *sp, pmask, counters_out);
        hipLaunchKernelGGL(brick_dilate_kernel, dim3(cdiv((long)bc->max_bricks * 27, 256)), dim3(256), 0, s, *bc, pmask, dm, counters_out);
        hipLaunchKernelGGL(brick_point_mask_kernel, dim3(pb), dim3(256), 0, s, *bc, *sp, prune, tmp);
        hipLaunchKernelGGL(brick_bases_kernel, dim3(cdiv(bc->max_bricks, 256)), dim3(256), 0, s, *bc, counters_out);
        hipLaunchKernelGGL(brick_point_entries_kernel, dim3(pb), dim3(256), 0, s, *bc, *sp, tmp);
    } else {
        const int mark_units = cdiv(sp->n_points, 256), fill_units = cdiv(bc->max_bricks, 4 * FILL_PER_WAVE);
        hipLaunchKernelGGL(brick_mark_kernel, grid_of(mark_units), dim3(256), 0, s, *bc, *sp, bc->n_dilate, counters_out, mark_units);
        hipLaunchKernelGGL(brick_fill_kernel, grid_of(fill_units), dim3(256), 0, s, *bc, *sp, prune, counters_out, fill_units);
    }
    hipLaunchKernelGGL(brick_publish_kernel, grid_of(cdiv((long)bc->dir_mask + 1, 256)), dim3(256), 0, s, *bc);
    PIN_CHECK_LAUNCH();
    return 0;
}

static int knn_bricks(const pin_search_params* sp, const pin_brick_cache* bc, const float* query, int32_t n, int32_t k,
                      const float* pose_host, const double* state, float* query_out, float* nbr_out,
                      int32_t* nn_count_out, void* stream, float* coh_state = nullptr, uint32_t* coh_win = nullptr,
                      int coh_mode = 0);

extern "C" int pin_knn_query_bricks(const pin_search_params* sp, const pin_brick_cache* bc, const float* query,
                                    int32_t n, int32_t k, const float* pose_host, float* query_out, float* nbr_out,
                                    int32_t* nn_count_out, void* stream) {
    PIN_ENTER();
    return knn_bricks(sp, bc, query, n, k, pose_host, nullptr, query_out, nbr_out, nn_count_out, stream);
}

namespace pin {
int knn_bricks_dev(const pin_search_params* sp, const pin_brick_cache* bc, const float* query, int32_t n, int32_t k,
                   const double* state, float* query_out, float* nbr_out, int32_t* nn_count_out, void* stream) {
    return knn_bricks(sp, bc, query, n, k, nullptr, state, query_out, nbr_out, nn_count_out, stream);
}
}  // namespace pin

extern "C" int pin_gn_knn_coherent(const pin_search_params* sp, const pin_brick_cache* bc, const float* src, int32_t n, int32_t k,
                                   const double* state, float* cur_out, float* nbr_out, int32_t* nn_count_out,
                                   float* coh_state, uint32_t* coh_win, int32_t iteration, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(state && cur_out && bc, "state / cur_out / brick cache NULL");
    PIN_CHECK_ARG(iteration >= 0, "iteration < 0");
    PIN_CHECK_ARG(coh_state && coh_win, "coherent state NULL");
    return knn_bricks(sp, bc, src, n, k, nullptr, state, cur_out, nbr_out, nn_count_out, stream, coh_state, coh_win,
                      iteration == 0 ? 1 : 2);
}

extern "C" int32_t pin_knn_list_stride(int32_t n_cand) {
    const int rounds = cdiv(n_cand, 8);
    return 8 * (rounds <= 4 ? 4 : rounds <= 5 ? 5 : rounds <= 8 ? 8 : rounds <= 11 ? 11 : 16);
}

extern "C" int pin_gn_knn_listed(const pin_search_params* sp, const pin_brick_cache* bc, const float* src, int32_t n, int32_t k,
                                 const double* state, float* cur_out, float* nbr_out, int32_t* nn_count_out, int32_t* cell_state,
                                 uint32_t* cell_list, int32_t rebuild, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(sp && bc && state && cur_out && cell_state && cell_list, "NULL pointer");
    PIN_CHECK_ARG(n >= 0 && k >= 1 && k <= PIN_MAX_K, "bad sizes");
    if (n == 0) return 0;
    PIN_CHECK_ARG(src && nbr_out && nn_count_out && bc->cand_dx && bc->dir_pack, "NULL pointer");
    PIN_CHECK_ARG(sp->n_points > 0 && sp->n_cand > 0 && sp->n_cand <= 125 && bc->n_dilate >= 0 && bc->n_dilate <= 2, "bad search state");
    float4* nbr = reinterpret_cast<float4*>(nbr_out);
    hipStream_t s = as_stream(stream);
    xcd_mode_init();
    const dim3 grid(xcd_grid(cdiv((long)n * 8, BRICK_BLOCK))), block(BRICK_BLOCK);
    const int rounds = cdiv(sp->n_cand, 8);
    // (PIN_KNN_SPEC=0: no speculative list loads -- one more dependent round trip per wave, 8 registers fewer)
    static const bool spec_on = [] { const char* e = getenv("PIN_KNN_SPEC"); return e && e[0] == '1'; }();
#define PIN_LAUNCH_KL(R)                                                                                                                \
    do {                                                                                                                                \
        if (spec_on)                                                                                                                    \
            hipLaunchKernelGGL((knn_brick_listed_kernel<R, 8>), grid, block, 0, s, *sp, *bc, src, n, k, cur_out, nbr, nn_count_out, state, \
                               reinterpret_cast<int4*>(cell_state), cell_list, rebuild);                                                \
        else                                                                                                                            \
            hipLaunchKernelGGL((knn_brick_listed_kernel<R, 0>), grid, block, 0, s, *sp, *bc, src, n, k, cur_out, nbr, nn_count_out, state, \
                               reinterpret_cast<int4*>(cell_state), cell_list, rebuild);                                                \
    } while (0)
    if (rounds <= 4) PIN_LAUNCH_KL(4);
    else if (rounds <= 5) PIN_LAUNCH_KL(5);
    else if (rounds <= 8) PIN_LAUNCH_KL(8);
    else if (rounds <= 11) PIN_LAUNCH_KL(11);
    else PIN_LAUNCH_KL(16);
#undef PIN_LAUNCH_KL
    PIN_CHECK_LAUNCH();
    return 0;
}

static int knn_bricks(const pin_search_params* sp, const pin_brick_cache* bc, const float* query, int32_t n, int32_t k,
                      const float* pose_host, const double* state, float* query_out, float* nbr_out,
                      int32_t* nn_count_out, void* stream, float* coh_state, uint32_t* coh_win, int coh_mode) {
    PIN_CHECK_ARG(sp && bc, "NULL params");
    PIN_CHECK_ARG(n >= 0, "n < 0");
    PIN_CHECK_ARG(k >= 1 && k <= PIN_MAX_K, "k must be in [1, 8]");
    if (n == 0) return 0;
    PIN_CHECK_ARG(query && nbr_out && nn_count_out && bc->cand_dx && bc->dir_pack, "NULL pointer");
    PIN_CHECK_ARG(sp->n_points > 0 && sp->n_cand > 0 && sp->n_cand <= 256, "bad search state");
    PoseB pose;
    pose.on = pose_host != nullptr;
    if (pose.on) memcpy(pose.m, pose_host, sizeof(pose.m));
    float4* nbr = reinterpret_cast<float4*>(nbr_out);
    hipStream_t s = as_stream(stream);
    // eight lanes per query (the per-query setup and the k selection rounds are shared by 8 queries per wave)
    PIN_CHECK_ARG(bc->n_dilate >= 0 && bc->n_dilate <= 2 && sp->n_cand <= 125, "the brick cache covers num_nei_cells <= 2");
    xcd_mode_init();
    const dim3 grid(xcd_grid(cdiv((long)n * 8, BRICK_BLOCK))), block(BRICK_BLOCK);
    const int rounds = cdiv(sp->n_cand, 8);
    float4* cst = reinterpret_cast<float4*>(coh_state);
#define PIN_LAUNCH_KB(R)                                                                                                         \
    do {                                                                                                                         \
        if (coh_mode != 0)                                                                                                       \
            hipLaunchKernelGGL((knn_brick_kernel<R, true>), grid, block, 0, s, *sp, *bc, query, n, k, pose, query_out, nbr,      \
                               nn_count_out, state, cst, coh_win, coh_mode);                                                     \
        else                                                                                                                     \
            hipLaunchKernelGGL((knn_brick_kernel<R, false>), grid, block, 0, s, *sp, *bc, query, n, k, pose, query_out, nbr,     \
                               nn_count_out, state, cst, coh_win, 0);                                                            \
    } while (0)
    if (rounds <= 4) PIN_LAUNCH_KB(4);
    else if (rounds <= 5) PIN_LAUNCH_KB(5);
    else if (rounds <= 8) PIN_LAUNCH_KB(8);
    else if (rounds <= 11) PIN_LAUNCH_KB(11);
    else PIN_LAUNCH_KB(16);
#undef PIN_LAUNCH_KB
    PIN_CHECK_LAUNCH();
    return 0;
}

// pin_warmup (common.hip): asking for a kernel's attributes makes the runtime load this translation unit's code object now
// instead of inside the first frame that launches one of its kernels
namespace pin {
int pin_warm_brick() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&brick_clear_kernel)) == hipSuccess ? 0 : -2;
}
}  // namespace pin
