// Voxel-hash radius search and k-nearest selection over neural points (gfx950).
//
// Replaces NeuralPoints.radius_neighborhood_search (model/neural_points.py:950-1009) and the
// sort/top-k at the head of NeuralPoints.query_feature (neural_points.py:573-589).
//
// Design: the path is a chain of dependent random gathers (hash slot -> point index ->
// position), i.e. latency/HBM bound.  A group of 16 lanes owns one query; candidate cells
// are strided over the lanes so that a wave keeps 64 * R independent probes in flight, the
// [N, Kc, *] temporaries of the reference are never materialised, and the k best are picked
// by a k-round tournament on packed (d2 bits, candidate id) keys with xor-shuffles inside
// the 16-lane group -- no LDS, no sort.
#include "pin_common.h"

namespace pin {

constexpr int GROUP = 16;       // lanes per query
constexpr int KNN_BLOCK = 256;  // 16 queries per workgroup
constexpr unsigned long long KEY_NONE = ~0ull;

struct Pose34 {
    float m[12];
    int on;
};

template <int R>
__global__ __launch_bounds__(KNN_BLOCK) void knn_query_kernel(
    pin_search_params sp, const float* __restrict__ query, int n, int k, Pose34 pose,
    float* __restrict__ query_out, float4* __restrict__ nbr, int* __restrict__ nn_count,
    const double* __restrict__ state) {
    if (state != nullptr) {  // device-resident GN loop: pose from the state, stop when done
        if (state[PIN_GN_STATE_DONE] != 0.0) return;
#pragma unroll
        for (int i = 0; i < 12; ++i) pose.m[i] = (float)state[i];
        pose.on = 1;
    }
    const int sub = threadIdx.x & (GROUP - 1);
    const int qi = (blockIdx.x * KNN_BLOCK + threadIdx.x) / GROUP;
    const bool active = qi < n;
    const int qq = active ? qi : n - 1;

    float qx = query[3 * qq + 0], qy = query[3 * qq + 1], qz = query[3 * qq + 2];
    if (pose.on) {
        const float* m = pose.m;
        float tx = fmaf(qz, m[2], fmaf(qy, m[1], qx * m[0])) + m[3];
        float ty = fmaf(qz, m[6], fmaf(qy, m[5], qx * m[4])) + m[7];
        float tz = fmaf(qz, m[10], fmaf(qy, m[9], qx * m[8])) + m[11];
        qx = tx; qy = ty; qz = tz;
        if (query_out != nullptr && active && sub == 0) {
            query_out[3 * qi + 0] = qx; query_out[3 * qi + 1] = qy; query_out[3 * qi + 2] = qz;
        }
    }

    const uint32_t B = (uint32_t)sp.buffer_size;
    const uint32_t base = hash_base(qx, qy, qz, sp.resolution, sp.buffer_size);
    const float4* __restrict__ pos4 = reinterpret_cast<const float4*>(sp.pos4);
    const bool tfilter = sp.travel_dist != nullptr;
    const float d_cur = tfilter ? sp.travel_dist[sp.cur_ts] : 0.f;

    // phase 1: hash probes (R independent loads per lane)
    int j[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int c = r * GROUP + sub;
        j[r] = -1;
        if (c < sp.n_cand) {
            uint32_t s = base + (uint32_t)sp.cand_off[c];
            if (s >= B) s -= B;
            j[r] = sp.table[s];
        }
    }
    // phase 2: position gathers for occupied cells
    float4 P[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        P[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j[r] >= 0) P[r] = pos4[j[r]];
    }
    // phase 3: filters, distance, index-space mapping
    unsigned long long key[R];
    int li[R];
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        key[r] = KEY_NONE;
        li[r] = -1;
        if (j[r] >= 0) {
            bool ok = true;
            if (tfilter) {
                const float dts = sp.travel_dist[__float_as_int(P[r].w)];
                ok = fabsf(d_cur - dts) < sp.diff_travel_dist_local;
            }
            const float dx = P[r].x - qx, dy = P[r].y - qy, dz = P[r].z - qz;
            const float d2 = dist2_exact(dx, dy, dz);
            ok = ok && !(d2 > sp.max_valid_dist2);
            int l = j[r];
            if (ok && sp.global2local != nullptr) {
                l = sp.global2local[j[r]];
                if (l == PIN_NONLOCAL) l = 1 | PIN_NBR_QUIRK_BIT;  // reference maps non-local -> 1
            }
            if (ok && l >= 0) {
                key[r] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)(r * GROUP + sub);
                li[r] = l;
                P[r].x = -dx; P[r].y = -dy; P[r].z = -dz;  // q - P, exact
                ++cnt;
            }
        }
    }
#pragma unroll
    for (int o = GROUP / 2; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, GROUP);
    if (active && sub == 0) nn_count[qi] = cnt;

    // phase 4: k-round tournament; the owner of each winner stores it
    float4* __restrict__ out = nbr + (size_t)qq * k;
    for (int t = 0; t < k; ++t) {
        unsigned long long best = KEY_NONE;
#pragma unroll
        for (int r = 0; r < R; ++r) best = key[r] < best ? key[r] : best;
        unsigned long long win = best;
#pragma unroll
        for (int o = GROUP / 2; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(win, o, GROUP);
            win = other < win ? other : win;
        }
        if (win == KEY_NONE) {  // fewer than k valid neighbours: pad with (0,0,0,-1)
            if (active && sub == 0)
                for (int u = t; u < k; ++u) out[u] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            break;
        }
        if (best == win) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (key[r] == win) {
                    if (active) out[t] = make_float4(P[r].x, P[r].y, P[r].z, __int_as_float(li[r]));
                    key[r] = KEY_NONE;
                }
            }
        }
    }
}

// One thread per (query, candidate): exact [N, Kc] outputs of radius_neighborhood_search.
__global__ __launch_bounds__(256) void radius_search_kernel(
    pin_search_params sp, const float* __restrict__ query, long total, float* __restrict__ d2_out,
    long long* __restrict__ idx_out) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= total) return;
    const int qi = (int)(tid / sp.n_cand);
    const int c = (int)(tid - (long)qi * sp.n_cand);
    const float qx = query[3 * qi + 0], qy = query[3 * qi + 1], qz = query[3 * qi + 2];
    const uint32_t B = (uint32_t)sp.buffer_size;
    uint32_t s = hash_base(qx, qy, qz, sp.resolution, sp.buffer_size) + (uint32_t)sp.cand_off[c];
    if (s >= B) s -= B;
    int j = sp.n_points > 0 ? sp.table[s] : -1;
    float d2 = sp.max_valid_dist2;
    if (j >= 0) {
        const float4 P = reinterpret_cast<const float4*>(sp.pos4)[j];
        bool ok = true;
        if (sp.travel_dist != nullptr) {
            const float dts = sp.travel_dist[__float_as_int(P.w)];
            ok = fabsf(sp.travel_dist[sp.cur_ts] - dts) < sp.diff_travel_dist_local;
        }
        if (ok) {
            d2 = dist2_exact(P.x - qx, P.y - qy, P.z - qz);
            if (d2 > sp.max_valid_dist2) j = -1;  // d2 itself is returned as computed
        } else {
            j = -1;
        }
    }
    d2_out[tid] = d2;
    idx_out[tid] = j;
}

__global__ void pack_positions_kernel(const float* __restrict__ pos, const int* __restrict__ ts,
                                      int first, int n, float4* __restrict__ pos4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int g = first + i;
    pos4[g] = make_float4(pos[3 * g], pos[3 * g + 1], pos[3 * g + 2], __int_as_float(ts[g]));
}

static int check_search(const pin_search_params* sp) {
    PIN_CHECK_ARG(sp != nullptr, "search params NULL");
    PIN_CHECK_ARG(sp->buffer_size > 0 && sp->buffer_size < (1LL << 31), "buffer_size must be in (0, 2^31)");
    PIN_CHECK_ARG(sp->n_cand > 0 && sp->n_cand <= 256, "n_cand must be in [1, 256]");
    PIN_CHECK_ARG(sp->n_points >= 0, "n_points < 0");
    PIN_CHECK_ARG(sp->table && sp->cand_off, "table / cand_off NULL");
    PIN_CHECK_ARG(sp->n_points == 0 || sp->pos4, "pos4 NULL");
    return 0;
}

}  // namespace pin

using namespace pin;

extern "C" int pin_candidate_offsets(const int32_t* dx, int32_t n_cand, int64_t B, int32_t* out) {
    PIN_CHECK_ARG(dx && out && n_cand > 0 && B > 0, "bad arguments");
    for (int c = 0; c < n_cand; ++c) {
        long long h = dx[3 * c] * PRIME0 + dx[3 * c + 1] * PRIME1 + dx[3 * c + 2] * PRIME2;
        long long m = h % B;
        if (m < 0) m += B;
        out[c] = (int32_t)m;
    }
    return 0;
}

extern "C" int pin_pack_positions(const float* pos, const int32_t* ts_create, int32_t first, int32_t n,
                                  float* pos4, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n >= 0 && first >= 0, "negative size");
    if (n == 0) return 0;
    PIN_CHECK_ARG(pos && ts_create && pos4, "NULL pointer");
    hipLaunchKernelGGL(pack_positions_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), pos,
                       ts_create, first, n, reinterpret_cast<float4*>(pos4));
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_radius_search(const pin_search_params* sp, const float* query, int32_t n, float* d2_out,
                                 int64_t* idx_out, void* stream) {
    PIN_ENTER();
    if (int e = check_search(sp)) return e;
    PIN_CHECK_ARG(n >= 0, "n < 0");
    if (n == 0) return 0;
    PIN_CHECK_ARG(query && d2_out && idx_out, "NULL pointer");
    const long total = (long)n * sp->n_cand;
    hipLaunchKernelGGL(radius_search_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), *sp,
                       query, total, d2_out, reinterpret_cast<long long*>(idx_out));
    PIN_CHECK_LAUNCH();
    return 0;
}

static int knn_direct(const pin_search_params* sp, const float* query, int32_t n, int32_t k, const float* pose_host,
                      const double* state, float* query_out, float* nbr_out, int32_t* nn_count_out, void* stream);

extern "C" int pin_knn_query(const pin_search_params* sp, const float* query, int32_t n, int32_t k,
                             const float* pose_host, float* query_out, float* nbr_out,
                             int32_t* nn_count_out, void* stream) {
    PIN_ENTER();
    return knn_direct(sp, query, n, k, pose_host, nullptr, query_out, nbr_out, nn_count_out, stream);
}

namespace pin {
int knn_direct_dev(const pin_search_params* sp, const float* query, int32_t n, int32_t k, const double* state,
                   float* query_out, float* nbr_out, int32_t* nn_count_out, void* stream) {
    return knn_direct(sp, query, n, k, nullptr, state, query_out, nbr_out, nn_count_out, stream);
}
}  // namespace pin

static int knn_direct(const pin_search_params* sp, const float* query, int32_t n, int32_t k, const float* pose_host,
                      const double* state, float* query_out, float* nbr_out, int32_t* nn_count_out, void* stream) {
    if (int e = check_search(sp)) return e;
    PIN_CHECK_ARG(n >= 0, "n < 0");
    PIN_CHECK_ARG(k >= 1 && k <= PIN_MAX_K, "k must be in [1, 8]");
    if (n == 0) return 0;
    PIN_CHECK_ARG(query && nbr_out && nn_count_out, "NULL pointer");
    PIN_CHECK_ARG(sp->n_points > 0, "empty map");
    Pose34 pose;
    pose.on = pose_host != nullptr;
    if (pose.on) memcpy(pose.m, pose_host, sizeof(pose.m));
    const dim3 grid(cdiv((long)n * GROUP, KNN_BLOCK)), block(KNN_BLOCK);
    float4* nbr = reinterpret_cast<float4*>(nbr_out);
    hipStream_t s = as_stream(stream);
    const int rounds = cdiv(sp->n_cand, GROUP);
#define PIN_LAUNCH_KNN(R) \
    hipLaunchKernelGGL(knn_query_kernel<R>, grid, block, 0, s, *sp, query, n, k, pose, query_out, nbr, nn_count_out, state)
    if (rounds <= 3) PIN_LAUNCH_KNN(3);
    else if (rounds <= 6) PIN_LAUNCH_KNN(6);
    else if (rounds <= 10) PIN_LAUNCH_KNN(10);
    else PIN_LAUNCH_KNN(16);
#undef PIN_LAUNCH_KNN
    PIN_CHECK_LAUNCH();
    return 0;
}

// pin_warmup (common.hip): asking for a kernel's attributes makes the runtime load this translation unit's code object now
// instead of inside the first frame that launches one of its kernels
namespace pin {
int pin_warm_knn() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&pack_positions_kernel)) == hipSuccess ? 0 : -2;
}
}  // namespace pin
