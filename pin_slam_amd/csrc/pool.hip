// Mapper.process_frame data path on gfx950 (once per frame, HBM-streaming):
//   DataSampler.sample                   utils/data_sampler.py:18-260   (K12: 7-8 samples per ray, ray-wise order)
//   pool append + transform_torch        utils/mapper.py:275-300        (fused into K12: samples land at the pool tail)
//   distance window + random discard     utils/mapper.py:303-360        (K13: mask, discard, ordered compaction of all pools)
//   NeuralPoints.query_certainty         model/neural_points.py:1011    (K14)
//   new-sample index                     utils/mapper.py:405-416        (K14b: ordered index compaction)
// float32 operation order follows the reference's torch expressions one by one (no FMA
// contraction except where torch's CPU kernels fuse: linalg.norm accumulates with FMAs).
#include "pin_common.h"
#include "compact.h"

// every float expression below is evaluated operation by operation like the torch ops it
// restates; FMAs appear only where written explicitly (fmaf)
#pragma clang fp contract(off)

namespace pin {

struct SampleConsts {  // python-double config values cast to float32 where torch casts them
    int S, Ff, Fb, A;
    float range, half, begin, end_dist;
    int dist_weight_on;
    float w_c1, max_range, w_scale;
    int dropoff_on;
    float d_max, d_diff;
    int C, frame_id;
    float pose[12];
};

// ---- K12 ------------------------------------------------------------------------------------
__global__ __launch_bounds__(MB) void sample_rays_kernel(SampleConsts sc, const float* __restrict__ points,
                                                         const float* __restrict__ colors, int stride, int n,
                                                         const float* __restrict__ rnd_s, const float* __restrict__ rnd_f,
                                                         const float* __restrict__ rnd_b, pin_pool_arrays out,
                                                         const int* __restrict__ sem_labels) {
#pragma clang fp contract(off)
    const long idx = (long)blockIdx.x * MB + threadIdx.x;
    if (idx >= (long)n * sc.A) return;
    const int i = (int)(idx / sc.A), j = (int)(idx - (long)i * sc.A);
    const float* P = points + (size_t)i * stride;
    const float x = P[0], y = P[1], z = P[2];
    // torch.linalg.norm (CPU): FMA chain, then a correctly rounded sqrt.  v_sqrt_f32 is a 1-ulp
    // instruction and hipcc emits it bare; sqrt in float64 rounded to float32 is exact rounding
    // (53 >= 2*24 + 2 bits).
    const float dist = (float)sqrt((double)__fmaf_rn(z, z, __fmaf_rn(y, y, x * x)));
    float disp, ratio;
    if (j == 0) {  // the measured point itself
        disp = 0.0f; ratio = 1.0f;
    } else if (j <= sc.S) {  // close-to-surface, N(0, range^2) along the ray
        disp = rnd_s[(size_t)(j - 1) * n + i] * sc.range;
        ratio = __fdiv_rn(disp, dist) + 1.0f;
    } else if (j <= sc.S + sc.Ff) {  // free space in front of the surface
        const float fmax = 1.0f - __fdiv_rn(sc.half, dist);
        const float fdiff = fmax - sc.begin;
        ratio = rnd_f[(size_t)(j - 1 - sc.S) * n + i] * fdiff + sc.begin;
        disp = (ratio - 1.0f) * dist;
    } else {  // free space behind the surface
        const float bmax = __fdiv_rn(sc.end_dist, dist) + 1.0f;
        const float bmin = 1.0f + __fdiv_rn(sc.half, dist);
        const float bdiff = bmax - bmin;
        ratio = rnd_b[(size_t)(j - 1 - sc.S - sc.Ff) * n + i] * bdiff + bmin;
        disp = (ratio - 1.0f) * dist;
    }
    const float cx = x * ratio, cy = y * ratio, cz = z * ratio;
    float w = 1.0f;
    const bool surface = j <= sc.S;
    if (surface && sc.dist_weight_on) w = sc.w_c1 - __fdiv_rn(dist, sc.max_range) * sc.w_scale;
    if (sc.dropoff_on) {
        float dw = __fdiv_rn(sc.d_max - disp, sc.d_diff);
        dw = fminf(fmaxf(dw, 0.0f), 1.0f);
        dw = dw * 0.8f + 0.2f;
        w = w * dw;
    }
    if (!surface) w = w * -1.0f;  // sign flags free-space samples
    const size_t o = (size_t)idx;
    out.coord[3 * o] = cx; out.coord[3 * o + 1] = cy; out.coord[3 * o + 2] = cz;
    const float* m = sc.pose;  // transform_torch: homogeneous point x float32(T)^T
    out.global_coord[3 * o] = fmaf(cz, m[2], fmaf(cy, m[1], cx * m[0])) + m[3];
    out.global_coord[3 * o + 1] = fmaf(cz, m[6], fmaf(cy, m[5], cx * m[4])) + m[7];
    out.global_coord[3 * o + 2] = fmaf(cz, m[10], fmaf(cy, m[9], cx * m[8])) + m[11];
    out.sdf_label[o] = -disp;
    out.weight[o] = w;
    out.ts[o] = sc.frame_id;
    if (sc.C > 0) {
        for (int c = 0; c < sc.C; ++c) out.color[o * sc.C + c] = surface ? colors[(size_t)i * stride + c] : 0.0f;
    }
    // semantic label: the point's for the measured point and its close-to-surface samples, 0 (free space) for the others
    if (out.sem_label != nullptr) out.sem_label[o] = (surface && sem_labels != nullptr) ? sem_labels[i] : 0;
}

// ---- K13 ------------------------------------------------------------------------------------
// mask[i] = ||global_i - origin||^2 < r^2 in float64 (float32 pool - float64 pose column promotes)
__global__ __launch_bounds__(MB) void pool_window_kernel(const float* __restrict__ g, int n, double ox, double oy, double oz,
                                                         double r2, unsigned char* __restrict__ mask,
                                                         int* __restrict__ block_cnt) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * MB + threadIdx.x;
    bool f = false;
    if (i < n) {
        const double rx = (double)g[3 * (size_t)i] - ox, ry = (double)g[3 * (size_t)i + 1] - oy,
                     rz = (double)g[3 * (size_t)i + 2] - oz;
        const double d2 = (rx * rx + ry * ry) + rz * rz;
        f = d2 < r2;
        mask[i] = f ? 1 : 0;
    }
    int total;
    block_flag_scan(f, total);
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = total;
}

// true_index[rank] = i for kept elements (torch.nonzero(filter_mask), mapper.py:310)
__global__ __launch_bounds__(MB) void pool_true_index_kernel(const unsigned char* __restrict__ mask, int n,
                                                             const int* __restrict__ block_off,
                                                             int* __restrict__ true_index) {
    const int i = blockIdx.x * MB + threadIdx.x;
    const bool f = i < n && mask[i] != 0;
    int total;
    const int ex = block_flag_scan(f, total);
    if (f) true_index[block_off[blockIdx.x] + ex] = i;
}

__global__ __launch_bounds__(MB) void pool_discard_kernel(unsigned char* __restrict__ mask, const int* __restrict__ true_index,
                                                          const long long* __restrict__ discard, int nd) {
    const int d = blockIdx.x * MB + threadIdx.x;
    if (d < nd) mask[true_index[discard[d]]] = 0;  // duplicates write the same value
}

__global__ __launch_bounds__(MB) void pool_tail_count_kernel(const unsigned char* __restrict__ mask, int first, int n,
                                                             int* __restrict__ cnt) {
    __shared__ int red[MB / 64];
    int c = 0;
    for (int i = first + blockIdx.x * MB + threadIdx.x; i < n; i += gridDim.x * MB) c += mask[i] != 0 ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {  // one atomic per block, grid capped at 128 blocks: same-address atomics serialise
        int t = 0;
#pragma unroll
        for (int w = 0; w < MB / 64; ++w) t += red[w];
        if (t) atomicAdd(cnt, t);
    }
}

__global__ __launch_bounds__(MB) void pool_scatter_kernel(pin_pool_arrays src, pin_pool_arrays dst,
                                                          const unsigned char* __restrict__ mask, int n,
                                                          const int* __restrict__ block_off) {
    const int i = blockIdx.x * MB + threadIdx.x;
    const bool f = i < n && mask[i] != 0;
    int total;
    const int ex = block_flag_scan(f, total);
    if (!f) return;
    const size_t s = (size_t)i, d = (size_t)(block_off[blockIdx.x] + ex);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        dst.coord[3 * d + a] = src.coord[3 * s + a];
        dst.global_coord[3 * d + a] = src.global_coord[3 * s + a];
    }
    dst.sdf_label[d] = src.sdf_label[s];
    dst.weight[d] = src.weight[s];
    dst.ts[d] = src.ts[s];
    if (src.color_channels > 0)
        for (int c = 0; c < src.color_channels; ++c) dst.color[d * src.color_channels + c] = src.color[s * src.color_channels + c];
    if (src.sem_label != nullptr) dst.sem_label[d] = src.sem_label[s];
}

// ---- K14 ------------------------------------------------------------------------------------
__global__ __launch_bounds__(MB) void query_certainty_kernel(pin_search_params sp, const float* __restrict__ certainty,
                                                             const float* __restrict__ query, int n,
                                                             float* __restrict__ out) {
    const int qi = blockIdx.x * MB + threadIdx.x;
    if (qi >= n) return;
    const float qx = query[3 * (size_t)qi], qy = query[3 * (size_t)qi + 1], qz = query[3 * (size_t)qi + 2];
    const uint32_t B = (uint32_t)sp.buffer_size;
    const uint32_t base = hash_base(qx, qy, qz, sp.resolution, sp.buffer_size);
    float best = 0.0f;  // certainty[idx < 0] = 0, then max over the candidates
    if (sp.n_points > 0) {
        for (int c = 0; c < sp.n_cand; ++c) {
            uint32_t s = base + (uint32_t)sp.cand_off[c];
            if (s >= B) s -= B;
            const int j = sp.table[s];
            if (j < 0) continue;
            const float4 P = reinterpret_cast<const float4*>(sp.pos4)[j];
            const float d2 = dist2_exact(P.x - qx, P.y - qy, P.z - qz);
            if (d2 > sp.max_valid_dist2) continue;
            best = fmaxf(best, certainty[j]);
        }
    }
    out[qi] = best;
}

__global__ __launch_bounds__(MB) void new_sample_flags_kernel(const float* __restrict__ cert, const float* __restrict__ label,
                                                              int n, float cert_thre, float label_thre,
                                                              unsigned char* __restrict__ flags, int* __restrict__ block_cnt) {
    const int i = blockIdx.x * MB + threadIdx.x;
    const bool f = i < n && cert[i] < cert_thre && fabsf(label[i]) < label_thre;
    if (i < n) flags[i] = f ? 1 : 0;
    int total;
    block_flag_scan(f, total);
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = total;
}

__global__ __launch_bounds__(MB) void new_sample_index_kernel(const unsigned char* __restrict__ flags, int n,
                                                              const int* __restrict__ block_off, long long offset,
                                                              long long* __restrict__ idx_out) {
    const int i = blockIdx.x * MB + threadIdx.x;
    const bool f = i < n && flags[i] != 0;
    int total;
    const int ex = block_flag_scan(f, total);
    if (f) idx_out[block_off[blockIdx.x] + ex] = offset + i;
}

// rows[|label| < thr] (update_points of process_frame, utils/mapper.py:236-246), order preserved
__global__ __launch_bounds__(MB) void surface_flags_kernel(const float* __restrict__ label, int n, float thr,
                                                           unsigned char* __restrict__ flags, int* __restrict__ block_cnt) {
    const int i = blockIdx.x * MB + threadIdx.x;
    const bool f = i < n && fabsf(label[i]) < thr;
    if (i < n) flags[i] = f ? 1 : 0;
    int total;
    block_flag_scan(f, total);
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = total;
}

__global__ __launch_bounds__(MB) void select_rows3_kernel(const float* __restrict__ rows, const unsigned char* __restrict__ flags,
                                                          int n, const int* __restrict__ block_off, float* __restrict__ out) {
    const int i = blockIdx.x * MB + threadIdx.x;
    const bool f = i < n && flags[i] != 0;
    int total;
    const int ex = block_flag_scan(f, total);
    if (!f) return;
    const size_t d = (size_t)(block_off[blockIdx.x] + ex), s = (size_t)i;
    out[3 * d] = rows[3 * s]; out[3 * d + 1] = rows[3 * s + 1]; out[3 * d + 2] = rows[3 * s + 2];
}

__global__ __launch_bounds__(MB) void gather_rows_kernel(const float* __restrict__ src, int width, const int* __restrict__ index,
                                                         int n, float* __restrict__ out) {
    const long t = (long)blockIdx.x * MB + threadIdx.x;
    if (t >= (long)n * width) return;
    const int i = (int)(t / width), c = (int)(t - (long)i * width);
    out[t] = src[(size_t)index[i] * width + c];
}

// transform_torch (utils/tools.py:534-553): homogeneous point x float32(T)^T
struct Pose12 { float m[12]; };
__global__ __launch_bounds__(MB) void transform_points_kernel(const float* __restrict__ pts, int stride, int n, Pose12 T,
                                                              float* __restrict__ out) {
    const int i = blockIdx.x * MB + threadIdx.x;
    if (i >= n) return;
    const float x = pts[(size_t)i * stride], y = pts[(size_t)i * stride + 1], z = pts[(size_t)i * stride + 2];
    const float* m = T.m;
    out[3 * (size_t)i] = fmaf(z, m[2], fmaf(y, m[1], x * m[0])) + m[3];
    out[3 * (size_t)i + 1] = fmaf(z, m[6], fmaf(y, m[5], x * m[4])) + m[7];
    out[3 * (size_t)i + 2] = fmaf(z, m[10], fmaf(y, m[9], x * m[8])) + m[11];
}

// transform_batch_torch (utils/tools.py:556-580) with the per-point matrix picked by a frame index, plus the
// quaternion update of NeuralPoints.adjust_map (quat_multiply(dq[frame], q), utils/tools.py:499-514)
__global__ __launch_bounds__(MB) void transform_by_frame_kernel(float* __restrict__ pts, int n, const int* __restrict__ frame,
                                                                const float* __restrict__ T, int n_frames,
                                                                float* __restrict__ quat, const float* __restrict__ dq) {
    const int i = blockIdx.x * MB + threadIdx.x;
    if (i >= n) return;
    int fr = frame[i];
    fr = fr < 0 ? fr + n_frames : fr;  // torch indexing semantics for negative indices
    const float* m = T + 12 * (size_t)fr;
    const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
    pts[3 * (size_t)i] = fmaf(z, m[2], fmaf(y, m[1], x * m[0])) + m[3];
    pts[3 * (size_t)i + 1] = fmaf(z, m[6], fmaf(y, m[5], x * m[4])) + m[7];
    pts[3 * (size_t)i + 2] = fmaf(z, m[10], fmaf(y, m[9], x * m[8])) + m[11];
    if (quat != nullptr) {
        const float4 a = reinterpret_cast<const float4*>(dq)[fr];  // w, x, y, z
        const float4 b = reinterpret_cast<const float4*>(quat)[i];
        float4 r;
        r.x = a.x * b.x - a.y * b.y - a.z * b.z - a.w * b.w;
        r.y = a.x * b.y + a.y * b.x + a.z * b.w - a.w * b.z;
        r.z = a.x * b.z - a.y * b.w + a.z * b.x + a.w * b.y;
        r.w = a.x * b.w + a.y * b.z - a.z * b.y + a.w * b.x;
        reinterpret_cast<float4*>(quat)[i] = r;
    }
}

static int check_pool(const pin_pool_arrays* p, const char* what) {
    if (!(p && p->coord && p->global_coord && p->sdf_label && p->weight && p->ts)) return fail(-1, "%s: NULL pool array", what);
    if (p->color_channels < 0 || p->color_channels > 4 || (p->color_channels > 0 && !p->color))
        return fail(-1, "%s: colour pool NULL or more than 4 channels", what);
    return 0;
}

}  // namespace pin

using namespace pin;

extern "C" int64_t pin_pool_workspace_bytes(int64_t n) {
    return (int64_t)(((size_t)cdiv(n + 1, MB) + 8) * sizeof(int) + 4 * 256 + 64);
}

extern "C" int pin_sample_rays(const pin_sample_params* p, const float* points, const float* colors, int32_t row_stride,
                               int32_t n, const float* rnd_surface, const float* rnd_front, const float* rnd_behind,
                               const pin_pool_arrays* out, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(p != nullptr && n >= 0, "params NULL or n < 0");
    PIN_CHECK_ARG(p->surface_n >= 0 && p->front_n >= 0 && p->behind_n >= 0, "negative sample counts");
    if (n == 0) return 0;
    if (int e = check_pool(out, "pin_sample_rays")) return e;
    PIN_CHECK_ARG(points && row_stride >= 3, "points NULL or row stride < 3");
    PIN_CHECK_ARG((p->surface_n == 0 || rnd_surface) && (p->front_n == 0 || rnd_front) && (p->behind_n == 0 || rnd_behind),
                  "random draws NULL");
    PIN_CHECK_ARG(out->color_channels == 0 || colors, "colour pool without scan colours");
    PIN_CHECK_ARG(p->sem_labels == nullptr || out->sem_label != nullptr, "semantic labels without a label pool");
    SampleConsts sc;
    sc.S = p->surface_n; sc.Ff = p->front_n; sc.Fb = p->behind_n; sc.A = sc.S + sc.Ff + sc.Fb + 1;
    PIN_CHECK_ARG((long)n * sc.A < (1L << 31), "too many samples for one call");
    sc.range = (float)p->surface_range;
    sc.half = (float)(2.0 * p->surface_range);  // sigma_ratio * surface_sample_range (data_sampler.py:71)
    sc.begin = (float)p->free_begin_ratio;
    sc.end_dist = (float)p->free_end_dist;
    sc.dist_weight_on = p->dist_weight_on;
    sc.w_c1 = (float)(1 + p->dist_weight_scale * 0.5);
    sc.max_range = (float)p->max_range;
    sc.w_scale = (float)p->dist_weight_scale;
    sc.dropoff_on = p->behind_dropoff_on;
    sc.d_max = (float)p->free_end_dist;
    sc.d_diff = (float)(p->free_end_dist - 0.2 * p->free_end_dist);
    sc.C = out->color_channels;
    sc.frame_id = p->frame_id;
    for (int i = 0; i < 12; ++i) sc.pose[i] = (float)p->pose[i];
    hipLaunchKernelGGL(sample_rays_kernel, dim3(cdiv((long)n * sc.A, MB)), dim3(MB), 0, as_stream(stream), sc, points, colors,
                       row_stride, n, rnd_surface, rnd_front, rnd_behind, *out, p->sem_labels);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_pool_window_mask(const float* global_coord, int32_t n, const double* origin, double radius,
                                    uint8_t* mask, int32_t* true_index, int32_t* count_out, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n >= 0 && origin && count_out, "n < 0 or NULL pointer");
    hipStream_t s = as_stream(stream);
    if (n == 0) { PIN_CHECK_HIP(hipMemsetAsync(count_out, 0, sizeof(int), s)); return 0; }
    PIN_CHECK_ARG(global_coord && mask && workspace, "NULL pointer");
    PIN_CHECK_ARG(workspace_bytes >= pin_pool_workspace_bytes(n), "workspace too small (pin_pool_workspace_bytes)");
    Carver cv{static_cast<char*>(workspace), static_cast<char*>(workspace) + workspace_bytes};
    const int nb = cdiv(n, MB);
    int* block_cnt = cv.take<int>(nb);
    hipLaunchKernelGGL(pool_window_kernel, dim3(nb), dim3(MB), 0, s, global_coord, n, origin[0], origin[1], origin[2],
                       radius * radius, mask, block_cnt);
    hipLaunchKernelGGL(scan_block_counts_kernel, dim3(1), dim3(1024), 0, s, block_cnt, nb, count_out);
    if (true_index) hipLaunchKernelGGL(pool_true_index_kernel, dim3(nb), dim3(MB), 0, s, mask, n, block_cnt, true_index);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_pool_discard(uint8_t* mask, const int32_t* true_index, const int64_t* discard_index,
                                int32_t n_discard, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n_discard >= 0, "n_discard < 0");
    if (n_discard == 0) return 0;
    PIN_CHECK_ARG(mask && true_index && discard_index, "NULL pointer");
    hipLaunchKernelGGL(pool_discard_kernel, dim3(cdiv(n_discard, MB)), dim3(MB), 0, as_stream(stream), mask, true_index,
                       reinterpret_cast<const long long*>(discard_index), n_discard);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_pool_compact(const pin_pool_arrays* src, const pin_pool_arrays* dst, const uint8_t* mask, int32_t n,
                                int32_t n_cur, int32_t* counts_out, void* workspace, int64_t workspace_bytes,
                                void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n >= 0 && n_cur >= 0 && n_cur <= n && counts_out, "bad sizes or NULL counts");
    hipStream_t s = as_stream(stream);
    PIN_CHECK_HIP(hipMemsetAsync(counts_out, 0, 2 * sizeof(int), s));
    if (n == 0) return 0;
    if (int e = check_pool(src, "pin_pool_compact(src)")) return e;
    if (int e = check_pool(dst, "pin_pool_compact(dst)")) return e;
    PIN_CHECK_ARG(src->color_channels == dst->color_channels, "colour channel mismatch");
    PIN_CHECK_ARG((src->sem_label == nullptr) == (dst->sem_label == nullptr), "semantic label pool on one side only");
    PIN_CHECK_ARG(src->coord != dst->coord, "compaction is out of place: src and dst must differ");
    PIN_CHECK_ARG(mask && workspace && workspace_bytes >= pin_pool_workspace_bytes(n), "mask NULL or workspace too small");
    Carver cv{static_cast<char*>(workspace), static_cast<char*>(workspace) + workspace_bytes};
    const int nb = cdiv(n, MB);
    int* block_cnt = cv.take<int>(nb);
    hipLaunchKernelGGL(block_counts_kernel, dim3(nb), dim3(MB), 0, s, mask, n, block_cnt);
    hipLaunchKernelGGL(scan_block_counts_kernel, dim3(1), dim3(1024), 0, s, block_cnt, nb, counts_out);
    if (n_cur > 0)
        hipLaunchKernelGGL(pool_tail_count_kernel, dim3(min(cdiv(n_cur, MB), 128)), dim3(MB), 0, s, mask, n - n_cur, n, counts_out + 1);
    hipLaunchKernelGGL(pool_scatter_kernel, dim3(nb), dim3(MB), 0, s, *src, *dst, mask, n, block_cnt);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_query_certainty(const pin_search_params* sp, const float* certainty, const float* query, int32_t n,
                                   float* certainty_out, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(sp && sp->table && sp->cand_off && sp->n_cand > 0, "search params incomplete");
    PIN_CHECK_ARG(sp->buffer_size > 0 && sp->buffer_size < (1LL << 31), "buffer_size must be in (0, 2^31)");
    PIN_CHECK_ARG(n >= 0, "n < 0");
    if (n == 0) return 0;
    PIN_CHECK_ARG(query && certainty_out && (sp->n_points == 0 || (sp->pos4 && certainty)), "NULL pointer");
    hipLaunchKernelGGL(query_certainty_kernel, dim3(cdiv(n, MB)), dim3(MB), 0, as_stream(stream), *sp, certainty, query, n,
                       certainty_out);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_new_sample_index(const float* certainty, const float* sdf_label, int32_t n, float certainty_thre,
                                    float label_thre, int64_t offset, int64_t* index_out, int32_t* count_out,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n >= 0 && count_out, "n < 0 or NULL count");
    hipStream_t s = as_stream(stream);
    if (n == 0) { PIN_CHECK_HIP(hipMemsetAsync(count_out, 0, sizeof(int), s)); return 0; }
    PIN_CHECK_ARG(certainty && sdf_label && index_out && workspace, "NULL pointer");
    PIN_CHECK_ARG(workspace_bytes >= pin_pool_workspace_bytes(n) + n, "workspace too small (pin_pool_workspace_bytes(n) + n)");
    Carver cv{static_cast<char*>(workspace), static_cast<char*>(workspace) + workspace_bytes};
    const int nb = cdiv(n, MB);
    int* block_cnt = cv.take<int>(nb);
    unsigned char* flags = cv.take<unsigned char>(n);
    PIN_CHECK_ARG(flags != nullptr, "workspace too small");
    hipLaunchKernelGGL(new_sample_flags_kernel, dim3(nb), dim3(MB), 0, s, certainty, sdf_label, n, certainty_thre, label_thre,
                       flags, block_cnt);
    hipLaunchKernelGGL(scan_block_counts_kernel, dim3(1), dim3(1024), 0, s, block_cnt, nb, count_out);
    hipLaunchKernelGGL(new_sample_index_kernel, dim3(nb), dim3(MB), 0, s, flags, n, block_cnt, (long long)offset,
                       reinterpret_cast<long long*>(index_out));
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_select_surface_points(const float* rows, const float* sdf_label, int32_t n, float label_thre,
                                         float* out, int32_t* count_out, void* workspace, int64_t workspace_bytes,
                                         void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n >= 0 && count_out, "n < 0 or NULL count");
    hipStream_t s = as_stream(stream);
    if (n == 0) { PIN_CHECK_HIP(hipMemsetAsync(count_out, 0, sizeof(int), s)); return 0; }
    PIN_CHECK_ARG(rows && sdf_label && out && workspace, "NULL pointer");
    PIN_CHECK_ARG(workspace_bytes >= pin_pool_workspace_bytes(n) + n, "workspace too small (pin_pool_workspace_bytes(n) + n)");
    Carver cv{static_cast<char*>(workspace), static_cast<char*>(workspace) + workspace_bytes};
    const int nb = cdiv(n, MB);
    int* block_cnt = cv.take<int>(nb);
    unsigned char* flags = cv.take<unsigned char>(n);
    PIN_CHECK_ARG(flags != nullptr, "workspace too small");
    hipLaunchKernelGGL(surface_flags_kernel, dim3(nb), dim3(MB), 0, s, sdf_label, n, label_thre, flags, block_cnt);
    hipLaunchKernelGGL(scan_block_counts_kernel, dim3(1), dim3(1024), 0, s, block_cnt, nb, count_out);
    hipLaunchKernelGGL(select_rows3_kernel, dim3(nb), dim3(MB), 0, s, rows, flags, n, block_cnt, out);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_transform_points(const float* points, int32_t row_stride, int32_t n, const double* pose, float* out,
                                    void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n >= 0 && row_stride >= 3, "n < 0 or row stride < 3");
    if (n == 0) return 0;
    PIN_CHECK_ARG(points && pose && out, "NULL pointer");
    Pose12 T;
    for (int i = 0; i < 12; ++i) T.m[i] = (float)pose[i];
    hipLaunchKernelGGL(transform_points_kernel, dim3(cdiv(n, MB)), dim3(MB), 0, as_stream(stream), points, row_stride, n, T, out);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_transform_by_frame(float* points, int32_t n, const int32_t* frame, const float* T, int32_t n_frames,
                                      float* quat, const float* dquat, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n >= 0 && n_frames > 0, "n < 0 or no frames");
    if (n == 0) return 0;
    PIN_CHECK_ARG(points && frame && T && (quat == nullptr || dquat != nullptr), "NULL pointer");
    hipLaunchKernelGGL(transform_by_frame_kernel, dim3(cdiv(n, MB)), dim3(MB), 0, as_stream(stream), points, n, frame, T,
                       n_frames, quat, dquat);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_gather_rows(const float* src, int32_t width, const int32_t* index, int32_t n, float* out, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n >= 0 && width >= 1, "n < 0 or width < 1");
    if (n == 0) return 0;
    PIN_CHECK_ARG(src && index && out, "NULL pointer");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv((long)n * width, MB)), dim3(MB), 0, as_stream(stream), src, width, index, n, out);
    PIN_CHECK_LAUNCH();
    return 0;
}

// pin_warmup (common.hip): asking for a kernel's attributes makes the runtime load this translation unit's code object now
// instead of inside the first frame that launches one of its kernels
namespace pin {
int pin_warm_pool() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&pool_true_index_kernel)) == hipSuccess ? 0 : -2;
}
}  // namespace pin
