// SLAMDataset.preprocess_frame data path on gfx950 (once per frame, HBM-streaming):
//   crop_frame            dataset/slam_dataset.py:1229-1247   (range / height window, ordered compaction of the rows)
//   intrinsic_correct     dataset/slam_dataset.py:1251-1269   (KITTI vertical-angle calibration)
//   deskewing             utils/tools.py:747-779              (per-point motion undistortion)
// The two voxel_down_sample_torch passes are pin_voxel_downsample (maint.hip) + pin_gather_rows.
#include "pin_common.h"
#include "compact.h"

#pragma clang fp contract(off)

namespace pin {

__device__ __forceinline__ float norm3_rn(float x, float y, float z) {
    // torch.norm (CPU) accumulates with FMAs; sqrt through float64 is correctly rounded
    return (float)sqrt((double)__fmaf_rn(z, z, __fmaf_rn(y, y, x * x)));
}

__global__ __launch_bounds__(MB) void crop_flags_kernel(const float* __restrict__ pts, int width, int n, float min_z,
                                                        float max_z, float min_r, float max_r,
                                                        unsigned char* __restrict__ flags, int* __restrict__ block_cnt) {
    const int i = blockIdx.x * MB + threadIdx.x;
    bool f = false;
    if (i < n) {
        const float* P = pts + (size_t)i * width;
        const float d = norm3_rn(P[0], P[1], P[2]);
        f = d > min_r && d < max_r && P[2] > min_z && P[2] < max_z;
        flags[i] = f ? 1 : 0;
    }
    int total;
    block_flag_scan(f, total);
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = total;
}

__global__ __launch_bounds__(MB) void crop_scatter_kernel(const float* __restrict__ pts, int width, int n,
                                                          const float* __restrict__ ts,
                                                          const unsigned char* __restrict__ flags,
                                                          const int* __restrict__ block_off, float* __restrict__ out,
                                                          float* __restrict__ ts_out) {
    const int i = blockIdx.x * MB + threadIdx.x;
    const bool f = i < n && flags[i] != 0;
    int total;
    const int ex = block_flag_scan(f, total);
    if (!f) return;
    const size_t d = (size_t)(block_off[blockIdx.x] + ex);
    for (int c = 0; c < width; ++c) out[d * width + c] = pts[(size_t)i * width + c];
    if (ts != nullptr) ts_out[d] = ts[i];
}

__global__ __launch_bounds__(MB) void intrinsic_correct_kernel(float* __restrict__ pts, int width, int n, float ang) {
    const int i = blockIdx.x * MB + threadIdx.x;
    if (i >= n) return;
    float* P = pts + (size_t)i * width;
    const float dist = norm3_rn(P[0], P[1], P[2]);
    const float v = asinf(__fdiv_rn(P[2], dist));
    const float vc = v + ang;
    const float hs = __fdiv_rn(cosf(vc), cosf(v));
    P[0] = P[0] * hs;
    P[1] = P[1] * hs;
    P[2] = dist * sinf(vc);
}

struct TsRange { unsigned int lo, hi; };  // order-preserving encodings of float min / max

__device__ __forceinline__ unsigned int enc_order(float f) {
    const unsigned int b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_order(unsigned int e) {
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}

__global__ void ts_range_init_kernel(TsRange* r) { r->lo = 0xffffffffu; r->hi = 0u; }

__global__ __launch_bounds__(MB) void ts_range_kernel(const float* __restrict__ ts, int n, TsRange* __restrict__ r) {
    const int i = blockIdx.x * MB + threadIdx.x;
    unsigned int lo = 0xffffffffu, hi = 0u;
    if (i < n) lo = hi = enc_order(ts[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, (unsigned int)__shfl_xor((int)lo, o, 64));
        hi = max(hi, (unsigned int)__shfl_xor((int)hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&r->lo, lo); atomicMax(&r->hi, hi); }
}

struct Deskew { float ax[3]; float theta; float tr[3]; float mid; };

// points_i <- exp(t_i * log R) p_i + t_i * trans, t_i = (ts_i - min)/(max - min) - mid (tools.py:763-777)
__global__ __launch_bounds__(MB) void deskew_kernel(float* __restrict__ pts, int width, int n, const float* __restrict__ ts,
                                                    const TsRange* __restrict__ r, Deskew dk) {
    const int i = blockIdx.x * MB + threadIdx.x;
    if (i >= n) return;
    const float lo = dec_order(r->lo), hi = dec_order(r->hi);
    float t = __fdiv_rn(ts[i] - lo, hi - lo);
    t = t - dk.mid;
    float* P = pts + (size_t)i * width;
    const float x = P[0], y = P[1], z = P[2];
    float s, c;
    sincosf(t * dk.theta, &s, &c);
    const float ax = dk.ax[0], ay = dk.ax[1], az = dk.ax[2];
    const float kx = ay * z - az * y, ky = az * x - ax * z, kz = ax * y - ay * x;  // a x p
    const float dot = (ax * x + ay * y + az * z) * (1.0f - c);
    P[0] = (x * c + kx * s + ax * dot) + t * dk.tr[0];  // Rodrigues
    P[1] = (y * c + ky * s + ay * dot) + t * dk.tr[1];
    P[2] = (z * c + kz * s + az * dot) + t * dk.tr[2];
}

}  // namespace pin

using namespace pin;

extern "C" int pin_crop_frame(const float* points, int32_t width, int32_t n, const float* ts, float min_z, float max_z,
                              float min_range, float max_range, float* points_out, float* ts_out, int32_t* count_out,
                              void* workspace, int64_t workspace_bytes, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n >= 0 && width >= 3 && count_out, "n < 0, width < 3 or NULL count");
    hipStream_t s = as_stream(stream);
    if (n == 0) { PIN_CHECK_HIP(hipMemsetAsync(count_out, 0, sizeof(int), s)); return 0; }
    PIN_CHECK_ARG(points && points_out && workspace && (ts == nullptr || ts_out), "NULL pointer");
    PIN_CHECK_ARG(workspace_bytes >= pin_pool_workspace_bytes(n) + n, "workspace too small (pin_pool_workspace_bytes(n) + n)");
    Carver cv{static_cast<char*>(workspace), static_cast<char*>(workspace) + workspace_bytes};
    const int nb = cdiv(n, MB);
    int* block_cnt = cv.take<int>(nb);
    unsigned char* flags = cv.take<unsigned char>(n);
    PIN_CHECK_ARG(flags != nullptr, "workspace too small");
    hipLaunchKernelGGL(crop_flags_kernel, dim3(nb), dim3(MB), 0, s, points, width, n, min_z, max_z, min_range, max_range,
                       flags, block_cnt);
    hipLaunchKernelGGL(scan_block_counts_kernel, dim3(1), dim3(1024), 0, s, block_cnt, nb, count_out);
    hipLaunchKernelGGL(crop_scatter_kernel, dim3(nb), dim3(MB), 0, s, points, width, n, ts, flags, block_cnt, points_out, ts_out);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_intrinsic_correct(float* points, int32_t width, int32_t n, double correct_deg, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n >= 0 && width >= 3, "n < 0 or width < 3");
    if (n == 0 || correct_deg == 0.0) return 0;
    PIN_CHECK_ARG(points, "NULL pointer");
    const float ang = (float)(correct_deg / 180.0 * 3.141592653589793);
    hipLaunchKernelGGL(intrinsic_correct_kernel, dim3(cdiv(n, MB)), dim3(MB), 0, as_stream(stream), points, width, n, ang);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_deskew(float* points, int32_t width, int32_t n, const float* ts, const double* pose, double ts_mid_pose,
                          void* workspace, int64_t workspace_bytes, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n >= 0 && width >= 3, "n < 0 or width < 3");
    if (n == 0) return 0;
    PIN_CHECK_ARG(points && ts && pose && workspace && workspace_bytes >= 64, "NULL pointer or workspace < 64 bytes");
    // log map of float32(R) on the host in float64
    double R[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[3 * r + c] = (double)(float)pose[4 * r + c];
    double cth = (R[0] + R[4] + R[8] - 1.0) / 2.0;
    cth = cth > 1.0 ? 1.0 : (cth < -1.0 ? -1.0 : cth);
    const double th = acos(cth);
    Deskew dk;
    dk.theta = (float)th;
    if (th < 1e-12) {
        dk.ax[0] = 1.f; dk.ax[1] = 0.f; dk.ax[2] = 0.f; dk.theta = 0.f;
    } else {
        const double k = 1.0 / (2.0 * sin(th));
        dk.ax[0] = (float)((R[7] - R[5]) * k); dk.ax[1] = (float)((R[2] - R[6]) * k); dk.ax[2] = (float)((R[3] - R[1]) * k);
    }
    for (int i = 0; i < 3; ++i) dk.tr[i] = (float)pose[4 * i + 3];
    dk.mid = (float)ts_mid_pose;
    hipStream_t s = as_stream(stream);
    TsRange* r = reinterpret_cast<TsRange*>(workspace);
    hipLaunchKernelGGL(ts_range_init_kernel, dim3(1), dim3(1), 0, s, r);
    hipLaunchKernelGGL(ts_range_kernel, dim3(cdiv(n, MB)), dim3(MB), 0, s, ts, n, r);
    hipLaunchKernelGGL(deskew_kernel, dim3(cdiv(n, MB)), dim3(MB), 0, s, points, width, n, ts, r, dk);
    PIN_CHECK_LAUNCH();
    return 0;
}

// pin_warmup (common.hip): asking for a kernel's attributes makes the runtime load this translation unit's code object now
// instead of inside the first frame that launches one of its kernels
namespace pin {
int pin_warm_prep() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&intrinsic_correct_kernel)) == hipSuccess ? 0 : -2;
}
}  // namespace pin
