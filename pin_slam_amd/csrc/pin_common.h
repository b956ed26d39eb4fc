// Shared device/host helpers for libpinhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/pin_abi.h"

namespace pin {

// ---- error reporting ---------------------------------------------------------------
char* last_error_buf();
int fail(int code, const char* fmt, ...);

#define PIN_CHECK_ARG(cond, msg)                                     \
    do {                                                             \
        if (!(cond)) return ::pin::fail(-1, "%s: %s", __func__, msg); \
    } while (0)

// hipGetLastError() is sticky per thread: another library's failed probe (e.g. a device
// query before the runtime is initialised) must not be reported as ours.
#define PIN_ENTER() ((void)hipGetLastError())

#define PIN_CHECK_LAUNCH()                                                              \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) return ::pin::fail(-2, "%s: %s", __func__, hipGetErrorString(e_)); \
    } while (0)

#define PIN_CHECK_HIP(expr)                                                             \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) return ::pin::fail(-2, "%s: %s", __func__, hipGetErrorString(e_)); \
    } while (0)

int* status_word();  // (common.hip) device word of sticky PIN_STATUS_* flags, nullptr if it could not be allocated

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// internal: kNN launches whose pose / stop flag live in the device-resident GN state
int knn_direct_dev(const pin_search_params* sp, const float* query, int32_t n, int32_t k, const double* state,
                   float* query_out, float* nbr_out, int32_t* nn_count_out, void* stream);
int knn_bricks_dev(const pin_search_params* sp, const pin_brick_cache* bc, const float* query, int32_t n, int32_t k,
                   const double* state, float* query_out, float* nbr_out, int32_t* nn_count_out, void* stream);

// internal: pin_voxel_downsample_fast with the number of points on the device (pin_preprocess_frame chains its stages without
// a host read-back in between); n bounds *n_dev
int vds_fast_dev(const float* points, int32_t n, const int32_t* n_dev, float voxel_size, int32_t* sel_out, int32_t* count_out,
                 void* workspace, int64_t workspace_bytes, hipStream_t s);

// ---- device helpers ----------------------------------------------------------------
constexpr long long PRIME0 = 73856093LL, PRIME1 = 19349669LL, PRIME2 = 83492791LL;
constexpr float IDW_EPS = 1e-15f;

// Voxel coordinate of one axis: floor(p / res) in IEEE fp32 (true division, like the CPU
// reference neural_points.py:963), as a 64-bit integer.
__device__ __forceinline__ long long voxel_coord(float p, float res) {
#pragma clang fp contract(off)
    return (long long)floorf(__fdiv_rn(p, res));
}

// h mod B in [0, B) for |h| < 2^62, 0 < B < 2^40.  The 64-bit signed `%` is ~200 emulated vector instructions;
// here the quotient is estimated in fp64 (exact to +-1: the estimate errs by < 2^-40 of itself) and the
// remainder fixed up in exact integer arithmetic.
__device__ __forceinline__ long long mod_nonneg(long long h, long long B) {
    const double inv = 1.0 / (double)B;  // loop-invariant at every call site
    const long long q = (long long)floor((double)h * inv);
    long long r = h - q * B;
    if (r < 0) r += B;
    if (r >= B) r -= B;
    return r;
}

// Mathematical (non-negative) modulus of the spatial hash = fmod + negative-index wrap of
// the reference (neural_points.py:972-978).
__device__ __forceinline__ uint32_t hash_base(float x, float y, float z, float res, long long B) {
    const long long h = voxel_coord(x, res) * PRIME0 + voxel_coord(y, res) * PRIME1 + voxel_coord(z, res) * PRIME2;
    return (uint32_t)mod_nonneg(h, B);
}

// (dx*dx + dy*dy) + dz*dz with one rounding per operation -- never contracted to FMA, so
// the bits equal torch's CPU float32 result (neural_points.py:992-995).
__device__ __forceinline__ float dist2_exact(float dx, float dy, float dz) {
#pragma clang fp contract(off)  // hipcc defaults to -ffp-contract=fast; __fmul_rn alone does not stop fusion
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    const float xy = xx + yy;
    return xy + zz;
}

// One Adam element, one step (torch.optim.Adam, tools.py:198-199).  Contraction is off so that every kernel
// that applies a step -- dense, row-flagged, lazy replay, halo rows -- performs the same roundings.
__device__ __forceinline__ void adam_elem(float& p, float& m, float& v, float g, float lr_over_bc1, float inv_sqrt_bc2,
                                          float b1, float b2, float eps) {
#pragma clang fp contract(off)
    m = m + (g - m) * (1.f - b1);               // exp_avg.lerp_(grad, 1-beta1)
    v = v * b2 + (1.f - b2) * g * g;            // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
    const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
    p = p - lr_over_bc1 * (m / denom);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// v + (lane ^ 16's v) + ... over the four 16-lane rows of the wave, in every lane: the sum over the four lanes
// (n, g = 0..3) of a query of the quad layout.  gfx950's row swaps instead of two ds_bpermute round trips through the LDS
// pipeline: v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second -- with both
// operands = v they come back as [r0 r0 r2 r2] and [r1 r1 r3 r3], whose sum is v + xor16(v) in every lane;
// v_permlane32_swap does the same for the wave's halves.  Same additions as `v += __shfl_xor(v, 16); v += __shfl_xor(v, 32)`
// (fp32 addition commutes: identical bits), no lane-index arithmetic, no LDS traffic.
typedef unsigned int swap2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float rows_pair_sum16(float v) {
    const swap2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float rows_pair_sum32(float v) {
    const swap2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// the same sum through the LDS crossbar (two ds_bpermute): no vector-ALU slots for the exchange itself.  The registration
// tile kernels are bound by their vector-ALU issue and keep this form (measured, same box, 4 runs each: 37.7 vs 38.3 us per
// launch at C3, 50.4 vs 51.2 at the per-neighbour workload); everything else uses the swaps (fewer registers: the lane-index
// arithmetic of the permute is loop-invariant and used to be spilled).
__device__ __forceinline__ float rows_sum_lds(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
#ifdef PIN_AB_ROWS_SHFL  // (A/B builds only, scripts/build_variant.sh)
__device__ __forceinline__ float rows_sum(float v) { return rows_sum_lds(v); }
#else
__device__ __forceinline__ float rows_sum(float v) { return rows_pair_sum32(rows_pair_sum16(v)); }
#endif

// float wave sum on the DPP path (no LDS crossbar): quad xor 1, 2, row_half_mirror, row_mirror
// give every lane its 16-lane row sum; the 4 row sums are then added through readlane.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_f32(float v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror
    return (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)) +
            __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)) +
            __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48)));
}

}  // namespace pin
