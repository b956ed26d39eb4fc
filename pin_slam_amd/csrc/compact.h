// Ordered stream compaction helpers shared by the maintenance kernels (maint.hip, pool.hip):
// per-wave ballot/popcount prefix inside a block, per-block offsets from a one-block scan.  The
// relative order of kept elements is the global order, as boolean-mask indexing gives in the
// reference.
#pragma once
#include "pin_common.h"

namespace pin {

constexpr int MB = 256;  // block size of the streaming kernels

// ---- ordered compaction helpers ---------------------------------------------------------
// exclusive prefix of a per-thread flag inside a 256-thread block; returns block total
__device__ __forceinline__ int block_flag_scan(bool flag, int& total) {
    __shared__ int wave_cnt[MB / 64];
    const unsigned long long bal = __ballot(flag);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < MB / 64; ++w) {
        if (w < wave) off += wave_cnt[w];
        tot += wave_cnt[w];
    }
    __syncthreads();
    total = tot;
    return off + before;
}

static __global__ __launch_bounds__(MB) void block_counts_kernel(const unsigned char* __restrict__ flags, int n,
                                                          int* __restrict__ block_cnt) {
    const int i = blockIdx.x * MB + threadIdx.x;
    int total;
    block_flag_scan(i < n && flags[i] != 0, total);
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = total;
}

// single-block exclusive scan of the per-block counts; total -> *count_out
static __global__ __launch_bounds__(1024) void scan_block_counts_kernel(int* __restrict__ block_cnt, int nblocks,
                                                                 int* __restrict__ count_out) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (nblocks + 1023) / 1024;
    const int b0 = t * per, b1 = min(b0 + per, nblocks);
    int s = 0;
    for (int b = b0; b < b1; ++b) s += block_cnt[b];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int b = b0; b < b1; ++b) {
        const int c = block_cnt[b];
        block_cnt[b] = run;
        run += c;
    }
    if (t == 1023) *count_out = part[1023];
}

// ---- workspace carving --------------------------------------------------------------------------
struct Carver {
    char* p;
    char* end;
    template <typename T>
    T* take(size_t n) {
        size_t a = (reinterpret_cast<size_t>(p) + 255) & ~size_t(255);
        char* q = reinterpret_cast<char*>(a);
        p = q + n * sizeof(T);
        return p <= end ? reinterpret_cast<T*>(q) : nullptr;
    }
};


}  // namespace pin
