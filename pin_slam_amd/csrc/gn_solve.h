// Normal-equation solve + loop control of one Gauss-Newton iteration (implicit_reg, utils/tracker.py:656-679; the bookkeeping of
// Tracker.tracking, :147-184) by ONE WAVE, the 6x7 system spread over its lanes.  Included by sdf.hip (outside namespace pin).
//
// r01-r05 ran this on lane 0 of a one-wave kernel: ~2.4 us of dependent float64 arithmetic (six pivots with conditional row
// exchanges, twelve divisions, sin / cos / acos from the device library) behind a launch of its own -- 8 us per iteration in the
// frame (scripts/exp/gn_loop_parts.py: odometry 3.17 -> 2.77 ms with the launch taken out), 7 % of it.  Here lane r < 6 keeps ROW r
// of the augmented system in registers; a Gauss-Jordan step is one pivot search over six lanes (readlane), one broadcast of the
// pivot row and ONE fused multiply-add per column in all rows at once -- no row exchanges (a row that has served as a pivot is
// marked, the solution is read from the row that took column c), one division per pivot.  The same function is the tail of the
// tile kernels' LAST block (gn_tail_last_block below): the block that draws the last ticket solves, so the iteration is two
// launches, not three (pin_gn_loop_init + pin_gn_accumulate_solve).
#pragma once

namespace pin {

// value of lane `lane` (wave-uniform) in every lane
__device__ __forceinline__ double rl64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane), hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// TAIL = false: a kernel of its own (gn_solve_kernel) -- parameters and status word come as kernel arguments, the sums were
// written by an earlier launch, the stop flag is requested with everything else and looked at before anything is written.
// TAIL = true: the last block of a tile kernel -- the parameters are read out of the state (pin_gn_loop_init put them there: a
// tile kernel at its register limit gets ONE more argument, not twelve), the sums were added to by the other blocks of THIS
// launch (device-scope atomics: read with device-scope loads)
template <bool TAIL>
__device__ __forceinline__ void gn_solve_wave(double* __restrict__ sums, double* __restrict__ st, const pin_gn_loop_params& lp_arg,
                                              const int* __restrict__ status_arg) {
    const int lane = threadIdx.x & 63;
    // everything the wave reads is requested up front: one memory round trip.  Lanes 0..23: pose and loop scalars; TAIL: lanes
    // 24..32 the loop parameters and the address of the status word
    const double st_pre = st[lane <= PIN_GN_STATE_MSE ? lane : (TAIL && lane <= PIN_GN_STATE_MSE + 9) ? PIN_GN_STATE_LP + lane - (PIN_GN_STATE_MSE + 1) : 0];
    const double nsrc = st[PIN_GN_STATE_NSRC];
    const double done_in = TAIL ? 0.0 : st[PIN_GN_STATE_DONE];
    pin_gn_loop_params lp;
    const int* status;
    int flags = 0;
    if constexpr (TAIL) {
        constexpr int L0 = PIN_GN_STATE_MSE + 1;
        lp.lm_lambda = rl64(st_pre, L0); lp.term_thre_deg = rl64(st_pre, L0 + 1); lp.term_thre_m = rl64(st_pre, L0 + 2);
        lp.min_valid_ratio = rl64(st_pre, L0 + 3); lp.max_increment_ratio = rl64(st_pre, L0 + 4);
        lp.min_valid_points = (int)rl64(st_pre, L0 + 5); lp.iter_n = (int)rl64(st_pre, L0 + 6); lp.early_exit = (int)rl64(st_pre, L0 + 7);
        status = reinterpret_cast<const int*>((unsigned long long)__double_as_longlong(rl64(st_pre, L0 + 8)));
        if (status != nullptr) flags = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (needed at the very end only)
    } else {
        lp = lp_arg;
        status = status_arg;
        if (status != nullptr) flags = *status;
    }
    constexpr bool COHERENT = TAIL, CHECK_DONE = !TAIL;
    double a = 0.0;
    {   // replica sum: two lanes per sum, GN_REPLICAS / 2 independent loads each
        const int i = lane & 31, h = lane >> 5;
        double v[GN_REPLICAS / 2];
#pragma unroll
        for (int r = 0; r < GN_REPLICAS / 2; ++r) {
            double* p = sums + (h * (GN_REPLICAS / 2) + r) * PIN_GN_NSUMS + i;
            if constexpr (COHERENT) v[r] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v[r] = *p;
        }
#pragma unroll
        for (int r = 0; r < GN_REPLICAS / 2; ++r) a += v[r];
        a += __shfl_xor(a, 32, 64);
    }
    if (CHECK_DONE && done_in != 0.0) return;  // (uniform)
    // lanes i and i + 32 hold sum i.  The replicas are cleared for the next iteration (nobody else touches them any more: the
    // other blocks of a fused launch have drawn their tickets behind their atomics), and so is the ticket
#pragma unroll
    for (int i = 0; i < GN_REPLICAS * PIN_GN_NSUMS / 64; ++i) sums[i * 64 + lane] = 0.0;
    if (lane == 0) {
        st[PIN_GN_STATE_TICKET] = 0.0;
        st[PIN_GN_STATE_STATUS] = (double)flags;  // (sticky flags of the library, e.g. a decoder outside the fp16 range)
    }
    const double cnt = rint(rl64(a, 29));
    double dR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, dt[3] = {0, 0, 0};
    double res_cm = 0.0, mse = rl64(st_pre, PIN_GN_STATE_MSE);
    if (cnt >= 10.0) {  // tracker.py:430-432 (wave-uniform)
        const double scale = cnt / (2.0 * rl64(a, 27));  // w /= 2*mean(w)
        const int r = lane < 6 ? lane : 5;  // lanes 6.. carry a copy of row 5 that nobody looks at
        double M[7];
#pragma unroll
        for (int c = 0; c < 6; ++c) {  // N[r][c] out of the packed upper triangle: o(a <= b) = a (13 - a) / 2 + b - a
            const int lo = r < c ? r : c, hi = r < c ? c : r;
            M[c] = scale * __shfl(a, lo * (13 - lo) / 2 + hi - lo, 64);
        }
        M[6] = -scale * __shfl(a, 21 + r, 64);
        if (lane < 6) {
#pragma unroll
            for (int c = 0; c < 6; ++c) st[PIN_GN_STATE_NRAW + lane * 6 + c] = M[c];
        }
        mse = scale * rl64(a, 30) / cnt;
#pragma unroll
        for (int c = 0; c < 6; ++c) M[c] = (c == r) ? fma(lp.lm_lambda, M[c], M[c]) : M[c];
        // Gauss-Jordan, partial pivoting over the rows that have not been a pivot yet (float64)
        bool used = lane >= 6;
        int prow[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const double key = used ? -1.0 : fabs(M[c]);
            double best = rl64(key, 0);
            int p = 0;
#pragma unroll
            for (int l = 1; l < 6; ++l) {
                const double v = rl64(key, l);
                if (v > best) { best = v; p = l; }
            }
            p = __builtin_amdgcn_readfirstlane(p);
            prow[c] = p;
            double P[7];
#pragma unroll
            for (int b = c; b < 7; ++b) P[b] = rl64(M[b], p);
            const double inv = 1.0 / P[c];
            const bool mine = lane == p;
            const double f = M[c] * inv;
#pragma unroll
            for (int b = c; b < 7; ++b) M[b] = mine ? P[b] * inv : fma(-f, P[b], M[b]);
            used = used || mine;
        }
        double t[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) t[c] = rl64(M[6], prow[c]);
        // expmap (tracker.py:784-795): R = I + sin(a) S + (1 - cos(a)) S^2.  Steps of a registration are small: both factors
        // from their series below half a radian (truncation < 2e-23), the device library beyond
        const double ang2 = t[0] * t[0] + t[1] * t[1] + t[2] * t[2];
        const double ang = sqrt(ang2);
        const double ax = t[0] / ang, ay = t[1] / ang, az = t[2] / ang;
        double sn, cs;
        if (ang < 0.5) {
            const double x = ang2;
            sn = ang * (1.0 + x * (-1.0 / 6 + x * (1.0 / 120 + x * (-1.0 / 5040 + x * (1.0 / 362880 + x * (-1.0 / 39916800 + x * (1.0 / 6227020800.0 +
                 x * (-1.0 / 1307674368000.0 + x * (1.0 / 355687428096000.0)))))))));
            cs = x * (0.5 + x * (-1.0 / 24 + x * (1.0 / 720 + x * (-1.0 / 40320 + x * (1.0 / 3628800 + x * (-1.0 / 479001600 + x * (1.0 / 87178291200.0 +
                 x * (-1.0 / 20922789888000.0 + x * (1.0 / 6402373705728000.0)))))))));
        } else {
            sn = sin(ang);
            cs = 1.0 - cos(ang);
        }
        const double S[9] = {0, -az, ay, az, 0, -ax, -ay, ax, 0};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double ss = 0.0;
#pragma unroll
                for (int c = 0; c < 3; ++c) ss += S[i * 3 + c] * S[c * 3 + j];
                dR[i * 3 + j] = (i == j ? 1.0 : 0.0) + S[i * 3 + j] * sn + ss * cs;
            }
        dt[0] = t[3]; dt[1] = t[4]; dt[2] = t[5];
        res_cm = rl64(a, 28) / cnt * 100.0;
    }
    // T = dT @ T, one element per lane (lanes 0..15: row i = lane / 4, column j = lane % 4)
    double out = st_pre;  // lanes 16..23: the loop scalars keep their value unless something below says otherwise
    {
        const int i = (lane >> 2) & 3, j = lane & 3;
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double Tcj = __shfl(st_pre, c * 4 + j, 64);
            const double d = c < 3 ? (i == 0 ? dR[c] : i == 1 ? dR[3 + c] : i == 2 ? dR[6 + c] : 0.0)
                                   : (i == 0 ? dt[0] : i == 1 ? dt[1] : i == 2 ? dt[2] : 1.0);
            acc = fma(d, Tcj, acc);
        }
        if (lane < 16) out = acc;
    }
    const double last = rl64(st_pre, PIN_GN_STATE_LAST_RES);
    bool valid = rl64(st_pre, PIN_GN_STATE_VALID) != 0.0;
    double last_new = last;
    if ((res_cm - last) / last > lp.max_increment_ratio) valid = false;  // tracker.py:150-159
    else last_new = res_cm;
    if (cnt < lp.min_valid_points || cnt / nsrc < lp.min_valid_ratio) valid = false;  // :161-169
    const int it = (int)rl64(st_pre, PIN_GN_STATE_ITERS);
    const bool was_converged = rl64(st_pre, PIN_GN_STATE_CONVERGED) != 0.0;
    const bool done = !valid || was_converged || it + 1 >= lp.iter_n;  // :171-172
    bool converged = was_converged;
    if (!done) {
        bool small = false;
        if (lp.early_exit) {
            const double rot_deg = acos((dR[0] + dR[4] + dR[8] - 1.0) / 2.0) * 180.0 / 3.14159265358979323846;
            const double tran = sqrt(dt[0] * dt[0] + dt[1] * dt[1] + dt[2] * dt[2]);
            small = fabs(rot_deg) < lp.term_thre_deg && tran < lp.term_thre_m;
        }
        if (small || it == lp.iter_n - 2) converged = true;  // :179-184
    }
    switch (lane) {
        case PIN_GN_STATE_LAST_RES: out = last_new; break;
        case PIN_GN_STATE_RES: out = res_cm; break;
        case PIN_GN_STATE_CNT: out = cnt; break;
        case PIN_GN_STATE_VALID: out = valid ? 1.0 : 0.0; break;
        case PIN_GN_STATE_CONVERGED: out = converged ? 1.0 : 0.0; break;
        case PIN_GN_STATE_DONE: out = done ? 1.0 : out; break;
        case PIN_GN_STATE_ITERS: out = (double)(it + 1); break;
        case PIN_GN_STATE_MSE: out = mse; break;
        default: break;
    }
    if (lane <= PIN_GN_STATE_MSE) st[lane] = out;
}

// Tail of a tile kernel's block (its wave 0, behind the block's atomics into `sums`): the block that draws the last of
// gridDim.x tickets finishes the iteration.  Release / acquire at device scope around the ticket: the last block sees every
// other block's atomics.
__device__ __forceinline__ void gn_tail_last_block(double* __restrict__ state, double* __restrict__ sums) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    unsigned int t = 0;
    if ((threadIdx.x & 63) == 0)
        t = (unsigned int)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(state + PIN_GN_STATE_TICKET), 1ull,
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
    if (t != gridDim.x - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    pin_gn_loop_params none;
    gn_solve_wave<true>(sums, state, none, nullptr);
}

}  // namespace pin
