// Normal-equation solve + loop control of one Gauss-Newton iteration (implicit_reg, utils/tracker.py:656-679; the bookkeeping of
// Tracker.tracking, :147-184) by ONE WAVE, the 6x7 system spread over its lanes.  Included by sdf.hip (outside namespace pin).
//
// r01-r05 ran this on lane 0 of a one-wave kernel: dependent float64 arithmetic in one lane (six pivots with conditional row
// exchanges, twelve IEEE divisions, sin / cos / acos from the device library) behind a launch of its own -- 8 us per iteration in
// the frame (scripts/exp/gn_loop_parts.py: odometry 3.17 -> 2.77 ms with the launch taken out), 7 % of it, and time stamps in the
// kernel (scripts/exp/gn_tail_stamps.py) put the ARITHMETIC, not the memory round trips around it (0.2-0.7 us each), in front:
// 3.0 us even with a row of the system per lane.  Here lane 8 r + c holds ELEMENT (r, c) of the augmented system.  A Gauss-Jordan
// step is: pivot out of lane 9 k (readlane), its reciprocal (v_rcp_f64 + two Newton steps), row k and column k brought to every
// element (two ds_bpermute), ONE fused multiply-add per lane.  No pivot search: N + lambda diag(N) is symmetric positive
// definite (J^T W J with the Levenberg-Marquardt term), for which elimination in natural order is backward stable; a zero
// diagonal entry (no Jacobian in some direction at all) ends in non-finite numbers, which the host reports, as it does for the
// reference's torch.linalg.inv.  The scale w /= 2 mean(w) (tracker.py:524) cancels in the step and is applied to the outputs that
// keep it (N_raw, mse).  expmap: R = I + A K + B K^2 with K = skew(t), A = sin(a) / a, B = (1 - cos a) / a^2 as series in a^2 below
// half a radian (no square root, no division, no library call: steps of a registration are small), the device library beyond.
// The same function is the tail of the tile kernels' LAST block (gn_tail_last_block below): the block that draws the last ticket
// solves, so the iteration is two launches, not three (pin_gn_loop_init + pin_gn_accumulate_solve).
//
// Measured and not kept (r06): the solve as the PROLOGUE of the next iteration's search kernel -- the first wave of every block
// solves for itself and hands the pose to its block through LDS, state and sums alternating between two copies: odometry
// 3.15-3.18 -> 3.37-3.41 ms (same box, two runs each).  Every one of the 390 blocks then starts 3-4 us late, and seven blocks
// per compute unit run the float64 chain side by side on four SIMDs.
#pragma once

namespace pin {

// value of lane `lane` (wave-uniform) in every lane
__device__ __forceinline__ double rl64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane), hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// 1 / x to within an ulp or two: the hardware's estimate and two Newton steps (five dependent instructions; the IEEE division
// is a dozen, and six of them sit on the critical path of the elimination)
__device__ __forceinline__ double rcp64(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// TAIL = false: a kernel of its own (gn_solve_kernel) -- parameters and status word come as kernel arguments, the sums were
// written by an earlier launch, the stop flag is looked at before anything is written.
// TAIL = true: the last block of a tile kernel -- the parameters are read out of the state (pin_gn_loop_init put them there: a
// tile kernel at its register limit gets ONE more argument, not twelve), the sums were added to by the other blocks of THIS
// launch (device-scope atomics: read with device-scope loads)
template <bool TAIL>
__device__ __forceinline__ void gn_solve_wave(double* sums, double* st, const pin_gn_loop_params& lp_arg, const int* status_arg) {
    const int lane = threadIdx.x & 63;
#ifdef PIN_GN_STAMPS
    const long long ts0 = wall_clock64();
#endif
    // everything the wave reads is requested up front: one memory round trip.  v0: slots 0..63 of the state (pose, loop scalars,
    // ...), v1: slots 64..79 (loop parameters, address of the status word)
    const double v0 = st[lane];
    double v1 = 0.0;
    if constexpr (TAIL) v1 = st[64 + (lane & 15)];
    double a = 0.0;
    {   // replica sum: two lanes per sum, PIN_GN_REPLICAS / 2 independent loads each
        const int i = lane & 31, h = lane >> 5;
        double v[PIN_GN_REPLICAS / 2];
#pragma unroll
        for (int r = 0; r < PIN_GN_REPLICAS / 2; ++r) {
            double* p = sums + (h * (PIN_GN_REPLICAS / 2) + r) * PIN_GN_NSUMS + i;
            if constexpr (TAIL) v[r] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v[r] = *p;
        }
#pragma unroll
        for (int r = 0; r < PIN_GN_REPLICAS / 2; ++r) a += v[r];
        a += __shfl_xor(a, 32, 64);
    }
    pin_gn_loop_params lp;
    const int* status;
    if constexpr (TAIL) {
        constexpr int L0 = PIN_GN_STATE_LP - 64;
        lp.lm_lambda = rl64(v1, L0); lp.term_thre_deg = rl64(v1, L0 + 1); lp.term_thre_m = rl64(v1, L0 + 2);
        lp.min_valid_ratio = rl64(v1, L0 + 3); lp.max_increment_ratio = rl64(v1, L0 + 4);
        lp.min_valid_points = (int)rl64(v1, L0 + 5); lp.iter_n = (int)rl64(v1, L0 + 6); lp.early_exit = (int)rl64(v1, L0 + 7);
        status = reinterpret_cast<const int*>((unsigned long long)__double_as_longlong(rl64(v1, PIN_GN_STATE_STATUS_PTR - 64)));
    } else {
        lp = lp_arg;
        status = status_arg;
    }
    int flags = 0;  // (requested now, needed for the very last store)
    if (status != nullptr) flags = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef PIN_GN_STAMPS
    const long long ts1 = wall_clock64();  // (behind the shuffle that needs every load)
#endif
    if (!TAIL && rl64(v0, PIN_GN_STATE_DONE) != 0.0) return;  // (uniform; a tile kernel has looked already)
    // lanes i and i + 32 hold sum i.  The replicas are cleared for the next iteration (nobody else touches them any more: the
    // other blocks of a fused launch have drawn their tickets behind their atomics)
#pragma unroll
    for (int i = 0; i < PIN_GN_REPLICAS * PIN_GN_NSUMS / 64; ++i) sums[i * 64 + lane] = 0.0;
    const double cnt = rint(rl64(a, 29));
    const double nsrc = rl64(v0, PIN_GN_STATE_NSRC);
    double dR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, dt[3] = {0, 0, 0};
    double res_cm = 0.0, mse = rl64(v0, PIN_GN_STATE_MSE);
    if (cnt >= 10.0) {  // tracker.py:430-432 (wave-uniform)
        // element (r, c) of [N | -J^T W r] out of the packed upper triangle: o(a <= b) = a (13 - a) / 2 + b - a, right-hand side 21 + r
        const int r = (lane >> 3) < 6 ? (lane >> 3) : 5, c = lane & 7;  // (rows 6, 7 and column 7 carry copies nobody looks at)
        const int lo = r < c ? r : c, hi = r < c ? c : r;
        double M = __shfl(a, c < 6 ? lo * (13 - lo) / 2 + hi - lo : 21 + r, 64);
        const double inv_cnt = rcp64(cnt);
        const double scale = cnt / (2.0 * rl64(a, 27));  // w /= 2 mean(w): cancels in the step, stays in N_raw and mse
        if ((lane >> 3) < 6 && c < 6) st[PIN_GN_STATE_NRAW + r * 6 + c] = scale * M;
        mse = scale * rl64(a, 30) * inv_cnt;
        res_cm = rl64(a, 28) * inv_cnt * 100.0;
        M = c == 6 ? -M : (r == c ? fma(lp.lm_lambda, M, M) : M);
#pragma unroll
        for (int k = 0; k < 6; ++k) {  // Gauss-Jordan in natural order
            const double rowk = __shfl(M, 8 * k + c, 64), colk = __shfl(M, 8 * r + k, 64);
            const double inv = rcp64(rl64(M, 9 * k));
            M = r == k ? M * inv : fma(-(colk * inv), rowk, M);
        }
        double t[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) t[k] = rl64(M, 8 * k + 6);
        // expmap (tracker.py:784-795) on the unnormalised axis: R = I + A K + B K^2, K = skew(t), K^2 = t t^T - a^2 I
        const double x = t[0] * t[0] + t[1] * t[1] + t[2] * t[2];
        double A, B;
        if (x < 0.25) {  // truncation < 2e-23
            A = 1.0 + x * (-1.0 / 6 + x * (1.0 / 120 + x * (-1.0 / 5040 + x * (1.0 / 362880 + x * (-1.0 / 39916800 + x * (1.0 / 6227020800.0 +
                x * (-1.0 / 1307674368000.0 + x * (1.0 / 355687428096000.0))))))));
            B = 0.5 + x * (-1.0 / 24 + x * (1.0 / 720 + x * (-1.0 / 40320 + x * (1.0 / 3628800 + x * (-1.0 / 479001600 + x * (1.0 / 87178291200.0 +
                x * (-1.0 / 20922789888000.0 + x * (1.0 / 6402373705728000.0))))))));
        } else {
            const double ang = sqrt(x);
            A = sin(ang) / ang;
            B = (1.0 - cos(ang)) / x;
        }
        const double d = 1.0 - B * x;
        dR[0] = fma(B * t[0], t[0], d); dR[4] = fma(B * t[1], t[1], d); dR[8] = fma(B * t[2], t[2], d);
        dR[1] = fma(B * t[0], t[1], -A * t[2]); dR[3] = fma(B * t[0], t[1], A * t[2]);
        dR[2] = fma(B * t[0], t[2], A * t[1]); dR[6] = fma(B * t[0], t[2], -A * t[1]);
        dR[5] = fma(B * t[1], t[2], -A * t[0]); dR[7] = fma(B * t[1], t[2], A * t[0]);
        dt[0] = t[3]; dt[1] = t[4]; dt[2] = t[5];
    }
    // T = dT @ T, one element per lane (lanes 0..15: row i = lane / 4, column j = lane % 4); the other lanes keep their slot's value
    double out = v0;
    {
        const int i = (lane >> 2) & 3, j = lane & 3;
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double Tcj = __shfl(v0, c * 4 + j, 64);
            const double d = c < 3 ? (i == 0 ? dR[c] : i == 1 ? dR[3 + c] : i == 2 ? dR[6 + c] : 0.0)
                                   : (i == 0 ? dt[0] : i == 1 ? dt[1] : i == 2 ? dt[2] : 1.0);
            acc = fma(d, Tcj, acc);
        }
        if (lane < 16) out = acc;
    }
    const double last = rl64(v0, PIN_GN_STATE_LAST_RES);
    bool valid = rl64(v0, PIN_GN_STATE_VALID) != 0.0;
    double last_new = last;
    // (res - last) / last > ratio, without the division: last is 1e5 or a mean of absolute values; last == 0 gives inf > ratio
    // for any res > 0 and NaN (false) for res == 0 in the reference -- as does the product form
    if (res_cm - last > lp.max_increment_ratio * last) valid = false;  // tracker.py:150-159
    else last_new = res_cm;
    if (cnt < lp.min_valid_points || cnt < lp.min_valid_ratio * nsrc) valid = false;  // :161-169
    const int it = (int)rl64(v0, PIN_GN_STATE_ITERS);
    const bool was_converged = rl64(v0, PIN_GN_STATE_CONVERGED) != 0.0;
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
        case PIN_GN_STATE_STATUS: out = (double)flags; break;  // (sticky flags of the library, e.g. a decoder outside the fp16 range)
        case PIN_GN_STATE_TICKET: out = 0.0; break;
        default: break;
    }
#ifdef PIN_GN_STAMPS
    if (lane == 0) { st[74] = (double)ts0; st[75] = (double)ts1; st[76] = (double)wall_clock64(); }
#endif
    if (lane <= PIN_GN_STATE_MSE || lane == PIN_GN_STATE_STATUS || lane == PIN_GN_STATE_TICKET) st[lane] = out;
}

// Tail of a tile kernel's block (its wave 0, behind the block's atomics into `sums`): the block that draws the last of
// gridDim.x tickets finishes the iteration.  Release / acquire at device scope around the ticket: the last block sees every
// other block's atomics.
__device__ __forceinline__ void gn_tail_last_block(double* state, double* sums) {
#ifdef PIN_GN_STAMPS
    const long long tt0 = wall_clock64();
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#ifdef PIN_GN_STAMPS
    const long long tt1 = wall_clock64();
#endif
    unsigned int t = 0;
    if ((threadIdx.x & 63) == 0)
        t = (unsigned int)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(state + PIN_GN_STATE_TICKET), 1ull,
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
    if (t != gridDim.x - 1) return;
#ifdef PIN_GN_STAMPS
    if ((threadIdx.x & 63) == 0) { state[73] = (double)tt0; state[77] = (double)tt1; state[78] = (double)wall_clock64(); }
#endif
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    pin_gn_loop_params none;
    gn_solve_wave<true>(sums, state, none, nullptr);
}

}  // namespace pin
