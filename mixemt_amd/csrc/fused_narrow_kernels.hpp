// fused_narrow_kernels.hpp -- part of libmixemt_hip.so (gfx950); included by mixemt_hip.hip only.
// The whole run_em inner loop (em.py:126-143) of a NARROW matrix in ONE launch.
#ifndef MIXEMT_FUSED_NARROW_KERNELS_HPP
#define MIXEMT_FUSED_NARROW_KERNELS_HPP

// ------------------------------------------------------------------------------------------
// K7n  em_fused_narrow: the refinement EM runs on the contributors' columns only (bin/mixemt:311-320 after
// preprocess.py:230-251: R x H' with H' = 1 .. 10 in practice) -- 24 MB at 10^6 x 3, and an iteration of the
// per-iteration path was three dependent launches (estep_narrow 9.9 us + colreduce 8.7 + finalize 7.7) plus the host's
// enqueueing: 152 us per iteration for ~5 us of data movement (profiles/r04/pipeline_1m_records_kernel_stats.csv).
// Here the loop stays inside one persistent grid, as K7 / K7r do for wide matrices, and the matrix stays IN REGISTERS:
//
//   once     thread g holds the rows g, g + G, ... (G = threads of the grid, RPT rows per thread) as
//            e_rh = exp(M_rh - rowmax_r) and their weights: the exponentials are taken once per launch
//   phase A  Z_r = sum_h p_h e_rh ; acc_h += (w_r / Z_r) e_rh over the thread's rows (estep_narrow_kernel's
//            arithmetic); workgroup sum in fixed order -> partial[it & 1][h][wg]               [write-through]
//   ---- grid barrier (fused_grid_barrier: counter tree, bounded spins, give-up flag) ----
//   phase B  EVERY workgroup: T_h = sum_wg partial[h][wg] in ONE fixed order (thread t takes workgroup t: the grid
//            has at most as many workgroups as a workgroup has threads), then the reference's own update in its own
//            variable, the LOG proportions (em.py:87-89, :53-54), exactly as mxm_m_finalize forms it:
//                ln p'_h = ln p_h + ln T_h - ln sum_h p_h T_h ;  p'_h = exp(ln p'_h) ;  l1 = sum_h |p'_h - p_h|
//            identical bits in every workgroup, so all take the same stop decision: ONE barrier per iteration
//            (the partial sums are double buffered by iteration parity: a workgroup that is already in the next
//            iteration's phase A writes the other buffer while a slower one still reads this one; it cannot get
//            two iterations ahead, because the next barrier needs the slower one's arrival).
//
// One workgroup of 512 per CU.  H <= HMAX columns sit in per-thread registers (HMAX = 4, 8, 16) beside the inlined
// fp64 log / exp (105 registers before the first matrix cell): the instances that compile without scratch are
// RPT <= 8 at HMAX = 4, RPT <= 4 at 8, RPT = 1 at 16 -- 1.05 * 10^6 rows at H <= 4, 5.2 * 10^5 at H <= 8, 1.3 * 10^5 at
// H <= 16 on 256 CUs.  Wider refinements (17 .. 64 columns) and taller matrices keep the per-iteration kernels.
// Stop / resume / give-up contract as K7.
// ------------------------------------------------------------------------------------------
#define FNARROW_THREADS 512
#define FNARROW_MAX_WG 512                 // phase B reads one partial per thread
#define FNARROW_MAX_CELLS 32               // RPT * HMAX doubles of matrix per thread (and RPT = 1 at HMAX = 16): spill-free instances

template <int HMAX, int RPT>
__global__ __launch_bounds__(FNARROW_THREADS, 2) void em_fused_narrow_kernel(
    const double *__restrict__ M, int64_t ldm, const double *__restrict__ w, int64_t R, int H, int B,
    double *__restrict__ ln_cur, double *__restrict__ ln_new, double *__restrict__ props_cur,
    mxm_em_state *__restrict__ state, double tol, int max_iter, int chunk, double *__restrict__ partial /* [2][H][ldn] */,
    int ldn, fused_sync *__restrict__ sync) {
    constexpr int THREADS = FNARROW_THREADS, NW = THREADS / 64;
    static_assert(HMAX * RPT <= FNARROW_MAX_CELLS, "matrix cells per thread");
    __shared__ double s_red[HMAX][NW];
    __shared__ int ok_flag;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int nwg = (int)gridDim.x;

    // ---- the thread's rows, linearised once: e = exp(M - rowmax), pad columns and rows past the matrix 0 ----
    double e[RPT][HMAX], wr[RPT];
    const int64_t gstride = (int64_t)nwg * THREADS;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int64_t r = (int64_t)blockIdx.x * THREADS + t + k * gstride;
        const bool live = r < R;
        double x[HMAX];
        double m = -INFINITY;
#pragma unroll
        for (int h = 0; h < HMAX; ++h) {
            x[h] = (live && h < H) ? M[r * ldm + h] : -INFINITY;
            m = fmax(m, x[h]);
        }
        const double shift = isfinite(m) ? m : 0.0;
#pragma unroll
        for (int h = 0; h < HMAX; ++h) e[k][h] = exp(x[h] - shift);          // exp(-inf) = 0
        wr[k] = live ? (w != nullptr ? w[r] : 1.0) : 0.0;
    }

    unsigned epoch = 0;
    for (int b = 0; b < B; ++b) {
        mxm_em_state *st = state + b;
        if (st->done != 0) continue;                       // written before the launch: plain load is fine
        int iters = st->iters, done = 0;
        double l1 = 0.0;
        double lc[HMAX], p[HMAX], ln_next[HMAX];
#pragma unroll
        for (int h = 0; h < HMAX; ++h) {
            lc[h] = (h < H) ? ln_cur[(int64_t)b * H + h] : -INFINITY;
            // a resumed restart continues with the very proportions it stopped with (= exp(ln_cur), mxm_m_finalize's)
            p[h] = (h < H) ? (iters > 0 ? props_cur[(int64_t)b * H + h] : exp(lc[h])) : 0.0;
            ln_next[h] = lc[h];
        }
        for (int it = 0; it < chunk && done == 0; ++it) {
            // ================= phase A: the thread's rows =================
            double acc[HMAX];
#pragma unroll
            for (int h = 0; h < HMAX; ++h) acc[h] = 0.0;
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                double z = 0.0;
#pragma unroll
                for (int h = 0; h < HMAX; ++h) z = fma(p[h], e[k][h], z);
                const double c = weight_over_norm(wr[k], z);
#pragma unroll
                for (int h = 0; h < HMAX; ++h) acc[h] = fma(c, e[k][h], acc[h]);
            }
#pragma unroll
            for (int h = 0; h < HMAX; ++h) {
                const double a = wave_sum_lane63(acc[h]);
                if (lane == 63) s_red[h][wv] = a;
            }
            __syncthreads();
            if (t < H) {                                   // column t of this workgroup's sums, waves in order
                double a = s_red[t][0];
#pragma unroll
                for (int q = 1; q < NW; ++q) a += s_red[t][q];
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(partial + ((int64_t)(epoch & 1u) * H + t) * ldn, 0, nwg * 8, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fin_u2, a), rs, (int)blockIdx.x * 8, 0, FUSED_SC1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned parity = epoch & 1u;            // the buffer this iteration wrote (epoch counts iterations)
            if (!fused_grid_barrier(sync, ++epoch, nwg, &ok_flag)) { done = -1; break; }

            // ================= phase B: T_h, the log update, the L1 test (every workgroup alike) =================
            for (int h = 0; h < H; ++h) {                  // uniform
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(partial + ((int64_t)parity * H + h) * ldn, 0, nwg * 8, 0x00020000);
                const fin_u2 raw = __builtin_amdgcn_raw_buffer_load_b64(rs, t * 8, 0, FUSED_SC1);     // past the grid: 0
                const double v = wave_sum_lane63(__hiloint2double((int)raw.y, (int)raw.x));
                if (lane == 63) s_red[h][wv] = v;
            }
            __syncthreads();
            double T[HMAX], tot = 0.0;
#pragma unroll
            for (int h = 0; h < HMAX; ++h) {
                double a = 0.0;
                if (h < H) {
                    a = s_red[h][0];
#pragma unroll
                    for (int q = 1; q < NW; ++q) a += s_red[h][q];
                }
                T[h] = a;
                tot = fma(p[h], a, tot);                   // p = 0 past the matrix
            }
            __syncthreads();                               // s_red is free for the next iteration's phase A
            const double ltot = log(tot);
            l1 = 0.0;
            double pn[HMAX];
#pragma unroll
            for (int h = 0; h < HMAX; ++h) {
                if (h < H) {
                    ln_next[h] = lc[h] + log(T[h]) - ltot; // em.py:87-89
                    pn[h] = exp(ln_next[h]);
                    l1 += fabs(pn[h] - p[h]);              // em.py:53-54
                } else {
                    pn[h] = 0.0;
                }
            }
            ++iters;
            const bool conv = l1 < tol;
            done = conv ? 1 : (iters >= max_iter ? 2 : 0);
            if (done == 0) {                               // em.py:140: props <- new_props
#pragma unroll
                for (int h = 0; h < HMAX; ++h) {
                    lc[h] = ln_next[h];
                    p[h] = pn[h];
                }
            }
        }
        // ---- results: ln_cur = log theta_k, props_cur = exp of it, ln_new = log theta_{k+1} (workgroup 0) ----
        if (blockIdx.x == 0) {
#pragma unroll
            for (int h = 0; h < HMAX; ++h) {
                if (t == h && h < H) {
                    ln_cur[(int64_t)b * H + h] = lc[h];
                    props_cur[(int64_t)b * H + h] = p[h];
                    if (done > 0) ln_new[(int64_t)b * H + h] = ln_next[h];
                }
            }
            if (t == 0) {
                st->iters = iters;
                st->l1 = l1;
                st->done = done;
            }
        }
        if (done < 0) return;                              // the grid gave up: every workgroup leaves
    }
}

#endif  // MIXEMT_FUSED_NARROW_KERNELS_HPP
