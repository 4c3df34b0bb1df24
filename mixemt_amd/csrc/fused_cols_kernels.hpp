// fused_cols_kernels.hpp -- part of libmixemt_hip.so (gfx950); included by mixemt_hip.hip only.
// The one-launch EM loop (em.py:126-143) for the SMALLEST matrices, transposed: columns are split
// over the workgroups, the matrix lives in registers, and only row sums cross the fabric.
#ifndef MIXEMT_FUSED_COLS_KERNELS_HPP
#define MIXEMT_FUSED_COLS_KERNELS_HPP

// ------------------------------------------------------------------------------------------
// K7c em_fused_cols: em_fused_loop_kernel (K7) splits ROWS over the workgroups; what it pays per
// iteration at a few rows per workgroup is the exchange of column partials -- 43 KB published and
// drained per workgroup, read back slice-wise, T / ln T out and in: 16 us of which the arithmetic is 1
// (DESIGN.md 4.4).  For matrices of up to RPT * 512 rows (1536) this kernel turns the decomposition
// around:
//
//   * workgroup j owns the columns [j*cp, (j+1)*cp), cp = ceil(H / grid) (22 at H = 5408), for ALL
//     rows; thread t keeps rows t, t + 512, ... of that slice in VGPRs for the whole launch
//     (RPT x CP doubles: the whole matrix sits in the chip's register files, nothing is re-read);
//   * per iteration   z_r[j] = sum_{c in slice} p_c P_rc            thread local, no reduction
//                     publish z[j][0..R)                            4.8 KB at 600 rows (not 43 KB)
//                     ---- grid barrier 1 ----
//                     owner of rows [m*rp, (m+1)*rp): Z_r = sum_j z_r[j] (fixed order),
//                     c_r = w_r / Z_r; publish c_r                  a few values per workgroup
//                     ---- grid barrier 2 ----
//                     every workgroup reads c[0..R) (4.8 KB), T_c = sum_r c_r P_rc for ITS columns
//                     (thread local products, one workgroup reduction) -- T never leaves the workgroup
//                     p'_c = p_c T_c / W,  ln p'_c = ln p_c + ln T_c - ln W,   W = sum_r w_r
//   * W is the M-step's normaliser: sum_h p_h T_h = sum_r w_r (sum_h p_h P_rh) / Z_r = sum_r w_r up to
//     the rounding of Z_r, so no exchange is needed for it (em.py:89 computes the same number from the
//     sums themselves; the difference is one rounding, and it cannot accumulate: whatever factor the
//     proportions are off by, Z_r carries it and T divides it out again);
//   * the L1 test (em.py:53-54) needs a sum over all workgroups: its partials ride on the NEXT
//     iteration's first exchange.  The step is committed speculatively, half an iteration later every
//     workgroup sums the same 256 partials in the same order and either goes on or stops with exactly
//     the (theta_k, theta_{k+1}) of the iteration that passed the test (kept in a backup) -- the same
//     iteration the reference stops on.
// Hand-offs, barrier, bounded spins, state / resume contract: as K7 (fused_kernels.hpp).
// ------------------------------------------------------------------------------------------
#define FCOLS_THREADS 512
#define FCOLS_MAX_RPT 3                     // rows per thread: R <= 1536
#define FCOLS_MAX_CP 24                     // columns per workgroup: H <= 24 * grid
#define FCOLS_NQ 3                          // passes of the Z reduce, two owned rows each: ceil(R / grid) <= 6

template <int CP, int RPT>
__global__ __launch_bounds__(FCOLS_THREADS, 2) void em_fused_cols_kernel(
    const double *__restrict__ P, int64_t ldp, const double *__restrict__ w, int64_t R, int H, int B,
    double *ln_cur, double *ln_new, double *props_cur, mxm_em_state *state, double tol, int max_iter,
    int chunk, double *zpart, int64_t ldz, double *cbuf, double *l1part, fused_sync *sync) {
    constexpr int THREADS = FCOLS_THREADS, NW = THREADS / 64;
    __shared__ double s_p[CP], s_lnp[CP], s_p_prev[CP], s_lnp_prev[CP];   // this workgroup's columns (state)
    __shared__ double s_t[NW][CP];                         // per-wave column sums
    __shared__ double s_red[NW];                           // block sums (W, L1, Z)
    __shared__ double s_bc[4];                             // broadcast scalars
    __shared__ int ok_flag;
    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int nwg = (int)gridDim.x;
    const int cp = (H + nwg - 1) / nwg;                    // columns per workgroup (<= CP)
    const int c0 = (int)blockIdx.x * cp;
    const int ncm = (c0 < H) ? ((H - c0) < cp ? (H - c0) : cp) : 0;        // my columns
    const int rp = (int)((R + nwg - 1) / nwg);             // rows whose Z this workgroup owns
    const int64_t m0 = (int64_t)blockIdx.x * rp;

    // fixed-order sum over the workgroup; every thread gets the result (one barrier pair)
    auto block_sum = [&](double v) -> double {
        v = wave_sum_lane63(v);
        __syncthreads();
        if (lane == 63) s_red[wv] = v;
        __syncthreads();
        double r = s_red[0];
#pragma unroll
        for (int q = 1; q < NW; ++q) r += s_red[q];
        return r;
    };

    // ---- this thread's rows of this workgroup's column slice: loaded once, kept in registers ----------
    double x[RPT][CP];
    double wsum_mine = 0.0;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int64_t r = (int64_t)t + (int64_t)i * THREADS;
        const bool live = r < R;
        const double *row = P + (live ? r : 0) * ldp + c0;
#pragma unroll
        for (int c = 0; c < CP; ++c) x[i][c] = (live && c < ncm) ? row[c] : 0.0;
        if (live) wsum_mine += (w != nullptr) ? w[r] : 1.0;
    }
    const double wtot = block_sum(wsum_mine);              // W = sum_r w_r, the same bits in every workgroup
    const double ln_wtot = log(wtot);
    const double r_wtot = 1.0 / wtot;

    const auto z_rsrc = __builtin_amdgcn_make_buffer_rsrc(zpart, 0, (int)((int64_t)nwg * ldz * 8), 0x00020000);
    const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc(cbuf, 0, (int)(ldz * 8), 0x00020000);
    // L1 partials are double buffered by publish: a workgroup that leaves a restart right after the test (stop
    // or end of chunk) publishes the NEXT restart's partials with no barrier in between, while a slower one
    // may still be summing this test's -- they must not share a buffer.  (Two publishes later a barrier has
    // passed that the slow one could only reach after its reads.)
    const auto l_rsrc = __builtin_amdgcn_make_buffer_rsrc(l1part, 0, 2 * nwg * 8, 0x00020000);
    int lbuf = 0;
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    auto ld_f64 = [&](decltype(z_rsrc) rsrc, int byte_off) -> double {
        return __builtin_bit_cast(double, (u2v)__builtin_amdgcn_raw_buffer_load_b64(rsrc, byte_off, 0, FUSED_SC1));
    };
    auto st_f64 = [&](decltype(z_rsrc) rsrc, int byte_off, double v) {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, v), rsrc, byte_off, 0, FUSED_SC1);
    };

    unsigned epoch = 0;
    for (int b = 0; b < B; ++b) {
        mxm_em_state *st = state + b;
        if (st->done != 0) continue;                       // written before the launch: plain load is fine
        double *lc_g = ln_cur + (int64_t)b * H, *ln_g = ln_new + (int64_t)b * H, *pc_g = props_cur + (int64_t)b * H;
        int iters = st->iters;
        __syncthreads();
        if (t < CP) {
            const bool own = t < ncm;
            const double l = own ? lc_g[c0 + t] : -INFINITY;
            s_lnp[t] = l;
            s_p[t] = own ? (iters > 0 ? pc_g[c0 + t] : exp(l)) : 0.0;     // a resumed restart continues bit for bit
            s_lnp_prev[t] = l;
            s_p_prev[t] = s_p[t];
        }
        __syncthreads();
        int done = 0, decided = 0;
        bool pending = false;                              // a committed step is waiting for its L1 test
        double l1 = st->l1, l1_mine = 0.0;
        for (;;) {
            // ---- row sums of my columns under the current proportions; published with the pending L1 partial
            double z[RPT];
#pragma unroll
            for (int i = 0; i < RPT; ++i) z[i] = 0.0;
#pragma unroll
            for (int c = 0; c < CP; ++c) {
                const double pc = s_p[c];
#pragma unroll
                for (int i = 0; i < RPT; ++i) z[i] = fma(x[i][c], pc, z[i]);
            }
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int64_t r = (int64_t)t + (int64_t)i * THREADS;
                if (r < R) st_f64(z_rsrc, (int)(((int64_t)blockIdx.x * ldz + r) * 8), z[i]);
            }
            lbuf ^= 1;
            if (t == 0) st_f64(l_rsrc, (lbuf * nwg + (int)blockIdx.x) * 8, l1_mine);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!fused_grid_barrier(sync, ++epoch, nwg, &ok_flag)) { done = -1; break; }

            // ---- the pending step's L1 test: every workgroup sums the same partials in the same order
            double zsum[FCOLS_NQ];                         // my owned rows' partials (loads in flight)
            const int g = t & 255, half = t >> 8;          // 256 partials per row, two rows per pass
#pragma unroll
            for (int q = 0; q < FCOLS_NQ; ++q) {
                const int64_t m = m0 + 2 * q + half;
                zsum[q] = 0.0;
                if (2 * q < rp && 2 * q + half < rp && m < R) {
                    double acc = 0.0;
                    for (int gg = g; gg < nwg; gg += 256) acc += ld_f64(z_rsrc, (int)(((int64_t)gg * ldz + m) * 8));
                    zsum[q] = acc;
                }
            }
            if (pending) {
                double v = 0.0;
                for (int gg = t; gg < nwg; gg += THREADS) v += ld_f64(l_rsrc, (lbuf * nwg + gg) * 8);
                l1 = block_sum(v);
                ++iters;
                ++decided;
                done = (l1 < tol) ? 1 : (iters >= max_iter ? 2 : 0);
                if (done != 0) break;                      // (theta_k, theta_{k+1}) = (backup, current)
                if (decided >= chunk) break;               // hand the committed state back to the host
            }
            // ---- Z_r of the rows this workgroup owns, c_r = w_r / Z_r ------------------------------------
#pragma unroll
            for (int q = 0; q < FCOLS_NQ; ++q) {
                if (2 * q < rp) {                          // workgroup uniform
                    double v = wave_sum_lane63(zsum[q]);
                    __syncthreads();
                    if (lane == 63) s_red[wv] = v;
                    __syncthreads();
                    if (t < 2) {                           // waves 0-3 hold row 2q, waves 4-7 row 2q + 1
                        const int64_t m = m0 + 2 * q + t;
                        if (2 * q + t < rp && m < R) {
                            const double zr = ((s_red[4 * t] + s_red[4 * t + 1]) + s_red[4 * t + 2]) + s_red[4 * t + 3];
                            st_f64(c_rsrc, (int)(m * 8), weight_over_norm((w != nullptr) ? w[m] : 1.0, zr));
                        }
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!fused_grid_barrier(sync, ++epoch, nwg, &ok_flag)) { done = -1; break; }

            // ---- T_c for my columns: thread-local products, one reduction over the workgroup -----------
            double cr[RPT];
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int64_t r = (int64_t)t + (int64_t)i * THREADS;
                cr[i] = (r < R) ? ld_f64(c_rsrc, (int)(r * 8)) : 0.0;
            }
#pragma unroll
            for (int c = 0; c < CP; ++c) {
                double tv = 0.0;
#pragma unroll
                for (int i = 0; i < RPT; ++i) tv = fma(cr[i], x[i][c], tv);
                tv = wave_sum_lane63(tv);
                if (lane == 63) s_t[wv][c] = tv;
            }
            __syncthreads();
            // ---- the step, committed speculatively (em.py:87-89, :140); backup for a stop at the next test
            double term = 0.0;
            if (t < CP) {
                double tc = s_t[0][t];
#pragma unroll
                for (int q = 1; q < NW; ++q) tc += s_t[q][t];
                const double pc = s_p[t], lc = s_lnp[t];
                const double pn = (t < ncm) ? pc * tc * r_wtot : 0.0;
                s_p_prev[t] = pc;
                s_lnp_prev[t] = lc;
                s_p[t] = pn;
                s_lnp[t] = (t < ncm) ? lc + log(tc) - ln_wtot : -INFINITY;
                term = (t < ncm) ? fabs(pn - pc) : 0.0;
            }
            if (wv == 0) {
                term = wave_sum_lane63(term);              // CP <= 24 lanes carry a term, fixed tree
                if (lane == 63) s_bc[0] = term;
            }
            __syncthreads();
            l1_mine = s_bc[0];
            pending = true;
        }
        // ---- results of this restart: every workgroup writes its columns ---------------------------------
        __syncthreads();
        if (t < ncm) {
            if (done > 0) {                                // stopped: theta_k from the backup, theta_{k+1} current
                lc_g[c0 + t] = s_lnp_prev[t];
                pc_g[c0 + t] = s_p_prev[t];
                ln_g[c0 + t] = s_lnp[t];
            } else {                                       // chunk over (or gave up): the committed state
                lc_g[c0 + t] = s_lnp[t];
                pc_g[c0 + t] = s_p[t];
            }
        }
        if (blockIdx.x == 0 && t == 0) {
            st->iters = iters;
            st->l1 = l1;
            st->done = done;
        }
        if (done < 0) return;                              // the grid gave up: every workgroup leaves
    }
}

#endif  // MIXEMT_FUSED_COLS_KERNELS_HPP
