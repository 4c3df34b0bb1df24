// fused_coded_kernels.hpp -- part of libmixemt_hip.so (gfx950); included by mixemt_hip.hip only.
// The whole run_em inner loop (em.py:126-143) over ROW-DICTIONARY records in ONE launch.
#ifndef MIXEMT_FUSED_CODED_KERNELS_HPP
#define MIXEMT_FUSED_CODED_KERNELS_HPP

// ------------------------------------------------------------------------------------------
// K7r  em_fused_coded: the records are what run_em iterates by default above 5e7 cells, and there an iteration of
// the per-iteration path was four dependent launches (coded rows -> dense leftover rows -> column reduce ->
// finalize): 39 / 63 / 91 / 200 us at 600 / 10^4 / 3*10^4 / 10^5 rows where the coded kernel's own share is
// 1 / 14 / 43 / 144 us (profiles/r03/small_runs.txt).  This kernel is em_fused_loop_kernel's scheme (K7,
// fused_kernels.hpp: persistent grid, write-through hand-offs, counter-tree grid barriers, bounded spins, give-up
// flag) with the coded row pass (coded_row_pass, coded_kernels.hpp -- the very code em_iter_coded_kernel runs, wide
// rows included) as phase A, two workgroups of 256 per CU like that kernel:
//
//   phase A  acc_c += (w_r / Z_r) P_rc over this workgroup's dealt rows; acc -> partial[wg][c]      [write-through]
//   ---- grid barrier 1 ----
//   phase B  workgroup j owns ceil(H / 2 / grid) column pairs: T_c = sum_wg partial[wg][c] in fixed order -> tbuf
//            [write-through].  The owner ALSO keeps the log proportions of its columns -- the loop's state, as in
//            the reference (em.py:123-124, :140) -- in registers: nobody else needs them during the loop (the row pass
//            multiplies with the linear proportions), so ln T is formed by the owner alone and only T crosses the
//            fabric (K7 publishes T and ln T and keeps the whole log vector in every workgroup's LDS: 43 KB that
//            two workgroups per CU cannot afford beside the row pass's buffers).
//   ---- grid barrier 2 ----
//   phase C  EVERY workgroup, for its register columns: tot = sum_c p_c T_c, p'_c = p_c T_c / tot,
//            l1 = sum |p' - p| (em.py:53-54) -- identical bits everywhere, so all take the same stop decision;
//            the owners: ln p'_c = ln p_c + ln T_c - ln tot (em.py:87-89).
//
// Results: each owner writes its slice of ln_cur (log theta_k) and ln_new (log theta_{k+1}); workgroup 0 writes
// props_cur and the state.  Stop / resume contract as K7 (a resumed restart continues from props_cur bit for bit).
// RESIDENT: every workgroup's rows fit the row pass's LDS metadata blocks (R <= 256 x grid): fetched once per launch.
// ------------------------------------------------------------------------------------------
#define FCODED_THREADS 256
#define FCODED_MAX_M 4                     // column pairs per slice <= 16 * FCODED_MAX_M

// Register budget: the row pass alone takes 224 VGPRs and ~98 SGPRs of the 256 / 102 a wave has at two waves per SIMD.
// Everything the other phases need (hand-off descriptors, result pointers, slice geometry) is therefore parked in LDS
// by thread 0 and read back AFTER the row pass's barriers -- a first version that kept it in scalar registers spilled
// 163 SGPRs and 22 VGPRs into the row loop.  Column bounds are enforced by buffer descriptors of exactly H doubles
// (out-of-range loads return 0, out-of-range stores are dropped) instead of per-column predicates.
struct fcoded_args {
    double *partial, *tbuf, *ln_cur, *ln_new, *props_cur;
    mxm_em_state *state;
    fused_sync *sync;
    long long ldpart;
    double tol;
    int max_iter, H;
};

// (Round 5: a variant whose phase A is shared out between the quad pass and the records' pass, as in
// em_iter_quad_coded_kernel, was built and measured: 1.41 ms per iteration at 10^6 rows against 1.45 for this kernel and
// 1.35 for the per-iteration kernels with quads -- both passes' scalar state beside phases B / C spilled 77 SGPRs.  Not
// kept; beside a quad dictionary mxm_em_loop_coded takes the per-iteration kernels.  profiles/r05/quads_product_fused_1m.txt)
template <int NCH, int NBUF, bool RESIDENT>
__global__ __launch_bounds__(FCODED_THREADS, 2) void em_fused_coded_kernel(
    const uint8_t *__restrict__ rec, const int64_t *__restrict__ rec_off, const int32_t *__restrict__ ndist, int ldc,
    const double *__restrict__ w, const int64_t *__restrict__ wide_rows, int64_t n_wide, int64_t R, int B, int chunk,
    fcoded_args args) {
    constexpr int THREADS = FCODED_THREADS, NW = THREADS / 64;
    __shared__ d2 cred[NW][FCODED_MAX_M][16];              // column reduce: per-wave slice sums
    __shared__ double bred[2][NW];                         // phase C block sums
    __shared__ int ok_flag;
    __shared__ d2 s_own_ln[16 * FCODED_MAX_M];             // the owners' log proportions (kept out of the row pass's registers)
    __shared__ fcoded_args s_a;
    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    if (t == 0) s_a = args;
    __syncthreads();

    unsigned epoch = 0;
    bool meta_ready = false;
#ifdef FUSED_STAMPS
    unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_amdgcn_s_memrealtime();
#endif
    for (int b = 0; b < B; ++b) {
        int iters, done = 0;
        double l1 = 0.0;
        double p[NCH][4];
        {
            const int H = s_a.H;
            const mxm_em_state *st = s_a.state + b;
            if (st->done != 0) continue;                   // written before the launch: plain load is fine
            const double *lc_g = s_a.ln_cur + (int64_t)b * H, *pc_g = s_a.props_cur + (int64_t)b * H;
            iters = st->iters;
            // the linear proportions, replicated in every workgroup: thread t holds the columns 4 (t + THREADS k) + e
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 4 * (t + k * THREADS) + e;
                    // a resumed restart continues with the very proportions it stopped with
                    p[k][e] = (c < H) ? (iters > 0 ? pc_g[c] : exp(lc_g[c])) : 0.0;
                }
            }
            const int nwg = (int)gridDim.x, ncol2 = H >> 1;
            const int cp2 = (ncol2 + nwg - 1) / nwg;
            const int GW = (cp2 <= 8) ? 8 : 16;
            const int own_pi = (t < GW * ((cp2 + GW - 1) / GW)) ? ((t % GW) + GW * (t / GW)) : cp2;
            const int own_c2 = (int)blockIdx.x * cp2 + own_pi;
            if (own_pi < cp2 && own_c2 < ncol2) s_own_ln[t] = d2{lc_g[2 * own_c2], lc_g[2 * own_c2 + 1]};   // own slot: no barrier needed
        }
        for (int it = 0; it < chunk && done == 0; ++it) {
#ifdef FUSED_STAMPS
            if (blockIdx.x == 0 && threadIdx.x == 0) { stamp_last = __builtin_amdgcn_s_memrealtime(); stamp_acc[5] += 1; }
#endif
            // ================= phase A: row pass over the records =================
            double acc[NCH][4];
#pragma unroll
            for (int k = 0; k < NCH; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[k][e] = 0.0;
            coded_row_pass<THREADS, NCH, NBUF, !RESIDENT, RESIDENT, false>(rec, rec_off, ndist, ldc, w, wide_rows, n_wide, R, p,
                                                                    acc, meta_ready);
            __syncthreads();                               // (LDS reads below must not move above the row pass)
            // slice geometry and hand-off descriptors, from LDS (see "register budget" above)
            const int H = s_a.H;
            const int64_t ldpart = s_a.ldpart;
            const int ncol2 = H >> 1;                      // H is even (records need it)
            const int nwg = (int)gridDim.x;
            const int cp2 = (ncol2 + nwg - 1) / nwg;       // column pairs per workgroup slice (<= 16 * FCODED_MAX_M)
            // slice walk: sub-groups of GW lanes take the partial rows g, g + G, ...; lane l of a sub-group the pairs l + GW m
            const int GW = (cp2 <= 8) ? 8 : 16;
            const int nm = (cp2 + GW - 1) / GW;
            const int G = THREADS / GW;
            const int lg = t & (GW - 1), gsub = t / GW;
            // the slice column pair this thread owns in phases B / C (threads t < GW * nm, one pair each)
            const int own_pi = (t < GW * nm) ? ((t % GW) + GW * (t / GW)) : cp2;
            const int own_c2 = (int)blockIdx.x * cp2 + own_pi;
            const bool owner = own_pi < cp2 && own_c2 < ncol2;
            fused_sync *sync = s_a.sync;
            {
                // this workgroup's partial row: a descriptor of exactly H doubles drops the columns past the row
                const auto mine = __builtin_amdgcn_make_buffer_rsrc(s_a.partial + (int64_t)blockIdx.x * ldpart, 0, H * 8, 0x00020000);
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const int c = 4 * (t + k * THREADS);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu4, d2{acc[k][0], acc[k][1]}), mine, c * 8, 0, FUSED_SC1);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu4, d2{acc[k][2], acc[k][3]}), mine, c * 8 + 16, 0, FUSED_SC1);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FUSED_STAMP(0);
            if (!fused_grid_barrier(sync, ++epoch, nwg, &ok_flag)) { done = -1; break; }
            FUSED_STAMP(1);

            // ================= phase B: this workgroup's slice of the column sums =================
            const auto part_rsrc = __builtin_amdgcn_make_buffer_rsrc(s_a.partial, 0, (int)((int64_t)nwg * ldpart * 8), 0x00020000);
            const auto t_rsrc = __builtin_amdgcn_make_buffer_rsrc(s_a.tbuf, 0, H * 8, 0x00020000);
#pragma unroll 1
            for (int m = 0; m < nm; ++m) {
                const int pi = lg + GW * m;
                const int c2 = (int)blockIdx.x * cp2 + pi;
                const bool valid = (pi < cp2) && (c2 < ncol2);
                d2 s = d2{0.0, 0.0};
                if (valid) {
                    auto load_part = [&](int gg) {
                        return __builtin_bit_cast(d2, (fu4)__builtin_amdgcn_raw_buffer_load_b128(
                                                          part_rsrc, (int)((int64_t)gg * ldpart * 8) + c2 * 16, 0, FUSED_SC1));
                    };
                    int g = gsub;
                    for (; g + 7 * G < nwg; g += 8 * G) {   // eight loads in flight per lane
                        d2 v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = load_part(g + G * u);
#pragma unroll
                        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; }
                    }
                    for (; g + 3 * G < nwg; g += 4 * G) {
                        d2 v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = load_part(g + G * u);
#pragma unroll
                        for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; }
                    }
                    for (; g < nwg; g += G) {
                        const d2 v = load_part(g);
                        s.x += v.x; s.y += v.y;
                    }
                }
                // the wave's sub-groups (lanes l, l + GW, ...), fixed tree; all lanes take part
                if (GW == 8) { s.x += __shfl_xor(s.x, 8, 64); s.y += __shfl_xor(s.y, 8, 64); }
                s.x += __shfl_xor(s.x, 16, 64); s.y += __shfl_xor(s.y, 16, 64);
                s.x += __shfl_xor(s.x, 32, 64); s.y += __shfl_xor(s.y, 32, 64);
                if (lane < GW) cred[wv][m][lane] = s;
            }
            __syncthreads();
            d2 own_lt = d2{0.0, 0.0};
            if (owner) {
                const int m = t / GW, l = t % GW;
                d2 tot = cred[0][m][l];
#pragma unroll
                for (int q = 1; q < NW; ++q) { tot.x += cred[q][m][l].x; tot.y += cred[q][m][l].y; }
                own_lt = d2{log(tot.x), log(tot.y)};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu4, tot), t_rsrc, own_c2 * 16, 0, FUSED_SC1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FUSED_STAMP(2);
            if (!fused_grid_barrier(sync, ++epoch, nwg, &ok_flag)) { done = -1; break; }
            FUSED_STAMP(3);

            // ================= phase C: normalise, convergence test (every workgroup alike) =================
            double T[NCH][4];
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int c = 4 * (t + k * THREADS);       // columns past the row: the descriptor returns 0, and p = 0 there
                const d2 a = __builtin_bit_cast(d2, (fu4)__builtin_amdgcn_raw_buffer_load_b128(t_rsrc, c * 8, 0, FUSED_SC1));
                const d2 bq = __builtin_bit_cast(d2, (fu4)__builtin_amdgcn_raw_buffer_load_b128(t_rsrc, c * 8 + 16, 0, FUSED_SC1));
                T[k][0] = a.x; T[k][1] = a.y; T[k][2] = bq.x; T[k][3] = bq.y;
            }
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < NCH; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) s = fma(p[k][e], T[k][e], s);
            s = wave_sum_lane63(s);
            if (lane == 63) bred[0][wv] = s;
            __syncthreads();
            double tot = bred[0][0];
#pragma unroll
            for (int q = 1; q < NW; ++q) tot += bred[0][q];
            const double ltot = log(tot);
            const double rtot = 1.0 / tot;
            double d = 0.0;
#pragma unroll
            for (int k = 0; k < NCH; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) d += fabs(p[k][e] * T[k][e] * rtot - p[k][e]);     // em.py:53-54 (0 past the row)
            d = wave_sum_lane63(d);
            if (lane == 63) bred[1][wv] = d;
            __syncthreads();
            l1 = bred[1][0];
#pragma unroll
            for (int q = 1; q < NW; ++q) l1 += bred[1][q];
            ++iters;
            const bool conv = l1 < s_a.tol;
            done = conv ? 1 : (iters >= s_a.max_iter ? 2 : 0);
            d2 own_next = d2{0.0, 0.0};
            if (owner) {
                const d2 own_ln = s_own_ln[t];
                own_next = d2{own_ln.x + own_lt.x - ltot, own_ln.y + own_lt.y - ltot};          // em.py:87-89
            }
            if (done == 0) {                               // em.py:140: props <- new_props
                if (owner) s_own_ln[t] = own_next;
#pragma unroll
                for (int k = 0; k < NCH; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) p[k][e] = p[k][e] * T[k][e] * rtot;              // stays 0 past the row
            } else if (owner) {                            // stopped: ln_new = log theta_{k+1}; ln_cur / p stay at theta_k
                double *ln_g = s_a.ln_new + (int64_t)b * H;
                ln_g[2 * own_c2] = own_next.x;
                ln_g[2 * own_c2 + 1] = own_next.y;
            }
            FUSED_STAMP(4);
        }
        // ---- results of this restart: ln_cur = log theta_k (the owners' slices), props_cur (workgroup 0) ----
        {
            const int H = s_a.H;
            const int nwg = (int)gridDim.x, ncol2 = H >> 1;
            const int cp2 = (ncol2 + nwg - 1) / nwg;
            const int GW = (cp2 <= 8) ? 8 : 16;
            const int own_pi = (t < GW * ((cp2 + GW - 1) / GW)) ? ((t % GW) + GW * (t / GW)) : cp2;
            const int own_c2 = (int)blockIdx.x * cp2 + own_pi;
            if (own_pi < cp2 && own_c2 < ncol2) {
                double *lc_g = s_a.ln_cur + (int64_t)b * H;
                const d2 own_ln = s_own_ln[t];
                lc_g[2 * own_c2] = own_ln.x;
                lc_g[2 * own_c2 + 1] = own_ln.y;
            }
            if (blockIdx.x == 0) {
                double *pc_g = s_a.props_cur + (int64_t)b * H;
#pragma unroll
                for (int k = 0; k < NCH; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = 4 * (t + k * THREADS) + e;
                        if (c < H) pc_g[c] = p[k][e];
                    }
                if (t == 0) {
                    mxm_em_state *st = s_a.state + b;
                    st->iters = iters;
                    st->l1 = l1;
                    st->done = done;
                }
            }
        }
        if (done < 0) return;                              // the grid gave up: every workgroup leaves
    }
#ifdef FUSED_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int i = 0; i < 6; ++i) s_a.sync->stamps[i] = stamp_acc[i];
#endif
}

#endif  // MIXEMT_FUSED_CODED_KERNELS_HPP
