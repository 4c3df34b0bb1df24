// fused_kernels.hpp -- part of libmixemt_hip.so (gfx950); included by mixemt_hip.hip only.
// The whole run_em inner loop (em.py:126-143) of a cache-resident matrix in ONE launch.
#ifndef MIXEMT_FUSED_KERNELS_HPP
#define MIXEMT_FUSED_KERNELS_HPP

// ------------------------------------------------------------------------------------------
// K7  em_fused_loop: real mixemt inputs are de-duplicated signatures (preprocess.py:163-174,
// :218-220): R ~ 10^3..10^5 rows, a matrix of tens of MB that lives in L2 / Infinity Cache.
// There the per-iteration kernels (stream -> column reduce -> finalize) are three DEPENDENT
// launches of a few microseconds each (31 us per iteration at 600 x 5408, 13.5 of them one
// workgroup doing 5408 fp64 log + exp).  This kernel keeps the iteration inside one persistent
// grid, one workgroup of 512 per CU, and runs until the restart has stopped:
//
//   phase A  row pass, exactly em_iter_wide_kernel's arithmetic (Z_r, acc_h += (w_r/Z_r) P_rh)
//            over this workgroup's dealt rows -- the same rows every iteration, default cache
//            policy, so they are served from this XCD's L2 after the first pass;
//            acc -> partial[wg][h]                                               [write-through]
//   ---- grid barrier 1 ----
//   phase B  workgroup j owns a slice of ceil(H / 2 / grid) column pairs: T_h = sum_wg partial
//            (fixed order), ln T_h; both written to tbuf                        [write-through]
//   ---- grid barrier 2 ----
//   phase C  EVERY workgroup, for its own register columns: tot = sum_h p_h T_h,
//            ln p'_h = ln p_h + ln T_h - ln tot (em.py:87-89; the loop's state, exact for
//            proportions that underflow), p'_h = p_h T_h / tot, l1 = sum |p' - p| (em.py:53-54)
//            -- all workgroups compute identical bits, so all take the same stop decision
//            without a third exchange.  The H logarithms are spread over the grid in phase B
//            (one per thread there); phase C has one log per workgroup.
//
// Cross-workgroup traffic follows the guide's write-through hand-off (MI355X_MICROARCH.md,
// "Valid forms", row 1): every handed-off byte is stored sc1 and drained (s_waitcnt vmcnt(0))
// by the storing wave, then ONE lane per workgroup arrives with an agent-scope atomic; every
// load of handed-off bytes is an sc1 buffer load issued after the polling lane's workgroup
// barrier.  No fences, no L2 write-back, no L1 invalidate.  The barrier is a two-level counter
// tree over groups of 16 workgroups (placement independent: nothing assumes which XCD a
// workgroup runs on); counters are monotonic in the barrier epoch, so nothing is reset inside
// the launch; the host zeroes the block before every launch.  Every spin is bounded: a
// workgroup that gives up poisons the generation word and raises `abort`, all others leave
// their spin at once, and the launch returns with state.done = -1 (the host reports it).
// ------------------------------------------------------------------------------------------
// Shape: one workgroup per CU.  Measured per iteration at 600 / 2400 / 10 000 x 5408
// (profiles/r02/fused_shapes.txt): 512 threads, ring 2: 17.2 / 26.1 / 80.0 us; 1024 threads, ring 3:
// 18.6 / 28.3 / 82.2; 1024 threads, ring 4: 18.6 / 28.9 / 81.8.  More rows in flight or more waves do not
// shorten the row pass: per workgroup it moves 43 KB per row at the rate one CU gets from the Infinity
// Cache / HBM (~1.5-1.9 us per row, i.e. 23-29 GB/s per CU, 6-7.4 TB/s over the chip), whatever the ring.
#ifndef FUSED_THREADS
#define FUSED_THREADS 512
#endif
#ifndef FUSED_NBUF
#define FUSED_NBUF 2                       // row ring: NBUF - 1 rows in flight per workgroup
#endif
#define FUSED_SPIN_LIMIT (1u << 22)        // poll rounds (>= ~1 us each) before a workgroup gives up: seconds
#define FUSED_MAX_M 4                      // column pairs per slice <= 16 * FUSED_MAX_M
#define FUSED_MAX_NCH (FUSED_THREADS == 1024 ? 3 : 6)   // column chunks per thread whose row loop compiles without scratch (H <= 6144)

#ifndef FUSED_BARRIER
#define FUSED_BARRIER 0                    // 0: two-level counter tree; 1: one flat counter; 2: flag word per workgroup
#endif
#define FUSED_GROUP 16                     // workgroups per first-level counter
#define FUSED_MAX_GROUPS (MXM_MAX_WG / FUSED_GROUP)

struct fused_sync {                        // every polled word on a 128-byte line of its own
    unsigned grp[FUSED_MAX_GROUPS][32];
    unsigned top[32];
    unsigned gen[32];
    unsigned abort_[32];
    unsigned flag[MXM_MAX_WG];             // FUSED_BARRIER == 2 only
    unsigned long long stamps[8];          // -DFUSED_STAMPS diagnostic build only: time per phase, workgroup 0
};

// Diagnostic build (-DFUSED_STAMPS, never the shipped library): thread 0 of workgroup 0 adds up the
// constant-rate clock (s_memrealtime, 100 MHz) over each phase of every iteration and leaves the sums
// in sync->stamps: [0] row pass incl. partial stores, [1] barrier 1, [2] slice reduce, [3] barrier 2,
// [4] normalise + test, [5] iterations.  The stamps serialise what the real kernel overlaps: read the
// shares, not the total (tools/time_small_runs.py --stamps).
#ifdef FUSED_STAMPS
#define FUSED_STAMP(i)                                                             \
    do {                                                                           \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                 \
            const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();      \
            stamp_acc[i] += now_ - stamp_last;                                     \
            stamp_last = now_;                                                     \
        }                                                                          \
    } while (0)
#else
#define FUSED_STAMP(i) do { } while (0)
#endif

// Grid barrier.  Contract (MI355X_MICROARCH.md, "Valid forms", row 1): all waves of the workgroup have
// drained their write-through payload stores (s_waitcnt vmcnt(0)) before calling; the other waves load
// handed-off bytes only after the __syncthreads() the polling lane joins at the end.  Returns false if
// the grid has given up (every workgroup then leaves the kernel).
// Measured per EM iteration at 600 x 5408 (two barriers each; profiles/r02/small_runs*.txt):
//   counter tree 17.0 us -- one agent-scope atomic per workgroup on its group's counter, the group's last
//     arriver adds to the top counter, the last of those publishes the epoch in `gen`, everybody polls
//     that ONE word;
//   a flag word per workgroup polled by a wave of every workgroup 22.6 us -- 256 pollers sweeping 1 KB
//     of write-through lines each cost more than the three dependent round trips of the tree.
// Counters and the generation word only grow with the epoch, so nothing is reset inside a launch.
__device__ __forceinline__ bool fused_grid_barrier(fused_sync *s, unsigned epoch, int nwg, int *lds_flag) {
    __syncthreads();
#if FUSED_BARRIER == 2
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane == 0) __hip_atomic_store(&s->flag[blockIdx.x], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
        unsigned spins = 0;
        for (;;) {
            bool here = true;
            for (int i = lane; i < nwg; i += 64)
                here = here && (__hip_atomic_load(&s->flag[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch);
            const unsigned gave_up = __hip_atomic_load(&s->abort_[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (gave_up != 0u) { ok = false; break; }                        // wave-uniform (one address)
            if (__builtin_amdgcn_ballot_w64(!here) == 0ull) break;           // every flag has arrived
            if (++spins > FUSED_SPIN_LIMIT) {
                if (lane == 0) __hip_atomic_store(&s->abort_[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) *lds_flag = ok ? 1 : 0;
    }
#else
    if (threadIdx.x == 0) {
#if FUSED_BARRIER == 1
        const unsigned old = __hip_atomic_fetch_add(&s->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == (unsigned)nwg * epoch)
            __hip_atomic_store(&s->gen[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        const int g = blockIdx.x / FUSED_GROUP;
        const int first = g * FUSED_GROUP;
        const int ngroups = (nwg + FUSED_GROUP - 1) / FUSED_GROUP;
        const unsigned gsize = (unsigned)((nwg - first) < FUSED_GROUP ? (nwg - first) : FUSED_GROUP);
        const unsigned old = __hip_atomic_fetch_add(&s->grp[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == gsize * epoch) {
            const unsigned o2 = __hip_atomic_fetch_add(&s->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (o2 + 1u == (unsigned)ngroups * epoch)
                __hip_atomic_store(&s->gen[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#endif
        unsigned spins = 0;
        for (;;) {
            const unsigned seen = __hip_atomic_load(&s->gen[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen >= epoch) break;                      // a poisoned word (0xffffffff) passes too
            if (++spins > FUSED_SPIN_LIMIT) {
                __hip_atomic_store(&s->abort_[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&s->gen[0], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        *lds_flag = (__hip_atomic_load(&s->abort_[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) ? 1 : 0;
    }
#endif
    __syncthreads();
    return *lds_flag != 0;
}

typedef unsigned int fu4 __attribute__((ext_vector_type(4)));
#define FUSED_SC1 16                       // aux bit of raw buffer loads / stores: sc1 (device scope, write-through)

template <int NCH, int NBUF>
__global__ __launch_bounds__(FUSED_THREADS, FUSED_THREADS / 256) void em_fused_loop_kernel(
    const double *__restrict__ P, int64_t ldp, const double *__restrict__ w, int64_t R, int H, int B,
    double *ln_cur, double *ln_new, double *props_cur, mxm_em_state *state, double tol, int max_iter,
    int chunk, double *partial, int64_t ldpart, double *tbuf, fused_sync *sync) {
    constexpr int THREADS = FUSED_THREADS, NW = THREADS / 64;
    __shared__ double red[2][NW];                          // row pass: per-wave partial dot products
    __shared__ d2 cred[NW][FUSED_MAX_M][16];               // column reduce: per-wave slice sums
    __shared__ double bred[2][NW];                         // phase C block sums
    __shared__ int ok_flag;
    __shared__ d2 lds_lc[NCH][THREADS];                    // log proportions (the loop's state), [k][thread]
    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int ncol2 = (H + 1) >> 1;
    const int nwg = (int)gridDim.x;
    const int cp2 = (ncol2 + nwg - 1) / nwg;               // column pairs per workgroup slice (<= 16 * FUSED_MAX_M)
    const int64_t nq = (R > (int64_t)blockIdx.x) ? (R - blockIdx.x + nwg - 1) / nwg : 0;   // rows dealt to this workgroup

    // ---- addressing of the row loads (as em_iter_wide_kernel) -------------------------------
    const int row_bytes = (int)(ldp * 8);
    const int voff = t * 16;
    int last_c2 = t + (NCH - 1) * THREADS;
    const bool last_own = last_c2 < ncol2;
    if (!last_own) last_c2 = ncol2 - 1;
    const int voff_last = last_c2 * 16;
    auto load_row = [&](d2(&xr)[NCH], int64_t q) {
        const int64_t r = (int64_t)blockIdx.x + (q < nq ? q : nq - 1) * (int64_t)nwg;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(P + r * ldp), 0, row_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < NCH - 1; ++k)
            xr[k] = __builtin_bit_cast(d2, (fu4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, k * THREADS * 16, 0));
        xr[NCH - 1] = __builtin_bit_cast(d2, (fu4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_last, 0, 0));
    };
    // hand-off buffers: descriptors over the whole region, per-lane byte offsets
    const auto part_rsrc = __builtin_amdgcn_make_buffer_rsrc(partial, 0, (int)((int64_t)nwg * ldpart * 8), 0x00020000);
    const auto t_rsrc = __builtin_amdgcn_make_buffer_rsrc(tbuf, 0, (int)(2 * ldpart * 8), 0x00020000);
    const int my_part = (int)((int64_t)blockIdx.x * ldpart * 8);

    unsigned epoch = 0;
#ifdef FUSED_STAMPS
    unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_amdgcn_s_memrealtime();
#endif
    for (int b = 0; b < B; ++b) {
        mxm_em_state *st = state + b;
        if (st->done != 0) continue;                       // written before the launch: plain load is fine
        double *lc_g = ln_cur + (int64_t)b * H, *ln_g = ln_new + (int64_t)b * H, *pc_g = props_cur + (int64_t)b * H;
        int iters = st->iters;
        // state of the loop, replicated in every workgroup: thread t holds column pairs {t + THREADS k};
        // the linear proportions stay in VGPRs (the row pass multiplies with them), the log proportions
        // -- touched once per iteration -- wait in LDS (own slot per thread: no barrier needed)
        d2 p[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = 2 * (t + k * THREADS);
            d2 l;
            l.x = (c < H) ? lc_g[c] : -INFINITY;
            l.y = (c + 1 < H) ? lc_g[c + 1] : -INFINITY;
            lds_lc[k][t] = l;
            // a resumed restart continues with the very proportions it stopped with
            p[k].x = (c < H) ? (iters > 0 ? pc_g[c] : exp(l.x)) : 0.0;
            p[k].y = (c + 1 < H) ? (iters > 0 ? pc_g[c + 1] : exp(l.y)) : 0.0;
        }
        int done = 0;
        double l1 = 0.0;
        for (int it = 0; it < chunk && done == 0; ++it) {
#ifdef FUSED_STAMPS
            if (blockIdx.x == 0 && threadIdx.x == 0) { stamp_last = __builtin_amdgcn_s_memrealtime(); stamp_acc[5] += 1; }
#endif
            // ================= phase A: row pass =================
            d2 acc[NCH];
#pragma unroll
            for (int k = 0; k < NCH; ++k) acc[k] = d2{0.0, 0.0};
            if (nq > 0) {
                d2 x[NBUF][NCH];                            // row ring: NBUF - 1 rows in flight
                int buf = 0;
                auto process = [&](d2(&xr)[NCH], int64_t q) {
                    // the weight's scalar load is issued before the dot product, not after the barrier
                    const int64_t r = (int64_t)blockIdx.x + (q < nq ? q : nq - 1) * (int64_t)nwg;
                    double wr = (q < nq) ? (w != nullptr ? w[r] : 1.0) : 0.0;
                    asm volatile("" : "+s"(wr));
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < NCH; ++k) {
                        s = fma(xr[k].x, p[k].x, s);
                        s = fma(xr[k].y, p[k].y, s);
                    }
                    s = wave_sum_lane63(s);
                    if (lane == 63) red[buf][wv] = s;
                    __syncthreads();
                    double cs[1];
                    group_ratio_to_sgpr<NW, 1>(&red[buf][0], lane, wr, cs);
                    const double c = cs[0];
#pragma unroll
                    for (int k = 0; k < NCH; ++k) {
                        acc[k].x = fma(c, xr[k].x, acc[k].x);
                        acc[k].y = fma(c, xr[k].y, acc[k].y);
                    }
                    buf ^= 1;
                };
#pragma unroll
                for (int j = 0; j < NBUF - 1; ++j) load_row(x[j], j);
                for (int64_t q = 0; q < nq; q += NBUF) {
#pragma unroll
                    for (int j = 0; j < NBUF; ++j) {
                        load_row(x[(j + NBUF - 1) % NBUF], q + j + NBUF - 1);
                        if (q + j < nq) process(x[j], q + j);             // workgroup uniform
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int c2 = t + k * THREADS;
                if (c2 < ncol2)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu4, acc[k]), part_rsrc, my_part + c2 * 16, 0,
                                                           FUSED_SC1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FUSED_STAMP(0);
            if (!fused_grid_barrier(sync, ++epoch, nwg, &ok_flag)) { done = -1; break; }
            FUSED_STAMP(1);

            // ================= phase B: this workgroup's slice of the column sums =================
            {
                constexpr int G = THREADS / 16;             // sub-groups of 16 lanes that walk the partial rows
                const int l16 = t & 15, gsub = t >> 4;
#pragma unroll 1
                for (int m = 0; m < FUSED_MAX_M; ++m) {
                    if (m * 16 < cp2) {
                        const int pi = l16 + 16 * m;
                        int c2 = (int)blockIdx.x * cp2 + pi;
                        const bool valid = (pi < cp2) && (c2 < ncol2);
                        if (!valid) c2 = 0;
                        d2 s = d2{0.0, 0.0};
                        int g = gsub;
                        auto load_part = [&](int gg) {
                            return __builtin_bit_cast(d2, (fu4)__builtin_amdgcn_raw_buffer_load_b128(
                                                              part_rsrc, (int)((int64_t)gg * ldpart * 8) + c2 * 16, 0, FUSED_SC1));
                        };
                        if constexpr (G <= 32) {
                            for (; g + 7 * G < nwg; g += 8 * G) {   // eight loads in flight per lane
                                d2 v[8];
#pragma unroll
                                for (int u = 0; u < 8; ++u) v[u] = load_part(g + G * u);
#pragma unroll
                                for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; }
                            }
                        }
                        for (; g + 3 * G < nwg; g += 4 * G) {   // four (all of them at 256 workgroups x 1024 threads)
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
                        // the wave's four sub-groups (lanes l, l+16, l+32, l+48), fixed tree
                        s.x += __shfl_xor(s.x, 16, 64); s.y += __shfl_xor(s.y, 16, 64);
                        s.x += __shfl_xor(s.x, 32, 64); s.y += __shfl_xor(s.y, 32, 64);
                        if (lane < 16) cred[wv][m][lane] = s;
                    }
                }
                __syncthreads();
                if (t < 16 * FUSED_MAX_M) {
                    const int m = t >> 4, l = t & 15;
                    const int pi = l + 16 * m;
                    const int c2 = (int)blockIdx.x * cp2 + pi;
                    if (pi < cp2 && c2 < ncol2) {
                        d2 tot = cred[0][m][l];
#pragma unroll
                        for (int q = 1; q < NW; ++q) { tot.x += cred[q][m][l].x; tot.y += cred[q][m][l].y; }
                        const d2 lt = d2{log(tot.x), log(tot.y)};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu4, tot), t_rsrc, c2 * 16, 0, FUSED_SC1);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu4, lt), t_rsrc, (int)(ldpart * 8) + c2 * 16, 0,
                                                               FUSED_SC1);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            FUSED_STAMP(2);
            if (!fused_grid_barrier(sync, ++epoch, nwg, &ok_flag)) { done = -1; break; }
            FUSED_STAMP(3);

            // ================= phase C: normalise, convergence test (every workgroup alike) =================
            d2 T[NCH], LT[NCH];
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int c2 = (k < NCH - 1) ? (t + k * THREADS) : last_c2;
                T[k] = __builtin_bit_cast(d2, (fu4)__builtin_amdgcn_raw_buffer_load_b128(t_rsrc, c2 * 16, 0, FUSED_SC1));
                LT[k] = __builtin_bit_cast(d2, (fu4)__builtin_amdgcn_raw_buffer_load_b128(t_rsrc, (int)(ldpart * 8) + c2 * 16, 0,
                                                                                          FUSED_SC1));
            }
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                if (k < NCH - 1 || last_own) {
                    s = fma(p[k].x, T[k].x, s);
                    s = fma(p[k].y, T[k].y, s);
                }
            }
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
            for (int k = 0; k < NCH; ++k) {
                if (k < NCH - 1 || last_own)                                                   // em.py:53-54
                    d += fabs(p[k].x * T[k].x * rtot - p[k].x) + fabs(p[k].y * T[k].y * rtot - p[k].y);
            }
            d = wave_sum_lane63(d);
            if (lane == 63) bred[1][wv] = d;
            __syncthreads();
            l1 = bred[1][0];
#pragma unroll
            for (int q = 1; q < NW; ++q) l1 += bred[1][q];
            ++iters;
            const bool conv = l1 < tol;
            done = conv ? 1 : (iters >= max_iter ? 2 : 0);
            if (done == 0) {                               // em.py:140: props <- new_props
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const d2 l = lds_lc[k][t];
                    lds_lc[k][t] = d2{l.x + LT[k].x - ltot, l.y + LT[k].y - ltot};              // em.py:87-89
                    // an odd H's pad column and the clamped tail stay at p = 0
                    const int c = 2 * (t + k * THREADS);
                    p[k].x = (c < H) ? p[k].x * T[k].x * rtot : 0.0;
                    p[k].y = (c + 1 < H) ? p[k].y * T[k].y * rtot : 0.0;
                }
            } else if (blockIdx.x == 0) {                  // stopped: ln_new = log theta_{k+1}; lc / p stay at theta_k
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const int c = 2 * (t + k * THREADS);
                    const d2 l = lds_lc[k][t];
                    if (c < H) ln_g[c] = l.x + LT[k].x - ltot;
                    if (c + 1 < H) ln_g[c + 1] = l.y + LT[k].y - ltot;
                }
            }
            FUSED_STAMP(4);
        }
        // ---- results of this restart (workgroup 0): ln_cur = log theta_k, ln_new = log theta_{k+1} ----
        if (blockIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int c = 2 * (t + k * THREADS);
                const d2 l = lds_lc[k][t];
                if (c < H) { lc_g[c] = l.x; pc_g[c] = p[k].x; }
                if (c + 1 < H) { lc_g[c + 1] = l.y; pc_g[c + 1] = p[k].y; }
            }
            if (t == 0) {
                st->iters = iters;
                st->l1 = l1;
                st->done = done;
            }
        }
        if (done < 0) return;                              // the grid gave up: every workgroup leaves
    }
#ifdef FUSED_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int i = 0; i < 6; ++i) sync->stamps[i] = stamp_acc[i];
#endif
}

#endif  // MIXEMT_FUSED_KERNELS_HPP
