// records_kernels.hpp -- part of libmixemt_hip.so (gfx950); included by mixemt_hip.hip only.
// The contributor vote (assemble.py:103-123, stats.py:34-45) straight from a matrix in record form -- no dense
// matrix, no posterior matrix.
#ifndef MIXEMT_RECORDS_KERNELS_HPP
#define MIXEMT_RECORDS_KERNELS_HPP

// ------------------------------------------------------------------------------------------
// best[r] = first index of max_h res_read_mix[r][h], where res_read_mix is what run_em returns (em.py:145-161):
// the per-run log posteriors  x_k[r][h] = ln_props[k][h] + M[r][h] - lse_k[r]  folded over the n_runs runs with
// logaddexp IN RUN ORDER (em.py:156; the final -log n shifts every column alike and drops out of the argmax).
//   * one run: the row's normaliser drops out too -- best = argmax_h (ln_props[h] + M[r][h]);
//   * several runs: each run's row normaliser weighs that run's columns, so it is needed: lse_k[r] from the
//     records in the loop's own variables (coded_row_lse), from the log values for the dense leftover rows.
// CODED = true: the rows with a record (ndist[r] > 0), read from their codes + log table.
// CODED = false: R_rest dense rows M_rest[i][:] whose row index is rest_rows[i] (the rows without a record).
// One workgroup of 256 per row; numpy.argmax semantics (first maximum; a NaN wins, the first one).
// ------------------------------------------------------------------------------------------
#define RECK_MAX_RUNS 4096            // runs folded per launch (their lse sit in dynamic LDS: 8 bytes per run)

template <bool CODED>
__global__ __launch_bounds__(256) void posterior_argmax_kernel(
    const uint8_t *__restrict__ rec, const int64_t *__restrict__ rec_off, const int32_t *__restrict__ ndist, int ldc,
    const double *__restrict__ M_rest, int64_t ldm_rest, const int64_t *__restrict__ rest_rows, int64_t R, int H,
    int n_runs, const double *__restrict__ ln_props, const double *__restrict__ props, const double *__restrict__ rowmax,
    int32_t *__restrict__ best) {
    __shared__ double s_p[CODED ? ENC_MAX_WIDE : 1], s_m[CODED ? ENC_MAX_WIDE : 1];
    __shared__ double s_red[4];
    extern __shared__ double s_lse[];                        // [n_runs] (dynamic: any number of runs)
    __shared__ double s_val[4];
    __shared__ int s_idx[4], s_nan[4];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (int64_t i = blockIdx.x; i < R; i += gridDim.x) {
        int64_t r = i;
        const uint8_t *codes = nullptr;
        const double *row = nullptr;
        bool wide = false;
        if constexpr (CODED) {
            const int nd = ndist[r];
            if (nd <= 0) {                                   // uniform: no record -- the dense pass fills it in,
                if (t == 0) best[r] = -1;                    // and a row the caller did not hand over stays "no vote"
                continue;
            }
            wide = nd > ENC_MAX_CODES;
            codes = rec + rec_off[r];
            const double *ptab = reinterpret_cast<const double *>(codes + rec_code_bytes(nd, ldc));
            for (int j = t; j < nd; j += 256) {
                s_p[j] = ptab[j];
                s_m[j] = ptab[nd + j];
            }
            __syncthreads();
        } else {
            r = rest_rows[i];
            row = M_rest + i * ldm_rest;
        }
        auto logv = [&](int h) -> double {
            if constexpr (CODED) return s_m[rec_code_at(codes, h, wide)];
            else return row[h];
        };
        if (n_runs > 1) {
            for (int k = 0; k < n_runs; ++k) {               // uniform loop; the helpers hold barriers
                const double *lnp = ln_props + (int64_t)k * H;
                double lse;
                if constexpr (CODED) {
                    lse = coded_row_lse(codes, wide, s_p, s_m, props + (int64_t)k * H, lnp, rowmax[r], H, s_red);
                } else {
                    double m = -INFINITY;
                    for (int h = t; h < H; h += 256) m = fmax(m, lnp[h] + row[h]);
                    m = wave_max(m);
                    __syncthreads();
                    if (lane == 0) s_red[wv] = m;
                    __syncthreads();
                    m = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
                    const double shift = (m > -INFINITY && m < INFINITY) ? m : 0.0;
                    double sacc = 0.0;
                    for (int h = t; h < H; h += 256) sacc += exp(lnp[h] + row[h] - shift);
                    sacc = wave_sum(sacc);
                    __syncthreads();
                    if (lane == 0) s_red[wv] = sacc;
                    __syncthreads();
                    sacc = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
                    lse = log(sacc) + m;                         // m, not shift: -inf rows stay -inf (estep_kernels.hpp)
                }
                if (t == 0) s_lse[k] = lse;
            }
            __syncthreads();
        }
        int cn = 0, ci = 0x7fffffff;
        double cv = -INFINITY;
        for (int h = t; h < H; h += 256) {                   // increasing h per thread: the first maximum is kept
            const double m = logv(h);
            double v = ln_props[h] + m;
            if (n_runs > 1) {
                v -= s_lse[0];
                for (int k = 1; k < n_runs; ++k) v = logaddexp_f64(v, (ln_props[(int64_t)k * H + h] + m) - s_lse[k]);
            }
            const int vn = (v != v) ? 1 : 0;
            if (cand_better(vn, v, h, cn, cv, ci)) { cn = vn; cv = v; ci = h; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(cv, off, 64);
            const int oi = __shfl_xor(ci, off, 64);
            const int on = __shfl_xor(cn, off, 64);
            if (cand_better(on, ov, oi, cn, cv, ci)) { cn = on; cv = ov; ci = oi; }
        }
        if (lane == 0) { s_val[wv] = cv; s_idx[wv] = ci; s_nan[wv] = cn; }
        __syncthreads();
        if (t == 0) {
            for (int q = 1; q < 4; ++q)
                if (cand_better(s_nan[q], s_val[q], s_idx[q], cn, cv, ci)) { cn = s_nan[q]; cv = s_val[q]; ci = s_idx[q]; }
            best[r] = (ci >= H) ? 0 : ci;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// The single-run case of the above, laid out like the EM iteration (round 4): best[r] = first index of
// max_h (ln_props[h] + M[r][h]).  Thread t owns the columns 4 (t + 256 k) + e for ALL rows and keeps their log
// proportions in registers; a row arrives as 4 code bytes (8 for a wide record) per thread and chunk plus its log table
// through LDS; the next row's loads are in flight while this one is reduced.  The one-workgroup-per-row kernel above
// read single code bytes and every ln_props[h] from memory per row: 14.4 ms at 10^6 rows (profiles/r04), this one 2.
// Rows without a record get best[r] = -1 (the dense pass fills them in).
// ------------------------------------------------------------------------------------------
// `rows` (nullable): the kernel then takes the rows rows[0 .. R) instead of 0 .. R-1.
// ONLY_WIDE: behind records_argmax_narrow_kernel, which leaves the wide rows (more than 256 values) to this kernel: every
// block of 256 rows is looked at and only its wide rows are taken (compacted in LDS; in any order -- rows are
// independent); the others are neither read nor written.  It finds the wide rows from ndist itself: the vote never
// depends on a caller's list.
template <int NCH, bool ONLY_WIDE = false>
__global__ __launch_bounds__(256) void records_argmax_kernel(const uint8_t *__restrict__ rec, const int64_t *__restrict__ rec_off,
                                                             const int32_t *__restrict__ ndist, int ldc, int64_t R, int H,
                                                             const double *__restrict__ ln_props, int32_t *__restrict__ best,
                                                             const int64_t *__restrict__ rows) {
    constexpr int THREADS = 256, TPT = ENC_MAX_WIDE / THREADS;
    __shared__ double s_m[2][ENC_MAX_WIDE];
    __shared__ double s_val[2][4];
    __shared__ int s_idx[2][4], s_nan[2][4];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    double lp[NCH][4];
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * (t + k * THREADS) + e;
            lp[k][e] = (c < H) ? ln_props[c] : -INFINITY;     // columns past the row can never win (log values are < +inf)
        }
    const int nword = ldc >> 2;
    u2v cw[NCH];
    double tn[TPT];
    // record offsets and table sizes of 256 of this workgroup's rows at a time, gathered by one thread per row (as
    // scalar loads inside the row loop they were two dependent memory round trips per row: 14 us a row)
    __shared__ long long s_off[THREADS];
    __shared__ long long s_row[THREADS];
    __shared__ int s_nd[THREADS];
    const int64_t grid = gridDim.x;
    const int64_t nq = (R > (int64_t)blockIdx.x) ? (R - blockIdx.x + grid - 1) / grid : 0;
    auto fetch = [&](int i) {                                // codes and this thread's entries of the LOG table of step i
        const int nd = __builtin_amdgcn_readfirstlane(s_nd[i]);
        if (nd <= 0) return;                                 // uniform
        const long long off = s_off[i];
        const uint8_t *base = rec + (((long long)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) |
                                     (unsigned int)__builtin_amdgcn_readfirstlane((int)off));
        const bool wide = nd > ENC_MAX_CODES;
        const int cbytes = wide ? 2 * ldc : ldc;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, cbytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            int wd = t + k * THREADS;
            if (wd > nword - 1) wd = nword - 1;              // words past the row: clamped (their columns are masked below)
            if (wide) {
                cw[k] = __builtin_amdgcn_raw_buffer_load_b64(rs, wd * 8, 0, 0);
            } else {
                cw[k].x = __builtin_amdgcn_raw_buffer_load_b32(rs, wd * 4, 0, 0);
                cw[k].y = 0;
            }
        }
        const auto rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base + cbytes + 8 * (int64_t)nd), 0, nd * 8, 0x00020000);
#pragma unroll
        for (int j = 0; j < TPT; ++j) {
            const u2v x = __builtin_amdgcn_raw_buffer_load_b64(rt, (t + j * THREADS) * 8, 0, 0);
            tn[j] = __hiloint2double((int)x.y, (int)x.x);
        }
    };
    int buf = 0;
    __shared__ int s_cnt;
    for (int64_t q0 = 0; q0 < nq; q0 += THREADS) {
        __syncthreads();                                     // the block before has been read
        int n_here = (int)((nq - q0) < THREADS ? (nq - q0) : THREADS);
        {
            int64_t r = (int64_t)blockIdx.x + (q0 + t) * grid;
            if (rows != nullptr && q0 + t < nq) r = rows[r];
            const int nd = (q0 + t < nq) ? ndist[r] : 0;
            const long long off = (q0 + t < nq) ? rec_off[r] : 0;
            if constexpr (ONLY_WIDE) {
                if (t == 0) s_cnt = 0;
                __syncthreads();
                if (nd > ENC_MAX_CODES) {
                    const int k = atomicAdd(&s_cnt, 1);
                    s_nd[k] = nd;
                    s_off[k] = off;
                    s_row[k] = r;
                }
            } else {
                s_nd[t] = nd;
                s_off[t] = off;
                s_row[t] = r;
            }
        }
        __syncthreads();
        if constexpr (ONLY_WIDE) n_here = s_cnt;
        if (n_here == 0) continue;                           // uniform
        fetch(0);
        for (int i = 0; i < n_here; ++i, buf ^= 1) {
            const int64_t r = s_row[i];
            const int nd = __builtin_amdgcn_readfirstlane(s_nd[i]);
            if (nd <= 0) {                                   // uniform
                if (t == 0) best[r] = -1;
                if (i + 1 < n_here) fetch(i + 1);
                continue;
            }
            const bool wide = nd > ENC_MAX_CODES;
            u2v cc[NCH];
#pragma unroll
            for (int k = 0; k < NCH; ++k) cc[k] = cw[k];
#pragma unroll
            for (int j = 0; j < TPT; ++j) s_m[buf][t + j * THREADS] = tn[j];
            if (i + 1 < n_here) fetch(i + 1);                // in flight under this row's reduction
            __syncthreads();                                 // the table is in LDS (double buffered: the row before is done with the other)
            int cn = 0, ci = 0x7fffffff;
            double cv = -INFINITY;
            auto cell = [&](int k, int e, int code) {
                const int c = 4 * (t + k * THREADS) + e;     // increasing per thread: the first maximum is kept
                const double v = lp[k][e] + s_m[buf][code];
                // a later column only replaces a strictly smaller non-NaN maximum (a NaN wins, the first one)
                const bool take = (cn == 0) & !(v <= cv);
                cn = take ? ((v != v) ? 1 : 0) : cn;
                cv = take ? v : cv;
                ci = take ? c : ci;
            };
            if (wide) {                                      // uniform
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    cell(k, 0, (int)(cc[k].x & 0xffffu));
                    cell(k, 1, (int)(cc[k].x >> 16));
                    cell(k, 2, (int)(cc[k].y & 0xffffu));
                    cell(k, 3, (int)(cc[k].y >> 16));
                }
            } else {
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    cell(k, 0, (int)(cc[k].x & 0xffu));
                    cell(k, 1, (int)((cc[k].x >> 8) & 0xffu));
                    cell(k, 2, (int)((cc[k].x >> 16) & 0xffu));
                    cell(k, 3, (int)(cc[k].x >> 24));
                }
            }
            // the wave's candidate by DPP ladders (no LDS permutes): a NaN wins, else the maximum; among equals the
            // smallest column.  No NaN lane -> every cv is a number and the max ladder is exact.
            {
                const bool any_nan = __builtin_amdgcn_ballot_w64(cn != 0) != 0ull;          // uniform
                double m = 0.0;
                bool cand;
                if (any_nan) {
                    cand = cn != 0;
                } else {
                    m = readlane_f64(wave_max_lane63(cv), 63);
                    cand = cv == m;
                }
                const int wi = __builtin_amdgcn_readlane(wave_min_lane63_i32(cand ? ci : 0x7fffffff), 63);
                if (lane == 0) {
                    s_val[buf][wv] = any_nan ? __builtin_nan("") : m;
                    s_idx[buf][wv] = wi;
                    s_nan[buf][wv] = any_nan ? 1 : 0;
                }
                cn = any_nan ? 1 : 0;                        // thread 0 continues from its wave's candidate
                cv = any_nan ? __builtin_nan("") : m;
                ci = wi;
            }
            __syncthreads();
            if (t == 0) {
                for (int q = 1; q < 4; ++q)
                    if (cand_better(s_nan[buf][q], s_val[buf][q], s_idx[buf][q], cn, cv, ci)) { cn = s_nan[buf][q]; cv = s_val[buf][q]; ci = s_idx[buf][q]; }
                best[r] = (ci >= H) ? 0 : ci;
            }
        }
    }
}

// The byte-coded rows alone (97.9 % of a build_em_matrix matrix), THREE rows in flight: a row is 4 code bytes per thread
// and chunk plus ONE table entry per thread, so three of them cost 24 registers and the kernel fits four workgroups per
// CU.  records_argmax_kernel, with room for a wide record per row in flight, ran one row ahead at 154 VGPRs and was
// bound by the memory latency per row (4.7 us per row and workgroup: 6.1 ms at 10^6 rows).  Wide rows are left to that
// kernel (launched over their list); rows without a record get best[r] = -1.
template <int NCH>
__global__ __launch_bounds__(256, 4) void records_argmax_narrow_kernel(const uint8_t *__restrict__ rec,
                                                                       const int64_t *__restrict__ rec_off,
                                                                       const int32_t *__restrict__ ndist, int ldc, int64_t R, int H,
                                                                       const double *__restrict__ ln_props,
                                                                       int32_t *__restrict__ best) {
    constexpr int THREADS = 256;
    __shared__ double s_m[3][ENC_MAX_CODES];             // one table per row in flight (slot)
    __shared__ double s_val[3][4];
    __shared__ int s_idx[3][4], s_nan[3][4];
    __shared__ long long s_off[THREADS];
    __shared__ int s_nd[THREADS];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    double lp[NCH][4];
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * (t + k * THREADS) + e;
            lp[k][e] = (c < H) ? ln_props[c] : -INFINITY;     // columns past the row can never win
        }
    const int nword = ldc >> 2;
    int last = t + (NCH - 1) * THREADS;
    if (last > nword - 1) last = nword - 1;
    const int64_t grid = gridDim.x;
    const int64_t nq = (R > (int64_t)blockIdx.x) ? (R - blockIdx.x + grid - 1) / grid : 0;
    unsigned int cwa[NCH], cwb[NCH], cwc[NCH];
    double tna = 0.0, tnb = 0.0, tnc = 0.0;
    auto fetch = [&](int i, unsigned int(&cw)[NCH], double &tn) {
        int nd = __builtin_amdgcn_readfirstlane(s_nd[i]);
        if (nd > ENC_MAX_CODES) nd = 0;                      // a wide row: not ours (nothing is read)
        const long long off = s_off[i];
        const uint8_t *base = rec + (((long long)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) |
                                     (unsigned int)__builtin_amdgcn_readfirstlane((int)off));
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, nd > 0 ? ldc : 0, 0x00020000);
#pragma unroll
        for (int k = 0; k < NCH - 1; ++k) cw[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, t * 4, k * THREADS * 4, 0);
        cw[NCH - 1] = __builtin_amdgcn_raw_buffer_load_b32(rs, last * 4, 0, 0);
        const auto rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base + ldc + 8 * (int64_t)(nd > 0 ? nd : 0)), 0,
                                                          (nd > 0 ? nd : 0) * 8, 0x00020000);
        const u2v x = __builtin_amdgcn_raw_buffer_load_b64(rt, t * 8, 0, 0);
        tn = __hiloint2double((int)x.y, (int)x.x);
    };
    // A row: its table into the slot's LDS buffer (the slot is a compile-time constant: the buffer's base is the
    // lookups' immediate offset), then the thread's first maximum over its 24 cells.  The lookups of chunk k + 1 are
    // issued before chunk k is compared (the scheduling barriers keep them there): issued where they are used, every
    // one of them waited for its own LDS round trip -- 24 of them in a row, 3.2 ms at 10^6 rows.
    // NaN-free form first: code byte, add, compare, index, maximum; the column index is kept WITHOUT the thread's 4 t,
    // so that every candidate is a literal.  A NaN (none in a matrix of log-likelihood sums; numpy.argmax lets the
    // first one win) sends the wave through the general form.
    auto process = [&](auto SLOT, int64_t r, int nd, const unsigned int(&cw)[NCH], double tn) {
        constexpr int slot = decltype(SLOT)::value;
        if (nd <= 0 || nd > ENC_MAX_CODES) {                 // uniform: no record, or a wide one -- the launch over the
            if (t == 0) best[r] = -1;                        // wide rows' list, behind this one on the stream, fills those in
            return;                                          // (a descriptor without the list leaves them at -1: no vote)
        }
        s_m[slot][t] = tn;
        __syncthreads();                                     // the table is in LDS
        const char *tb = reinterpret_cast<const char *>(&s_m[slot][0]);
        auto lookup4 = [&](unsigned int word, double(&out)[4]) {
            out[0] = *reinterpret_cast<const double *>(tb + code_byte_x8<0>(word));
            out[1] = *reinterpret_cast<const double *>(tb + code_byte_x8<1>(word));
            out[2] = *reinterpret_cast<const double *>(tb + code_byte_x8<2>(word));
            out[3] = *reinterpret_cast<const double *>(tb + code_byte_x8<3>(word));
        };
        int cn = 0, ci = 0x7fffffff;
        double cv = -INFINITY;
        bool seen_nan = false;
        double cur[4], nxt[4];
        lookup4(cw[0], cur);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (k + 1 < NCH) lookup4(cw[k + 1], nxt);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double v = lp[k][e] + cur[e];
                seen_nan |= (v != v);
                ci = (v > cv) ? 4 * k * THREADS + e : ci;    // increasing per thread: the first maximum is kept
                cv = (v > cv) ? v : cv;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) cur[e] = nxt[e];
        }
        ci = (ci == 0x7fffffff) ? ci : ci + 4 * t;
        if (__builtin_amdgcn_ballot_w64(seen_nan) != 0ull) {  // wave uniform
            cn = 0;
            ci = 0x7fffffff;
            cv = -INFINITY;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                unsigned int cx = cw[k];
                asm volatile("" : "+v"(cx));                 // looked up again: the values of the loop above are not kept for this
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 4 * (t + k * THREADS) + e;
                    const double v = lp[k][e] + s_m[slot][(cx >> (8 * e)) & 0xffu];
                    const bool take = (cn == 0) & !(v <= cv);    // a NaN wins, the first one; else a strictly larger value
                    cn = take ? ((v != v) ? 1 : 0) : cn;
                    cv = take ? v : cv;
                    ci = take ? c : ci;
                }
            }
        }
        const bool any_nan = __builtin_amdgcn_ballot_w64(cn != 0) != 0ull;      // uniform
        double m = 0.0;
        bool cand;
        if (any_nan) {
            cand = cn != 0;
        } else {
            m = readlane_f64(wave_max_lane63(cv), 63);
            cand = cv == m;
        }
        const int wi = __builtin_amdgcn_readlane(wave_min_lane63_i32(cand ? ci : 0x7fffffff), 63);
        if (lane == 0) {
            s_val[slot][wv] = any_nan ? __builtin_nan("") : m;
            s_idx[slot][wv] = wi;
            s_nan[slot][wv] = any_nan ? 1 : 0;
        }
        __syncthreads();
        if (t == 0) {
            int bn = s_nan[slot][0], bi = s_idx[slot][0];
            double bv = s_val[slot][0];
            for (int q = 1; q < 4; ++q)
                if (cand_better(s_nan[slot][q], s_val[slot][q], s_idx[slot][q], bn, bv, bi)) { bn = s_nan[slot][q]; bv = s_val[slot][q]; bi = s_idx[slot][q]; }
            best[r] = (bi >= H) ? 0 : bi;
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    for (int64_t q0 = 0; q0 < nq; q0 += THREADS) {
        __syncthreads();                                     // the block before has been read
        {
            const int64_t r = (int64_t)blockIdx.x + (q0 + t) * grid;
            s_nd[t] = (q0 + t < nq) ? ndist[r] : 0;
            s_off[t] = (q0 + t < nq) ? rec_off[r] : 0;
        }
        __syncthreads();
        const int n_here = (int)((nq - q0) < THREADS ? (nq - q0) : THREADS);
        auto row_of = [&](int i) { return (int64_t)blockIdx.x + (q0 + i) * grid; };
        auto nd_of = [&](int i) { return __builtin_amdgcn_readfirstlane(s_nd[i]); };
        fetch(0, cwa, tna);
        if (n_here > 1) fetch(1, cwb, tnb);
        for (int i = 0; i < n_here; i += 3) {                // every branch below is workgroup uniform
            if (i + 2 < n_here) fetch(i + 2, cwc, tnc);
            process(S0{}, row_of(i), nd_of(i), cwa, tna);
            if (i + 1 >= n_here) break;
            if (i + 3 < n_here) fetch(i + 3, cwa, tna);
            process(S1{}, row_of(i + 1), nd_of(i + 1), cwb, tnb);
            if (i + 2 >= n_here) break;
            if (i + 4 < n_here) fetch(i + 4, cwb, tnb);
            process(S2{}, row_of(i + 2), nd_of(i + 2), cwc, tnc);
        }
    }
}

// votes[h] = sum of w[r] over the rows with best[r] == h (assemble.py:116-119) without float atomics: workgroup g takes
// the contiguous rows [g * per, (g + 1) * per), ONE thread adds them in row order into the workgroup's row of
// `vote_part` (zeroed here first); colreduce_kernel then sums the workgroups in fixed order -- fractional weights
// reproduce bit for bit.
__global__ __launch_bounds__(256) void votes_from_best_kernel(const int32_t *__restrict__ best, const double *__restrict__ w,
                                                              int64_t R, int H, double *__restrict__ vote_part,
                                                              int64_t ldpart) {
    // round 4: the one thread that adds in row order adds into LDS (a 30 ns round trip; the workgroup's row of
    // `vote_part` in global memory cost ~0.7 us per row: 1.4 ms at 10^6 rows, 14 ms at 10^7), then all threads write
    // the row out.  Same order, same bits.
    // Round 5: INTEGER weights (the reference's only case: fragments per signature, preprocess.py:220) need no order --
    // sums of whole numbers below 2^53 are exact however they are formed -- so all 256 threads add (LDS atomics) and
    // watch for a weight that is not a whole number; only then is the range redone by one thread in row order.  The
    // serial walk was 0.26 ms at 10^6 rows but linear in R (2.6 ms at 10^7).
    extern __shared__ double s_votes[];
    __shared__ int s_fractional;
    const int64_t per = (R + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = (lo + per < R) ? lo + per : R;
    for (int h = threadIdx.x; h < H; h += 256) s_votes[h] = 0.0;
    if (threadIdx.x == 0) s_fractional = 0;
    __syncthreads();
    bool frac = false;
    for (int64_t r = lo + threadIdx.x; r < hi; r += 256) {
        const int b = best[r];
        if ((unsigned)b >= (unsigned)H) continue;            // -1: a row without a record that nobody supplied densely
        const double wr = (w != nullptr) ? w[r] : 1.0;
        frac = frac || !(wr == floor(wr)) || !(fabs(wr) < 4503599627370496.0);     // (NaN and infinities count as fractional)
        atomicAdd(&s_votes[b], wr);
    }
    if (frac) s_fractional = 1;
    __syncthreads();
    if (s_fractional != 0) {                                 // uniform: fractional weights -> the fixed row order
        for (int h = threadIdx.x; h < H; h += 256) s_votes[h] = 0.0;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int64_t r = lo; r < hi; ++r) {
                const int b = best[r];
                if ((unsigned)b >= (unsigned)H) continue;
                s_votes[b] += (w != nullptr ? w[r] : 1.0);
            }
        }
        __syncthreads();
    }
    double *mine = vote_part + (int64_t)blockIdx.x * ldpart;
    for (int h = threadIdx.x; h < H; h += 256) mine[h] = s_votes[h];
}

// first[h] = the smallest row index r with best[r] == h (R where nobody voted for h): the insertion order of the
// reference's vote table (assemble.py:116-119, a dict filled row by row) without bringing best[] to the host.
// An integer minimum is order independent, so atomics are exact here.  A workgroup takes a contiguous range of rows,
// keeps the minima of ITS range in LDS (offsets into the range: 32 bits) and sends one global atomic per haplogroup
// it saw -- a handful; a first version with one global atomic per row took 3.6 ms at 10^6 rows (three hot addresses).
#define FSEEN_MAX_H 8192
// One launch covers the haplogroups [h0, h0 + span), span <= FSEEN_MAX_H (wider matrices: one launch per block).
__global__ __launch_bounds__(256) void first_seen_kernel(const int32_t *__restrict__ best, int64_t R, int h0, int span,
                                                         unsigned long long *__restrict__ first) {
    __shared__ unsigned int s_first[FSEEN_MAX_H];
    const int64_t per = (R + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = (lo + per < R) ? lo + per : R;
    if (lo >= hi) return;
    for (int h = threadIdx.x; h < span; h += 256) s_first[h] = 0xFFFFFFFFu;
    __syncthreads();
    for (int64_t r = lo + threadIdx.x; r < hi; r += 256) {
        const int b = best[r] - h0;
        if ((unsigned)b < (unsigned)span) atomicMin(&s_first[b], (unsigned int)(r - lo));
    }
    __syncthreads();
    for (int h = threadIdx.x; h < span; h += 256) {
        const unsigned int v = s_first[h];
        if (v != 0xFFFFFFFFu) atomicMin(&first[h0 + h], (unsigned long long)(lo + v));
    }
}

__global__ __launch_bounds__(256) void fill_u64_kernel(unsigned long long *__restrict__ x, int64_t n, unsigned long long v) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] = v;
}

#endif  // MIXEMT_RECORDS_KERNELS_HPP
