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

// votes[h] = sum of w[r] over the rows with best[r] == h (assemble.py:116-119) without float atomics: workgroup g takes
// the contiguous rows [g * per, (g + 1) * per), ONE thread adds them in row order into the workgroup's row of
// `vote_part` (zeroed here first); colreduce_kernel then sums the workgroups in fixed order -- fractional weights
// reproduce bit for bit.
__global__ __launch_bounds__(256) void votes_from_best_kernel(const int32_t *__restrict__ best, const double *__restrict__ w,
                                                              int64_t R, int H, double *__restrict__ vote_part,
                                                              int64_t ldpart) {
    double *mine = vote_part + (int64_t)blockIdx.x * ldpart;
    for (int h = threadIdx.x; h < H; h += 256) mine[h] = 0.0;
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int64_t per = (R + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = (lo + per < R) ? lo + per : R;
    for (int64_t r = lo; r < hi; ++r) {
        const int b = best[r];
        if ((unsigned)b >= (unsigned)H) continue;            // -1: a row without a record that nobody supplied densely
        volatile double *slot = mine + b;                    // the same thread wrote the zero / the last sum
        *slot = *slot + (w != nullptr ? w[r] : 1.0);
    }
}

#endif  // MIXEMT_RECORDS_KERNELS_HPP
