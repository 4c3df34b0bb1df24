// estep_kernels.hpp -- part of libmixemt_hip.so (gfx950); included by mixemt_hip.hip only.
// The E-step with the reference's log-space semantics (em.py:80-83, fold :156): generic and register-resident forms.
#ifndef MIXEMT_ESTEP_KERNELS_HPP
#define MIXEMT_ESTEP_KERNELS_HPP

// ------------------------------------------------------------------------------------------
// K6  estep_log: the reference's E-step verbatim in log space (em.py:80-83), optional posterior
// write / logaddexp fold (em.py:156) and optional M-step sums (em.py:87-88, linear space).
// One workgroup per row (grid-stride), column sums in LDS (each thread owns its columns).
// Used for em_step() and the posterior pass when the rows are not 16-byte aligned / H is odd, and
// for anything wider than the register-resident kernels cover; narrow matrices take K6n.
// ------------------------------------------------------------------------------------------
template <bool ITER>
__global__ __launch_bounds__(ROW_THREADS) void estep_log_kernel(
    const double *__restrict__ M, int64_t ldm, const double *__restrict__ w,
    const double *__restrict__ ln_props, int64_t R, int H, double *__restrict__ out, int64_t ldo,
    int mode, double *__restrict__ partial, int64_t ldpart,
    const mxm_em_state *__restrict__ state, const int64_t *__restrict__ out_rows = nullptr) {
    // out_rows (nullable): row r's posterior goes to row out_rows[r] of `out` (a compact side matrix whose rows
    // belong at scattered places of the result: the dense leftover rows of a matrix in record form).
    // ITER = false: the reference's E-step verbatim (posterior written / folded; `partial` gets the
    //               M-step sums  sum_r w_r exp(posterior)).
    // ITER = true:  one loop iteration for narrow matrices, nothing written but `partial`, which
    //               gets the UNSCALED sums T_h = sum_r (w_r / Z_r) exp(M_rh - rowmax_r) with
    //               Z_r = sum_h exp(ln p_h + M_rh - rowmax_r)  -- same quantity as the streaming
    //               kernel's, so finalize_kernel treats both alike.
    extern __shared__ double dyn[];            // [H] ln props + [H] column sums
    __shared__ double scratch[ROW_THREADS / 64];
    if (state != nullptr && state->done != 0) return;
    const int t = threadIdx.x;
    double *lnp = dyn;
    double *acc = dyn + H;
    for (int h = t; h < H; h += ROW_THREADS) {
        lnp[h] = ln_props[h];
        if (partial != nullptr) acc[h] = 0.0;
    }
    __syncthreads();
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
        const double *src = M + r * ldm;
        const double wr = (w != nullptr) ? w[r] : 1.0;
        if constexpr (ITER) {
            double m = -INFINITY;
            for (int h = t; h < H; h += ROW_THREADS) m = fmax(m, src[h]);
            m = block_reduce<ROW_THREADS, true>(m, scratch);
            const double shift = isfinite(m) ? m : 0.0;
            double z = 0.0;
            for (int h = t; h < H; h += ROW_THREADS) z += exp(lnp[h] + (src[h] - shift));
            z = block_reduce<ROW_THREADS, false>(z, scratch);
            const double c = weight_over_norm(wr, z);
            for (int h = t; h < H; h += ROW_THREADS) acc[h] += c * exp(src[h] - shift);
        } else {
            double m = -INFINITY;
            for (int h = t; h < H; h += ROW_THREADS) m = fmax(m, lnp[h] + src[h]);
            m = block_reduce<ROW_THREADS, true>(m, scratch);
            const double shift = isfinite(m) ? m : 0.0;
            double s = 0.0;
            for (int h = t; h < H; h += ROW_THREADS) s += exp((lnp[h] + src[h]) - shift);
            s = block_reduce<ROW_THREADS, false>(s, scratch);
            const double lse = log(s) + m;          // m (not shift): -inf rows give -inf, as scipy does
            for (int h = t; h < H; h += ROW_THREADS) {
                const double v = (lnp[h] + src[h]) - lse;
                if (out != nullptr) {
                    double *o = out + (out_rows != nullptr ? out_rows[r] : r) * ldo + h;
                    *o = (mode == 1) ? logaddexp_f64(*o, v) : v;
                }
                if (partial != nullptr && wr != 0.0) acc[h] += wr * exp(v);   // scipy drops zero-weight rows, NaN or not
            }
        }
    }
    if (partial != nullptr) {
        double *dst = partial + (int64_t)blockIdx.x * ldpart;
        for (int h = t; h < H; h += ROW_THREADS) dst[h] = acc[h];
    }
}

// ------------------------------------------------------------------------------------------
// K6g estep_global: estep_log_kernel for ANY width (round 5: the reference takes whatever Phylotree build and custom
// haplogroups it is given, phylotree.py:231-250; K6 keeps two vectors of H doubles in LDS and stops at 9600 columns).
// The log proportions are read from global memory and the column sums live in the workgroup's OWN partial row there
// (thread t owns the columns t, t + 256, ...: a plain read-modify-write, no atomics, fixed order): everything is
// L2-resident, nothing depends on H but the loop counts.  The fallback of the fallback: correctness, not speed.
// ------------------------------------------------------------------------------------------
template <bool ITER>
__global__ __launch_bounds__(ROW_THREADS) void estep_global_kernel(
    const double *__restrict__ M, int64_t ldm, const double *__restrict__ w,
    const double *__restrict__ ln_props, int64_t R, int H, double *__restrict__ out, int64_t ldo,
    int mode, double *__restrict__ partial, int64_t ldpart,
    const mxm_em_state *__restrict__ state, const int64_t *__restrict__ out_rows = nullptr) {
    __shared__ double scratch[ROW_THREADS / 64];
    if (state != nullptr && state->done != 0) return;
    const int t = threadIdx.x;
    double *acc = partial != nullptr ? partial + (int64_t)blockIdx.x * ldpart : nullptr;
    if (acc != nullptr)
        for (int h = t; h < H; h += ROW_THREADS) acc[h] = 0.0;
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
        const double *src = M + r * ldm;
        const double wr = (w != nullptr) ? w[r] : 1.0;
        if constexpr (ITER) {
            double m = -INFINITY;
            for (int h = t; h < H; h += ROW_THREADS) m = fmax(m, src[h]);
            m = block_reduce<ROW_THREADS, true>(m, scratch);
            const double shift = isfinite(m) ? m : 0.0;
            double z = 0.0;
            for (int h = t; h < H; h += ROW_THREADS) z += exp(ln_props[h] + (src[h] - shift));
            z = block_reduce<ROW_THREADS, false>(z, scratch);
            const double c = weight_over_norm(wr, z);
            for (int h = t; h < H; h += ROW_THREADS) acc[h] += c * exp(src[h] - shift);
        } else {
            double m = -INFINITY;
            for (int h = t; h < H; h += ROW_THREADS) m = fmax(m, ln_props[h] + src[h]);
            m = block_reduce<ROW_THREADS, true>(m, scratch);
            const double shift = isfinite(m) ? m : 0.0;
            double s = 0.0;
            for (int h = t; h < H; h += ROW_THREADS) s += exp((ln_props[h] + src[h]) - shift);
            s = block_reduce<ROW_THREADS, false>(s, scratch);
            const double lse = log(s) + m;          // m (not shift): -inf rows give -inf, as scipy does
            for (int h = t; h < H; h += ROW_THREADS) {
                const double v = (ln_props[h] + src[h]) - lse;
                if (out != nullptr) {
                    double *o = out + (out_rows != nullptr ? out_rows[r] : r) * ldo + h;
                    *o = (mode == 1) ? logaddexp_f64(*o, v) : v;
                }
                if (acc != nullptr && wr != 0.0) acc[h] += wr * exp(v);   // scipy drops zero-weight rows, NaN or not
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// K6n estep_narrow: the same two computations for matrices with a handful of columns -- the
// refinement EM on the contributors' columns (bin/mixemt:311-320, H = 2..10) and its posterior.
// One THREAD per row (grid-stride): the row, the proportions and the column sums sit in
// registers, nothing is exchanged per row (estep_log_kernel spends a workgroup, two barriers and
// two block reductions on every 24-byte row: 2.75 ms per pass over 10^6 x 3, this kernel is
// bandwidth-bound on the same 24 MB).  Column sums: per-thread registers -> wave sum -> LDS ->
// partial[wg][h], reduced in fixed order by colreduce_kernel like everywhere else.
// ------------------------------------------------------------------------------------------
// LPR = lanes per row: 1 (H <= 32), or 2 adjacent lanes that own the columns [0, HMAX) and
// [HMAX, 2 HMAX) of one row and combine their row maximum / row sum through a DPP lane swap.
template <int HMAX, bool ITER, int LPR>
__global__ __launch_bounds__(256) void estep_narrow_kernel(
    const double *__restrict__ M, int64_t ldm, const double *__restrict__ w,
    const double *__restrict__ ln_props, int64_t R, int H, double *__restrict__ out, int64_t ldo,
    int mode, double *__restrict__ partial, int64_t ldpart,
    const mxm_em_state *__restrict__ state) {
    static_assert(LPR == 1 || LPR == 2, "lanes per row");
    __shared__ double red[4][HMAX * LPR];
    if (state != nullptr && state->done != 0) return;
    const int t = threadIdx.x;
    const int c0 = (LPR == 2) ? (t & 1) * HMAX : 0; // first column of this lane
    double lp[HMAX], acc[HMAX];                     // ITER: lp holds the LINEAR proportions
#pragma unroll
    for (int h = 0; h < HMAX; ++h) {
        const double l = (c0 + h < H) ? ln_props[c0 + h] : -INFINITY;
        lp[h] = ITER ? exp(l) : l;
        acc[h] = 0.0;
    }
    const int64_t stride = (int64_t)gridDim.x * (256 / LPR);
    for (int64_t r = (int64_t)blockIdx.x * (256 / LPR) + t / LPR; r < R; r += stride) {
        const double *src = M + r * ldm + c0;
        const double wr = (w != nullptr) ? w[r] : 1.0;
        double x[HMAX];
#pragma unroll
        for (int h = 0; h < HMAX; ++h) x[h] = (c0 + h < H) ? src[h] : -INFINITY;
        if constexpr (ITER) {
            // T_h += (w_r / Z_r) e_h,  e_h = exp(M_rh - rowmax_r),  Z_r = sum_h p_h e_h
            double m = x[0];
#pragma unroll
            for (int h = 1; h < HMAX; ++h) m = fmax(m, x[h]);
            if constexpr (LPR == 2) m = fmax(m, dpp_mov_f64<0xB1>(m));      // the row's other half
            const double shift = isfinite(m) ? m : 0.0;
            double z = 0.0;
#pragma unroll
            for (int h = 0; h < HMAX; ++h) {
                x[h] = exp(x[h] - shift);           // pad columns: exp(-inf) = 0
                z = fma(lp[h], x[h], z);
            }
            if constexpr (LPR == 2) z += dpp_mov_f64<0xB1>(z);
            const double c = weight_over_norm(wr, z);
#pragma unroll
            for (int h = 0; h < HMAX; ++h) acc[h] = fma(c, x[h], acc[h]);
        } else {
            // em.py:80-83 verbatim: z = ln p + M;  z -= logsumexp(z)
#pragma unroll
            for (int h = 0; h < HMAX; ++h) x[h] += lp[h];
            double m = x[0];
#pragma unroll
            for (int h = 1; h < HMAX; ++h) m = fmax(m, x[h]);
            if constexpr (LPR == 2) m = fmax(m, dpp_mov_f64<0xB1>(m));
            const double shift = isfinite(m) ? m : 0.0;
            double ssum = 0.0;
#pragma unroll
            for (int h = 0; h < HMAX; ++h) ssum += exp(x[h] - shift);
            if constexpr (LPR == 2) ssum += dpp_mov_f64<0xB1>(ssum);
            const double lse = log(ssum) + m;       // m (not shift): -inf rows stay -inf, as scipy does
#pragma unroll
            for (int h = 0; h < HMAX; ++h) {
                if (c0 + h < H) {
                    const double v = x[h] - lse;
                    if (out != nullptr) {
                        double *o = out + r * ldo + c0 + h;
                        *o = (mode == 1) ? logaddexp_f64(*o, v) : v;
                    }
                    if (partial != nullptr && wr != 0.0) acc[h] += wr * exp(v);   // scipy drops zero-weight rows, NaN or not
                }
            }
        }
    }
    if (partial != nullptr) {
#pragma unroll
        for (int h = 0; h < HMAX; ++h) {
            double a = acc[h];
            if constexpr (LPR == 2) {               // sum the lanes of equal parity: every xor step but the last
#pragma unroll
                for (int off = 32; off > 1; off >>= 1) a += __shfl_xor(a, off, 64);
            } else {
                a = wave_sum(a);
            }
            if ((t & 63) < LPR) red[t >> 6][c0 + h] = a;
        }
        __syncthreads();
        if (t < H) partial[(int64_t)blockIdx.x * ldpart + t] = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
    }
}

// ------------------------------------------------------------------------------------------
// K6b estep_wide: the same E-step (em.py:80-83, fold :156, M-step sums :87-88) for wide rows,
// one HBM read + one write per cell: the row is held in VGPRs across the two row reductions
// (max, then sum of exp), exactly like the streaming kernel holds it across its dot product.
// Needs H even, 16-byte aligned rows in M and out; everything else takes estep_log_kernel.
// ------------------------------------------------------------------------------------------
template <int NCH, bool COLSUM>
__global__ __launch_bounds__(256, 2) void estep_wide_kernel(
    const double *__restrict__ M, int64_t ldm, const double *__restrict__ w,
    const double *__restrict__ lnp_in, int64_t R, int H,
    double *__restrict__ out, int64_t ldo, int mode, double *__restrict__ partial, int64_t ldpart) {
    constexpr int THREADS = 256, NW = THREADS / 64;
    __shared__ double red[2][2][NW];               // [ring][max|sum][wave]
    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int ncol2 = H >> 1;

    const row_deal deal(R);                         // step q of this workgroup = row b + q * grid

    // with the M-step sums the exponentials have to survive the second reduction: that variant
    // gives up the register double buffer (two workgroups per CU still overlap load and math)
    constexpr int NBUF = COLSUM ? 1 : 2;
    d2 lp[NCH], acc[COLSUM ? NCH : 1];
    bool own[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c2 = t + k * THREADS;
        own[k] = c2 < ncol2;
        // a clamped (not owned) lane carries -inf log-proportions: it adds exp(-inf) = 0
        lp[k].x = own[k] ? lnp_in[2 * c2] : -INFINITY;
        lp[k].y = own[k] ? lnp_in[2 * c2 + 1] : -INFINITY;
        if constexpr (COLSUM) acc[k] = d2{0.0, 0.0};
    }

    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const int row_bytes = (int)(ldm * 8);
    const int voff = t * 16;
    int last_c2 = t + (NCH - 1) * THREADS;
    if (last_c2 > ncol2 - 1) last_c2 = ncol2 - 1;
    const int voff_last = last_c2 * 16;

    d2 x[NBUF][NCH];
    auto load_row = [&](d2(&xr)[NCH], int64_t q) {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(M + deal.row(q) * ldm), 0,
                                                            row_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < NCH - 1; ++k)
            xr[k] = __builtin_bit_cast(d2, (u4)__builtin_amdgcn_raw_buffer_load_b128(
                                               rsrc, voff, k * THREADS * 16, 2));
        xr[NCH - 1] = __builtin_bit_cast(d2, (u4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_last, 0, 2));
    };

    int ring = 0;
    auto process = [&](d2(&xr)[NCH], int64_t q) {
        const bool live = deal.live(q);
        const int64_t r = deal.row(q);
        double m = -INFINITY;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            xr[k].x += lp[k].x;                    // z = ln p + M   (em.py:80)
            xr[k].y += lp[k].y;
            m = fmax(m, fmax(xr[k].x, xr[k].y));
        }
        m = wave_max_lane63(m);
        if (lane == 63) red[ring][0][wv] = m;
        __syncthreads();
        m = red[ring][0][0];
#pragma unroll
        for (int q = 1; q < NW; ++q) m = fmax(m, red[ring][0][q]);
        const double shift = isfinite(m) ? m : 0.0;
        d2 e[COLSUM ? NCH : 1];
        double ssum = 0.0;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const double ex = exp(xr[k].x - shift), ey = exp(xr[k].y - shift);
            if constexpr (COLSUM) e[k] = d2{ex, ey};
            ssum += ex + ey;
        }
        ssum = wave_sum_lane63(ssum);
        if (lane == 63) red[ring][1][wv] = ssum;
        __syncthreads();
        ssum = red[ring][1][0];
#pragma unroll
        for (int q = 1; q < NW; ++q) ssum += red[ring][1][q];
        ring ^= 1;
        const double lse = log(ssum) + m;          // em.py:81-83 (m, not shift: -inf rows stay -inf)
        if (out != nullptr && live) {
            d2 *orow = reinterpret_cast<d2 *>(out + r * ldo);
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                if (own[k]) {
                    d2 v = d2{xr[k].x - lse, xr[k].y - lse};
                    d2 *dst = orow + t + k * THREADS;
                    if (mode == 1) {
                        const d2 old = *dst;
                        v.x = logaddexp_f64(old.x, v.x);
                        v.y = logaddexp_f64(old.y, v.y);
                    }
                    __builtin_nontemporal_store(v, dst);
                }
            }
        }
        if constexpr (COLSUM) {
            const double wr = live ? (w != nullptr ? w[r] : 1.0) : 0.0;
            const double c = weight_over_norm(wr, ssum);         // w * exp(z - lse) = w * e / sum
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                acc[k].x = fma(c, e[k].x, acc[k].x);
                acc[k].y = fma(c, e[k].y, acc[k].y);
            }
        }
    };

    if constexpr (NBUF == 2) {
        load_row(x[0], 0);
        for (int64_t q = 0; q < deal.nq; q += 2) {
            load_row(x[1], q + 1);
            process(x[0], q);
            load_row(x[0], q + 2);
            process(x[1], q + 1);
        }
    } else {
        for (int64_t q = 0; q < deal.nq; ++q) {
            load_row(x[0], q);
            process(x[0], q);
        }
    }
    if constexpr (COLSUM) {
        d2 *dst = reinterpret_cast<d2 *>(partial + (int64_t)blockIdx.x * ldpart);
#pragma unroll
        for (int k = 0; k < NCH; ++k)
            if (own[k]) dst[t + k * THREADS] = acc[k];
    }
}

#endif  // MIXEMT_ESTEP_KERNELS_HPP
