// aux_kernels.hpp -- part of libmixemt_hip.so (gfx950); included by mixemt_hip.hip only.
// Small vector kernels, consumers of the posterior (assemble.py:103-123, :284-334) and the bare-read diagnostic.
#ifndef MIXEMT_AUX_KERNELS_HPP
#define MIXEMT_AUX_KERNELS_HPP

// ------------------------------------------------------------------------------------------
// small vector kernels (one workgroup)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FIN_THREADS) void log_normalize_kernel(const double *__restrict__ colsum,
                                                                    int H, double *__restrict__ ln_new) {
    __shared__ double scratch[FIN_THREADS / 64];
    double s = 0.0;
    for (int h = threadIdx.x; h < H; h += FIN_THREADS) s += colsum[h];
    const double lt = log(block_reduce<FIN_THREADS, false>(s, scratch));
    for (int h = threadIdx.x; h < H; h += FIN_THREADS) ln_new[h] = log(colsum[h]) - lt;
}

__global__ __launch_bounds__(FIN_THREADS) void l1_exp_diff_kernel(const double *__restrict__ a,
                                                                  const double *__restrict__ b, int H,
                                                                  double *__restrict__ out) {
    __shared__ double scratch[FIN_THREADS / 64];
    double s = 0.0;
    for (int h = threadIdx.x; h < H; h += FIN_THREADS) s += fabs(exp(a[h]) - exp(b[h]));
    s = block_reduce<FIN_THREADS, false>(s, scratch);
    if (threadIdx.x == 0) out[0] = s;
}

__global__ __launch_bounds__(256) void add_scalar_kernel(double *__restrict__ x, int64_t ld, int64_t R,
                                                         int H, double delta) {
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
        double *row = x + r * ld;
        for (int h = threadIdx.x; h < H; h += 256) row[h] += delta;
    }
}

// first index of the row maximum (numpy.argmax: a NaN counts as the maximum, first one wins)
// + weighted votes (assemble.py:115-123).  Candidate order: NaN before numbers, then larger
// value, then smaller index.
__device__ __forceinline__ bool cand_better(int an, double av, int ai, int bn, double bv, int bi) {
    // branch-free (bitwise on purpose): as early returns this compiled to five exec-masked branches per comparison
    const bool nan_wins = an > bn, same_kind = an == bn;
    const bool by_value = (an == 0) & (av != bv);
    return nan_wins | (same_kind & (by_value ? (av > bv) : (ai < bi)));
}

__global__ __launch_bounds__(ROW_THREADS) void row_argmax_votes_kernel(
    const double *__restrict__ X, int64_t ldx, const double *__restrict__ w, int64_t R, int H,
    int32_t *__restrict__ best, double *__restrict__ vote_part, int64_t ldpart) {
    // vote_part[wg][h]: this workgroup's votes, added up by ONE thread in row order (no atomics:
    // fractional weights give the same bits on every run); colreduce_kernel sums the workgroups.
    constexpr int NW = ROW_THREADS / 64;
    __shared__ double s_val[NW];
    __shared__ int s_idx[NW];
    __shared__ int s_nan[NW];
    const int t = threadIdx.x;
    double *my_votes = vote_part != nullptr ? vote_part + (int64_t)blockIdx.x * ldpart : nullptr;
    if (my_votes != nullptr) {
        for (int h = t; h < H; h += ROW_THREADS) my_votes[h] = 0.0;
        __threadfence_block();
        __syncthreads();
    }
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
        const double *row = X + r * ldx;
        int cn = 0, ci = 0x7fffffff;
        double cv = -INFINITY;
        for (int h = t; h < H; h += ROW_THREADS) {
            const double v = row[h];
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
        __syncthreads();
        if ((t & 63) == 0) { s_val[t >> 6] = cv; s_idx[t >> 6] = ci; s_nan[t >> 6] = cn; }
        __syncthreads();
        if (t == 0) {
            for (int q = 1; q < NW; ++q)
                if (cand_better(s_nan[q], s_val[q], s_idx[q], cn, cv, ci)) { cn = s_nan[q]; cv = s_val[q]; ci = s_idx[q]; }
            if (ci >= H) ci = 0;
            best[r] = ci;
            if (my_votes != nullptr) {
                volatile double *slot = my_votes + ci;          // same thread wrote the zero / the last sum
                *slot = *slot + ((w != nullptr) ? w[r] : 1.0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Wide rows (H even, 16-byte aligned): the same answer as row_argmax_votes_kernel with the
// streaming kernels' machinery -- rows dealt over the workgroups, 16-byte non-temporal buffer
// loads, the next row in flight while this one is reduced, DPP reductions, one barrier per row.
// Votes are collected per workgroup in LDS (ds_add_f64) and flushed once, instead of 10^6
// global atomics onto the handful of winning columns.
// ------------------------------------------------------------------------------------------
struct argmax_cand {
    double v;
    int i, n;                                       // value, column, is-NaN
};
__device__ __forceinline__ argmax_cand cand_pick(const argmax_cand &a, const argmax_cand &b) {
    return cand_better(b.n, b.v, b.i, a.n, a.v, a.i) ? b : a;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ argmax_cand cand_dpp(const argmax_cand &c) {   // masked-out rows keep their own
    argmax_cand o;
    o.v = dpp_mov_old_f64<CTRL, ROW_MASK>(c.v, c.v);
    o.i = __builtin_amdgcn_update_dpp(c.i, c.i, CTRL, ROW_MASK, 0xf, false);
    o.n = __builtin_amdgcn_update_dpp(c.n, c.n, CTRL, ROW_MASK, 0xf, false);
    return o;
}
__device__ __forceinline__ argmax_cand wave_best_lane63(argmax_cand c) {
    c = cand_pick(c, cand_dpp<0xB1, 0xF>(c));
    c = cand_pick(c, cand_dpp<0x4E, 0xF>(c));
    c = cand_pick(c, cand_dpp<0x141, 0xF>(c));
    c = cand_pick(c, cand_dpp<0x140, 0xF>(c));
    c = cand_pick(c, cand_dpp<0x142, 0xA>(c));
    c = cand_pick(c, cand_dpp<0x143, 0xC>(c));
    return c;
}

template <int NCH>
__global__ __launch_bounds__(256, 2) void row_argmax_wide_kernel(
    const double *__restrict__ X, int64_t ldx, const double *__restrict__ w, int64_t R, int H,
    int32_t *__restrict__ best, double *__restrict__ votes, int64_t ldpart) {
    // votes = vote_part[wg][h] (this workgroup's row is written whole; colreduce_kernel sums them)
    constexpr int THREADS = 256, NW = THREADS / 64;
    extern __shared__ double lds_votes[];           // [H] when votes are wanted
    __shared__ double s_val[2][NW];
    __shared__ int s_idx[2][NW], s_nan[2][NW];
    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int ncol2 = H >> 1;
    const row_deal deal(R);
    if (votes != nullptr) {
        for (int h = t; h < H; h += THREADS) lds_votes[h] = 0.0;
        __syncthreads();
    }
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const int row_bytes = (int)(ldx * 8);
    const int voff = t * 16;
    int last_c2 = t + (NCH - 1) * THREADS;
    const bool last_own = last_c2 < ncol2;
    if (!last_own) last_c2 = ncol2 - 1;             // a clamped lane re-reads a real element: harmless
    const int voff_last = last_c2 * 16;

    d2 x[2][NCH];
    auto load_row = [&](d2(&xr)[NCH], int64_t q) {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(X + deal.row(q) * ldx), 0,
                                                            row_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < NCH - 1; ++k)
            xr[k] = __builtin_bit_cast(d2, (u4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, k * THREADS * 16, 2));
        xr[NCH - 1] = __builtin_bit_cast(d2, (u4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_last, 0, 2));
    };
    int ring = 0;
    auto process = [&](d2(&xr)[NCH], int64_t q) {
        // this lane's candidate in two cheap sweeps over its registers: the maximum (v_max ignores
        // NaNs) and whether there is a NaN at all; then the first column holding it (descending,
        // so the smallest column is written last).  NaN rows are rare: the wave branches only then.
        double m = -INFINITY;
        bool any_nan = false;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            m = fmax(m, fmax(xr[k].x, xr[k].y));
            any_nan = any_nan || (xr[k].x != xr[k].x) || (xr[k].y != xr[k].y);
        }
        argmax_cand c{m, 0x7fffffff, 0};
#pragma unroll
        for (int k = NCH - 1; k >= 0; --k) {
            const int col = 2 * ((k < NCH - 1) ? (t + k * THREADS) : last_c2);
            c.i = (xr[k].y == m) ? col + 1 : c.i;
            c.i = (xr[k].x == m) ? col : c.i;
        }
        if (__builtin_amdgcn_ballot_w64(any_nan) != 0) {
            if (any_nan) {
                c.n = 1;
                c.i = 0x7fffffff;
#pragma unroll
                for (int k = NCH - 1; k >= 0; --k) {
                    const int col = 2 * ((k < NCH - 1) ? (t + k * THREADS) : last_c2);
                    c.i = (xr[k].y != xr[k].y) ? col + 1 : c.i;
                    c.i = (xr[k].x != xr[k].x) ? col : c.i;
                }
            }
        }
        c = wave_best_lane63(c);
        if (lane == 63) { s_val[ring][wv] = c.v; s_idx[ring][wv] = c.i; s_nan[ring][wv] = c.n; }
        __syncthreads();
        if (t == 0 && deal.live(q)) {
            argmax_cand g{s_val[ring][0], s_idx[ring][0], s_nan[ring][0]};
#pragma unroll
            for (int j = 1; j < NW; ++j) g = cand_pick(g, argmax_cand{s_val[ring][j], s_idx[ring][j], s_nan[ring][j]});
            const int64_t r = deal.row(q);
            int ci = g.i;
            if (ci >= H) ci = 0;
            best[r] = ci;
            if (votes != nullptr) lds_votes[ci] += (w != nullptr) ? w[r] : 1.0;   // one thread, row order
        }
        ring ^= 1;
    };
    load_row(x[0], 0);
    for (int64_t q = 0; q < deal.nq; q += 2) {
        load_row(x[1], q + 1);
        process(x[0], q);
        load_row(x[0], q + 2);
        process(x[1], q + 1);
    }
    if (votes != nullptr) {
        __syncthreads();
        double *dst = votes + (int64_t)blockIdx.x * ldpart;
        for (int h = t; h < H; h += THREADS) dst[h] = lds_votes[h];
    }
}

// Read -> contributor assignment (assemble.py:284-334): per row, among the contributor columns
// only, the two largest  X[r][c] - log p_c ; assigned to the best one if the gap reaches
// log(min_fold), else unassigned (-1).  One thread per row; the row touches nC scattered cells.
// Order among exactly equal values follows numpy.argsort(...)[::-1]: the larger column wins.
__global__ __launch_bounds__(256) void assign_reads_kernel(const double *__restrict__ X, int64_t ldx,
                                                           const double *__restrict__ log_props,
                                                           const int32_t *__restrict__ cols, int nC, int64_t R,
                                                           double log_min_fold, int32_t *__restrict__ assigned) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const double *row = X + r * ldx;
    double v1 = -INFINITY, v2 = -INFINITY;      // best, runner-up
    int i1 = -1, c1 = -1, c2 = -1;
    for (int i = 0; i < nC; ++i) {
        const int c = cols[i];
        const double v = row[c] - log_props[c];
        if (i1 < 0 || v > v1 || (v == v1 && c > c1)) {
            v2 = v1; c2 = c1;
            v1 = v; c1 = c; i1 = i;
        } else if (c2 < 0 || v > v2 || (v == v2 && c > c2)) {
            v2 = v; c2 = c;
        }
    }
    assigned[r] = (nC >= 2 && (v1 - v2) >= log_min_fold) ? i1 : -1;
}

// Column subset for the refinement EM (preprocess.py:247-251: em_mat[:, indexes]):
// out[r][i] = M[r][cols[i]].  A wave covers 64 / nC' rows (nC' = nC rounded up to a power of two,
// capped at 64), so the writes of a wave are contiguous; the reads touch nC sectors per row.
__global__ __launch_bounds__(256) void gather_columns_kernel(const double *__restrict__ M, int64_t ldm, int64_t R,
                                                             const int32_t *__restrict__ cols, int nC,
                                                             double *__restrict__ out, int64_t ldo,
                                                             const int64_t *__restrict__ out_rows = nullptr) {
    const int64_t n = R * (int64_t)nC;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / nC;
        const int c = (int)(i - r * nC);
        out[(out_rows != nullptr ? out_rows[r] : r) * ldo + c] = M[r * ldm + cols[c]];
    }
}

// Log-space fold of posterior blocks (em.py:156, :161 across ranks / runs):
//   acc[r][h] = logaddexp(...logaddexp(logaddexp(acc, in[0]), in[1])..., in[n_in - 1]) + delta
// in fixed order (deterministic).  Entries far below exp(-745) keep their finite logs, unlike a
// sum in linear space.  16-byte vectors when H, the leading dimensions and the pointers allow.
#define FOLD_MAX_IN 8
struct fold_inputs {
    const double *ptr[FOLD_MAX_IN];
    int64_t ld[FOLD_MAX_IN];
};
template <bool VEC>
__global__ __launch_bounds__(256) void fold_logaddexp_kernel(double *__restrict__ acc, int64_t lda, fold_inputs in,
                                                             int n_in, int64_t R, int H, double delta) {
    const int hw = VEC ? (H >> 1) : H;                  // work items per row
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
        for (int c = threadIdx.x; c < hw; c += 256) {
            if constexpr (VEC) {
                d2 *dst = reinterpret_cast<d2 *>(acc + r * lda) + c;
                d2 v = *dst;
                for (int k = 0; k < n_in; ++k) {
                    const d2 x = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(in.ptr[k] + r * in.ld[k]) + c);
                    v.x = logaddexp_f64(v.x, x.x);
                    v.y = logaddexp_f64(v.y, x.y);
                }
                *dst = d2{v.x + delta, v.y + delta};
            } else {
                double *dst = acc + r * lda + c;
                double v = *dst;
                for (int k = 0; k < n_in; ++k) v = logaddexp_f64(v, in.ptr[k][r * in.ld[k] + c]);
                *dst = v + delta;
            }
        }
    }
}

// Diagnostic only: bare streaming reads (16 B/lane, 8 loads in flight per lane, xor-folded so
// nothing is optimised away) -- the practical HBM read ceiling the streaming kernel's roofline
// fraction is judged against (tools/stream_ceiling.py).  BLOCKED = false: grid-stride, plain
// loads (the textbook pattern); true: one contiguous block per workgroup, non-temporal loads
// (the EM kernel's pattern).
template <bool BLOCKED>
__global__ __launch_bounds__(256) void diag_stream_read_kernel(const uint4 *__restrict__ src_in, int64_t n16,
                                                               unsigned int *__restrict__ sink) {
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    const u4v *src = reinterpret_cast<const u4v *>(src_in);
    unsigned int acc = 0;
    if (BLOCKED) {
        const int64_t per_wg = (n16 + gridDim.x - 1) / gridDim.x;
        const int64_t lo = (int64_t)blockIdx.x * per_wg;
        const int64_t hi = (lo + per_wg < n16) ? lo + per_wg : n16;
        int64_t i = lo + threadIdx.x;
        for (; i + 7 * 256 < hi; i += 8 * 256) {
            u4v v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = __builtin_nontemporal_load(src + i + q * 256);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc ^= v[q].x ^ v[q].y ^ v[q].z ^ v[q].w;
        }
        for (; i < hi; i += 256) {
            const u4v v = __builtin_nontemporal_load(src + i);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    } else {
        const int64_t stride = (int64_t)gridDim.x * 256;
        int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        for (; i + 7 * stride < n16; i += 8 * stride) {
            u4v v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = src[i + q * stride];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc ^= v[q].x ^ v[q].y ^ v[q].z ^ v[q].w;
        }
        for (; i < n16; i += stride) {
            const u4v v = src[i];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;          // practically never: keeps the loads alive
}

// Diagnostic only: the streaming EM kernel's access pattern with the arithmetic taken out --
// 512 threads, rows of `row16` 16-byte units dealt over the workgroups, per-row buffer
// descriptor, non-temporal loads, register ring of 3 rows; the loaded words are xor-folded.
template <int NCH>
__global__ __launch_bounds__(512, 2) void diag_stream_dealt_kernel(const double *__restrict__ src, int64_t R,
                                                                   int row16, unsigned int *__restrict__ sink) {
    constexpr int THREADS = 512, NBUF = 3;
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const int t = threadIdx.x;
    const row_deal deal(R);
    const int row_bytes = row16 * 16;
    const int voff = t * 16;
    int last = t + (NCH - 1) * THREADS;
    if (last > row16 - 1) last = row16 - 1;
    const int voff_last = last * 16;
    u4 x[NBUF][NCH];
    unsigned int acc = 0;
    auto load_row = [&](u4(&xr)[NCH], int64_t q) {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double *>(src + deal.row(q) * (int64_t)row16 * 2), 0, row_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < NCH - 1; ++k)
            xr[k] = (u4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, k * THREADS * 16, 2);
        xr[NCH - 1] = (u4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_last, 0, 2);
    };
#pragma unroll
    for (int j = 0; j < NBUF - 1; ++j) load_row(x[j], j);
    for (int64_t q = 0; q < deal.nq; q += NBUF) {
#pragma unroll
        for (int j = 0; j < NBUF; ++j) {
            load_row(x[(j + NBUF - 1) % NBUF], q + j + NBUF - 1);
#pragma unroll
            for (int k = 0; k < NCH; ++k) acc ^= x[j][k].x ^ x[j][k].y ^ x[j][k].z ^ x[j][k].w;
        }
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;
}

// Diagnostic only: em_iter_coded_kernel's access pattern with the arithmetic taken out -- 256 threads, records of
// `rec_bytes` bytes (code words as 4 B per lane, NCH of them per thread, then a table of `tbl` doubles read as 8 B per
// lane by the first `tbl` threads) dealt over the workgroups, non-temporal, three records in flight.  Used to calibrate
// the FETCH_SIZE counter for this access width (tools/pmc_summary.py) and as the bare-read ceiling of that pattern.
template <int NCH>
__global__ __launch_bounds__(256, 2) void diag_stream_records_kernel(const uint8_t *__restrict__ src, int64_t R, int rec_bytes,
                                                                     int code_bytes, int tbl, unsigned int *__restrict__ sink) {
    constexpr int THREADS = 256, NBUF = 3;
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    const int t = threadIdx.x;
    const row_deal deal(R);
    const int nword = code_bytes >> 2;
    int last = t + (NCH - 1) * THREADS;
    if (last > nword - 1) last = nword - 1;
    unsigned int x[NBUF][NCH];
    u2v y[NBUF];
    unsigned int acc = 0;
    auto load_rec = [&](unsigned int(&xr)[NCH], u2v &yr, int64_t q) {
        const uint8_t *base = src + deal.row(q) * (int64_t)rec_bytes;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, code_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < NCH - 1; ++k) xr[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, t * 4, k * THREADS * 4, 2);
        xr[NCH - 1] = __builtin_amdgcn_raw_buffer_load_b32(rs, last * 4, 0, 2);
        const auto rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base + code_bytes), 0, tbl * 8, 0x00020000);
        yr = __builtin_amdgcn_raw_buffer_load_b64(rt, t * 8, 0, 2);
    };
#pragma unroll
    for (int j = 0; j < NBUF - 1; ++j) load_rec(x[j], y[j], j);
    for (int64_t q = 0; q < deal.nq; q += NBUF) {
#pragma unroll
        for (int j = 0; j < NBUF; ++j) {
            load_rec(x[(j + NBUF - 1) % NBUF], y[(j + NBUF - 1) % NBUF], q + j + NBUF - 1);
#pragma unroll
            for (int k = 0; k < NCH; ++k) acc ^= x[j][k];
            acc ^= y[j].x ^ y[j].y;
        }
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;
}

// Diagnostic only: em_iter_coded_kernel's OWN loads over a real record buffer -- every byte-coded row's code words
// (4 B per lane) and P table (8 B per lane), the wide rows' 16-bit codes and tables, through the same per-row
// descriptors, non-temporal, three rows in flight -- and nothing else.  What FETCH_SIZE shows for this kernel is what
// the counter makes of exactly the bytes the EM kernel has to read; the EM kernel's own reading divided by it is its
// traffic ratio, whatever the counter's factor for these access widths is (tools/pmc_calibrate_coded.py).
template <int NCH>
__global__ __launch_bounds__(256, 2) void diag_stream_coded_kernel(const uint8_t *__restrict__ rec, const int64_t *__restrict__ rec_off,
                                                                   const int32_t *__restrict__ ndist, int ldc, int64_t R,
                                                                   unsigned int *__restrict__ sink) {
    constexpr int THREADS = 256, NBUF = 3;
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    const int t = threadIdx.x;
    const row_deal deal(R);
    const int nword = ldc >> 2;
    int last = t + (NCH - 1) * THREADS;
    if (last > nword - 1) last = nword - 1;
    u2v x[NBUF][NCH];
    u2v y[NBUF][4];
    unsigned int acc = 0;
    auto load_rec = [&](u2v(&xr)[NCH], u2v(&yr)[4], int64_t q) {
        const int64_t r = deal.row(q);
        const int nd = ndist[r];
        const uint8_t *base = rec + rec_off[r];
        const bool wide = nd > 256;
        const int cbytes = nd > 0 ? (wide ? 2 * ldc : ldc) : 0;      // a row without a record: nothing is read
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, cbytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int wd = (k < NCH - 1) ? t + k * THREADS : last;
            if (wide) {
                xr[k] = __builtin_amdgcn_raw_buffer_load_b64(rs, wd * 8, 0, 2);
            } else {
                xr[k].x = __builtin_amdgcn_raw_buffer_load_b32(rs, wd * 4, 0, 2);
                xr[k].y = 0;
            }
        }
        const auto rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base + cbytes), 0, (nd > 0 ? nd : 0) * 8, 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j) yr[j] = __builtin_amdgcn_raw_buffer_load_b64(rt, (t + j * THREADS) * 8, 0, 2);
    };
#pragma unroll
    for (int j = 0; j < NBUF - 1; ++j) load_rec(x[j], y[j], j);
    for (int64_t q = 0; q < deal.nq; q += NBUF) {
#pragma unroll
        for (int j = 0; j < NBUF; ++j) {
            load_rec(x[(j + NBUF - 1) % NBUF], y[(j + NBUF - 1) % NBUF], q + j + NBUF - 1);
#pragma unroll
            for (int k = 0; k < NCH; ++k) acc ^= x[j][k].x ^ x[j][k].y;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc ^= y[j][k].x ^ y[j][k].y;
        }
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;
}

// ------------------------------------------------------------------------------------------
// expand_tables: the dense expected-base table E[S][lde] (or its 4-bit code form Ecode, through `map`) from the marker form
// (maj + the (site, haplogroup, base) triples that differ from it: 0.4 MB) ON THE DEVICE -- the host used to upload the
// 22 MB table and gather the codes with torch operators (a 176 MB index temporary; the operators' first uses were most of a
// cold process's 118 ms "tables" stage).  One workgroup per site row: fill with the majority base, then the markers.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void expand_tables_kernel(const uint8_t *__restrict__ maj, const int32_t *__restrict__ mk_ptr,
                                                           const uint16_t *__restrict__ mk_hap, const uint8_t *__restrict__ mk_base,
                                                           const uint8_t *__restrict__ map, int S, int H, int64_t lde,
                                                           uint8_t *__restrict__ out) {
    for (int s = blockIdx.x; s < S; s += gridDim.x) {
        uint8_t *row = out + (int64_t)s * lde;
        const uint8_t fill = (map != nullptr) ? map[maj[s]] : maj[s];
        for (int64_t h = threadIdx.x; h < lde; h += 256) row[h] = (h < H) ? fill : (uint8_t)0;
        __syncthreads();
        for (int j = mk_ptr[s] + (int)threadIdx.x; j < mk_ptr[s + 1]; j += 256) {
            const int hap = mk_hap[j];
            if (hap < H) row[hap] = (map != nullptr) ? map[mk_base[j]] : mk_base[j];
        }
        __syncthreads();
    }
}

#endif  // MIXEMT_AUX_KERNELS_HPP
