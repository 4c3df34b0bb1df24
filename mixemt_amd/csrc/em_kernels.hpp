// em_kernels.hpp -- part of libmixemt_hip.so (gfx950); included by mixemt_hip.hip only.
// The EM iteration (em.py:57-91, :126-143): linearise, fused E+M streaming kernels, column reduce, finalize.
#ifndef MIXEMT_EM_KERNELS_HPP
#define MIXEMT_EM_KERNELS_HPP

// ------------------------------------------------------------------------------------------
// K2  linearize:  rowmax[r], P[r][h] = exp(M[r][h] - rowmax[r])   (one-time)
// ------------------------------------------------------------------------------------------
#define ROW_THREADS 256

__global__ __launch_bounds__(ROW_THREADS) void linearize_kernel(const double *__restrict__ M,
                                                                int64_t ldm, int64_t R, int H,
                                                                double *__restrict__ P, int64_t ldp,
                                                                double *__restrict__ rowmax) {
    __shared__ double scratch[ROW_THREADS / 64];
    const int t = threadIdx.x;
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
        const double *src = M + r * ldm;
        double m = -INFINITY;
        for (int h = t; h < H; h += ROW_THREADS) m = fmax(m, src[h]);
        m = block_reduce<ROW_THREADS, true>(m, scratch);
        const double shift = isfinite(m) ? m : 0.0;
        double *dst = P + r * ldp;
        for (int h = t; h < (int)ldp; h += ROW_THREADS) dst[h] = (h < H) ? exp(src[h] - shift) : 0.0;
        if (t == 0) rowmax[r] = shift;
    }
}

// ------------------------------------------------------------------------------------------
// K2w linearize for wide, 16-byte aligned rows: one read + one write per cell (the row waits in
// VGPRs for its maximum, like the streaming kernel's row waits for its dot product).
// ST = double or float (the opt-in storage variant).  H even, M rows 16-byte aligned.
// ------------------------------------------------------------------------------------------
template <int NCH, typename ST>
__global__ __launch_bounds__(256, 2) void linearize_wide_kernel(const double *__restrict__ M, int64_t ldm,
                                                                int64_t R, int H,
                                                                ST *__restrict__ P, int64_t ldp,
                                                                double *__restrict__ rowmax) {
    constexpr int THREADS = 256, NW = THREADS / 64;
    __shared__ double red[2][NW];
    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int ncol2 = H >> 1;
    const row_deal deal(R);

    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const int row_bytes = (int)(ldm * 8);
    const int voff = t * 16;
    int last_c2 = t + (NCH - 1) * THREADS;
    const bool last_own = last_c2 < ncol2;
    if (!last_own) last_c2 = ncol2 - 1;
    const int voff_last = last_c2 * 16;

    d2 x[2][NCH];
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
        double m = -INFINITY;                   // a clamped lane repeats a real element: harmless for a max
#pragma unroll
        for (int k = 0; k < NCH; ++k) m = fmax(m, fmax(xr[k].x, xr[k].y));
        m = wave_max_lane63(m);
        if (lane == 63) red[ring][wv] = m;
        __syncthreads();
        m = red[ring][0];
#pragma unroll
        for (int q = 1; q < NW; ++q) m = fmax(m, red[ring][q]);
        ring ^= 1;
        if (!deal.live(q)) return;
        const int64_t r = deal.row(q);
        const double shift = isfinite(m) ? m : 0.0;
        if (t == 0) rowmax[r] = shift;
        ST *prow = P + r * ldp;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (k < NCH - 1 || last_own) {
                const int c = 2 * (t + k * THREADS);
                const double ex = exp(xr[k].x - shift), ey = exp(xr[k].y - shift);
                if constexpr (sizeof(ST) == 8) {
                    __builtin_nontemporal_store(d2{ex, ey}, reinterpret_cast<d2 *>(prow + c));
                } else {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    __builtin_nontemporal_store(f2{(float)ex, (float)ey}, reinterpret_cast<f2 *>(prow + c));
                }
            }
        }
        // pad columns [H, ldp) of P are zero by contract
        for (int c = H + t; c < (int)ldp; c += THREADS) prow[c] = (ST)0;
    };
    load_row(x[0], 0);
    for (int64_t q = 0; q < deal.nq; q += 2) {
        load_row(x[1], q + 1);
        process(x[0], q);
        load_row(x[0], q + 2);
        process(x[1], q + 1);
    }
}

// ------------------------------------------------------------------------------------------
// K3  em_iter_wide: fused E+M step in linear space, BT restarts per pass over the matrix.
//
//   Z_r[b]     = sum_h p_h[b] P_rh                  (row reduction, per restart)
//   acc_h[b]  += (w_r / Z_r[b]) * P_rh              (column accumulation, per workgroup)
//   T_h[b]     = sum_wg acc_h[b]                    (colreduce_kernel; finalize applies p_h)
//
// which is em.py:80-88 with exp(M - rowmax) hoisted out of the loop:
//   posterior_rh = p_h P_rh / Z_r,   colsum_h = sum_r w_r posterior_rh = p_h T_h.
//
// Rows are dealt round-robin over the workgroups (row_deal, common.hpp).  Thread t owns the
// double2 column pairs {t + THREADS k}, k < NCH: one 16-byte buffer load per pair per row (a wave
// instruction covers 1 KiB contiguous); the row stays in VGPRs between the dot products and the
// accumulation, so the matrix is read from HBM exactly once per pass.  NBUF - 1 further rows are
// in flight in a register ring.  Per row: dot products -> in-wave DPP sum -> one LDS exchange +
// barrier -> w/Z through SGPRs (group_ratio_to_sgpr) -> accumulate.  BT restarts share each row:
// every restart adds accumulator registers and a proportion vector (PREG of them in VGPRs, the
// rest in LDS), the bytes read stay the same.  Column partials live in registers for the whole
// kernel and are written once: partial[wg][b][h], summed in fixed order afterwards -> bitwise
// reproducible.  Shapes in use (mixemt_hip.hip): BT = 1 <512 threads, ring 3>, BT = 2..4
// <512, ring 2>, one workgroup per CU; <256, ring 2> x 2 per CU is the alternative BT = 1 shape.
// ------------------------------------------------------------------------------------------
#ifndef MXM_V1_MINW
#define MXM_V1_MINW 2                 // min waves/SIMD the ring-2 BT = 1 shape is compiled for (2 WGs of 256 per CU)
#endif
#ifndef MXM_LOAD_AUX
#define MXM_LOAD_AUX 2                // cache policy of the row loads: 2 = non-temporal (streamed once per pass)
#endif
// batched shapes run one workgroup per CU: min waves/SIMD = THREADS / 256

// PREG of the BT restarts keep their proportions in VGPRs, the other BT - PREG in LDS.
template <int THREADS, int NCH, int BT, int NBUF, int PREG>
__global__ __launch_bounds__(THREADS, ((BT == 1 && NBUF == 2) ? MXM_V1_MINW : THREADS / 256)) void em_iter_wide_kernel(
    const double *__restrict__ P, int64_t ldp, const double *__restrict__ w,
    const double *__restrict__ props, int64_t R, int H,
    double *__restrict__ partial, int64_t ldpart, const mxm_em_state *__restrict__ state, mxm_slots slots) {
    // props / state are the bases of the loop vectors; restart b of this tile is slots.s[b]
    constexpr int NW = THREADS / 64;
    __shared__ double red[2][BT][NW];
    if (state != nullptr) {
        bool any = false;
#pragma unroll
        for (int b = 0; b < BT; ++b) any = any || (state[slots.s[b]].done == 0);
        if (!any) return;                           // every restart of this tile has stopped
    }

    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int ncol2 = (H + 1) >> 1;                 // d2 pairs per row (pad column is 0 in P)

    // proportions: registers for a single restart; a batch keeps (most of) them in LDS as
    // [b][k][thread] pairs (one conflict-free ds_read_b128 per use) so that the VGPR
    // budget goes to the accumulators and the row ring
    static_assert(PREG >= 0 && PREG <= BT, "restarts with register-resident proportions");
    extern __shared__ d2 lds_p[];
    constexpr bool P_IN_LDS = PREG < BT;
    d2 p[PREG > 0 ? PREG : 1][NCH], acc[BT][NCH];
#pragma unroll
    for (int b = 0; b < BT; ++b) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = 2 * (t + k * THREADS);
            d2 v;
            v.x = (c < H) ? props[(int64_t)slots.s[b] * H + c] : 0.0;
            v.y = (c + 1 < H) ? props[(int64_t)slots.s[b] * H + c + 1] : 0.0;
            if (b < PREG) p[b < PREG ? b : 0][k] = v;
            else lds_p[((b - PREG) * NCH + k) * THREADS + t] = v;
            acc[b][k] = d2{0.0, 0.0};
        }
    }

    const row_deal deal(R);                         // step q of this workgroup = row b + q * grid

    // Row loads: buffer_load_dwordx4 through a per-row descriptor (scalar registers only).
    // Per-lane offset = one VGPR (t * 16), the chunk offset is an immediate, so no 64-bit
    // per-load addresses and no exec-masked branches: steps past the workgroup's last row and
    // column pairs past the row are CLAMPED to a valid element instead of skipped -- a clamped
    // step gets weight 0 below, a clamped column has p = 0 and its accumulator is never stored.
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const int row_bytes = (int)(ldp * 8);
    const int voff = t * 16;
    int last_c2 = t + (NCH - 1) * THREADS;
    if (last_c2 > ncol2 - 1) last_c2 = ncol2 - 1;
    const int voff_last = last_c2 * 16;

    d2 x[NBUF][NCH];                                // register ring: NBUF - 1 rows in flight

    auto load_row = [&](d2(&xr)[NCH], int64_t q) {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(P + deal.row(q) * ldp), 0,
                                                            row_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < NCH - 1; ++k)
            xr[k] = __builtin_bit_cast(d2, (u4)__builtin_amdgcn_raw_buffer_load_b128(
                                               rsrc, voff, k * THREADS * 16, MXM_LOAD_AUX /* 2 = nt */));
        xr[NCH - 1] = __builtin_bit_cast(
            d2, (u4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_last, 0, MXM_LOAD_AUX /* 2 = nt */));
    };

    int buf = 0;
    auto process = [&](d2(&xr)[NCH], int64_t q) {
        double d[BT];
        // the row's weight is a scalar load with a cache line of its own (rows of a workgroup are a grid
        // apart): asked for here, it arrives during the dot products; asked for after the barrier it was an
        // L2 round trip on every row's critical path (batches of 3 / 4: 6.47 -> 6.21 / 6.80 -> 6.54 ms per pass)
        double wr = deal.live(q) ? (w != nullptr ? w[deal.row(q)] : 1.0) : 0.0;
        asm volatile("" : "+s"(wr));
        // keep the batch's proportions IN LDS: without this the loads are loop-invariant
        // and get hoisted back into (BT * NCH * 4) VGPRs
        if constexpr (P_IN_LDS) asm volatile("" ::: "memory");
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                d2 pk;
                if (b < PREG) pk = p[b < PREG ? b : 0][k];
                else pk = lds_p[((b - PREG) * NCH + k) * THREADS + t];     // own slot: no barrier needed
                s = fma(xr[k].x, pk.x, s);
                s = fma(xr[k].y, pk.y, s);
            }
            d[b] = s;
        }
#pragma unroll
        for (int b = 0; b < BT; ++b) d[b] = wave_sum_lane63(d[b]);
        if (lane == 63) {
#pragma unroll
            for (int b = 0; b < BT; ++b) red[buf][b][wv] = d[b];
        }
        __syncthreads();
        double cs[BT];                                  // w_r / Z_r per restart
        group_ratio_to_sgpr<NW, BT>(&red[buf][0][0], lane, wr, cs);
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            const double c = cs[b];
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                acc[b][k].x = fma(c, xr[k].x, acc[b][k].x);
                acc[b][k].y = fma(c, xr[k].y, acc[b][k].y);
            }
        }
        buf ^= 1;
    };

#pragma unroll
    for (int j = 0; j < NBUF - 1; ++j) load_row(x[j], j);
    for (int64_t q = 0; q < deal.nq; q += NBUF) {
#pragma unroll
        for (int j = 0; j < NBUF; ++j) {
            load_row(x[(j + NBUF - 1) % NBUF], q + j + NBUF - 1);
            process(x[j], q + j);
        }
    }

#pragma unroll
    for (int b = 0; b < BT; ++b) {
        d2 *dst = reinterpret_cast<d2 *>(partial + ((int64_t)blockIdx.x * BT + b) * ldpart);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c2 = t + k * THREADS;
            if (c2 < ncol2) dst[c2] = acc[b][k];
        }
    }
}

// ------------------------------------------------------------------------------------------
// K3f  fp32-STORAGE variant of the streaming kernel (opt-in, labelled as such everywhere):
// P is kept as float (half the HBM bytes per iteration), every product and sum stays fp64.
// Same structure as em_iter_wide_kernel with 4 columns per 16-byte load; one restart per pass.
// ------------------------------------------------------------------------------------------
typedef float f4 __attribute__((ext_vector_type(4)));

template <int THREADS, int NCH, int NBUF>
__global__ __launch_bounds__(THREADS, 2) void em_iter_wide_f32_kernel(
    const float *__restrict__ P, int64_t ldp, const double *__restrict__ w,
    const double *__restrict__ props, int64_t R, int H,
    double *__restrict__ partial, int64_t ldpart, const mxm_em_state *__restrict__ state) {
    constexpr int NW = THREADS / 64;
    __shared__ double red[2][NW];
    if (state != nullptr && state->done != 0) return;
    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int ncol4 = (H + 3) >> 2;                 // float4 groups per row (pad columns are 0 in P)

    double p[NCH][4], acc[NCH][4];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * (t + k * THREADS) + e;
            p[k][e] = (c < H) ? props[c] : 0.0;
            acc[k][e] = 0.0;
        }
    }
    const row_deal deal(R);

    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const int row_bytes = (int)(ldp * 4);
    const int voff = t * 16;
    int last_c4 = t + (NCH - 1) * THREADS;
    if (last_c4 > ncol4 - 1) last_c4 = ncol4 - 1;
    const int voff_last = last_c4 * 16;

    f4 x[NBUF][NCH];
    auto load_row = [&](f4(&xr)[NCH], int64_t q) {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P + deal.row(q) * ldp), 0,
                                                            row_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < NCH - 1; ++k)
            xr[k] = __builtin_bit_cast(f4, (u4)__builtin_amdgcn_raw_buffer_load_b128(
                                               rsrc, voff, k * THREADS * 16, 2));
        xr[NCH - 1] = __builtin_bit_cast(f4, (u4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_last, 0, 2));
    };

    int buf = 0;
    auto process = [&](f4(&xr)[NCH], int64_t q) {
        double wr = deal.live(q) ? (w != nullptr ? w[deal.row(q)] : 1.0) : 0.0;   // asked for before the barrier
        asm volatile("" : "+s"(wr));
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s = fma((double)xr[k][e], p[k][e], s);
        }
        s = wave_sum_lane63(s);
        if (lane == 63) red[buf][wv] = s;
        __syncthreads();
        double cs[1];
        group_ratio_to_sgpr<NW, 1>(&red[buf][0], lane, wr, cs);
        const double c = cs[0];
        buf ^= 1;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // re-convert from the float (1 v_cvt) instead of keeping the doubles of the dot
                // product alive across the barrier (2 VGPRs each): opaque to CSE on purpose
                float xf = xr[k][e];
                asm volatile("" : "+v"(xf));
                acc[k][e] = fma(c, (double)xf, acc[k][e]);
            }
        }
    };

#pragma unroll
    for (int j = 0; j < NBUF - 1; ++j) load_row(x[j], j);
    for (int64_t q = 0; q < deal.nq; q += NBUF) {
#pragma unroll
        for (int j = 0; j < NBUF; ++j) {
            load_row(x[(j + NBUF - 1) % NBUF], q + j + NBUF - 1);
            process(x[j], q + j);
        }
    }
    double *dst = partial + (int64_t)blockIdx.x * ldpart;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = 4 * (t + k * THREADS);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < (int)ldpart) dst[c + e] = acc[k][e];
    }
}

__global__ __launch_bounds__(ROW_THREADS) void linearize_f32_kernel(const double *__restrict__ M, int64_t ldm,
                                                                    int64_t R, int H, float *__restrict__ P,
                                                                    int64_t ldp, double *__restrict__ rowmax) {
    __shared__ double scratch[ROW_THREADS / 64];
    const int t = threadIdx.x;
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
        const double *src = M + r * ldm;
        double m = -INFINITY;
        for (int h = t; h < H; h += ROW_THREADS) m = fmax(m, src[h]);
        m = block_reduce<ROW_THREADS, true>(m, scratch);
        const double shift = isfinite(m) ? m : 0.0;
        float *dst = P + r * ldp;
        for (int h = t; h < (int)ldp; h += ROW_THREADS) dst[h] = (h < H) ? (float)exp(src[h] - shift) : 0.0f;
        if (t == 0) rowmax[r] = shift;
    }
}

// ------------------------------------------------------------------------------------------
// K4  colreduce: colsum[h] = scale_h * sum_{g < nwg} partial[g][h]   (fixed order)
// 64 columns per workgroup; 4 waves take interleaved quarters of the partial rows.
// scale_h = props[h] where the caller wants the M-step sums themselves (em_step), 1 in the loop
// (the loop's finalize works from the unscaled sums, see finalize_kernel).
// ------------------------------------------------------------------------------------------
#define COLRED_THREADS 1024
// What the records' row pass leaves beside its partial rows (coded_kernels.hpp, CHECK): {wide rows met, list fault} per
// workgroup.  The list of wide rows is exactly the set of rows with more than 256 values iff no fault was seen and the
// counts add up to n_wide; otherwise rows were skipped or read twice and the sums are NOT the matrix's: poisoned.
struct wide_check {
    const int *chk;                   // nullptr: nothing to check (dense matrices, votes)
    int n_chk;
    long long n_wide;
};
__device__ __forceinline__ bool wide_check_failed(const wide_check &wc) {
    if (wc.chk == nullptr) return false;                    // uniform
    __shared__ int s_v[2];
    if (threadIdx.x == 0) s_v[0] = s_v[1] = 0;
    __syncthreads();
    int c = 0, bad = 0;
    for (int i = threadIdx.x; i < wc.n_chk; i += blockDim.x) {
        c += wc.chk[2 * i];
        bad |= wc.chk[2 * i + 1];
    }
    if (c) atomicAdd(&s_v[0], c);
    if (bad) s_v[1] = 1;
    __syncthreads();
    return s_v[1] != 0 || (long long)s_v[0] != wc.n_wide;
}

__global__ __launch_bounds__(COLRED_THREADS) void colreduce_kernel(const double *__restrict__ partial,
                                                                   int64_t ldpart, int nwg, int nb, int H,
                                                                   const double *__restrict__ props,
                                                                   double *__restrict__ colsum,
                                                                   mxm_em_state *__restrict__ state,
                                                                   mxm_slots slots, wide_check wc) {
    // grid = (ceil(H/64), nb); partial is [nwg][nb][ldpart] (tile-local b); props / colsum / state are
    // the loop vectors' bases, indexed by the restart slots.s[b].
    // 16 waves take interleaved sixteenths of the partial rows, four independent chains each
    // (the loads are what this kernel waits for); every order below is fixed -> deterministic.
    constexpr int NW = COLRED_THREADS / 64;
    __shared__ double part[NW][64];
    const int b = blockIdx.y;
    const int run = slots.s[b];
    if (state != nullptr && state[run].done != 0) return;
    const bool poisoned = wide_check_failed(wc);
    if (poisoned && state != nullptr && blockIdx.x == 0 && threadIdx.x == 0) state[run].error = 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int h = blockIdx.x * 64 + lane;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (h < H) {
        const double *src = partial + (int64_t)b * ldpart + h;
        const int64_t step = (int64_t)nb * ldpart;
        int g = wv;
        for (; g + 3 * NW < nwg; g += 4 * NW) {
            s0 += src[(int64_t)g * step];
            s1 += src[(int64_t)(g + NW) * step];
            s2 += src[(int64_t)(g + 2 * NW) * step];
            s3 += src[(int64_t)(g + 3 * NW) * step];
        }
        for (; g < nwg; g += NW) s0 += src[(int64_t)g * step];
    }
    part[wv][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wv == 0 && h < H) {
        double tot = part[0][lane];
#pragma unroll
        for (int q = 1; q < NW; ++q) tot += part[q][lane];
        if (poisoned) tot = __builtin_nan("");
        colsum[(int64_t)run * H + h] = (props != nullptr) ? props[(int64_t)run * H + h] * tot : tot;
    }
}

// ------------------------------------------------------------------------------------------
// K5  finalize: em.py:89 (normalise), :39-54 (L1 test), :133-143 (loop state).
//
// The state of the loop is the LOG proportions, as in the reference (em.py:123-124, :140):
//   colsum_h = T_h = sum_r (w_r / Z_r) P_rh      (no p_h factor: representable however small p_h is)
//   ln p'_h  = ln p_h + ln T_h - ln sum_h p_h T_h        == em.py:87-89
//   l1       = sum_h |exp(ln p'_h) - exp(ln p_h)|        == em.py:53-54
// so a proportion that underflows in linear space keeps a finite log, exactly like the reference's;
// props_cur = exp(ln_cur) is what the streaming kernel multiplies with.
//
// Round 4: rounds 1-3 ran this as ONE workgroup per restart -- 5408 software fp64 logarithms and as many
// exponentials on one CU, 12 us of an iteration whose streaming kernel takes 14 us at 10^4 rows.  Now
//   * every workgroup of the grid does the part that needs nothing from the others: u_h = ln p_h + ln T_h for its own
//     columns (the logarithms, spread over the chip), published through ln_new (write-through);
//   * the workgroup that ARRIVES LAST (agent-scope ticket in the restart's state: nobody spins, nothing can deadlock)
//     does what needs all columns: tot = sum p T, the exponentials, the L1 test, the decision and the writes that
//     depend on it.
// Every sum has ONE fixed order whatever the grid: 64-column blocks, xor butterfly inside a block, the blocks' sums
// added in block order -- so the stand-alone kernel (behind a multi-GPU all-reduce) and the form fused behind the column
// reduce give the same bits.
// ------------------------------------------------------------------------------------------
#define FIN_THREADS 1024              // the stand-alone kernel: one thread per column in phase 1, 16 waves in the tail
#define FIN_MAX_BLOCKS 160            // 64-column blocks a restart may have (H <= 10240; the log-space kernel's LDS stops at 9600)
#define FIN_KEEP 6                    // blocks per wave whose new values stay in registers (H <= 6144 with 16 waves)

typedef unsigned int fin_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double fin_load_sc1(const double *p) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p), 0, 8, 0x00020000);
    const fin_u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, 0, 0, 16 /* sc1 */);
    return __hiloint2double((int)v.y, (int)v.x);
}

// Fixed-order sum of n <= FIN_MAX_BLOCKS block sums held in LDS, by wave 0 (result broadcast through s_out): lane l adds
// v[l] + v[l + 64] + v[l + 128] in that order, then the DPP ladder of wave_sum_lane63.  Contains barriers; uniform.
__device__ __forceinline__ double fin_sum_blocks(const double *s_v, int n, double *s_out) {
    static_assert(FIN_MAX_BLOCKS <= 192, "three values per lane");
    __syncthreads();                                        // the block sums are in place
    if (threadIdx.x < 64) {
        const int l = threadIdx.x;
        double v = (l < n) ? s_v[l] : 0.0;
        if (l + 64 < n) v += s_v[l + 64];
        if (l + 128 < n) v += s_v[l + 128];
        v = wave_sum_lane63(v);
        if (l == 63) *s_out = v;
    }
    __syncthreads();
    return *s_out;
}

// The last arriver's part.  THREADS / 64 waves; wave w takes the blocks w, w + NW, ...  `u` = ln_new holds u_h.
// cs / u were written by other workgroups of this launch (write-through, drained before their ticket): sc1 loads.
// All of a wave's loads are issued up front (a first version loaded block by block inside the reduction loop and paid
// the fabric's latency six times over: 19 us for the fused kernel), the block sums are DPP ladders (no LDS permutes).
template <int THREADS>
__device__ __forceinline__ void finalize_tail(const double *__restrict__ cs, double *__restrict__ lc, double *__restrict__ ln,
                                              double *__restrict__ pc, int H, double tol, int max_iter,
                                              mxm_em_state *__restrict__ st) {
    constexpr int NW = THREADS / 64;
    __shared__ double s_blk[FIN_MAX_BLOCKS], s_blk2[FIN_MAX_BLOCKS];
    __shared__ double s_bc[2];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int nblk = (H + 63) >> 6;
    const auto cs_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(cs), 0, H * 8, 0x00020000);
    const auto u_rs = __builtin_amdgcn_make_buffer_rsrc(ln, 0, H * 8, 0x00020000);
    const auto pc_rs = __builtin_amdgcn_make_buffer_rsrc(pc, 0, H * 8, 0x00020000);
    auto ld = [&](decltype(cs_rs) rs, int h, auto AUX) -> double {           // past the vector: 0
        const fin_u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, h * 8, 0, decltype(AUX)::value);
        return __hiloint2double((int)v.y, (int)v.x);
    };
    using SC1 = std::integral_constant<int, 16>;                             // bypasses L1 (other workgroups' stores)
    using PLAIN = std::integral_constant<int, 0>;
    // ---- this wave's first FIN_KEEP blocks: every load in flight at once -------------------------
    double cv[FIN_KEEP], uv[FIN_KEEP], pv[FIN_KEEP];
#pragma unroll
    for (int q = 0; q < FIN_KEEP; ++q) {
        const int h = (wv + q * NW) * 64 + lane;            // (blocks past the vector read zeros)
        cv[q] = ld(cs_rs, h, SC1{});
        uv[q] = ld(u_rs, h, SC1{});
        pv[q] = ld(pc_rs, h, PLAIN{});
    }
    // ---- tot = sum_h p_h T_h --------------------------------------------------------------------
#pragma unroll
    for (int q = 0; q < FIN_KEEP; ++q) {
        const int blk = wv + q * NW;
        const double sum = wave_sum_lane63(pv[q] * cv[q]);
        if (blk < nblk && lane == 63) s_blk[blk] = sum;
    }
    for (int blk = wv + FIN_KEEP * NW; blk < nblk; blk += NW) {               // wider than the register window
        const int h = blk * 64 + lane;
        const double sum = wave_sum_lane63(ld(pc_rs, h, PLAIN{}) * ld(cs_rs, h, SC1{}));
        if (lane == 63) s_blk[blk] = sum;
    }
    const double ltot = log(fin_sum_blocks(s_blk, nblk, &s_bc[0]));
    // ---- ln p' = u - ln tot; the exponential of it is the L1 test's term AND the next pass's proportion ----
    double ek[FIN_KEEP];
#pragma unroll
    for (int q = 0; q < FIN_KEEP; ++q) {
        const int blk = wv + q * NW;
        const int h = blk * 64 + lane;
        uv[q] = uv[q] - ltot;
        ek[q] = exp(uv[q]);
        const double sum = wave_sum_lane63((h < H) ? fabs(ek[q] - pv[q]) : 0.0);
        if (blk < nblk && lane == 63) s_blk2[blk] = sum;
    }
    for (int blk = wv + FIN_KEEP * NW; blk < nblk; blk += NW) {
        const int h = blk * 64 + lane;
        const double e = exp(ld(u_rs, h, SC1{}) - ltot);
        const double sum = wave_sum_lane63((h < H) ? fabs(e - ld(pc_rs, h, PLAIN{})) : 0.0);
        if (lane == 63) s_blk2[blk] = sum;
    }
    const double l1 = fin_sum_blocks(s_blk2, nblk, &s_bc[1]);
    const int iters = st->iters + 1;
    const bool conv = l1 < tol;
    const bool stop = conv || iters >= max_iter;
#pragma unroll
    for (int q = 0; q < FIN_KEEP; ++q) {
        const int h = (wv + q * NW) * 64 + lane;
        if (h < H) {
            ln[h] = uv[q];
            if (!stop) {
                lc[h] = uv[q];
                pc[h] = ek[q];
            }
        }
    }
    for (int blk = wv + FIN_KEEP * NW; blk < nblk; blk += NW) {               // formed again, same bits
        const int h = blk * 64 + lane;
        if (h < H) {
            const double v = ld(u_rs, h, SC1{}) - ltot;
            ln[h] = v;
            if (!stop) {
                lc[h] = v;
                pc[h] = exp(v);
            }
        }
    }
    // The columns beyond the register window are formed twice from u = ln[h] (once for the L1 sum, once here).  That is
    // race-free without a barrier in between: column h is read AND rewritten by the one thread that owns it (h = blk * 64
    // + lane, blk = wv + k * NW) -- no other thread of this workgroup touches ln[h], and the other workgroups finished
    // with ln before their ticket (finalize_arrive), which this workgroup saw as the last arriver.  The barrier below
    // only orders the state write after every wave's vector writes.
    __syncthreads();
    if (t == 0) {
        st->iters = iters;
        st->l1 = l1;
        st->done = conv ? 1 : (stop ? 2 : 0);
        __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next call
    }
}

// Phase 1 of a restart's finalize for the columns [h0, h0 + n): u_h = ln p_h + ln T_h -> ln_new (write-through), then the
// ticket.  Returns true in the workgroup whose ticket came last (uniform).  T_h is passed in by the caller's lanes.
template <int THREADS>
__device__ __forceinline__ bool finalize_arrive(mxm_em_state *__restrict__ st, int nwg_expected) {
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's write-through stores have left
    __syncthreads();                                        // ... and every other wave's of the workgroup
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (old + 1u == (unsigned)nwg_expected) ? 1 : 0;
    }
    __syncthreads();
    return s_last != 0;
}

__global__ __launch_bounds__(FIN_THREADS) void finalize_kernel(const double *__restrict__ colsum,
                                                               double *__restrict__ ln_cur,
                                                               double *__restrict__ ln_new,
                                                               double *__restrict__ props_cur, int H,
                                                               double tol, int max_iter,
                                                               mxm_em_state *__restrict__ state, int base,
                                                               int use_slots, mxm_slots slots) {
    // grid = (ceil(H / FIN_THREADS), restarts)
    const int b = use_slots ? slots.s[blockIdx.y] : base + (int)blockIdx.y;       // which restart
    mxm_em_state *st = state + b;
    if (st->done != 0) return;
    const double *cs = colsum + (int64_t)b * H;
    double *lc = ln_cur + (int64_t)b * H;
    double *ln = ln_new + (int64_t)b * H;
    double *pc = props_cur + (int64_t)b * H;
    const int h = (int)blockIdx.x * FIN_THREADS + (int)threadIdx.x;
    if (h < H) {
        const double u = lc[h] + log(cs[h]);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(ln, 0, H * 8, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fin_u2, u), rs, h * 8, 0, 16 /* sc1 */);
    }
    if (!finalize_arrive<FIN_THREADS>(st, (int)gridDim.x)) return;
    finalize_tail<FIN_THREADS>(cs, lc, ln, pc, H, tol, max_iter, st);
}

// K5g  finalize for ANY width (round 5; finalize_kernel's block bookkeeping stops at 64 * FIN_MAX_BLOCKS columns): one
// workgroup per restart, every thread its own columns t, t + 1024, ..., the two sums through block_reduce -- ONE fixed
// order whatever the caller, so every rank of a sharded run still takes the same decision.  Same update, same state.
__global__ __launch_bounds__(FIN_THREADS) void finalize_any_kernel(const double *__restrict__ colsum, double *__restrict__ ln_cur,
                                                                   double *__restrict__ ln_new, double *__restrict__ props_cur,
                                                                   int H, double tol, int max_iter, mxm_em_state *__restrict__ state,
                                                                   int base, int use_slots, mxm_slots slots) {
    __shared__ double scratch[FIN_THREADS / 64];
    const int b = use_slots ? slots.s[blockIdx.y] : base + (int)blockIdx.y;
    mxm_em_state *st = state + b;
    if (st->done != 0) return;
    const double *cs = colsum + (int64_t)b * H;
    double *lc = ln_cur + (int64_t)b * H, *ln = ln_new + (int64_t)b * H, *pc = props_cur + (int64_t)b * H;
    const int t = threadIdx.x;
    double s = 0.0;
    for (int h = t; h < H; h += FIN_THREADS) s = fma(pc[h], cs[h], s);
    const double ltot = log(block_reduce<FIN_THREADS, false>(s, scratch));
    double d = 0.0;
    for (int h = t; h < H; h += FIN_THREADS) {
        const double v = lc[h] + log(cs[h]) - ltot;       // em.py:87-89
        ln[h] = v;
        d += fabs(exp(v) - pc[h]);                         // em.py:53-54
    }
    const double l1 = block_reduce<FIN_THREADS, false>(d, scratch);
    const int iters = st->iters + 1;
    const bool conv = l1 < tol;
    const bool stop = conv || iters >= max_iter;
    if (!stop)
        for (int h = t; h < H; h += FIN_THREADS) {         // (each thread rewrites the columns it wrote itself)
            const double v = ln[h];
            lc[h] = v;
            pc[h] = exp(v);
        }
    __syncthreads();
    if (t == 0) {
        st->iters = iters;
        st->l1 = l1;
        st->done = conv ? 1 : (stop ? 2 : 0);
    }
}

// ------------------------------------------------------------------------------------------
// K4+K5  column reduce with the finalize behind it in ONE launch (the single-GPU per-iteration path, where no
// collective sits between the two): colreduce_kernel's sums, bit for bit, then phase 1 of the finalize for the block's
// 64 columns by the wave that holds them, ticket, and the last workgroup's tail.  grid = (ceil(H/64), nb).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(COLRED_THREADS) void colreduce_finalize_kernel(
    const double *__restrict__ partial, int64_t ldpart, int nwg, int nb, int H, double *__restrict__ colsum,
    double *__restrict__ ln_cur, double *__restrict__ ln_new, double *__restrict__ props_cur, double tol, int max_iter,
    mxm_em_state *__restrict__ state, mxm_slots slots, wide_check wc) {
    constexpr int NW = COLRED_THREADS / 64;
    __shared__ double part[NW][64];
    const int b = blockIdx.y;
    const int run = slots.s[b];
    mxm_em_state *st = state + run;
    if (st->done != 0) return;
    const bool poisoned = wide_check_failed(wc);
    if (poisoned && blockIdx.x == 0 && threadIdx.x == 0) st->error = 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int h = blockIdx.x * 64 + lane;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (h < H) {
        const double *src = partial + (int64_t)b * ldpart + h;
        const int64_t step = (int64_t)nb * ldpart;
        int g = wv;
        for (; g + 3 * NW < nwg; g += 4 * NW) {
            s0 += src[(int64_t)g * step];
            s1 += src[(int64_t)(g + NW) * step];
            s2 += src[(int64_t)(g + 2 * NW) * step];
            s3 += src[(int64_t)(g + 3 * NW) * step];
        }
        for (; g < nwg; g += NW) s0 += src[(int64_t)g * step];
    }
    part[wv][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    double *cs = colsum + (int64_t)run * H;
    double *lc = ln_cur + (int64_t)run * H;
    double *ln = ln_new + (int64_t)run * H;
    double *pc = props_cur + (int64_t)run * H;
    if (wv == 0 && h < H) {
        double tot = part[0][lane];
#pragma unroll
        for (int q = 1; q < NW; ++q) tot += part[q][lane];
        if (poisoned) tot = __builtin_nan("");
        const auto cs_rs = __builtin_amdgcn_make_buffer_rsrc(cs, 0, H * 8, 0x00020000);
        const auto ln_rs = __builtin_amdgcn_make_buffer_rsrc(ln, 0, H * 8, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fin_u2, tot), cs_rs, h * 8, 0, 16 /* sc1 */);
        const double u = lc[h] + log(tot);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fin_u2, u), ln_rs, h * 8, 0, 16 /* sc1 */);
    }
    if (!finalize_arrive<COLRED_THREADS>(st, (int)gridDim.x)) return;
    finalize_tail<COLRED_THREADS>(cs, lc, ln, pc, H, tol, max_iter, st);
}

#endif  // MIXEMT_EM_KERNELS_HPP
