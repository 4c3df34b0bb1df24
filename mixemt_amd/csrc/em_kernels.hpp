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
__global__ __launch_bounds__(COLRED_THREADS) void colreduce_kernel(const double *__restrict__ partial,
                                                                   int64_t ldpart, int nwg, int nb, int H,
                                                                   const double *__restrict__ props,
                                                                   double *__restrict__ colsum,
                                                                   const mxm_em_state *__restrict__ state,
                                                                   mxm_slots slots) {
    // grid = (ceil(H/64), nb); partial is [nwg][nb][ldpart] (tile-local b); props / colsum / state are
    // the loop vectors' bases, indexed by the restart slots.s[b].
    // 16 waves take interleaved sixteenths of the partial rows, four independent chains each
    // (the loads are what this kernel waits for); every order below is fixed -> deterministic.
    constexpr int NW = COLRED_THREADS / 64;
    __shared__ double part[NW][64];
    const int b = blockIdx.y;
    const int run = slots.s[b];
    if (state != nullptr && state[run].done != 0) return;
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
        colsum[(int64_t)run * H + h] = (props != nullptr) ? props[(int64_t)run * H + h] * tot : tot;
    }
}

// ------------------------------------------------------------------------------------------
// K5  finalize: em.py:89 (normalise), :39-54 (L1 test), :133-143 (loop state). One WG per restart.
// ------------------------------------------------------------------------------------------
#define FIN_THREADS 1024

__global__ __launch_bounds__(FIN_THREADS) void finalize_kernel(const double *__restrict__ colsum,
                                                               double *__restrict__ ln_cur,
                                                               double *__restrict__ ln_new,
                                                               double *__restrict__ props_cur, int H,
                                                               double tol, int max_iter,
                                                               mxm_em_state *__restrict__ state, int base,
                                                               int use_slots, mxm_slots slots) {
    // The state of the loop is the LOG proportions, as in the reference (em.py:123-124, :140):
    //   colsum_h = T_h = sum_r (w_r / Z_r) P_rh      (no p_h factor: representable however small p_h is)
    //   ln p'_h  = ln p_h + ln T_h - ln sum_h p_h T_h        == em.py:87-89
    //   l1       = sum_h |exp(ln p'_h) - exp(ln p_h)|        == em.py:53-54
    // so a proportion that underflows in linear space keeps a finite log, exactly like the
    // reference's; props_cur = exp(ln_cur) is what the streaming kernel multiplies with.
    __shared__ double scratch[FIN_THREADS / 64];
    const int b = use_slots ? slots.s[blockIdx.x] : base + (int)blockIdx.x;       // which restart
    mxm_em_state *st = state + b;
    if (st->done != 0) return;
    const double *cs = colsum + (int64_t)b * H;
    double *lc = ln_cur + (int64_t)b * H;
    double *ln = ln_new + (int64_t)b * H;
    double *pc = props_cur + (int64_t)b * H;
    const int t = threadIdx.x;
    double s = 0.0;
    for (int h = t; h < H; h += FIN_THREADS) s += pc[h] * cs[h];
    const double ltot = log(block_reduce<FIN_THREADS, false>(s, scratch));
    double l1 = 0.0;
    constexpr int FIN_KEEP = 8;                     // columns per thread whose new value stays in registers (8192 columns)
    double vk[FIN_KEEP], ek[FIN_KEEP];
#pragma unroll
    for (int q = 0; q < FIN_KEEP; ++q) {
        const int h = t + q * FIN_THREADS;
        vk[q] = 0.0;
        ek[q] = 0.0;
        if (h < H) {
            vk[q] = lc[h] + log(cs[h]) - ltot;
            ek[q] = exp(vk[q]);
            ln[h] = vk[q];
            l1 += fabs(ek[q] - pc[h]);
        }
    }
    for (int h = t + FIN_KEEP * FIN_THREADS; h < H; h += FIN_THREADS) {       // wider than 8192 (log-space paths only)
        const double v = lc[h] + log(cs[h]) - ltot;
        ln[h] = v;
        l1 += fabs(exp(v) - pc[h]);
    }
    l1 = block_reduce<FIN_THREADS, false>(l1, scratch);
    const int iters = st->iters + 1;
    const bool conv = l1 < tol;
    const bool stop = conv || iters >= max_iter;
    if (!stop) {
        // the exponential computed for the L1 test IS the next pass's proportion: same bits, one exp less
#pragma unroll
        for (int q = 0; q < FIN_KEEP; ++q) {
            const int h = t + q * FIN_THREADS;
            if (h < H) {
                lc[h] = vk[q];
                pc[h] = ek[q];
            }
        }
        for (int h = t + FIN_KEEP * FIN_THREADS; h < H; h += FIN_THREADS) {
            const double v = ln[h];
            lc[h] = v;
            pc[h] = exp(v);
        }
    }
    __syncthreads();
    if (t == 0) {
        st->iters = iters;
        st->l1 = l1;
        st->done = conv ? 1 : (stop ? 2 : 0);
    }
}

#endif  // MIXEMT_EM_KERNELS_HPP
