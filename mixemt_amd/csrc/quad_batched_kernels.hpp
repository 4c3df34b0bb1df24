// quad_batched_kernels.hpp -- part of libmixemt_hip.so (gfx950); included by mixemt_hip.hip only.
// Several restarts through ONE pass over the records beside a quad dictionary (round 6; BASELINE configs 3 and 5,
// reference: em.py:117-161 runs its restarts one after another over the same matrix).
#ifndef MIXEMT_QUAD_BATCHED_KERNELS_HPP
#define MIXEMT_QUAD_BATCHED_KERNELS_HPP

// ------------------------------------------------------------------------------------------
// Why / what.  Ten restarts over records read them ten times (em_iter_quad_coded_kernel: one restart per pass).  A row's
// loads, its table in LDS and the lookups do not depend on the restart; only the two FMAs per cell, the wave sums and the
// division do.  Here a workgroup of 512 threads takes a row (thread t' owns the quads tl + 256 (2 k + half), tl = t' & 255,
// half = t' >> 8, k < NCH: the odd or the even code bytes of the 8-byte word thread tl of the one-restart pass loads, so
// the quad records serve unchanged) and keeps the proportions and column sums of BT restarts in registers (BT x NCH x 16
// VGPRs each way): a row's codes, table and lookups are fetched once, the dot products / wave sums / updates run BT times.
// Sized as an experiment first (tools/experiments/quad_experiment.hip, profiles/r05/quad_batched_1m.txt): BT = 3 with two
// rows in flight fills the 256 registers a wave of a 512-thread workgroup may have; BT = 4 spills and loses everything.
//
// The same step takes ALL coded rows, class after class: the rows with quads (quad records), then
// the byte-coded rows without (the records' byte codes: one dword = the same four columns a quad names), then the wide
// rows (16-bit codes) -- the one-restart kernel runs the last two as a pass of their own on a share of the grid, which
// here would idle the 512-thread workgroups' second half.  Per-thread column ownership is the same in all three classes
// (word / quad index W = tl + 256 (2 k + half) <-> columns 4 W .. 4 W + 3), so the registers carry over.
// The three row lists are checked where they are used, as in the one-restart kernels: an entry out of range, not above
// its predecessor or not of its class is never dereferenced and raises `fault`, which the column reduce turns into NaN
// sums + mxm_em_state.error (wide_check); that the lists add up to all coded rows is coded_check's business (host).
// Partial rows: partial[(workgroup * BT + b)][column] -- colreduce_kernel's layout for a tile of BT restarts.
// ------------------------------------------------------------------------------------------
#define QB_THREADS 512
#define QB_NW (QB_THREADS / 64)
#define QB_TBL_DOUBLES 1024               // 8 KB: 256 quads x 32 B, or the 1024 values of a wide row

template <int BT, int NCH, int NBUF, int CLS>
__device__ __forceinline__ void quadb_class_pass(const uint8_t *__restrict__ rbase,        // qrec (CLS 0) or rec (CLS 1, 2)
                                                 const int64_t *__restrict__ roff,        // qoff or rec_off, by row
                                                 const int32_t *__restrict__ ndist, const int32_t *__restrict__ nquad,
                                                 const int64_t *__restrict__ list, int64_t n_list, int64_t R, int ldc,
                                                 const double *__restrict__ w, const double (&p)[BT][NCH][4],
                                                 double (&acc)[BT][NCH][4], double (*s_tbl)[QB_TBL_DOUBLES],
                                                 double (*red)[BT * QB_NW], long long (*s_off)[QB_THREADS],
                                                 double (*s_wr)[QB_THREADS], int (*s_nd)[QB_THREADS], int *s_fault) {
    static_assert(NBUF == 2 || NBUF == 3, "row q + 2's record is asked for in step q; NBUF slots of codes and table entries");
    static_assert(CLS >= 0 && CLS <= 2, "0 quad rows, 1 byte-coded rows without quads, 2 wide rows");
    constexpr int THREADS = QB_THREADS, NW = QB_NW, AUX = 2;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, half = t >> 8, tl = t & 255;
    const row_deal deal(n_list);
    if (n_list <= 0 || deal.nq <= 0) return;             // (workgroup uniform)

    auto fetch_meta = [&](int hf, int64_t q0) {          // steps q0 .. q0 + THREADS - 1, thread t takes step q0 + t
        const int64_t q = q0 + t;
        const int64_t e = deal.row(q);                   // list index
        int64_t r = list[e];
        int nd = 0;
        if (r < 0 || r >= R || (e > 0 && list[e - 1] >= r)) {
            *s_fault = 1;
            r = -1;
        } else if constexpr (CLS == 0) {
            nd = nquad[r];
            if (nd <= 0 || nd > QUAD_MAX || ndist[r] < 1 || ndist[r] > ENC_MAX_CODES) {
                *s_fault = 1;
                nd = 0;
            }
        } else if constexpr (CLS == 1) {
            nd = ndist[r];
            if (nd <= 0 || nd > ENC_MAX_CODES || nquad[r] != 0) {      // (with quads the first class takes it too)
                *s_fault = 1;
                nd = 0;
            }
        } else {
            nd = ndist[r];
            if (nd <= ENC_MAX_CODES || nd > ENC_MAX_WIDE) {
                *s_fault = 1;
                nd = 0;
            }
        }
        s_off[hf][t] = (nd > 0) ? roff[r] : 0;
        s_nd[hf][t] = nd;
        s_wr[hf][t] = (deal.live(q) && nd > 0) ? (w != nullptr ? w[r] : 1.0) : 0.0;
    };
    // a row in flight: its code words and this thread's share of its table (a row with nd = 0 -- a faulty entry -- reads
    // nothing: descriptors of 0 bytes return zeros)
    constexpr int CW = CLS == 0 ? 2 : (CLS == 1 ? NCH : 2 * NCH);      // dwords of codes per thread and row
    unsigned int cw[NBUF][CW];
    quad_d2 tring[NBUF];
    int pre_off_lo, pre_off_hi, pre_nd;                  // uniform (SGPRs)
    double pre_wr;
    auto read_meta = [&](int64_t q_load, int64_t q_weight) {
        const int hf = (int)((q_load / THREADS) & 1), idx = (int)(q_load % THREADS);
        const long long off = s_off[hf][idx];
        pre_nd = __builtin_amdgcn_readfirstlane(s_nd[hf][idx]);
        pre_off_hi = __builtin_amdgcn_readfirstlane((int)(off >> 32));
        pre_off_lo = __builtin_amdgcn_readfirstlane((int)off);
        pre_wr = s_wr[(q_weight / THREADS) & 1][q_weight % THREADS];
    };
    auto load_row = [&](auto SLOT) {
        constexpr int slot = decltype(SLOT)::value;
        const int nd = pre_nd;
        const uint8_t *base = rbase + (((long long)pre_off_hi << 32) | (unsigned int)pre_off_lo);
        if constexpr (CLS == 0) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, nd > 0 ? QUAD_CODE_BYTES : 0, 0x00020000);
            const quad_u2 c = __builtin_amdgcn_raw_buffer_load_b64(rs, tl * 8, 0, AUX);
            cw[slot][0] = c.x;
            cw[slot][1] = c.y;
            const auto rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base + QUAD_CODE_BYTES), 0, nd * 32, 0x00020000);
            const quad_u4 a = __builtin_amdgcn_raw_buffer_load_b128(rt, t * 16, 0, AUX);
            tring[slot] = quad_d2{__hiloint2double((int)a.y, (int)a.x), __hiloint2double((int)a.w, (int)a.z)};
        } else if constexpr (CLS == 1) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, nd > 0 ? ldc : 0, 0x00020000);
#pragma unroll
            for (int k = 0; k < NCH; ++k) cw[slot][k] = __builtin_amdgcn_raw_buffer_load_b32(rs, (tl + 256 * half) * 4, k * 2048, AUX);
            const auto rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base + ldc), 0, nd * 8, 0x00020000);
            const quad_u2 a = __builtin_amdgcn_raw_buffer_load_b64(rt, t * 8, 0, AUX);     // (threads past the table: zeros)
            tring[slot] = quad_d2{__hiloint2double((int)a.y, (int)a.x), 0.0};
        } else {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, nd > 0 ? 2 * ldc : 0, 0x00020000);
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const quad_u2 c = __builtin_amdgcn_raw_buffer_load_b64(rs, (tl + 256 * half) * 8, k * 4096, AUX);
                cw[slot][2 * k] = c.x;
                cw[slot][2 * k + 1] = c.y;
            }
            const auto rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base + 2 * ldc), 0, nd * 8, 0x00020000);
            const quad_u2 a = __builtin_amdgcn_raw_buffer_load_b64(rt, t * 8, 0, AUX), b = __builtin_amdgcn_raw_buffer_load_b64(rt, t * 8, THREADS * 8, AUX);
            tring[slot] = quad_d2{__hiloint2double((int)a.y, (int)a.x), __hiloint2double((int)b.y, (int)b.x)};
        }
    };
    auto publish = [&](auto SLOT) {
        constexpr int slot = decltype(SLOT)::value;
        if constexpr (CLS == 0) {
            // the table in LDS: first halves of the entries, then second halves (quad_kernels.hpp, QUAD_SPLIT_TABLE); this
            // thread holds half t & 1 of entry t >> 1
            *reinterpret_cast<quad_d2 *>(&s_tbl[slot][(t & 1) * (QUAD_MAX * 2) + (t >> 1) * 2]) = tring[slot];
        } else if constexpr (CLS == 1) {
            if (t < ENC_MAX_CODES) s_tbl[slot][t] = tring[slot].x;
        } else {
            s_tbl[slot][t] = tring[slot].x;
            s_tbl[slot][t + THREADS] = tring[slot].y;
        }
    };
    double v[NCH][4];
    auto lookup_row = [&](const char *tb, const unsigned int(&c)[CW]) {
        if constexpr (CLS == 0) {
            // the thread's code bytes: 0, 2, 4 of the word shifted down by `half` bytes
            const unsigned long long word = (((unsigned long long)c[1] << 32) | c[0]) >> (8 * half);
            const unsigned int lo = (unsigned int)word, hi = (unsigned int)(word >> 32);
            unsigned int off[3] = {quad_byte_x16<0>(lo), quad_byte_x16<2>(lo), quad_byte_x16<0>(hi)};
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const quad_d2 a = *reinterpret_cast<const quad_d2 *>(tb + off[k]), b = *reinterpret_cast<const quad_d2 *>(tb + off[k] + QUAD_MAX * 16);
                v[k][0] = a.x;
                v[k][1] = a.y;
                v[k][2] = b.x;
                v[k][3] = b.y;
            }
        } else if constexpr (CLS == 1) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                v[k][0] = *reinterpret_cast<const double *>(tb + code_byte_x8<0>(c[k]));
                v[k][1] = *reinterpret_cast<const double *>(tb + code_byte_x8<1>(c[k]));
                v[k][2] = *reinterpret_cast<const double *>(tb + code_byte_x8<2>(c[k]));
                v[k][3] = *reinterpret_cast<const double *>(tb + code_byte_x8<3>(c[k]));
            }
        } else {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                v[k][0] = *reinterpret_cast<const double *>(tb + ((c[2 * k] & 0xffffu) << 3));
                v[k][1] = *reinterpret_cast<const double *>(tb + ((c[2 * k] >> 16) << 3));
                v[k][2] = *reinterpret_cast<const double *>(tb + ((c[2 * k + 1] & 0xffffu) << 3));
                v[k][3] = *reinterpret_cast<const double *>(tb + ((c[2 * k + 1] >> 16) << 3));
            }
        }
    };

    auto step = [&](auto J, int64_t q) {
        constexpr int j = decltype(J)::value;
        constexpr int jn = (j + 1) % NBUF, jl = (j + 2) % NBUF;
        if ((q % THREADS) == 0) fetch_meta((int)((q / THREADS + 1) & 1), q + THREADS);   // the block after this one
        load_row(std::integral_constant<int, jl>{});     // row q + 2; its slot held row q - 1 (NBUF = 3) or row q (2): consumed
        const double wr = pre_wr;
        double s[BT];
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            double s2[2] = {0.0, 0.0};
#pragma unroll
            for (int k = 0; k < NCH; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) s2[e & 1] = fma(v[k][e], p[b][k][e], s2[e & 1]);
            s[b] = s2[0] + s2[1];
        }
        __builtin_amdgcn_s_setprio(1);
        if constexpr (BT == 3) {
            // the three sums in one folded ladder (common.hpp): row 0 of the wave ends with restart 0's, row 2 with
            // restart 1's, rows 1 and 3 with restart 2's
            const double z = wave_sum3_by_row(s[0], s[1], s[2]);
            if ((lane & 15) == 0 && lane < 48) red[j][(lane == 0 ? 0 : (lane == 32 ? 1 : 2)) * NW + wv] = z;
        } else {
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                s[b] = wave_sum_lane63(s[b]);
                if (lane == 63) red[j][b * NW + wv] = s[b];
            }
        }
        publish(std::integral_constant<int, jn>{});      // row q + 1's table, published by the same barrier
        __syncthreads();
        read_meta(q + 3, q + 1);                         // for the next step; returns under the division
        double cf[BT];
        group_ratio_to_sgpr<NW, BT>(&red[j][0], lane, wr, cf);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int b = 0; b < BT; ++b)
#pragma unroll
            for (int k = 0; k < NCH; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[b][k][e] = fma(cf[b], v[k][e], acc[b][k][e]);
        lookup_row(reinterpret_cast<const char *>(&s_tbl[jn][0]), cw[jn]);
    };

    __syncthreads();                                     // the class before this one is done with the LDS blocks
    fetch_meta(0, 0);
    __syncthreads();
    read_meta(0, 0);
    load_row(std::integral_constant<int, 0>{});
    read_meta(1, 0);
    load_row(std::integral_constant<int, 1>{});
    publish(std::integral_constant<int, 0>{});
    __syncthreads();
    read_meta(2, 0);
    lookup_row(reinterpret_cast<const char *>(&s_tbl[0][0]), cw[0]);
    for (int64_t q = 0; q < deal.nq; q += NBUF) {
        step(std::integral_constant<int, 0>{}, q);
        step(std::integral_constant<int, 1>{}, q + 1);
        if constexpr (NBUF > 2) step(std::integral_constant<int, 2>{}, q + 2);
    }
}

// K3qB  em_iter_quad_batched_kernel<BT, NCH, CLASSES>: one pass over the rows of the classes named by CLASSES (bit 0 quad
// rows, bit 1 byte-coded rows without quads, bit 2 wide rows) for the BT restarts slots.s[0 .. BT); grid <= one workgroup
// per CU.  Two launches per tile: <.., 1> starts the partial rows, <.., 6> (the SAME grid: same thread <-> column map) takes
// them up as its initial sums and adds the leftover rows -- as ONE kernel the three row loops cost 12-19 spilled registers
// whose reloads wait (s_waitcnt vmcnt) for the rows just asked for: 5.35 ms per pass against 3.35 + 0.17 for the parts
// (profiles/r06/time_quads_batched_*.txt).  {0, list fault} per workgroup behind the partial rows, where the one-restart
// kernels leave theirs (the second launch ORs its fault in).
template <int BT, int NCH, int CLASSES>
__global__ __launch_bounds__(QB_THREADS, 1) void em_iter_quad_batched_kernel(
    const uint8_t *__restrict__ rec, const int64_t *__restrict__ rec_off, const int32_t *__restrict__ ndist, int ldc,
    const int64_t *__restrict__ wide_rows, int64_t n_wide, const int64_t *__restrict__ byte_rows, int64_t n_byte_rows,
    const uint8_t *__restrict__ qrec, const int64_t *__restrict__ qoff, const int32_t *__restrict__ nquad,
    const int64_t *__restrict__ quad_rows, int64_t n_quad_rows, int64_t R, const double *__restrict__ w,
    const double *__restrict__ props, int H, double *__restrict__ partial, int64_t ldpart,
    const mxm_em_state *__restrict__ state, mxm_slots slots) {
    static_assert(BT >= 2 && BT <= MXM_MAX_BT && BT * QB_NW <= 64, "one wave partial per lane in the ratio");
    static_assert(NCH >= 1 && NCH <= 3, "three code bytes per thread: H <= 6144");
    static_assert(CLASSES == 1 || CLASSES == 6, "the quad rows start the sums, the leftover rows are added to them");
    constexpr int NBUF = 3, NBUF_LEFT = 2;               // (the few leftover rows: one slot less)
    constexpr bool FIRST = (CLASSES & 1) != 0;
    __shared__ __attribute__((aligned(16))) double s_tbl[NBUF][QB_TBL_DOUBLES];
    __shared__ __attribute__((aligned(16))) double red[NBUF][BT * QB_NW];
    __shared__ long long s_off[2][QB_THREADS];
    __shared__ double s_wr[2][QB_THREADS];
    __shared__ int s_nd[2][QB_THREADS];
    __shared__ int s_fault;
    if (state != nullptr) {
        bool any = false;
#pragma unroll
        for (int b = 0; b < BT; ++b) any = any || (state[slots.s[b]].done == 0);
        if (!any) return;                                // every restart of this tile has stopped
    }
    const int t = threadIdx.x, half = t >> 8, tl = t & 255;
    if (t == 0) s_fault = 0;
    double p[BT][NCH][4], acc[BT][NCH][4];
#pragma unroll
    for (int b = 0; b < BT; ++b) {
        const double *pb = props + (int64_t)slots.s[b] * H;
        const double *src = partial + ((int64_t)blockIdx.x * BT + b) * ldpart;
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * (tl + 256 * (2 * k + half)) + e;
                p[b][k][e] = (c < H) ? pb[c] : 0.0;
                acc[b][k][e] = (!FIRST && c < H) ? src[c] : 0.0;
            }
    }
    if constexpr ((CLASSES & 1) != 0) quadb_class_pass<BT, NCH, NBUF, 0>(qrec, qoff, ndist, nquad, quad_rows, n_quad_rows, R, ldc, w, p, acc, s_tbl, red, s_off, s_wr, s_nd, &s_fault);
    if constexpr ((CLASSES & 2) != 0) quadb_class_pass<BT, NCH, NBUF_LEFT, 1>(rec, rec_off, ndist, nquad, byte_rows, n_byte_rows, R, ldc, w, p, acc, s_tbl, red, s_off, s_wr, s_nd, &s_fault);
    if constexpr ((CLASSES & 4) != 0) quadb_class_pass<BT, NCH, NBUF_LEFT, 2>(rec, rec_off, ndist, nquad, wide_rows, n_wide, R, ldc, w, p, acc, s_tbl, red, s_off, s_wr, s_nd, &s_fault);
    __syncthreads();
    if (t == 0) {
        int *out = reinterpret_cast<int *>(partial + (int64_t)MXM_MAX_WG * ldpart) + 2 * blockIdx.x;
        if constexpr (FIRST) {
            out[0] = 0;
            out[1] = s_fault;
        } else if (s_fault != 0) {
            out[1] = 1;
        }
    }
#pragma unroll
    for (int b = 0; b < BT; ++b) {
        double *dst = partial + ((int64_t)blockIdx.x * BT + b) * ldpart;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = 4 * (tl + 256 * (2 * k + half));
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < H) dst[c + e] = acc[b][k][e];
        }
    }
}

#endif  // MIXEMT_QUAD_BATCHED_KERNELS_HPP
