// coded3_experiment.hip -- NOT part of libmixemt_hip.so.  Round-5 experiments on the records row pass (VERDICT r4 #3:
// "take the cross-wave chain off every row's critical path"; coded_kernels.hpp, coded_row_pass).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I include -I mixemt_amd/csrc \
//         tools/experiments/coded3_experiment.hip -o tools/experiments/_build/libcoded3.so
// Driver: tools/experiments/time_coded3.py; results and reasoning: profiles/r05/experiments.md.
//
// Part 1 -- bare readers of the records (no arithmetic): what limits the LOADS of this access pattern?  The product
//           kernel's loads alone run at 5.0 TB/s (profiles/r04/records_read_ceiling.txt) against 6.8-6.9 for the dense
//           matrix's; variants: rows in flight, 16-byte loads, cache policy, records taken in address order.
// Part 2 -- the row pass with the accumulation DELAYED by one row: the wave sums of row q are exchanged and its
//           coefficient w_r / Z_r is formed while the wave already multiplies row q + 1, and row q is accumulated one
//           step later -- the ladder -> LDS -> barrier -> division chain no longer sits between a row's two halves.
//           Costs a second set of row values in registers (48); the proportions move to LDS to pay for it.
// Byte-coded rows only (wide rows and rows without a record are skipped: weight 0), as coded2_experiment.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <cmath>
#include <functional>
#include <type_traits>
#include "mixemt_hip.h"
#include "common.hpp"

#define C3_MAX_CODES 256
typedef unsigned int c3_u2 __attribute__((ext_vector_type(2)));
typedef unsigned int c3_u4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------------
// Part 1: bare readers.  WIDE16 = false: the product kernel's loads (4 B per lane, NCH per thread and row);
// WIDE16 = true: 16 B per lane, one load per thread and row (threads past the row idle).  `order` (nullable): the rows
// are taken in that order instead of 0..R-1 (rows sorted by record address).  AUX: 2 = nt, 0 = default policy.
// ------------------------------------------------------------------------------------------------------------------
template <int NCH, int NBUF, bool WIDE16, int AUX, int WG_PER_CU>
__global__ __launch_bounds__(256, WG_PER_CU) void reader_kernel(const uint8_t *__restrict__ rec, const int64_t *__restrict__ rec_off,
                                                                const int32_t *__restrict__ ndist, int ldc, int64_t R,
                                                                const int64_t *__restrict__ order, unsigned int *__restrict__ sink) {
    constexpr int THREADS = 256;
    const int t = threadIdx.x;
    const row_deal deal(R);
    const int nword = ldc >> 2;
    int last = t + (NCH - 1) * THREADS;
    if (last > nword - 1) last = nword - 1;
    __shared__ long long s_off[2][THREADS];
    __shared__ int s_nd[2][THREADS];
    auto fetch_meta = [&](int half, int64_t q0) {
        int64_t r = deal.row(q0 + t);
        if (order != nullptr) r = order[r];
        int nd = ndist[r];
        if (nd > C3_MAX_CODES) nd = 0;
        s_off[half][t] = rec_off[r];
        s_nd[half][t] = nd;
    };
    unsigned int x[NBUF][WIDE16 ? 4 : NCH];
    c3_u2 y[NBUF];
    unsigned int acc = 0;
    auto load_rec = [&](unsigned int(&xr)[WIDE16 ? 4 : NCH], c3_u2 &yr, int64_t q) {
        const int half = (int)((q / THREADS) & 1), idx = (int)(q % THREADS);
        const long long off = s_off[half][idx];
        const int nd = __builtin_amdgcn_readfirstlane(s_nd[half][idx]);
        const uint8_t *base = rec + (((long long)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) |
                                     (unsigned int)__builtin_amdgcn_readfirstlane((int)off));
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, nd > 0 ? ldc : 0, 0x00020000);
        if constexpr (WIDE16) {
            const c3_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, t * 16, 0, AUX);     // past the row: zeros
            const c3_u4 v2 = __builtin_amdgcn_raw_buffer_load_b128(rs, t * 16, 4096, AUX);  // bytes 4096.. (threads 0..81)
            xr[0] = v.x ^ v2.x; xr[1] = v.y ^ v2.y; xr[2] = v.z ^ v2.z; xr[3] = v.w ^ v2.w;
        } else {
#pragma unroll
            for (int k = 0; k < NCH - 1; ++k) xr[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, t * 4, k * THREADS * 4, AUX);
            xr[NCH - 1] = __builtin_amdgcn_raw_buffer_load_b32(rs, last * 4, 0, AUX);
        }
        const auto rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base + ldc), 0, nd * 8, 0x00020000);
        yr = __builtin_amdgcn_raw_buffer_load_b64(rt, t * 8, 0, AUX);
    };
    if (deal.nq <= 0) return;
    fetch_meta(0, 0);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NBUF - 1; ++j) load_rec(x[j], y[j], j);
    for (int64_t q = 0; q < deal.nq; q += NBUF) {
#pragma unroll
        for (int j = 0; j < NBUF; ++j) {
            const int64_t qq = q + j;
            if ((qq % THREADS) == 0) {
                __syncthreads();
                fetch_meta((int)((qq / THREADS + 1) & 1), qq + THREADS);
                __syncthreads();
            }
            load_rec(x[(j + NBUF - 1) % NBUF], y[(j + NBUF - 1) % NBUF], qq + NBUF - 1);
#pragma unroll
            for (int k = 0; k < (WIDE16 ? 4 : NCH); ++k) acc ^= x[j][k];
            acc ^= y[j].x ^ y[j].y;
        }
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;
}

// ------------------------------------------------------------------------------------------------------------------
// Part 2: the row pass with the accumulation one row behind.
//   thread t owns the columns 4 (t + 256 k) + e, k < NCH (as the product kernel);
//   P_LDS: the proportions sit in LDS as s_p[k][e][t] (lane-contiguous: conflict-free ds_read_b64 at immediate
//          offsets) and are read for every dot product; otherwise in registers (48 at NCH = 6);
//   step q:  loads of row q + NBUF - 1 | dot(q) from va | ladder(q), interleaved by the scheduler with
//            acc += cf(q-1) * vb and the lookups of row q + 1 into vb | red, table of row q + 2 -> LDS | barrier |
//            wave sums -> cf(q) (used in step q + 1) ; va <-> vb by unrolling.
// ------------------------------------------------------------------------------------------------------------------
template <int NCH, int NBUF, bool P_LDS, int WG_PER_CU>
__global__ __launch_bounds__(256, WG_PER_CU) void delayed_kernel(const uint8_t *__restrict__ rec, const int64_t *__restrict__ rec_off,
                                                                 const int32_t *__restrict__ ndist, int ldc,
                                                                 const double *__restrict__ w, const double *__restrict__ props,
                                                                 int64_t R, int H, double *__restrict__ partial, int64_t ldpart) {
    constexpr int THREADS = 256, NW = 4;
    static_assert(NBUF >= 4 && (NBUF % 2) == 0, "tables are published two rows ahead; the v roles alternate");
    __shared__ double s_tbl[NBUF][C3_MAX_CODES];
    __shared__ __attribute__((aligned(16))) double red[NBUF][NW];
    __shared__ long long s_off[2][THREADS];
    __shared__ double s_wr[2][THREADS];
    __shared__ int s_nd[2][THREADS];
    __shared__ double s_p[P_LDS ? NCH * 4 * THREADS : 1];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int nword = ldc >> 2;
    const row_deal deal(R);
    const int voff = t * 4;
    int last_w = t + (NCH - 1) * THREADS;
    if (last_w > nword - 1) last_w = nword - 1;
    const int voff_last = last_w * 4;
    const int tslot = t & (C3_MAX_CODES - 1);

    double p[P_LDS ? 1 : NCH][4], acc[NCH][4];
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * (t + k * THREADS) + e;
            const double pv = (c < H) ? props[c] : 0.0;
            if constexpr (P_LDS) s_p[(k * 4 + e) * THREADS + t] = pv;
            else p[k][e] = pv;
            acc[k][e] = 0.0;
        }
    auto fetch_meta = [&](int half, int64_t q0) {
        const int64_t q = q0 + t;
        const int64_t r = deal.row(q);
        int nd = ndist[r];
        if (nd > C3_MAX_CODES) nd = 0;
        s_off[half][t] = rec_off[r];
        s_nd[half][t] = nd;
        s_wr[half][t] = (deal.live(q) && nd > 0) ? (w != nullptr ? w[r] : 1.0) : 0.0;
    };
    unsigned int cw[NBUF][NCH];
    double tring[NBUF];
    int pre_off_lo, pre_off_hi, pre_nd;
    double pre_wr;
    auto read_meta = [&](int64_t q_load, int64_t q_weight) {
        const int half = (int)((q_load / THREADS) & 1), idx = (int)(q_load % THREADS);
        const long long off = s_off[half][idx];
        pre_nd = __builtin_amdgcn_readfirstlane(s_nd[half][idx]);
        pre_off_hi = __builtin_amdgcn_readfirstlane((int)(off >> 32));
        pre_off_lo = __builtin_amdgcn_readfirstlane((int)off);
        pre_wr = s_wr[(q_weight / THREADS) & 1][q_weight % THREADS];
    };
    auto load_row = [&](unsigned int(&cws)[NCH], double &tbl_entry) {
        const int nd = pre_nd;
        const uint8_t *base = rec + (((long long)pre_off_hi << 32) | (unsigned int)pre_off_lo);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, ldc, 0x00020000);
#pragma unroll
        for (int k = 0; k < NCH - 1; ++k) cws[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, k * THREADS * 4, 2);
        cws[NCH - 1] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff_last, 0, 2);
        const auto rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base + ldc), 0, nd * 8, 0x00020000);
        const c3_u2 v = __builtin_amdgcn_raw_buffer_load_b64(rt, tslot * 8, 0, 2);
        tbl_entry = __hiloint2double((int)v.y, (int)v.x);
    };
    auto lookup = [&](const char *tb, unsigned int word, auto E) -> double {
        return *reinterpret_cast<const double *>(tb + code_byte_x8<decltype(E)::value>(word));
    };
    using E0 = std::integral_constant<int, 0>;
    using E1 = std::integral_constant<int, 1>;
    using E2 = std::integral_constant<int, 2>;
    using E3 = std::integral_constant<int, 3>;
    double va[NCH][4], vb[NCH][4];                       // the row being multiplied / the row being accumulated
    double cf_prev = 0.0;                                // coefficient of the row in the "accumulated" set

    // One step.  VD: the set holding row q (dot product now, accumulated next step); VA: the set holding row q - 1
    // (accumulated now, refilled with row q + 1).
    auto step = [&](auto J, int64_t q, double(&vd)[NCH][4], double(&vacc)[NCH][4]) {
        constexpr int j = decltype(J)::value;
        constexpr int jn = (j + 1) % NBUF, jn2 = (j + 2) % NBUF, jl = (j + NBUF - 1) % NBUF;
        if ((q % THREADS) == 0) fetch_meta((int)((q / THREADS + 1) & 1), q + THREADS);
        load_row(cw[jl], tring[jl]);                     // row q + NBUF - 1
        const double wr = pre_wr;                        // weight of row q
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double pv;
                if constexpr (P_LDS) pv = s_p[(k * 4 + e) * THREADS + t];
                else pv = p[k][e];
                s4[e] = fma(vd[k][e], pv, s4[e]);
            }
        double s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        s = wave_sum_lane63(s);
        // row q - 1 into the accumulators, row q + 1's values into the registers it leaves (its table was published at
        // the barrier of the step before)
        const char *tbn = reinterpret_cast<const char *>(&s_tbl[jn][0]);
        const double cf = cf_prev;
        auto upd = [&](int k, auto E) {
            constexpr int e = decltype(E)::value;
            acc[k][e] = fma(cf, vacc[k][e], acc[k][e]);
            asm volatile("" : "+v"(acc[k][e]));
            vacc[k][e] = lookup(tbn, cw[jn][k], E);
        };
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            upd(k, E0{});
            upd(k, E1{});
            upd(k, E2{});
            upd(k, E3{});
        }
        if (lane == 63) red[j][wv] = s;
        s_tbl[jn2][tslot] = tring[jn2];                  // row q + 2's table, published by this step's barrier
        __syncthreads();
        read_meta(q + NBUF, q + 1);
        typedef double d2v __attribute__((ext_vector_type(2)));
        const d2v ra = *reinterpret_cast<const d2v *>(&red[j][0]), rb = *reinterpret_cast<const d2v *>(&red[j][2]);
        cf_prev = readlane_f64(weight_over_norm(wr, (ra.x + ra.y) + (rb.x + rb.y)), 0);       // used one step later
    };

    if (deal.nq > 0) {
        fetch_meta(0, 0);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NBUF - 1; ++j) {
            read_meta(j, 0);
            load_row(cw[j], tring[j]);
        }
        s_tbl[0][tslot] = tring[0];
        s_tbl[1][tslot] = tring[1];
        __syncthreads();
        read_meta(NBUF - 1, 0);
        const char *tb0 = reinterpret_cast<const char *>(&s_tbl[0][0]);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            va[k][0] = lookup(tb0, cw[0][k], E0{});
            va[k][1] = lookup(tb0, cw[0][k], E1{});
            va[k][2] = lookup(tb0, cw[0][k], E2{});
            va[k][3] = lookup(tb0, cw[0][k], E3{});
            vb[k][0] = vb[k][1] = vb[k][2] = vb[k][3] = 0.0;       // "row -1": coefficient 0
        }
        // rows past the workgroup's last re-read that row with weight 0 (fetch_meta), so running the ring to a
        // multiple of NBUF and one extra step for the delayed accumulation of the last row is harmless
        const int64_t nsteps = (deal.nq + 1 + NBUF - 1) / NBUF * NBUF;
        for (int64_t q = 0; q < nsteps; q += NBUF) {
            step(std::integral_constant<int, 0>{}, q, va, vb);
            step(std::integral_constant<int, 1>{}, q + 1, vb, va);
            step(std::integral_constant<int, 2>{}, q + 2, va, vb);
            step(std::integral_constant<int, 3>{}, q + 3, vb, va);
            if constexpr (NBUF > 4) {
                step(std::integral_constant<int, 4>{}, q + 4, va, vb);
                step(std::integral_constant<int, 5>{}, q + 5, vb, va);
            }
        }
    }
    double *dst = partial + (int64_t)blockIdx.x * ldpart;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = 4 * (t + k * THREADS);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < H) dst[c + e] = acc[k][e];
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Part 3: TWO rows per step -- rows 2s and 2s + 1 of the workgroup share one barrier; their ladders and divisions are
// independent instruction streams the scheduler can interleave (the product kernel's chain has nothing beside it).
// Costs a second set of row values (48 registers); proportions stay in registers.  NP pairs in the ring.
// ------------------------------------------------------------------------------------------------------------------
template <int NCH, int NP, int WG_PER_CU>
__global__ __launch_bounds__(256, WG_PER_CU) void pair_kernel(const uint8_t *__restrict__ rec, const int64_t *__restrict__ rec_off,
                                                              const int32_t *__restrict__ ndist, int ldc,
                                                              const double *__restrict__ w, const double *__restrict__ props,
                                                              int64_t R, int H, double *__restrict__ partial, int64_t ldpart) {
    constexpr int THREADS = 256, NW = 4;
    static_assert(NP >= 3, "codes NP - 1 pairs ahead, tables NP - 2");
    __shared__ double s_tbl[NP][2][C3_MAX_CODES];
    __shared__ __attribute__((aligned(16))) double red[NP][2][NW];
    __shared__ long long s_off[2][THREADS];
    __shared__ double s_wr[2][THREADS];
    __shared__ int s_nd[2][THREADS];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int nword = ldc >> 2;
    const row_deal deal(R);
    const int voff = t * 4;
    int last_w = t + (NCH - 1) * THREADS;
    if (last_w > nword - 1) last_w = nword - 1;
    const int voff_last = last_w * 4;
    const int tslot = t & (C3_MAX_CODES - 1);
    double p[NCH][4], acc[NCH][4];
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * (t + k * THREADS) + e;
            p[k][e] = (c < H) ? props[c] : 0.0;
            acc[k][e] = 0.0;
        }
    auto fetch_meta = [&](int half, int64_t q0) {
        const int64_t q = q0 + t;
        const int64_t r = deal.row(q);
        int nd = ndist[r];
        if (nd > C3_MAX_CODES) nd = 0;
        s_off[half][t] = rec_off[r];
        s_nd[half][t] = nd;
        s_wr[half][t] = (deal.live(q) && nd > 0) ? (w != nullptr ? w[r] : 1.0) : 0.0;
    };
    unsigned int cw[NP][2][NCH];
    double tring[NP][2];
    auto load_row = [&](int64_t q, unsigned int(&cws)[NCH], double &tbl_entry) {
        const int half = (int)((q / THREADS) & 1), idx = (int)(q % THREADS);
        const long long off = s_off[half][idx];
        const int nd = __builtin_amdgcn_readfirstlane(s_nd[half][idx]);
        const uint8_t *base = rec + (((long long)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) |
                                     (unsigned int)__builtin_amdgcn_readfirstlane((int)off));
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, ldc, 0x00020000);
#pragma unroll
        for (int k = 0; k < NCH - 1; ++k) cws[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, k * THREADS * 4, 2);
        cws[NCH - 1] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff_last, 0, 2);
        const auto rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base + ldc), 0, nd * 8, 0x00020000);
        const c3_u2 v = __builtin_amdgcn_raw_buffer_load_b64(rt, tslot * 8, 0, 2);
        tbl_entry = __hiloint2double((int)v.y, (int)v.x);
    };
    auto lookup = [&](const char *tb, unsigned int word, auto E) -> double {
        return *reinterpret_cast<const double *>(tb + code_byte_x8<decltype(E)::value>(word));
    };
    using E0 = std::integral_constant<int, 0>;
    using E1 = std::integral_constant<int, 1>;
    using E2 = std::integral_constant<int, 2>;
    using E3 = std::integral_constant<int, 3>;
    double va[NCH][4], vb[NCH][4];
    auto step = [&](auto J, int64_t q) {                  // q = first row of the pair
        constexpr int j = decltype(J)::value;
        constexpr int jn = (j + 1) % NP, jl = (j + NP - 1) % NP;
        // metadata blocks: the loads below read steps q + 2 (NP - 1), + 1; a block of 256 steps is fetched when the
        // pair index crosses into its second half (THREADS is even, so pairs never straddle a block)
        if ((q % THREADS) == 0) fetch_meta((int)((q / THREADS + 1) & 1), q + THREADS);
        load_row(q + 2 * (NP - 1), cw[jl][0], tring[jl][0]);
        load_row(q + 2 * (NP - 1) + 1, cw[jl][1], tring[jl][1]);
        const double wra = s_wr[(q / THREADS) & 1][q % THREADS], wrb = s_wr[((q + 1) / THREADS) & 1][(q + 1) % THREADS];
        double sa4[4] = {0.0, 0.0, 0.0, 0.0}, sb4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sa4[e] = fma(va[k][e], p[k][e], sa4[e]);
                sb4[e] = fma(vb[k][e], p[k][e], sb4[e]);
            }
        double sa = (sa4[0] + sa4[1]) + (sa4[2] + sa4[3]), sb = (sb4[0] + sb4[1]) + (sb4[2] + sb4[3]);
        __builtin_amdgcn_s_setprio(1);
        sa = wave_sum_lane63(sa);
        sb = wave_sum_lane63(sb);
        if (lane == 63) {
            red[j][0][wv] = sa;
            red[j][1][wv] = sb;
        }
        s_tbl[jn][0][tslot] = tring[jn][0];
        s_tbl[jn][1][tslot] = tring[jn][1];
        __syncthreads();
        typedef double d2v __attribute__((ext_vector_type(2)));
        const d2v a0 = *reinterpret_cast<const d2v *>(&red[j][0][0]), a1 = *reinterpret_cast<const d2v *>(&red[j][0][2]);
        const d2v b0 = *reinterpret_cast<const d2v *>(&red[j][1][0]), b1 = *reinterpret_cast<const d2v *>(&red[j][1][2]);
        const double cfa = readlane_f64(weight_over_norm(wra, (a0.x + a0.y) + (a1.x + a1.y)), 0);
        const double cfb = readlane_f64(weight_over_norm(wrb, (b0.x + b0.y) + (b1.x + b1.y)), 0);
        __builtin_amdgcn_s_setprio(0);
        const char *tna = reinterpret_cast<const char *>(&s_tbl[jn][0][0]);
        const char *tnb = reinterpret_cast<const char *>(&s_tbl[jn][1][0]);
        auto upd = [&](int k, auto E) {
            constexpr int e = decltype(E)::value;
            acc[k][e] = fma(cfa, va[k][e], acc[k][e]);
            acc[k][e] = fma(cfb, vb[k][e], acc[k][e]);
            asm volatile("" : "+v"(acc[k][e]));
            va[k][e] = lookup(tna, cw[jn][0][k], E);
            vb[k][e] = lookup(tnb, cw[jn][1][k], E);
        };
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            upd(k, E0{});
            upd(k, E1{});
            upd(k, E2{});
            upd(k, E3{});
        }
    };
    if (deal.nq > 0) {
        fetch_meta(0, 0);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NP - 1; ++j) {
            load_row(2 * j, cw[j][0], tring[j][0]);
            load_row(2 * j + 1, cw[j][1], tring[j][1]);
        }
        s_tbl[0][0][tslot] = tring[0][0];
        s_tbl[0][1][tslot] = tring[0][1];
        __syncthreads();
        const char *ta = reinterpret_cast<const char *>(&s_tbl[0][0][0]), *tb = reinterpret_cast<const char *>(&s_tbl[0][1][0]);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            va[k][0] = lookup(ta, cw[0][0][k], E0{}); va[k][1] = lookup(ta, cw[0][0][k], E1{});
            va[k][2] = lookup(ta, cw[0][0][k], E2{}); va[k][3] = lookup(ta, cw[0][0][k], E3{});
            vb[k][0] = lookup(tb, cw[0][1][k], E0{}); vb[k][1] = lookup(tb, cw[0][1][k], E1{});
            vb[k][2] = lookup(tb, cw[0][1][k], E2{}); vb[k][3] = lookup(tb, cw[0][1][k], E3{});
        }
        const int64_t nsteps = (deal.nq + 2 * NP - 1) / (2 * NP) * (2 * NP);
        for (int64_t q = 0; q < nsteps; q += 2 * NP) {
            step(std::integral_constant<int, 0>{}, q);
            step(std::integral_constant<int, 1>{}, q + 2);
            step(std::integral_constant<int, 2>{}, q + 4);
            if constexpr (NP > 3) step(std::integral_constant<int, 3>{}, q + 6);
        }
    }
    double *dst = partial + (int64_t)blockIdx.x * ldpart;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = 4 * (t + k * THREADS);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < H) dst[c + e] = acc[k][e];
    }
}

static float time_launches(int reps, const std::function<void()> &launch) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1.0f;
    launch();
    if (hipDeviceSynchronize() != hipSuccess) return -3.0f;
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b, 0);
    if (hipEventSynchronize(b) != hipSuccess) return -4.0f;
    float ms = 0.0f;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a);
    hipEventDestroy(b);
    return ms / reps;
}

// bare readers: variant -> {rows in flight, load width, policy, workgroups per CU, order}
extern "C" float coded3_reader(int variant, const uint8_t *rec, const int64_t *rec_off, const int32_t *ndist, int ldc, int64_t R,
                               const int64_t *order, unsigned int *sink, int n_cu, int reps) {
    if (ldc / 4 > 1536) return -2.0f;
#define RD(nbuf, wide, aux, wg, ord) [&]() { hipLaunchKernelGGL((reader_kernel<6, nbuf, wide, aux, wg>), dim3(n_cu * wg), dim3(256), 0, 0, rec, rec_off, ndist, ldc, R, ord, sink); }
    switch (variant) {
        case 0: return time_launches(reps, RD(3, false, 2, 2, nullptr));    // the product kernel's loads
        case 1: return time_launches(reps, RD(6, false, 2, 2, nullptr));    // five rows in flight
        case 2: return time_launches(reps, RD(3, true, 2, 2, nullptr));     // 16 B per lane
        case 3: return time_launches(reps, RD(6, true, 2, 2, nullptr));
        case 4: return time_launches(reps, RD(3, false, 0, 2, nullptr));    // default cache policy
        case 5: return time_launches(reps, RD(3, false, 2, 2, order));      // records in address order
        case 6: return time_launches(reps, RD(6, true, 2, 2, order));
        case 7: return time_launches(reps, RD(6, true, 2, 4, nullptr));     // four workgroups per CU
        case 8: return time_launches(reps, RD(6, true, 0, 2, nullptr));
        default: return -5.0f;
    }
#undef RD
}

extern "C" float coded3_time(int variant, const uint8_t *rec, const int64_t *rec_off, const int32_t *ndist, int ldc,
                             const double *w, const double *props, int64_t R, int H, double *partial, int64_t ldpart,
                             int n_cu, int reps, int *grid_out) {
    if (ldc / 4 > 1536) return -2.0f;
    int grid = n_cu * 2;
#define DL(nbuf, plds, wg) [&]() { hipLaunchKernelGGL((delayed_kernel<6, nbuf, plds, wg>), dim3(grid), dim3(256), 0, 0, rec, rec_off, ndist, ldc, w, props, R, H, partial, ldpart); }
    float ms;
    switch (variant) {
        case 0: grid = n_cu * 2; ms = time_launches(reps, DL(4, true, 2)); break;      // p in LDS, two workgroups per CU
        case 1: grid = n_cu * 2; ms = time_launches(reps, DL(6, true, 2)); break;      // ... five rows in flight
        case 2: grid = n_cu * 2; ms = time_launches(reps, DL(4, false, 2)); break;     // p in registers (spills expected)
        case 3: grid = n_cu; ms = time_launches(reps, DL(4, false, 1)); break;         // p in registers, one workgroup per CU
        case 4: grid = n_cu * 2; ms = time_launches(reps, [&]() { hipLaunchKernelGGL((pair_kernel<6, 3, 2>), dim3(grid), dim3(256), 0, 0, rec, rec_off, ndist, ldc, w, props, R, H, partial, ldpart); }); break;
        case 5: grid = n_cu * 2; ms = time_launches(reps, [&]() { hipLaunchKernelGGL((pair_kernel<6, 4, 2>), dim3(grid), dim3(256), 0, 0, rec, rec_off, ndist, ldc, w, props, R, H, partial, ldpart); }); break;
        default: return -5.0f;
    }
#undef DL
    *grid_out = grid;
    return ms;
}
