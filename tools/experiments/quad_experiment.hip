// quad_experiment.hip -- NOT part of libmixemt_hip.so.  Round-5 experiment on the records row pass: a QUAD dictionary
// (profiles/r05/experiments.md, 1d).  A byte-coded record spends, per cell, a shift (code byte -> table offset), a
// ds_read_b64 and two FMAs, and the pass is bound by instruction issue.  Here a code names FOUR consecutive columns'
// values at once: one shift and two ds_read_b128 per four cells; a row's record is
//     256 x 8 bytes of codes, thread-contiguous (byte j of thread t = quad t + 256 j, j < 6; the rest padding)
//     nq x 32 bytes of table (the row's distinct quads of values), nq <= 256
// Thread t owns the columns 4 (t + 256 k) + e as in the product kernel, so the column sums can be compared with its.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I include -I mixemt_amd/csrc \
//         tools/experiments/quad_experiment.hip -o tools/experiments/_build/libquad.so
// Driver: tools/experiments/time_quad.py (builds the quad records from the product's records with torch).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <cmath>
#include <functional>
#include <type_traits>
#include "mixemt_hip.h"
#include "common.hpp"

#define Q_MAX 256
#define Q_CODE_BYTES 2048
typedef unsigned int q_u2 __attribute__((ext_vector_type(2)));
typedef unsigned int q_u4 __attribute__((ext_vector_type(4)));
typedef double q_d2 __attribute__((ext_vector_type(2)));

template <int B>
__device__ __forceinline__ unsigned int quad_byte_x32(unsigned int word) {
    const unsigned int five = 5;
    unsigned int r;
    if constexpr (B == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(five), "v"(word));
    else if constexpr (B == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(five), "v"(word));
    else if constexpr (B == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(five), "v"(word));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(five), "v"(word));
    return r;
}

// The wave's sum on the matrix core: D = A x ones gives every lane of a 16-lane group the sums over the four lanes
// {i, i + 16, i + 32, i + 48} of four i; folding the four registers and a second product leaves the total in EVERY lane --
// two v_mfma_f64_16x16x4_f64 and three v_add_f64 instead of twelve v_mov_dpp, six v_add_f64 and the DPP hazard nops.
typedef double q_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double wave_sum_mfma(double v) {
    const q_d4 zero = {0.0, 0.0, 0.0, 0.0};
    const q_d4 a = __builtin_amdgcn_mfma_f64_16x16x4f64(v, 1.0, zero, 0, 0, 0);
    const double t = (a.x + a.y) + (a.z + a.w);
    const q_d4 b = __builtin_amdgcn_mfma_f64_16x16x4f64(t, 1.0, zero, 0, 0, 0);
    return b.x;
}

// DMA: the table goes global -> LDS directly (buffer_load_dwordx4 ... lds, gfx950), no registers, no ds_write
template <int NBUF, bool DMA, int WG_PER_CU, bool ONEWAIT = false, bool MFMA_SUM = false>
__global__ __launch_bounds__(256, WG_PER_CU) void quad_kernel(const uint8_t *__restrict__ qrec, const int64_t *__restrict__ qoff,
                                                              const int32_t *__restrict__ nquad, const double *__restrict__ w,
                                                              const double *__restrict__ props, int64_t R, int H,
                                                              double *__restrict__ partial, int64_t ldpart) {
    constexpr int THREADS = 256, NW = 4, NCH = 6, AUX = 2;
    static_assert(NBUF >= 3, "codes NBUF - 1 rows ahead");
    __shared__ __attribute__((aligned(16))) double s_tbl[NBUF][Q_MAX * 4];
    __shared__ __attribute__((aligned(16))) double red[NBUF][NW];
    __shared__ long long s_off[2][THREADS];
    __shared__ double s_wr[2][THREADS];
    __shared__ int s_nd[2][THREADS];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const row_deal deal(R);

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
        int nd = nquad[r];
        if (nd > Q_MAX || nd < 0) nd = 0;
        s_off[half][t] = qoff[r];
        s_nd[half][t] = nd;
        s_wr[half][t] = (deal.live(q) && nd > 0) ? (w != nullptr ? w[r] : 1.0) : 0.0;
    };
    q_u2 cw[NBUF];
    double tring[DMA ? 1 : NBUF][4];
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
    auto load_row = [&](auto SLOT) {
        constexpr int slot = decltype(SLOT)::value;
        const int nd = pre_nd;
        const uint8_t *base = qrec + (((long long)pre_off_hi << 32) | (unsigned int)pre_off_lo);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, Q_CODE_BYTES, 0x00020000);
        cw[slot] = __builtin_amdgcn_raw_buffer_load_b64(rs, t * 8, 0, AUX);
        const auto rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base + Q_CODE_BYTES), 0, nd * 32, 0x00020000);
        if constexpr (DMA) {
            // lane l of wave wv writes 16 bytes at M0 + 16 l: the LDS image is the table's own layout
            char *dst = reinterpret_cast<char *>(&s_tbl[slot][0]) + wv * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void *)dst, 16, t * 16, 0, 0, AUX);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void *)(dst + 4096), 16, t * 16, 0, 4096, AUX);
        } else {
            const q_u4 a = __builtin_amdgcn_raw_buffer_load_b128(rt, t * 32, 0, AUX);
            const q_u4 b = __builtin_amdgcn_raw_buffer_load_b128(rt, t * 32, 16, AUX);
            tring[slot][0] = __hiloint2double((int)a.y, (int)a.x);
            tring[slot][1] = __hiloint2double((int)a.w, (int)a.z);
            tring[slot][2] = __hiloint2double((int)b.y, (int)b.x);
            tring[slot][3] = __hiloint2double((int)b.w, (int)b.z);
        }
    };
    auto publish = [&](auto SLOT) {
        if constexpr (!DMA) {
            constexpr int slot = decltype(SLOT)::value;
            q_d2 *dst = reinterpret_cast<q_d2 *>(&s_tbl[slot][t * 4]);
            dst[0] = q_d2{tring[slot][0], tring[slot][1]};
            dst[1] = q_d2{tring[slot][2], tring[slot][3]};
        }
    };
    double v[NCH][4];
    auto lookup_quad = [&](const char *tb, auto K, const q_u2 &c) {
        constexpr int k = decltype(K)::value;
        unsigned int off;
        if constexpr (k < 4) off = quad_byte_x32<k>(c.x);
        else off = quad_byte_x32<k - 4>(c.y);
        const q_d2 a = *reinterpret_cast<const q_d2 *>(tb + off), b = *reinterpret_cast<const q_d2 *>(tb + off + 16);
        v[k][0] = a.x;
        v[k][1] = a.y;
        v[k][2] = b.x;
        v[k][3] = b.y;
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;
    using K4 = std::integral_constant<int, 4>;
    using K5 = std::integral_constant<int, 5>;

    auto step = [&](auto J, int64_t q) {
        constexpr int j = decltype(J)::value;
        constexpr int jn = (j + 1) % NBUF, jl = (j + NBUF - 1) % NBUF;
        if ((q % THREADS) == 0) fetch_meta((int)((q / THREADS + 1) & 1), q + THREADS);
        load_row(std::integral_constant<int, jl>{});     // row q + NBUF - 1
        const double wr = pre_wr;
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
        // ONEWAIT: one wait for all of the row's lookups instead of the compiler's countdown (lgkmcnt(11), (10), ... (0):
        // a wait per arriving value -- a dozen issue slots to start the dot product a few cycles earlier)
        if constexpr (ONEWAIT) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) s4[e] = fma(v[k][e], p[k][e], s4[e]);
        double s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (MFMA_SUM) s = wave_sum_mfma(s);
        else s = wave_sum_lane63(s);
        if (lane == 63) red[j][wv] = s;
        publish(std::integral_constant<int, jn>{});      // row q + 1's table, published by the same barrier
        __syncthreads();
        read_meta(q + NBUF, q + 1);
        const q_d2 ra = *reinterpret_cast<const q_d2 *>(&red[j][0]), rb = *reinterpret_cast<const q_d2 *>(&red[j][2]);
        const double cf = readlane_f64(weight_over_norm(wr, (ra.x + ra.y) + (rb.x + rb.y)), 0);
        __builtin_amdgcn_s_setprio(0);
        const char *tbn = reinterpret_cast<const char *>(&s_tbl[jn][0]);
        auto upd = [&](auto K) {
            constexpr int k = decltype(K)::value;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[k][e] = fma(cf, v[k][e], acc[k][e]);
                asm volatile("" : "+v"(acc[k][e]));
            }
            lookup_quad(tbn, K, cw[jn]);
        };
        upd(K0{});
        upd(K1{});
        upd(K2{});
        upd(K3{});
        upd(K4{});
        upd(K5{});
    };

    if (deal.nq > 0) {
        fetch_meta(0, 0);
        __syncthreads();
        read_meta(0, 0);
        load_row(std::integral_constant<int, 0>{});
        read_meta(1, 0);
        load_row(std::integral_constant<int, 1>{});
        if constexpr (NBUF > 3) {
            read_meta(2, 0);
            load_row(std::integral_constant<int, 2>{});
        }
        if constexpr (NBUF > 4) {
            read_meta(3, 0);
            load_row(std::integral_constant<int, 3>{});
        }
        if constexpr (NBUF > 5) {
            read_meta(4, 0);
            load_row(std::integral_constant<int, 4>{});
        }
        publish(std::integral_constant<int, 0>{});
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        read_meta(NBUF - 1, 0);
        {
            const char *tb0 = reinterpret_cast<const char *>(&s_tbl[0][0]);
            lookup_quad(tb0, K0{}, cw[0]);
            lookup_quad(tb0, K1{}, cw[0]);
            lookup_quad(tb0, K2{}, cw[0]);
            lookup_quad(tb0, K3{}, cw[0]);
            lookup_quad(tb0, K4{}, cw[0]);
            lookup_quad(tb0, K5{}, cw[0]);
        }
        for (int64_t q = 0; q < deal.nq; q += NBUF) {
            step(std::integral_constant<int, 0>{}, q);
            step(std::integral_constant<int, 1>{}, q + 1);
            step(std::integral_constant<int, 2>{}, q + 2);
            if constexpr (NBUF > 3) step(std::integral_constant<int, 3>{}, q + 3);
            if constexpr (NBUF > 4) step(std::integral_constant<int, 4>{}, q + 4);
            if constexpr (NBUF > 5) step(std::integral_constant<int, 5>{}, q + 5);
        }
    }
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (no LDS write may land after the workgroup left)
    double *dst = partial + (int64_t)blockIdx.x * ldpart;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = 4 * (t + k * THREADS);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < H) dst[c + e] = acc[k][e];
    }
}

// ---- B restarts through ONE pass over the quad records (config 3: ten restarts on one matrix) -------------------------
// A workgroup of 512 threads takes a row (3 quads = 12 columns per thread: thread t' owns the quads tl + 256 (2 k + half),
// tl = t' & 255, half = t' >> 8 -- the odd or the even code bytes of the word thread tl of the product kernel loads) and
// keeps the proportions and the column sums of B restarts in registers (B x 48 VGPRs each); a row's codes, table and LDS
// lookups are fetched once for all B, the dot products / wave sums / updates run B times.  One workgroup per CU (two waves
// per SIMD, 256 registers each).
template <int B, int NBUF>
__global__ __launch_bounds__(512, 1) void quadB_kernel(const uint8_t *__restrict__ qrec, const int64_t *__restrict__ qoff,
                                                       const int32_t *__restrict__ nquad, const double *__restrict__ w,
                                                       const double *__restrict__ props, int64_t ldp, int64_t R, int H,
                                                       double *__restrict__ partial, int64_t ldpart) {
    constexpr int THREADS = 512, NW = 8, NCH = 3, AUX = 2;
    static_assert(NBUF >= 3 && NBUF <= 4, "codes NBUF - 1 rows ahead");
    static_assert(B * NW <= 64, "one partial per lane in the ratio");
    __shared__ __attribute__((aligned(16))) double s_tbl[NBUF][Q_MAX * 4];
    __shared__ __attribute__((aligned(16))) double red[NBUF][B * NW];
    __shared__ long long s_off[2][THREADS];
    __shared__ double s_wr[2][THREADS];
    __shared__ int s_nd[2][THREADS];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, half = t >> 8, tl = t & 255;
    const row_deal deal(R);

    double p[B][NCH][4], acc[B][NCH][4];
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * (tl + 256 * (2 * k + half)) + e;
                p[b][k][e] = (c < H) ? props[b * ldp + c] : 0.0;
                acc[b][k][e] = 0.0;
            }
    auto fetch_meta = [&](int hf, int64_t q0) {
        const int64_t q = q0 + t;
        const int64_t r = deal.row(q);
        int nd = nquad[r];
        if (nd > Q_MAX || nd < 0) nd = 0;
        s_off[hf][t] = nd > 0 ? qoff[r] : 0;
        s_nd[hf][t] = nd;
        s_wr[hf][t] = (deal.live(q) && nd > 0) ? (w != nullptr ? w[r] : 1.0) : 0.0;
    };
    q_u2 cw[NBUF];
    q_d2 tring[NBUF];
    int pre_off_lo, pre_off_hi, pre_nd;
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
        const uint8_t *base = qrec + (((long long)pre_off_hi << 32) | (unsigned int)pre_off_lo);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, nd > 0 ? Q_CODE_BYTES : 0, 0x00020000);
        cw[slot] = __builtin_amdgcn_raw_buffer_load_b64(rs, tl * 8, 0, AUX);
        const auto rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base + Q_CODE_BYTES), 0, nd * 32, 0x00020000);
        const q_u4 a = __builtin_amdgcn_raw_buffer_load_b128(rt, t * 16, 0, AUX);
        tring[slot] = q_d2{__hiloint2double((int)a.y, (int)a.x), __hiloint2double((int)a.w, (int)a.z)};
    };
    auto publish = [&](auto SLOT) {
        constexpr int slot = decltype(SLOT)::value;
        *reinterpret_cast<q_d2 *>(&s_tbl[slot][t * 2]) = tring[slot];
    };
    double v[NCH][4];
    // the thread's three code bytes: 0, 2, 4 of the word shifted down by `half` bytes
    auto lookup_row = [&](const char *tb, const q_u2 &c) {
        const unsigned long long word = (((unsigned long long)c.y << 32) | c.x) >> (8 * half);
        const unsigned int lo = (unsigned int)word, hi = (unsigned int)(word >> 32);
        const unsigned int off[NCH] = {quad_byte_x32<0>(lo), quad_byte_x32<2>(lo), quad_byte_x32<0>(hi)};
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const q_d2 a = *reinterpret_cast<const q_d2 *>(tb + off[k]), b = *reinterpret_cast<const q_d2 *>(tb + off[k] + 16);
            v[k][0] = a.x;
            v[k][1] = a.y;
            v[k][2] = b.x;
            v[k][3] = b.y;
        }
    };

    auto step = [&](auto J, int64_t q) {
        constexpr int j = decltype(J)::value;
        constexpr int jn = (j + 1) % NBUF, jl = (j + NBUF - 1) % NBUF;
        if ((q % THREADS) == 0) fetch_meta((int)((q / THREADS + 1) & 1), q + THREADS);
        load_row(std::integral_constant<int, jl>{});     // row q + NBUF - 1
        const double wr = pre_wr;
        double s[B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            double s2[2] = {0.0, 0.0};
#pragma unroll
            for (int k = 0; k < NCH; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) s2[e & 1] = fma(v[k][e], p[b][k][e], s2[e & 1]);
            s[b] = s2[0] + s2[1];
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int b = 0; b < B; ++b) {
            s[b] = wave_sum_lane63(s[b]);
            if (lane == 63) red[j][b * NW + wv] = s[b];
        }
        publish(std::integral_constant<int, jn>{});      // row q + 1's table, published by the same barrier
        __syncthreads();
        read_meta(q + NBUF, q + 1);
        double cf[B];
        group_ratio_to_sgpr<NW, B>(&red[j][0], lane, wr, cf);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int k = 0; k < NCH; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[b][k][e] = fma(cf[b], v[k][e], acc[b][k][e]);
        lookup_row(reinterpret_cast<const char *>(&s_tbl[jn][0]), cw[jn]);
    };

    if (deal.nq > 0) {
        fetch_meta(0, 0);
        __syncthreads();
        read_meta(0, 0);
        load_row(std::integral_constant<int, 0>{});
        read_meta(1, 0);
        load_row(std::integral_constant<int, 1>{});
        if constexpr (NBUF > 3) {
            read_meta(2, 0);
            load_row(std::integral_constant<int, 2>{});
        }
        publish(std::integral_constant<int, 0>{});
        __syncthreads();
        read_meta(NBUF - 1, 0);
        lookup_row(reinterpret_cast<const char *>(&s_tbl[0][0]), cw[0]);
        for (int64_t q = 0; q < deal.nq; q += NBUF) {
            step(std::integral_constant<int, 0>{}, q);
            step(std::integral_constant<int, 1>{}, q + 1);
            step(std::integral_constant<int, 2>{}, q + 2);
            if constexpr (NBUF > 3) step(std::integral_constant<int, 3>{}, q + 3);
        }
    }
#pragma unroll
    for (int b = 0; b < B; ++b) {
        double *dst = partial + ((int64_t)b * gridDim.x + blockIdx.x) * ldpart;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = 4 * (tl + 256 * (2 * k + half));
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < H) dst[c + e] = acc[b][k][e];
        }
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

extern "C" float quad_time(int variant, const uint8_t *qrec, const int64_t *qoff, const int32_t *nquad, const double *w,
                           const double *props, int64_t R, int H, double *partial, int64_t ldpart, int n_cu, int reps,
                           int *grid_out) {
    if (H > 4 * 256 * 6) return -2.0f;
    int grid = n_cu * 2;
#define QK(nbuf, dma, wg, ...) [&]() { hipLaunchKernelGGL((quad_kernel<nbuf, dma, wg, ##__VA_ARGS__>), dim3(grid), dim3(256), 0, 0, qrec, qoff, nquad, w, props, R, H, partial, ldpart); }
    float ms;
    switch (variant) {
        case 0: grid = n_cu * 2; ms = time_launches(reps, QK(3, false, 2)); break;
        case 1: grid = n_cu * 2; ms = time_launches(reps, QK(4, false, 2)); break;
        case 2: grid = n_cu * 2; ms = time_launches(reps, QK(4, true, 2)); break;
        case 3: grid = n_cu * 2; ms = time_launches(reps, QK(6, true, 2)); break;
        case 4: grid = n_cu * 3; ms = time_launches(reps, QK(4, true, 3)); break;
        case 5: grid = n_cu * 2; ms = time_launches(reps, QK(4, false, 2, true)); break;
        case 6: grid = n_cu * 2; ms = time_launches(reps, QK(3, false, 2, true)); break;
        case 7: grid = n_cu * 2; ms = time_launches(reps, QK(5, false, 2)); break;
        case 8: grid = n_cu * 2; ms = time_launches(reps, QK(6, false, 2)); break;
        case 9: grid = n_cu * 2; ms = time_launches(reps, QK(4, false, 2, false, true)); break;
        case 10: grid = n_cu * 2; ms = time_launches(reps, QK(3, false, 2, false, true)); break;
        default: return -5.0f;
    }
#undef QK
    *grid_out = grid;
    return ms;
}

// B restarts in one pass: props [B][ldp], partial [B][grid][ldpart]; -> ms per launch (all B restarts)
extern "C" float quadB_time(int B, int nbuf, const uint8_t *qrec, const int64_t *qoff, const int32_t *nquad, const double *w,
                            const double *props, int64_t ldp, int64_t R, int H, double *partial, int64_t ldpart, int n_cu,
                            int reps, int *grid_out) {
    if (H > 4 * 256 * 6) return -2.0f;
    const int grid = n_cu;
    *grid_out = grid;
#define QB(b, nb) [&]() { hipLaunchKernelGGL((quadB_kernel<b, nb>), dim3(grid), dim3(512), 0, 0, qrec, qoff, nquad, w, props, ldp, R, H, partial, ldpart); }
    switch (B * 10 + nbuf) {
        case 13: return time_launches(reps, QB(1, 3));
        case 23: return time_launches(reps, QB(2, 3));
        case 33: return time_launches(reps, QB(3, 3));
        case 43: return time_launches(reps, QB(4, 3));
        case 24: return time_launches(reps, QB(2, 4));
        case 34: return time_launches(reps, QB(3, 4));
        case 44: return time_launches(reps, QB(4, 4));
        default: return -5.0f;
    }
#undef QB
}
