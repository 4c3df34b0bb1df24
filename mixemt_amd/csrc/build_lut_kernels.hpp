// build_lut_kernels.hpp -- part of libmixemt_hip.so (gfx950); included by mixemt_hip.hip only.
// build_em_matrix (preprocess.py:177-198) with the hit/miss choice served by an LDS lookup.
#ifndef MIXEMT_BUILD_LUT_KERNELS_HPP
#define MIXEMT_BUILD_LUT_KERNELS_HPP

// ------------------------------------------------------------------------------------------
// K1c build_lut: same sums, same order, same bits as build_em_matrix_kernel (K1).
//
// K1 spends 4 VALU operations per (cell, site): byte compare, two 32-bit selects for the fp64
// hit / miss term, the fp64 add -- and it is VALU-bound (72 % pipe busy at 37 ms for 2e11 terms).
// Here the select is a table lookup on the LDS pipe, which K1 leaves idle:
//   * the expected-base table holds 4-bit codes PRE-SHIFTED by 3 (code << 3, one byte per cell),
//     the observation likewise; x = e ^ o is then 0 for a hit and a multiple of 8 below 128 otherwise
//     -- directly the byte offset into a 16-entry table of doubles;
//   * per observation of the row a 128-byte LUT {log hit, log miss x 15} is staged in LDS
//     (double buffered with the row's record list, filled while the previous row is added up);
//   * per (cell, site): one address add (x's byte + the site's LUT base), one ds_read_b64, one
//     fp64 add: 2 VALU + 1 LDS instead of 4 VALU.  Per 4 cells and site: one 4-byte table load through
//     a buffer descriptor (32-bit offset arithmetic), one xor, one 8-byte record read.
// The 16 entries of a site sit in 16 different 8-byte bank pairs, equal addresses broadcast:
// the reads are conflict free whatever the mix of hits and misses.
//
// (A variant that also emitted mxm_linearize's output from the sums in registers measured 65 ms against
// 24.6 + 15.0 ms for the two separate passes -- the kernel is instruction-bound, the exponentials land on its
// critical resource -- and was removed in round 3; profiles/r02/lut_variants.txt.)
// All stores are non-temporal: the matrix streams past the L2 instead of evicting the table.
//
// `order` (nullable): row processing order.  Rows that start at nearby positions touch the same
// table rows; handing them to the grid together keeps those rows in L2.  Results do not depend on it.
// ------------------------------------------------------------------------------------------
#define LUT_THREADS 256
#define LUT_CAP 128                // observations staged per pass (longer rows take several passes)
#define LUT_NOMATCH (15u << 3)     // observation code that equals no expected code
#ifndef LUT_CPL
#define LUT_CPL 4                  // columns per lane and tile: 4 (dword table loads) or 8 (dwordx2)
#endif
#ifndef LUT_UNROLL
#define LUT_UNROLL 8               // sites per block of the inner loop = table loads in flight per wave (x2 when pipelined)
#endif
#ifndef LUT_PIPE
#define LUT_PIPE 0                 // 1: the next block's table loads are issued before the current block is added up
#endif
// Measured at 10^6 x 5408, rows in position order (profiles/r02/lut_variants.txt), matrix only / with P:
//   CPL 4, blocks of 8, plain (default)  24.9 / 65.2 ms      CPL 8, blocks of 8, plain      40.0 / 65.9 ms
//   CPL 4, blocks of 8, pipelined        30.5 / 49.6 ms      CPL 8, blocks of 8, pipelined  36.6 / 60.6 ms
//   CPL 4, blocks of 16, plain           44.8 / 73.2 ms      CPL 8, blocks of 4, pipelined  35.0 / 58.5 ms
// More table bytes in flight per wave do not pay: the kernel is bound by its VALU + LDS instruction
// streams (2.9 VALU + 1.25 LDS per cell and site; the two pipes add up at 4 waves per SIMD), and every
// variant that holds more registers loses a wave per SIMD.

template <int NT, int CPL>
__global__ __launch_bounds__(LUT_THREADS) void build_lut_kernel(
    const uint8_t *__restrict__ E, int64_t lde, int64_t e_bytes, const double *__restrict__ lhit,
    const double *__restrict__ lmiss, const uint8_t *__restrict__ obsmap, const int64_t *__restrict__ row_ptr,
    const uint16_t *__restrict__ site, const uint8_t *__restrict__ obs, const int64_t *__restrict__ order,
    int64_t R, int H, double *__restrict__ M, int64_t ldm, int vec_ok, int compact) {
    // compact (with `order`): row order[i] is written to row i of M -- a compact side matrix of the listed rows (the
    // rows the marker kernel hands back, built a slab at a time: mxm_build_em_matrix_lut_rows) -- instead of to its own row
    static_assert(CPL == 4 || CPL == 8, "one 4- or 8-byte table load per lane, site and tile");
    constexpr int EW = CPL / 4;                         // dwords per table load
    __shared__ double s_lut[2][LUT_CAP][16];            // per observation: [0] = log hit, [1..15] = log miss
    __shared__ uint2 s_rec[2][LUT_CAP];                 // per observation: {table row byte offset, obs code << 3 in 4 bytes}
    const int t = threadIdx.x;
    const auto e_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(E), 0, (int)e_bytes, 0x00020000);

    auto stage = [&](int buf, int slot, int64_t at) {   // observation `at` of the CSR -> slot of buffer buf
        const uint32_t s0 = site[at];
        const uint32_t oc = obsmap[obs[at]];
        const double hv = lhit[s0], mv = lmiss[s0];
        s_rec[buf][slot] = make_uint2((uint32_t)(s0 * (uint32_t)lde), oc * 0x01010101u);
        d2 *dst = reinterpret_cast<d2 *>(&s_lut[buf][slot][0]);
        dst[0] = d2{hv, mv};
#pragma unroll
        for (int q = 1; q < 8; ++q) dst[q] = d2{mv, mv};
    };

    int cur = 0;
    bool have = false;                                  // buffer[cur] already holds this row's list
    for (int64_t i = blockIdx.x; i < R; i += gridDim.x, cur ^= 1) {
        const int64_t r = order != nullptr ? order[i] : i;
        const int64_t beg = row_ptr[r], end = row_ptr[r + 1];
        const int64_t n_all = end - beg;
        const bool fast = n_all <= LUT_CAP;
        const int64_t in = i + gridDim.x;
        const int64_t rn = (in < R) ? (order != nullptr ? order[in] : in) : 0;
        const int64_t begn = (in < R) ? row_ptr[rn] : 0;
        const int nn = (in < R) ? (int)((row_ptr[rn + 1] - begn) <= LUT_CAP ? (row_ptr[rn + 1] - begn) : 0) : 0;
        if (fast && !have && t < (int)n_all) stage(cur, t, beg + t);
        __syncthreads();
        // next row's list in three steps spread over this row's tiles, so neither of its two dependent
        // global loads (site -> lhit[site]) is ever waited for: here the first hop
        uint32_t nsite = 0, nobs = 0, ncode = 0;
        double nhit = 0.0, nmiss = 0.0;
        if (t < nn) {
            nsite = site[begn + t];
            nobs = obs[begn + t];
        }

#pragma unroll
        for (int tile = 0; tile < NT; ++tile) {
            const int h = tile * (LUT_THREADS * CPL) + CPL * t;
            double a[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) a[c] = 0.0;
            auto add_sites = [&](int n) {
                const char *lut = reinterpret_cast<const char *>(&s_lut[cur][0][0]);
                auto fetch = [&](uint32_t(&e)[EW], int j) {         // the table bytes of site j for this lane's columns
                    const uint32_t off = s_rec[cur][j].x + (uint32_t)h;
                    if constexpr (EW == 1) {
                        e[0] = __builtin_amdgcn_raw_buffer_load_b32(e_rsrc, (int)off, 0, 0);
                    } else {
                        typedef unsigned int u2v __attribute__((ext_vector_type(2)));
                        const u2v v = __builtin_amdgcn_raw_buffer_load_b64(e_rsrc, (int)off, 0, 0);
                        e[0] = v.x;
                        e[1] = v.y;
                    }
                };
                auto add_site = [&](const uint32_t(&e)[EW], int j) {
                    const uint32_t o4 = s_rec[cur][j].y;
                    const char *row = lut + j * 128;
#pragma unroll
                    for (int q = 0; q < EW; ++q) {
                        const uint32_t x = e[q] ^ o4;
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            a[4 * q + c] += *reinterpret_cast<const double *>(row + ((x >> (8 * c)) & 0xffu));
                    }
                };
#if LUT_PIPE
                // software pipeline over blocks of LUT_UNROLL sites: the table loads of block b + 1 are in
                // flight while block b goes through the LDS lookups and the adds (the loads are what a wave
                // waits for: L2 latency x bytes in flight).  Sites past n are clamped for the loads and
                // skipped for the adds (n is wave uniform: scalar branches).
                uint32_t ec[LUT_UNROLL][EW], en[LUT_UNROLL][EW];
#pragma unroll
                for (int u = 0; u < LUT_UNROLL; ++u) fetch(ec[u], u < n ? u : n - 1);
                for (int j0 = 0; j0 < n; j0 += LUT_UNROLL) {
                    if (j0 + LUT_UNROLL < n) {
#pragma unroll
                        for (int u = 0; u < LUT_UNROLL; ++u) {
                            const int j = j0 + LUT_UNROLL + u;
                            fetch(en[u], j < n ? j : n - 1);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < LUT_UNROLL; ++u)
                        if (j0 + u < n) add_site(ec[u], j0 + u);
#pragma unroll
                    for (int u = 0; u < LUT_UNROLL; ++u)
#pragma unroll
                        for (int q = 0; q < EW; ++q) ec[u][q] = en[u][q];
                }
#else
#pragma unroll LUT_UNROLL
                for (int j = 0; j < n; ++j) {
                    uint32_t e[EW];
                    fetch(e, j);
                    add_site(e, j);
                }
#endif
            };
            if (fast) {
                if (h < H) add_sites((int)n_all);
            } else {
                for (int64_t j0 = beg; j0 < end; j0 += LUT_CAP) {
                    const int n = (int)((end - j0) < LUT_CAP ? (end - j0) : LUT_CAP);
                    __syncthreads();
                    if (t < n) stage(cur, t, j0 + t);
                    __syncthreads();
                    if (h < H) add_sites(n);
                }
            }
            if (tile == 0 && t < nn) {                  // second hop of the next row's list
                nhit = lhit[nsite];
                nmiss = lmiss[nsite];
                ncode = obsmap[nobs];
            }
            if (h < H) {
                double *dst = M + (compact ? i : r) * ldm + h;
                if (vec_ok && h + CPL <= H) {
#pragma unroll
                    for (int c = 0; c < CPL; c += 2)
                        __builtin_nontemporal_store(d2{a[c], a[c + 1]}, reinterpret_cast<d2 *>(dst) + c / 2);
                } else {
#pragma unroll
                    for (int c = 0; c < CPL; ++c)
                        if (h + c < H) dst[c] = a[c];
                }
            }
        }
        if (t < nn) {                                   // third step: park it in the other buffer
            s_rec[cur ^ 1][t] = make_uint2((uint32_t)(nsite * (uint32_t)lde), ncode * 0x01010101u);
            d2 *dst = reinterpret_cast<d2 *>(&s_lut[cur ^ 1][t][0]);
            dst[0] = d2{nhit, nmiss};
#pragma unroll
            for (int q = 1; q < 8; ++q) dst[q] = d2{nmiss, nmiss};
        }
        have = nn > 0;

    }
}

#endif  // MIXEMT_BUILD_LUT_KERNELS_HPP
