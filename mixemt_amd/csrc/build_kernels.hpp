// build_kernels.hpp -- part of libmixemt_hip.so (gfx950); included by mixemt_hip.hip only.
// build_em_matrix (preprocess.py:177-198): byte-table kernel and LDS-staged packed-table kernel.
#ifndef MIXEMT_BUILD_KERNELS_HPP
#define MIXEMT_BUILD_KERNELS_HPP

// ------------------------------------------------------------------------------------------
// K1  build_em_matrix  (preprocess.py:177-198, :69-96)
// One workgroup per read row (grid-stride); a thread owns 4 adjacent haplogroup
// columns per 1024-column tile and adds the per-site terms IN SIGNATURE ORDER,
// so every cell is the same fp64 sum the reference forms.
// ------------------------------------------------------------------------------------------
#define BUILD_THREADS 256
#define BUILD_CAP 512              // observations staged in LDS per pass
#ifndef BUILD_UNROLL
#define BUILD_UNROLL 8             // table loads in flight per wave
#endif

__global__ __launch_bounds__(BUILD_THREADS) void build_em_matrix_kernel(
    const uint8_t *__restrict__ E, int64_t lde, const double *__restrict__ lhit,
    const double *__restrict__ lmiss, const int64_t *__restrict__ row_ptr,
    const uint16_t *__restrict__ site, const uint8_t *__restrict__ obs, int64_t R, int H,
    double *__restrict__ M, int64_t ldm, int vec_ok) {
    constexpr int BUILD_CPL = 4;                // columns per lane: one 4-byte table load per site
                                                // (8 per lane, as bytes or as 4-bit codes, measured slower)
    // The row's (table offset, observation, log hit, log miss) list lives in LDS, double
    // buffered: while row r is being added up, the threads fetch row r + grid's list (two
    // dependent global loads: site -> lhit[site]) and park it in the other buffer, so that
    // latency is never exposed.  Rows with more than BUILD_THREADS observations (multi-kb
    // fragments) take the staged-in-passes path below.
    __shared__ int64_t s_off[2][BUILD_CAP];     // site * lde
    __shared__ uint32_t s_obs[2][BUILD_CAP];
    __shared__ double s_hit[2][BUILD_CAP];
    __shared__ double s_miss[2][BUILD_CAP];
    const int t = threadIdx.x;
    int cur = 0;
    bool have = false;                          // buffer[cur] already holds this row's list
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x, cur ^= 1) {
        const int64_t beg = row_ptr[r], end = row_ptr[r + 1];
        const int64_t n_all = end - beg;
        const bool fast = n_all <= BUILD_THREADS;
        const int64_t rn = r + gridDim.x;
        const int64_t begn = (rn < R) ? row_ptr[rn] : 0;
        const int nn = (rn < R) ? (int)((row_ptr[rn + 1] - begn) <= BUILD_THREADS ? (row_ptr[rn + 1] - begn) : 0) : 0;
        if (fast && !have && t < (int)n_all) {
            const int s0 = site[beg + t];
            s_off[cur][t] = (int64_t)s0 * lde;
            s_obs[cur][t] = obs[beg + t];
            s_hit[cur][t] = lhit[s0];
            s_miss[cur][t] = lmiss[s0];
        }
        __syncthreads();
        // next row's list, first hop
        int sv = 0;
        uint32_t ov = 0;
        double hv = 0.0, mv = 0.0;
        if (t < nn) {
            sv = site[begn + t];
            ov = obs[begn + t];
        }
        for (int cb = 0; cb < H; cb += BUILD_THREADS * BUILD_CPL) {
            const int h = cb + BUILD_CPL * t;
            double a[BUILD_CPL];
#pragma unroll
            for (int c = 0; c < BUILD_CPL; ++c) a[c] = 0.0;
            if (fast) {
                const int n = (int)n_all;
                if (h < H) {
                    // independent table loads: unrolled so several are in flight per wave
#pragma unroll BUILD_UNROLL
                    for (int j = 0; j < n; ++j) {
                        const uint32_t e = *reinterpret_cast<const uint32_t *>(E + s_off[cur][j] + h);
                        const uint32_t o = s_obs[cur][j];
                        const double hit = s_hit[cur][j], miss = s_miss[cur][j];
#pragma unroll
                        for (int c = 0; c < BUILD_CPL; ++c)
                            a[c] += (((e >> (8 * c)) & 0xffu) == o) ? hit : miss;
                    }
                }
            } else {
                for (int64_t j0 = beg; j0 < end; j0 += BUILD_CAP) {
                    const int n = (int)((end - j0) < BUILD_CAP ? (end - j0) : BUILD_CAP);
                    __syncthreads();
                    for (int j = t; j < n; j += BUILD_THREADS) {
                        const int s0 = site[j0 + j];
                        s_off[cur][j] = (int64_t)s0 * lde;
                        s_obs[cur][j] = obs[j0 + j];
                        s_hit[cur][j] = lhit[s0];
                        s_miss[cur][j] = lmiss[s0];
                    }
                    __syncthreads();
                    if (h < H) {
#pragma unroll BUILD_UNROLL
                        for (int j = 0; j < n; ++j) {
                            const uint32_t e = *reinterpret_cast<const uint32_t *>(E + s_off[cur][j] + h);
                            const uint32_t o = s_obs[cur][j];
                            const double hit = s_hit[cur][j], miss = s_miss[cur][j];
#pragma unroll
                            for (int c = 0; c < BUILD_CPL; ++c)
                                a[c] += (((e >> (8 * c)) & 0xffu) == o) ? hit : miss;
                        }
                    }
                }
            }
            if (cb == 0 && t < nn) {            // second hop of the next row's list
                hv = lhit[sv];
                mv = lmiss[sv];
            }
            if (h < H) {
                double *dst = M + r * ldm + h;
                if (vec_ok && h + BUILD_CPL <= H) {
#pragma unroll
                    for (int c = 0; c < BUILD_CPL; c += 2) reinterpret_cast<d2 *>(dst)[c / 2] = d2{a[c], a[c + 1]};
                } else {
#pragma unroll
                    for (int c = 0; c < BUILD_CPL; ++c)
                        if (h + c < H) dst[c] = a[c];
                }
            }
        }
        if (t < nn) {
            s_off[cur ^ 1][t] = (int64_t)sv * lde;
            s_obs[cur ^ 1][t] = ov;
            s_hit[cur ^ 1][t] = hv;
            s_miss[cur ^ 1][t] = mv;
        }
        have = nn > 0;
    }
}

// ------------------------------------------------------------------------------------------
// K1b build_em_matrix, LDS-staged table form (alternative to K1: same sums, same order; measured
//     slower on MI355X -- 76.8 ms vs 40.4 ms at 1M x 5408 -- because it is VALU-bound, see
//     profiles/r01/build_kernels.txt; kept selectable and parity-tested).
//
// A workgroup owns a tile of 64 haplogroup columns and a chunk of rows.  Its slice of the
// expected-base table -- 4-bit codes, S sites x 32 bytes -- is staged into LDS once (130 KB
// for S = 4070), with the per-site (log hit, log miss) pair index and the observation-byte ->
// code map, so the R*H*k look-ups of preprocess.py:188-191 never leave the CU.  A wave works on
// 8 rows at a time: 8 lanes per row, each lane 8 adjacent columns (one LDS dword = 8 codes).
// The row's (site, observation) list is fetched 8 entries at a time, one entry per lane, and
// broadcast inside the 8-lane group with ds_bpermute; terms are added in signature order, so
// the result is bit-identical to the byte-table kernel and to the reference.
// Output: 64 B per lane, 512 B contiguous per row and tile.
// ------------------------------------------------------------------------------------------
#define TILE_THREADS 1024
#define TILE_COLS 64

__global__ __launch_bounds__(TILE_THREADS) void build_tile_kernel(
    const uint32_t *__restrict__ Epk, const uint8_t *__restrict__ muidx,
    const double *__restrict__ pairs, int n_mu, const uint8_t *__restrict__ obsmap,
    const int64_t *__restrict__ row_ptr, const uint16_t *__restrict__ site,
    const uint8_t *__restrict__ obs, int64_t R, int H, int S, double *__restrict__ M, int64_t ldm,
    int64_t rows_per_chunk, int vec_ok) {
    extern __shared__ uint32_t lds_tab[];
    // layout (dwords): E[(S+1)*8] | pairs[(n_mu+1)*4] | obsmap[64] | muidx[S+1 bytes]
    // Row S / pair n_mu are a NULL site: (hit, miss) = (+0.0, +0.0).  Rows shorter than the
    // longest row of their wave are padded with it -- x + 0.0 == x bit for bit -- so the
    // 8-entry inner block is straight-line code (no per-entry branch, LDS reads overlap).
    uint32_t *lds_e = lds_tab;
    double *lds_pairs = reinterpret_cast<double *>(lds_tab + ((size_t)S + 1) * 8);
    uint8_t *lds_map = reinterpret_cast<uint8_t *>(lds_tab + ((size_t)S + 1) * 8 + ((size_t)n_mu + 1) * 4);
    uint8_t *lds_mu = lds_map + 256;

    const int t = threadIdx.x;
    const int tile = blockIdx.x;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(Epk + (size_t)tile * S * 8);
        uint4 *dst = reinterpret_cast<uint4 *>(lds_e);
        for (int i = t; i < S * 2; i += TILE_THREADS) dst[i] = src[i];
        if (t < 8) lds_e[(size_t)S * 8 + t] = 0u;
        for (int i = t; i < n_mu * 2; i += TILE_THREADS) lds_pairs[i] = pairs[i];
        if (t < 2) lds_pairs[2 * n_mu + t] = 0.0;
        for (int i = t; i < 256; i += TILE_THREADS) lds_map[i] = obsmap[i];
        for (int i = t; i < S; i += TILE_THREADS) lds_mu[i] = muidx[i];
        if (t == 0) lds_mu[S] = (uint8_t)n_mu;
    }
    __syncthreads();

    const int lane = t & 63, wv = t >> 6;
    const int g = lane >> 3, j = lane & 7;
    const int group_base = lane & ~7;
    const uint32_t null_entry = (uint32_t)S | (15u << 16);
    const int64_t c0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t c1 = (c0 + rows_per_chunk < R) ? (c0 + rows_per_chunk) : R;
    // column of accumulator n: 16 * (n / 2) + 2 * j + (n % 2)  (so each store below covers a
    // full 128-byte line per row across the row's 8 lanes)
    const int col0 = tile * TILE_COLS + 2 * j;

    for (int64_t base = c0 + (int64_t)wv * 8; base < c1; base += (TILE_THREADS / 64) * 8) {
        const int64_t r = base + g;
        const bool live = r < c1;
        const int64_t beg = live ? row_ptr[r] : 0;
        const int n = live ? (int)(row_ptr[r + 1] - beg) : 0;
        int nmax = n;
#pragma unroll
        for (int off = 32; off >= 8; off >>= 1) {
            const int o = __shfl_xor(nmax, off, 64);
            nmax = o > nmax ? o : nmax;
        }
        double acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.0;

        auto fetch = [&](int k0) -> uint32_t {
            const int kk = k0 + j;
            if (kk < n) return (uint32_t)site[beg + kk] | ((uint32_t)lds_map[obs[beg + kk]] << 16);
            return null_entry;
        };
        uint32_t mine = fetch(0);
        for (int k0 = 0; k0 < nmax; k0 += 8) {
            const uint32_t next = fetch(k0 + 8);            // in flight while this block is added
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const uint32_t e = (uint32_t)__shfl((int)mine, group_base | jj, 64);
                const uint32_t s = e & 0xffffu, oc = e >> 16;
                const uint32_t ew = lds_e[s * 8 + j];
                const uint32_t mi = lds_mu[s];
                const double hit = lds_pairs[2 * mi], miss = lds_pairs[2 * mi + 1];
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    acc[c] += (((ew >> (4 * c)) & 15u) == oc) ? hit : miss;
            }
            mine = next;
        }
        if (live) {
            double *dst = M + r * ldm + col0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = col0 + 16 * i;
                if (vec_ok && col + 2 <= H) {
                    *reinterpret_cast<d2 *>(dst + 16 * i) = d2{acc[2 * i], acc[2 * i + 1]};
                } else {
                    if (col < H) dst[16 * i] = acc[2 * i];
                    if (col + 1 < H) dst[16 * i + 1] = acc[2 * i + 1];
                }
            }
        }
    }
}

#endif  // MIXEMT_BUILD_KERNELS_HPP
