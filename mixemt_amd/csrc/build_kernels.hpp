// build_kernels.hpp -- part of libmixemt_hip.so (gfx950); included by mixemt_hip.hip only.
// build_em_matrix (preprocess.py:177-198): byte-table kernel (any alphabet, any width; the fallback of the faster builds).
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

// (K1b, an LDS-staged 4-bit-table variant of this kernel, measured 76.8 ms against 36 ms and was removed in
// round 3; profiles/r01/build_kernels.txt keeps its numbers.)
#endif  // MIXEMT_BUILD_KERNELS_HPP
