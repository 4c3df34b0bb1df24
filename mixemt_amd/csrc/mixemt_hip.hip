// mixemt_hip.hip -- gfx950 (MI355X, CDNA4) kernels + C ABI for mixemt's EM hot path.
//
// Replaces, behind include/mixemt_hip.h:
//   preprocess.build_em_matrix   /root/reference/mixemt/preprocess.py:177-198
//   em.em_step / em.run_em loop  /root/reference/mixemt/em.py:57-91, :126-143
//
// Everything here is HBM-bound byte/fp64 streaming + reductions: 64-wide
// wavefronts, 16-byte coalesced loads, rows held in VGPRs between the row
// reduction and the column accumulation, deterministic two-stage column sums
// (no float atomics).  No MFMA: there is no contraction to feed it.
//
// Kernel map (DESIGN.md has the roofline of each):
//   build_em_matrix_kernel  R*H*8 B written, E table served from L2/MALL
//   linearize_kernel        one-time  P = exp(M - rowmax)
//   em_iter_wide_kernel     THE hot kernel: R*H*8 B read per EM iteration
//   estep_log_kernel        reference-semantics E-step (posterior pass, small H)
//   colreduce_kernel        [nWG][H] partials -> colsum[H], fixed order
//   finalize_kernel         normalise, L1 test, loop state
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

#include "mixemt_hip.h"

typedef double d2 __attribute__((ext_vector_type(2)));

#define MXM_MAX_WG 1024            // upper bound on the persistent grid (workspace sizing)
#define MXM_WIDE_THREADS 256
#define MXM_WIDE_MAX_NCH 16        // double2 chunks per thread -> H <= 2*256*16 = 8192
#define MXM_LINEAR_MIN_H 65        // below this the log-space kernel is used

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, const char *a = "", long long b = 0, long long c = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, b, c);
    return code;
}
#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail(-2, "HIP error: %s (line %lld)", hipGetErrorString(e_), __LINE__); \
    } while (0)

static int g_num_cu = 0;
static int num_cu() {
    if (g_num_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        g_num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return g_num_cu;
}

// ------------------------------------------------------------------------------------------
// wave / workgroup reductions (wave = 64 lanes on gfx950)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// all threads get the result; `scratch` holds THREADS/64 doubles; two barriers
template <int THREADS, bool IS_MAX>
__device__ __forceinline__ double block_reduce(double v, double *scratch) {
    constexpr int NW = THREADS / 64;
    v = IS_MAX ? wave_max(v) : wave_sum(v);
    __syncthreads();                       // scratch free (previous use finished)
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = scratch[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) r = IS_MAX ? fmax(r, scratch[i]) : r + scratch[i];
    return r;
}

__device__ __forceinline__ double logaddexp_f64(double a, double b) {
    // numpy.logaddexp semantics (em.py:156)
    if (a == b) return a + 0.693147180559945309417232121458176568;   // covers +-inf ties
    double d = a - b;
    if (d > 0) return a + log1p(exp(-d));
    if (d <= 0) return b + log1p(exp(d));
    return d;                                                         // NaN
}

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
#define BUILD_CPL 4                // haplogroup columns per lane (one 4-byte table load per site;
                                   // 8 per lane measured slower: 56 ms vs 40.6 ms at 1M x 5408)

__global__ __launch_bounds__(BUILD_THREADS) void build_em_matrix_kernel(
    const uint8_t *__restrict__ E, int64_t lde, const double *__restrict__ lhit,
    const double *__restrict__ lmiss, const int64_t *__restrict__ row_ptr,
    const uint16_t *__restrict__ site, const uint8_t *__restrict__ obs, int64_t R, int H,
    double *__restrict__ M, int64_t ldm, int vec_ok) {
    __shared__ int64_t s_off[BUILD_CAP];     // site * lde
    __shared__ uint32_t s_obs[BUILD_CAP];
    __shared__ double s_hit[BUILD_CAP];
    __shared__ double s_miss[BUILD_CAP];
    const int t = threadIdx.x;
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
        const int64_t beg = row_ptr[r], end = row_ptr[r + 1];
        for (int cb = 0; cb < H; cb += BUILD_THREADS * BUILD_CPL) {
            const int h = cb + BUILD_CPL * t;
            double a[BUILD_CPL];
#pragma unroll
            for (int c = 0; c < BUILD_CPL; ++c) a[c] = 0.0;
            for (int64_t j0 = beg; j0 < end; j0 += BUILD_CAP) {
                const int n = (int)((end - j0) < BUILD_CAP ? (end - j0) : BUILD_CAP);
                __syncthreads();
                for (int j = t; j < n; j += BUILD_THREADS) {
                    const int s = site[j0 + j];
                    s_off[j] = (int64_t)s * lde;
                    s_obs[j] = obs[j0 + j];
                    s_hit[j] = lhit[s];
                    s_miss[j] = lmiss[s];
                }
                __syncthreads();
                if (h < H) {
                    // independent table loads: unrolled so several are in flight per wave
#pragma unroll BUILD_UNROLL
                    for (int j = 0; j < n; ++j) {
                        const uint32_t *src = reinterpret_cast<const uint32_t *>(E + s_off[j] + h);
                        uint32_t e4[BUILD_CPL / 4];
#pragma unroll
                        for (int q = 0; q < BUILD_CPL / 4; ++q) e4[q] = src[q];
                        const uint32_t o = s_obs[j];
                        const double hit = s_hit[j], miss = s_miss[j];
#pragma unroll
                        for (int c = 0; c < BUILD_CPL; ++c)
                            a[c] += (((e4[c / 4] >> (8 * (c % 4))) & 0xffu) == o) ? hit : miss;
                    }
                }
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
    }
}

// ------------------------------------------------------------------------------------------
// K1b build_em_matrix, LDS-staged table form (the fast path; same sums, same order).
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
// K3  em_iter_wide: fused E+M step in linear space, one restart.
//
//   Z_r      = sum_h p_h P_rh                     (row reduction)
//   acc_h   += (w_r / Z_r) * P_rh                 (column accumulation, per workgroup)
//   colsum_h = p_h * sum_wg acc_h                 (colreduce_kernel)
//
// which is em.py:80-88 with exp(M - rowmax) hoisted out of the loop:
//   posterior_rh = p_h P_rh / Z_r,   colsum_h = sum_r w_r posterior_rh.
//
// A workgroup (256 threads) owns a contiguous block of rows.  Thread t owns the
// double2 column pairs {t + 256 k}, k < NCH: one 16-byte load per pair per row
// (a wave instruction covers 1 KiB contiguous), the row stays in VGPRs between
// the dot product and the accumulation, so the matrix is read from HBM exactly
// once per iteration.  The next row's loads are issued before the current row's
// reduction (register double buffer).  BT restarts can share each row: every
// restart adds its own p / accumulator registers, the bytes read stay the same
// (BT = 1: 256 threads, 2 workgroups per CU; BT = 2, 3: 512 threads, 1 per CU).
// Column partials live in registers for the whole kernel and are written once:
// partial[wg][h], summed in fixed order afterwards -> bitwise reproducible.
// ------------------------------------------------------------------------------------------
#ifndef MXM_V1_MINW
#define MXM_V1_MINW 2                 // min waves/SIMD the BT = 1 shape is compiled for (2 WGs of 256 per CU)
#endif
#ifndef MXM_V1_P_LDS
#define MXM_V1_P_LDS 0                // 1: the single-restart shape also keeps its proportions in LDS
#endif
#ifndef MXM_SCHED_FENCE
#define MXM_SCHED_FENCE 0
#endif
// batched shapes run one workgroup per CU: min waves/SIMD = THREADS / 256

template <int THREADS, int NCH, int BT, int NBUF>
__global__ __launch_bounds__(THREADS, (BT == 1 ? MXM_V1_MINW : THREADS / 256)) void em_iter_wide_kernel(
    const double *__restrict__ P, int64_t ldp, const double *__restrict__ w,
    const double *__restrict__ props, int64_t R, int H, int64_t rows_per_wg,
    double *__restrict__ partial, int64_t ldpart, const mxm_em_state *__restrict__ state) {
    constexpr int NW = THREADS / 64;
    __shared__ double red[2][BT][NW];
    if (state != nullptr) {
        bool any = false;
#pragma unroll
        for (int b = 0; b < BT; ++b) any = any || (state[b].done == 0);
        if (!any) return;                           // every restart of this tile has stopped
    }

    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int ncol2 = (H + 1) >> 1;                 // d2 pairs per row (pad column is 0 in P)

    // proportions: registers for a single restart; for a batch they sit in LDS as
    // [b][k][thread] pairs (one conflict-free ds_read_b128 per use) so that the VGPR
    // budget goes to the accumulators and the row double buffer
    extern __shared__ d2 lds_p[];
    constexpr bool P_IN_LDS = (BT > 1) || (MXM_V1_P_LDS != 0);
    d2 p[P_IN_LDS ? 1 : NCH], acc[BT][NCH];
#pragma unroll
    for (int b = 0; b < BT; ++b) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = 2 * (t + k * THREADS);
            d2 v;
            v.x = (c < H) ? props[(int64_t)b * H + c] : 0.0;
            v.y = (c + 1 < H) ? props[(int64_t)b * H + c + 1] : 0.0;
            if constexpr (!P_IN_LDS) p[k] = v;
            else lds_p[(b * NCH + k) * THREADS + t] = v;
            acc[b][k] = d2{0.0, 0.0};
        }
    }

    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t r1 = (r0 + rows_per_wg < R) ? (r0 + rows_per_wg) : R;
    if (r0 >= r1) return;

    // Row loads: buffer_load_dwordx4 through one descriptor over this workgroup's row block.
    // Per-lane offset = one VGPR (t * 16), row and chunk offsets are scalar, so no 64-bit
    // per-load addresses and no exec-masked branches: rows past the block and column pairs
    // past the row are CLAMPED to a valid element instead of skipped -- a clamped row gets
    // weight 0 below, a clamped column has p = 0 and its accumulator is never stored.
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(P + r0 * ldp), 0, (int)((r1 - r0) * ldp * 8), 0x00020000);
    const int row_bytes = (int)(ldp * 8);
    const int voff = t * 16;
    int last_c2 = t + (NCH - 1) * THREADS;
    if (last_c2 > ncol2 - 1) last_c2 = ncol2 - 1;
    const int voff_last = last_c2 * 16;

    d2 x[NBUF][NCH];                                // register ring: NBUF - 1 rows in flight

    auto load_row = [&](d2(&xr)[NCH], int64_t r) {
        const int64_t rr = (r < r1) ? r : (r1 - 1);
        const int soff = (int)(rr - r0) * row_bytes;
#pragma unroll
        for (int k = 0; k < NCH - 1; ++k)
            xr[k] = __builtin_bit_cast(d2, (u4)__builtin_amdgcn_raw_buffer_load_b128(
                                               rsrc, voff, soff + k * THREADS * 16, 2 /* nt */));
        xr[NCH - 1] = __builtin_bit_cast(
            d2, (u4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_last, soff, 2 /* nt */));
    };

    int buf = 0;
    auto process = [&](d2(&xr)[NCH], int64_t r) {
        double d[BT];
        // keep the batch's proportions IN LDS: without this the loads are loop-invariant
        // and get hoisted back into (BT * NCH * 4) VGPRs
        if constexpr (P_IN_LDS) asm volatile("" ::: "memory");
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                d2 pk;
                if constexpr (!P_IN_LDS) pk = p[k];
                else pk = lds_p[(b * NCH + k) * THREADS + t];     // own slot: no barrier needed
                s = fma(xr[k].x, pk.x, s);
                s = fma(xr[k].y, pk.y, s);
            }
            d[b] = s;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int b = 0; b < BT; ++b) d[b] += __shfl_xor(d[b], off, 64);
        }
        if (lane == 0) {
#pragma unroll
            for (int b = 0; b < BT; ++b) red[buf][b][wv] = d[b];
        }
        __syncthreads();
        const bool live = r < r1;
        const double wr = live ? (w != nullptr ? w[r] : 1.0) : 0.0;
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            double z = red[buf][b][0];
#pragma unroll
            for (int q = 1; q < NW; ++q) z += red[buf][b][q];
            const double c = (z > 0.0) ? wr / z : 0.0;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                acc[b][k].x = fma(c, xr[k].x, acc[b][k].x);
                acc[b][k].y = fma(c, xr[k].y, acc[b][k].y);
            }
        }
        buf ^= 1;
    };

#pragma unroll
    for (int j = 0; j < NBUF - 1; ++j) load_row(x[j], r0 + j);
    for (int64_t r = r0; r < r1; r += NBUF) {
#pragma unroll
        for (int j = 0; j < NBUF; ++j) {
            load_row(x[(j + NBUF - 1) % NBUF], r + j + NBUF - 1);
#if MXM_SCHED_FENCE
            // keep the scheduler from hoisting these loads above the previous row's last
            // uses of the same ring slot (it would need a second register set for it)
            __builtin_amdgcn_sched_barrier(0);
#endif
            process(x[j], r + j);
#if MXM_SCHED_FENCE
            __builtin_amdgcn_sched_barrier(0);
#endif
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
    const double *__restrict__ props, int64_t R, int H, int64_t rows_per_wg,
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
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t r1 = (r0 + rows_per_wg < R) ? (r0 + rows_per_wg) : R;
    if (r0 >= r1) return;

    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P + r0 * ldp), 0,
                                                        (int)((r1 - r0) * ldp * 4), 0x00020000);
    const int row_bytes = (int)(ldp * 4);
    const int voff = t * 16;
    int last_c4 = t + (NCH - 1) * THREADS;
    if (last_c4 > ncol4 - 1) last_c4 = ncol4 - 1;
    const int voff_last = last_c4 * 16;

    f4 x[NBUF][NCH];
    auto load_row = [&](f4(&xr)[NCH], int64_t r) {
        const int64_t rr = (r < r1) ? r : (r1 - 1);
        const int soff = (int)(rr - r0) * row_bytes;
#pragma unroll
        for (int k = 0; k < NCH - 1; ++k)
            xr[k] = __builtin_bit_cast(f4, (u4)__builtin_amdgcn_raw_buffer_load_b128(
                                               rsrc, voff, soff + k * THREADS * 16, 2));
        xr[NCH - 1] = __builtin_bit_cast(f4, (u4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_last, soff, 2));
    };

    int buf = 0;
    auto process = [&](f4(&xr)[NCH], int64_t r) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s = fma((double)xr[k][e], p[k][e], s);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) red[buf][wv] = s;
        __syncthreads();
        double z = red[buf][0];
#pragma unroll
        for (int q = 1; q < NW; ++q) z += red[buf][q];
        buf ^= 1;
        const bool live = r < r1;
        const double wr = live ? (w != nullptr ? w[r] : 1.0) : 0.0;
        const double c = (z > 0.0) ? wr / z : 0.0;
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
    for (int j = 0; j < NBUF - 1; ++j) load_row(x[j], r0 + j);
    for (int64_t r = r0; r < r1; r += NBUF) {
#pragma unroll
        for (int j = 0; j < NBUF; ++j) {
            load_row(x[(j + NBUF - 1) % NBUF], r + j + NBUF - 1);
            process(x[j], r + j);
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
// scale_h = props[h] for the linear kernel, 1 for the log-space kernel.
// ------------------------------------------------------------------------------------------
#define COLRED_THREADS 1024
__global__ __launch_bounds__(COLRED_THREADS) void colreduce_kernel(const double *__restrict__ partial,
                                                                   int64_t ldpart, int nwg, int nb, int H,
                                                                   const double *__restrict__ props,
                                                                   double *__restrict__ colsum,
                                                                   const mxm_em_state *__restrict__ state) {
    // grid = (ceil(H/64), nb); partial is [nwg][nb][ldpart]; props / colsum are [nb][H].
    // 16 waves take interleaved sixteenths of the partial rows, four independent chains each
    // (the loads are what this kernel waits for); every order below is fixed -> deterministic.
    constexpr int NW = COLRED_THREADS / 64;
    __shared__ double part[NW][64];
    const int b = blockIdx.y;
    if (state != nullptr && state[b].done != 0) return;
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
        colsum[(int64_t)b * H + h] = (props != nullptr) ? props[(int64_t)b * H + h] * tot : tot;
    }
}

// ------------------------------------------------------------------------------------------
// K5  finalize: em.py:89 (normalise), :39-54 (L1 test), :133-143 (loop state). One WG per restart.
// ------------------------------------------------------------------------------------------
#define FIN_THREADS 1024

__global__ __launch_bounds__(FIN_THREADS) void finalize_kernel(const double *__restrict__ colsum,
                                                               double *__restrict__ props_cur,
                                                               double *__restrict__ props_new, int H,
                                                               double tol, int max_iter,
                                                               mxm_em_state *__restrict__ state) {
    __shared__ double scratch[FIN_THREADS / 64];
    const int b = blockIdx.x;
    mxm_em_state *st = state + b;
    if (st->done != 0) return;
    const double *cs = colsum + (int64_t)b * H;
    double *pc = props_cur + (int64_t)b * H;
    double *pn = props_new + (int64_t)b * H;
    const int t = threadIdx.x;
    double s = 0.0;
    for (int h = t; h < H; h += FIN_THREADS) s += cs[h];
    const double total = block_reduce<FIN_THREADS, false>(s, scratch);
    double l1 = 0.0;
    for (int h = t; h < H; h += FIN_THREADS) {
        const double v = cs[h] / total;
        pn[h] = v;
        l1 += fabs(v - pc[h]);
    }
    l1 = block_reduce<FIN_THREADS, false>(l1, scratch);
    const int iters = st->iters + 1;
    const bool conv = l1 < tol;
    const bool stop = conv || iters >= max_iter;
    if (!stop)
        for (int h = t; h < H; h += FIN_THREADS) pc[h] = pn[h];
    __syncthreads();
    if (t == 0) {
        st->iters = iters;
        st->l1 = l1;
        st->done = conv ? 1 : (stop ? 2 : 0);
    }
}

// ------------------------------------------------------------------------------------------
// K6  estep_log: the reference's E-step verbatim in log space (em.py:80-83), optional posterior
// write / logaddexp fold (em.py:156) and optional M-step sums (em.py:87-88, linear space).
// One workgroup per row (grid-stride), column sums in LDS (each thread owns its columns).
// Used for em_step(), the final posterior pass, and EM iterations when H is tiny.
// ------------------------------------------------------------------------------------------
template <bool FROM_LINEAR_PROPS>
__global__ __launch_bounds__(ROW_THREADS) void estep_log_kernel(
    const double *__restrict__ M, int64_t ldm, const double *__restrict__ w,
    const double *__restrict__ pvec, int64_t R, int H, double *__restrict__ out, int64_t ldo,
    int mode, double *__restrict__ partial, int64_t ldpart,
    const mxm_em_state *__restrict__ state) {
    extern __shared__ double dyn[];            // [H] column sums (if partial) + [H] ln props
    __shared__ double scratch[ROW_THREADS / 64];
    if (state != nullptr && state->done != 0) return;
    const int t = threadIdx.x;
    double *lnp = dyn;
    double *acc = dyn + H;
    for (int h = t; h < H; h += ROW_THREADS) {
        lnp[h] = FROM_LINEAR_PROPS ? log(pvec[h]) : pvec[h];
        if (partial != nullptr) acc[h] = 0.0;
    }
    __syncthreads();
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
        const double *src = M + r * ldm;
        double m = -INFINITY;
        for (int h = t; h < H; h += ROW_THREADS) m = fmax(m, lnp[h] + src[h]);
        m = block_reduce<ROW_THREADS, true>(m, scratch);
        const double shift = isfinite(m) ? m : 0.0;
        double s = 0.0;
        for (int h = t; h < H; h += ROW_THREADS) s += exp((lnp[h] + src[h]) - shift);
        s = block_reduce<ROW_THREADS, false>(s, scratch);
        const double lse = log(s) + m;          // m (not shift): -inf rows give -inf, as scipy does
        const double wr = (w != nullptr) ? w[r] : 1.0;
        for (int h = t; h < H; h += ROW_THREADS) {
            const double v = (lnp[h] + src[h]) - lse;
            if (out != nullptr) {
                double *o = out + r * ldo + h;
                *o = (mode == 1) ? logaddexp_f64(*o, v) : v;
            }
            if (partial != nullptr) acc[h] += wr * exp(v);
        }
    }
    if (partial != nullptr) {
        double *dst = partial + (int64_t)blockIdx.x * ldpart;
        for (int h = t; h < H; h += ROW_THREADS) dst[h] = acc[h];
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
    const double *__restrict__ lnp_in, int64_t R, int H, int64_t rows_per_wg,
    double *__restrict__ out, int64_t ldo, int mode, double *__restrict__ partial, int64_t ldpart) {
    constexpr int THREADS = 256, NW = THREADS / 64;
    __shared__ double red[2][2][NW];               // [ring][max|sum][wave]
    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int ncol2 = H >> 1;

    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t r1 = (r0 + rows_per_wg < R) ? (r0 + rows_per_wg) : R;
    if (r0 >= r1) return;

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
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(M + r0 * ldm), 0,
                                                        (int)((r1 - r0) * ldm * 8), 0x00020000);
    const int row_bytes = (int)(ldm * 8);
    const int voff = t * 16;
    int last_c2 = t + (NCH - 1) * THREADS;
    if (last_c2 > ncol2 - 1) last_c2 = ncol2 - 1;
    const int voff_last = last_c2 * 16;

    d2 x[NBUF][NCH];
    auto load_row = [&](d2(&xr)[NCH], int64_t r) {
        const int64_t rr = (r < r1) ? r : (r1 - 1);
        const int soff = (int)(rr - r0) * row_bytes;
#pragma unroll
        for (int k = 0; k < NCH - 1; ++k)
            xr[k] = __builtin_bit_cast(d2, (u4)__builtin_amdgcn_raw_buffer_load_b128(
                                               rsrc, voff, soff + k * THREADS * 16, 2));
        xr[NCH - 1] = __builtin_bit_cast(d2, (u4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_last, soff, 2));
    };

    int ring = 0;
    auto process = [&](d2(&xr)[NCH], int64_t r) {
        const bool live = r < r1;
        double m = -INFINITY;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            xr[k].x += lp[k].x;                    // z = ln p + M   (em.py:80)
            xr[k].y += lp[k].y;
            m = fmax(m, fmax(xr[k].x, xr[k].y));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
        if (lane == 0) red[ring][0][wv] = m;
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
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ssum += __shfl_xor(ssum, off, 64);
        if (lane == 0) red[ring][1][wv] = ssum;
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
            const double c = (ssum > 0.0) ? wr / ssum : 0.0;     // w * exp(z - lse) = w * e / sum
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                acc[k].x = fma(c, e[k].x, acc[k].x);
                acc[k].y = fma(c, e[k].y, acc[k].y);
            }
        }
    };

    if constexpr (NBUF == 2) {
        load_row(x[0], r0);
        for (int64_t r = r0; r < r1; r += 2) {
            load_row(x[1], r + 1);
            process(x[0], r);
            load_row(x[0], r + 2);
            process(x[1], r + 1);
        }
    } else {
        for (int64_t r = r0; r < r1; ++r) {
            load_row(x[0], r);
            process(x[0], r);
        }
    }
    if constexpr (COLSUM) {
        d2 *dst = reinterpret_cast<d2 *>(partial + (int64_t)blockIdx.x * ldpart);
#pragma unroll
        for (int k = 0; k < NCH; ++k)
            if (own[k]) dst[t + k * THREADS] = acc[k];
    }
}

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
    if (an != bn) return an > bn;
    if (an == 0 && av != bv) return av > bv;
    return ai < bi;
}

__global__ __launch_bounds__(ROW_THREADS) void row_argmax_votes_kernel(
    const double *__restrict__ X, int64_t ldx, const double *__restrict__ w, int64_t R, int H,
    int32_t *__restrict__ best, double *__restrict__ votes) {
    constexpr int NW = ROW_THREADS / 64;
    __shared__ double s_val[NW];
    __shared__ int s_idx[NW];
    __shared__ int s_nan[NW];
    const int t = threadIdx.x;
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
            if (votes != nullptr) atomicAdd(votes + ci, (w != nullptr) ? w[r] : 1.0);
        }
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

// ------------------------------------------------------------------------------------------
// host side of the C ABI
// ------------------------------------------------------------------------------------------
static inline int clamp_grid(int64_t want, int cap) {
    if (want < 1) want = 1;
    return (int)(want < cap ? want : cap);
}

static inline int64_t part_ld(int H) { return ((int64_t)H + 1) & ~(int64_t)1; }

extern "C" int mxm_version(void) { return MXM_VERSION; }
extern "C" const char *mxm_last_error(void) { return g_err; }

extern "C" int mxm_linear_supported(int32_t H) {
    return (H >= MXM_LINEAR_MIN_H && H <= 8192) ? 1 : 0;
}

extern "C" size_t mxm_workspace_bytes(int64_t R, int32_t H, int32_t B) {
    (void)R;
    (void)B;                      // restart tiles are processed one after another over the same scratch
    return (size_t)MXM_MAX_WG * 3 * (size_t)part_ld(H) * sizeof(double);
}

extern "C" int mxm_build_em_matrix(const uint8_t *E, int64_t lde, const double *lhit,
                                   const double *lmiss, const int64_t *row_ptr, const uint16_t *site,
                                   const uint8_t *obs, int64_t R, int32_t H, int32_t S, double *M,
                                   int64_t ldm, void *stream) {
    if (R < 0 || H <= 0 || S <= 0) return fail(-1, "mxm_build_em_matrix: bad shape R=%s%lld H=%lld", "", R, H);
    if (lde < (((int64_t)H + 7) & ~(int64_t)7) || (lde & 7) != 0 || (reinterpret_cast<uintptr_t>(E) & 7) != 0)
        return fail(-1, "mxm_build_em_matrix: E must be 8-byte aligned with lde a multiple of 8 and >= H rounded up to 8%s (lde=%lld)", "", lde);
    if (ldm < H) return fail(-1, "mxm_build_em_matrix: ldm < H%s", "");
    if (S > 65536) return fail(-1, "mxm_build_em_matrix: more than 65536 variant sites%s", "");
    if (R == 0) return 0;
    const int grid = clamp_grid(R, num_cu() * 8);
    const int vec_ok = ((ldm & 1) == 0) && ((reinterpret_cast<uintptr_t>(M) & 15) == 0);
    hipLaunchKernelGGL(build_em_matrix_kernel, dim3(grid), dim3(BUILD_THREADS), 0, (hipStream_t)stream, E,
                       lde, lhit, lmiss, row_ptr, site, obs, R, (int)H, M, ldm, vec_ok);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" size_t mxm_build_packed_lds_bytes(int32_t S, int32_t n_mu) {
    return ((size_t)S + 1) * 32 + ((size_t)n_mu + 1) * 16 + 256 + (((size_t)S + 4) & ~(size_t)3);
}

extern "C" int mxm_build_em_matrix_packed(const uint32_t *Epk, const uint8_t *muidx, const double *pairs,
                                          int32_t n_mu, const uint8_t *obsmap, const int64_t *row_ptr,
                                          const uint16_t *site, const uint8_t *obs, int64_t R, int32_t H,
                                          int32_t S, double *M, int64_t ldm, void *stream) {
    if (R < 0 || H <= 0 || S <= 0 || n_mu <= 0 || n_mu > 255)
        return fail(-1, "mxm_build_em_matrix_packed: bad shape R=%s%lld H=%lld", "", R, H);
    if (ldm < H) return fail(-1, "mxm_build_em_matrix_packed: ldm < H%s", "");
    if (S > 65536) return fail(-1, "mxm_build_em_matrix_packed: more than 65536 variant sites%s", "");
    const size_t lds = mxm_build_packed_lds_bytes(S, n_mu);
    if (lds > 158 * 1024) return fail(-1, "mxm_build_em_matrix_packed: tables need %s%lld B of LDS (> 158 KiB); use mxm_build_em_matrix", "", (long long)lds);
    if (R == 0) return 0;
    static bool raised = false;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&build_tile_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised = true;
    }
    const int ntiles = (H + TILE_COLS - 1) / TILE_COLS;
    // one workgroup per CU (the table slice fills the LDS): aim at ~2 waves of workgroups
    int nchunks = (2 * num_cu() + ntiles - 1) / ntiles;
    if (nchunks < 1) nchunks = 1;
    const int64_t max_chunks = (R + 127) / 128;
    if (nchunks > max_chunks) nchunks = (int)max_chunks;
    int64_t rows_per_chunk = (R + nchunks - 1) / nchunks;
    rows_per_chunk = (rows_per_chunk + 127) / 128 * 128;
    nchunks = (int)((R + rows_per_chunk - 1) / rows_per_chunk);
    const int vec_ok = ((ldm & 1) == 0) && ((reinterpret_cast<uintptr_t>(M) & 15) == 0);
    hipLaunchKernelGGL(build_tile_kernel, dim3(ntiles, nchunks), dim3(TILE_THREADS), lds, (hipStream_t)stream,
                       Epk, muidx, pairs, (int)n_mu, obsmap, row_ptr, site, obs, R, (int)H, (int)S, M, ldm,
                       rows_per_chunk, vec_ok);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_linearize(const double *M, int64_t ldm, int64_t R, int32_t H, double *P, int64_t ldp,
                             double *rowmax, void *stream) {
    if (R < 0 || H <= 0) return fail(-1, "mxm_linearize: bad shape%s", "");
    if (ldm < H || ldp < H || (ldp & 1)) return fail(-1, "mxm_linearize: ldp must be even and >= H%s (ldp=%lld)", "", ldp);
    if (R == 0) return 0;
    const int grid = clamp_grid(R, num_cu() * 8);
    hipLaunchKernelGGL(linearize_kernel, dim3(grid), dim3(ROW_THREADS), 0, (hipStream_t)stream, M, ldm, R,
                       (int)H, P, ldp, rowmax);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- optional timing hook (bench.py): events recorded right around the dominant kernel --------
static hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
static int g_min_rows_per_wg = 8;     // measured: 1000 x 5408 31 us/step (vs 39 at 2); no effect from 10^4 rows up
static int g_max_bt = 3;              // restarts per matrix pass (1..MXM_MAX_BT); see mxm_set_batch_tile
extern "C" int mxm_set_timing_events(void *ev_start, void *ev_stop) {
    g_ev_start = (hipEvent_t)ev_start;
    g_ev_stop = (hipEvent_t)ev_stop;
    return 0;
}

extern "C" int mxm_set_min_rows_per_wg(int32_t n) {
    if (n < 1) return fail(-1, "mxm_set_min_rows_per_wg: n < 1%s", "");
    g_min_rows_per_wg = n;
    return 0;
}

extern "C" int mxm_set_batch_tile(int32_t bt) {
    if (bt < 1 || bt > 3) return fail(-1, "mxm_set_batch_tile: tile must be 1..3%s", "");
    g_max_bt = bt;
    return 0;
}

// ---- wide-kernel dispatch over NCH ------------------------------------------------------------
// Tuned shapes of the streaming kernel per batch size: workgroup threads, register ring depth
// (rows in flight = NBUF - 1 per workgroup), workgroups per CU of the persistent grid.
#ifndef MXM_V1_THREADS
#define MXM_V1_THREADS 256
#endif
#ifndef MXM_V1_NBUF
#define MXM_V1_NBUF 2
#endif
#ifndef MXM_V1_WG_PER_CU
#define MXM_V1_WG_PER_CU 2
#endif
#ifndef MXM_V2_THREADS
#define MXM_V2_THREADS 512
#endif
#ifndef MXM_V2_NBUF
#define MXM_V2_NBUF 3                 // measured: 6.22 ms vs 6.67 ms per pass at 1M x 5408 (profiles/r01/tune_sweep.txt)
#endif
#ifndef MXM_V3_THREADS
#define MXM_V3_THREADS 512
#endif
#ifndef MXM_V3_NBUF
#define MXM_V3_NBUF 2
#endif
#define MXM_MAX_BT 3                  // restarts sharing one read of the matrix
#define MXM_MAX_COL2 4096             // column pairs per row the register tiling covers (H <= 8192)
#define MXM_LDS_BUDGET (156 * 1024)   // of the CU's 160 KiB, leaving room for the exchange buffers

static inline int variant_threads(int nb) { return nb == 1 ? MXM_V1_THREADS : (nb == 2 ? MXM_V2_THREADS : MXM_V3_THREADS); }
static inline int variant_nbuf(int nb) { return nb == 1 ? MXM_V1_NBUF : (nb == 2 ? MXM_V2_NBUF : MXM_V3_NBUF); }

static size_t batch_lds_bytes(int H, int nb) {
    const int threads = variant_threads(nb);
    const int nch = ((H + 1) / 2 + threads - 1) / threads;
    return (size_t)nb * nch * threads * 16;
}

template <int THREADS, int NCH, int BT, int NBUF>
static int launch_wide(const double *P, int64_t ldp, const double *w, const double *props, int64_t R,
                       int H, int grid, int64_t rows_per_wg, double *partial, int64_t ldpart,
                       const mxm_em_state *state, hipStream_t stream) {
    if constexpr (NCH * THREADS > MXM_MAX_COL2) {
        return fail(-1, "mxm_em_iter: H=%s%lld outside the linear kernel's range", "", H);
    } else {
        const bool p_in_lds = (BT > 1) || (MXM_V1_P_LDS != 0);
        const size_t lds = p_in_lds ? (size_t)BT * NCH * THREADS * sizeof(d2) : 0;
        if (p_in_lds) {
            static bool raised = false;     // > 64 KiB of dynamic LDS must be opted into, once per kernel
            if (!raised) {
                (void)hipFuncSetAttribute(
                    reinterpret_cast<const void *>(&em_iter_wide_kernel<THREADS, NCH, BT, NBUF>),
                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                raised = true;
            }
        }
        hipLaunchKernelGGL((em_iter_wide_kernel<THREADS, NCH, BT, NBUF>), dim3(grid), dim3(THREADS), lds,
                           stream, P, ldp, w, props, R, H, rows_per_wg, partial, ldpart, state);
        return 0;
    }
}

template <int THREADS, int BT, int NBUF>
static int dispatch_wide(int nch, const double *P, int64_t ldp, const double *w, const double *props,
                         int64_t R, int H, int grid, int64_t rows_per_wg, double *partial,
                         int64_t ldpart, const mxm_em_state *state, hipStream_t stream) {
    switch (nch) {
#define WIDE_CASE(n) case n: return launch_wide<THREADS, n, BT, NBUF>(P, ldp, w, props, R, H, grid, rows_per_wg, partial, ldpart, state, stream);
        WIDE_CASE(1) WIDE_CASE(2) WIDE_CASE(3) WIDE_CASE(4) WIDE_CASE(5) WIDE_CASE(6) WIDE_CASE(7) WIDE_CASE(8)
        WIDE_CASE(9) WIDE_CASE(10) WIDE_CASE(11) WIDE_CASE(12) WIDE_CASE(13) WIDE_CASE(14) WIDE_CASE(15) WIDE_CASE(16)
#undef WIDE_CASE
        default: break;
    }
    return fail(-1, "mxm_em_iter: H=%s%lld outside the linear kernel's range", "", H);
}

// One tile of nb (<= MXM_MAX_BT) restarts over the linear matrix: streaming kernel + column reduce.
static int em_iter_linear_tile(const double *P, int64_t ldp, const double *w, const double *props, int64_t R,
                               int H, int nb, const mxm_em_state *state, double *colsum, double *partial,
                               hipStream_t stream, bool timed) {
    const int64_t ldpart = part_ld(H);
    const int ncol2 = (H + 1) / 2;
    const int threads = variant_threads(nb);
    const int nbuf = variant_nbuf(nb);
    const int wg_per_cu = (nb == 1) ? MXM_V1_WG_PER_CU : 1;
    const int nch = (ncol2 + threads - 1) / threads;
    int cap = num_cu() * wg_per_cu;
    if (cap > MXM_MAX_WG) cap = MXM_MAX_WG;
    // small matrices: fewer, longer row blocks (every workgroup pays 2 x H x 8 bytes of
    // proportion loads and partial stores, which colreduce then reads back)
    int nwg = clamp_grid((R + g_min_rows_per_wg - 1) / g_min_rows_per_wg, cap);
    int64_t rows_per_wg = (R + nwg - 1) / nwg;
    rows_per_wg = (rows_per_wg + nbuf - 1) / nbuf * nbuf;
    nwg = (int)((R + rows_per_wg - 1) / rows_per_wg);
    // a workgroup addresses its row block through one buffer descriptor with 32-bit offsets
    if ((double)rows_per_wg * (double)ldp * 8.0 >= 2147483648.0)
        return fail(-1, "mxm_em_iter: %s%lld rows of %lld doubles per workgroup exceed the 2 GiB descriptor range", "",
                    (long long)rows_per_wg, (long long)ldp);
    if (timed && g_ev_start != nullptr) HIP_TRY(hipEventRecord(g_ev_start, stream));
    int rc;
    if (nb == 1)
        rc = dispatch_wide<MXM_V1_THREADS, 1, MXM_V1_NBUF>(nch, P, ldp, w, props, R, H, nwg, rows_per_wg, partial, ldpart, state, stream);
    else if (nb == 2)
        rc = dispatch_wide<MXM_V2_THREADS, 2, MXM_V2_NBUF>(nch, P, ldp, w, props, R, H, nwg, rows_per_wg, partial, ldpart, state, stream);
    else
        rc = dispatch_wide<MXM_V3_THREADS, 3, MXM_V3_NBUF>(nch, P, ldp, w, props, R, H, nwg, rows_per_wg, partial, ldpart, state, stream);
    if (rc != 0) return rc;
    HIP_TRY(hipGetLastError());
    if (timed && g_ev_stop != nullptr) HIP_TRY(hipEventRecord(g_ev_stop, stream));
    hipLaunchKernelGGL(colreduce_kernel, dim3((H + 63) / 64, nb), dim3(COLRED_THREADS), 0, stream, partial, ldpart, nwg, nb,
                       H, props, colsum, state);
    HIP_TRY(hipGetLastError());
    return 0;
}

#ifndef MXM_F32_THREADS
#define MXM_F32_THREADS 256
#endif
#ifndef MXM_F32_NBUF
#define MXM_F32_NBUF 2
#endif
#ifndef MXM_F32_WG_PER_CU
#define MXM_F32_WG_PER_CU 2
#endif

template <int NCH>
static int launch_wide_f32(const float *P, int64_t ldp, const double *w, const double *props, int64_t R, int H,
                           int grid, int64_t rows_per_wg, double *partial, int64_t ldpart,
                           const mxm_em_state *state, hipStream_t stream) {
    if constexpr (NCH * MXM_F32_THREADS > 2048) {
        return fail(-1, "mxm_em_iter_f32: H=%s%lld outside the kernel's range", "", H);
    } else {
        hipLaunchKernelGGL((em_iter_wide_f32_kernel<MXM_F32_THREADS, NCH, MXM_F32_NBUF>), dim3(grid),
                           dim3(MXM_F32_THREADS), 0, stream, P, ldp, w, props, R, H, rows_per_wg, partial, ldpart,
                           state);
        return 0;
    }
}

static int em_iter_f32_one(const float *P, int64_t ldp, const double *w, const double *props, int64_t R, int H,
                           const mxm_em_state *state, double *colsum, double *partial, hipStream_t stream,
                           bool timed) {
    const int64_t ldpart = part_ld(H);
    const int nch = ((H + 3) / 4 + MXM_F32_THREADS - 1) / MXM_F32_THREADS;
    const int nbuf = MXM_F32_NBUF;
    int cap = num_cu() * MXM_F32_WG_PER_CU;
    if (cap > MXM_MAX_WG) cap = MXM_MAX_WG;
    int nwg = clamp_grid((R + nbuf - 1) / nbuf, cap);
    int64_t rows_per_wg = (R + nwg - 1) / nwg;
    rows_per_wg = (rows_per_wg + nbuf - 1) / nbuf * nbuf;
    nwg = (int)((R + rows_per_wg - 1) / rows_per_wg);
    if (timed && g_ev_start != nullptr) HIP_TRY(hipEventRecord(g_ev_start, stream));
    switch (nch) {
#define F32_CASE(n) case n: { const int lrc = launch_wide_f32<n>(P, ldp, w, props, R, H, nwg, rows_per_wg, partial, ldpart, state, stream); if (lrc != 0) return lrc; } break;
        F32_CASE(1) F32_CASE(2) F32_CASE(3) F32_CASE(4) F32_CASE(5) F32_CASE(6) F32_CASE(7) F32_CASE(8)
#undef F32_CASE
        default: return fail(-1, "mxm_em_iter_f32: H=%s%lld outside the kernel's range", "", H);
    }
    HIP_TRY(hipGetLastError());
    if (timed && g_ev_stop != nullptr) HIP_TRY(hipEventRecord(g_ev_stop, stream));
    hipLaunchKernelGGL(colreduce_kernel, dim3((H + 63) / 64, 1), dim3(COLRED_THREADS), 0, stream, partial, ldpart, nwg,
                       1, H, props, colsum, state);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int em_iter_log_one(const double *M, int64_t ldm, const double *w, const double *props, int64_t R, int H,
                           const mxm_em_state *state, double *colsum, double *partial, hipStream_t stream) {
    if (M == nullptr) return fail(-1, "mxm_em_iter: M is NULL and the linear path does not apply%s", "");
    const size_t lds = 2 * (size_t)H * sizeof(double);
    if (lds > 150 * 1024) return fail(-1, "mxm_em_iter: H=%s%lld too large for the log-space kernel", "", H);
    const int64_t ldpart = part_ld(H);
    const int nwg = clamp_grid(R, num_cu() * 2 < MXM_MAX_WG ? num_cu() * 2 : MXM_MAX_WG);
    hipLaunchKernelGGL((estep_log_kernel<true>), dim3(nwg), dim3(ROW_THREADS), lds, stream, M, ldm, w, props,
                       R, H, (double *)nullptr, (int64_t)0, 0, partial, ldpart, state);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(colreduce_kernel, dim3((H + 63) / 64, 1), dim3(COLRED_THREADS), 0, stream, partial, ldpart, nwg, 1,
                       H, (const double *)nullptr, colsum, state);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_em_iter(const double *M, int64_t ldm, const double *P, int64_t ldp, const double *w,
                           const double *props, int64_t R, int32_t H, int32_t B, const mxm_em_state *state,
                           double *colsum, void *ws, size_t ws_bytes, void *stream) {
    if (R <= 0 || H <= 0 || B <= 0) return fail(-1, "mxm_em_iter: bad shape R=%s%lld H=%lld", "", R, H);
    if (ws == nullptr || ws_bytes < mxm_workspace_bytes(R, H, B)) return fail(-1, "mxm_em_iter: workspace too small%s", "");
    if (P != nullptr && ((ldp & 1) || ldp < H)) return fail(-1, "mxm_em_iter: ldp must be even and >= H%s", "");
    if (M != nullptr && ldm < H) return fail(-1, "mxm_em_iter: ldm < H%s", "");
    const bool linear = (P != nullptr) && mxm_linear_supported(H);
    // restarts are taken in tiles of up to g_max_bt that share one pass over the matrix; the
    // scratch is reused tile after tile (same stream, so the passes are ordered)
    const int max_bt = linear ? g_max_bt : 1;
    for (int b = 0; b < B;) {
        int nb = B - b;
        if (nb > max_bt) nb = max_bt;
        // a batch keeps nb proportion vectors in LDS: shrink the tile until they fit
        while (nb > 1 && batch_lds_bytes((int)H, nb) > MXM_LDS_BUDGET) --nb;
        const mxm_em_state *st = state ? state + b : nullptr;
        int rc;
        if (linear)
            rc = em_iter_linear_tile(P, ldp, w, props + (int64_t)b * H, R, (int)H, nb, st,
                                     colsum + (int64_t)b * H, (double *)ws, (hipStream_t)stream, b == 0);
        else
            rc = em_iter_log_one(M, ldm, w, props + (int64_t)b * H, R, (int)H, st, colsum + (int64_t)b * H,
                                 (double *)ws, (hipStream_t)stream);
        if (rc != 0) return rc;
        b += nb;
    }
    return 0;
}

extern "C" int mxm_linearize_f32(const double *M, int64_t ldm, int64_t R, int32_t H, float *P, int64_t ldp,
                                 double *rowmax, void *stream) {
    if (R < 0 || H <= 0) return fail(-1, "mxm_linearize_f32: bad shape%s", "");
    if (ldm < H || ldp < H || (ldp & 3)) return fail(-1, "mxm_linearize_f32: ldp must be a multiple of 4 and >= H%s (ldp=%lld)", "", ldp);
    if (R == 0) return 0;
    const int grid = clamp_grid(R, num_cu() * 8);
    hipLaunchKernelGGL(linearize_f32_kernel, dim3(grid), dim3(ROW_THREADS), 0, (hipStream_t)stream, M, ldm, R, (int)H,
                       P, ldp, rowmax);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_em_iter_f32(const float *P, int64_t ldp, const double *w, const double *props, int64_t R,
                               int32_t H, int32_t B, const mxm_em_state *state, double *colsum, void *ws,
                               size_t ws_bytes, void *stream) {
    if (R <= 0 || H <= 0 || B <= 0) return fail(-1, "mxm_em_iter_f32: bad shape R=%s%lld H=%lld", "", R, H);
    if (!mxm_linear_supported(H)) return fail(-1, "mxm_em_iter_f32: H=%s%lld outside the linear kernel's range", "", H);
    if (P == nullptr || (ldp & 3) || ldp < H) return fail(-1, "mxm_em_iter_f32: ldp must be a multiple of 4 and >= H%s", "");
    if (ws == nullptr || ws_bytes < mxm_workspace_bytes(R, H, B)) return fail(-1, "mxm_em_iter_f32: workspace too small%s", "");
    for (int b = 0; b < B; ++b) {
        const int rc = em_iter_f32_one(P, ldp, w, props + (int64_t)b * H, R, (int)H, state ? state + b : nullptr,
                                       colsum + (int64_t)b * H, (double *)ws, (hipStream_t)stream, b == 0);
        if (rc != 0) return rc;
    }
    return 0;
}

extern "C" int mxm_m_finalize(const double *colsum, double *props_cur, double *props_new, int32_t H, int32_t B,
                              double tol, int32_t max_iter, mxm_em_state *state, void *stream) {
    if (H <= 0 || B <= 0 || state == nullptr) return fail(-1, "mxm_m_finalize: bad arguments%s", "");
    hipLaunchKernelGGL(finalize_kernel, dim3(B), dim3(FIN_THREADS), 0, (hipStream_t)stream, colsum, props_cur,
                       props_new, (int)H, tol, (int)max_iter, state);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- the run_em inner loop (em.py:126-143) as a native driver -----------------------------------
// Iterations are enqueued in chunks; for small matrices the chunk is captured once into a
// hipGraph and replayed (the loop is launch-bound there: 3 kernels of a few microseconds each),
// for large ones plain launches already run ahead of the GPU.  Either way the kernels of a
// finished restart are no-ops, so the state freezes on the iteration the reference stops on.
static int g_loop_graph = -1;          // -1 auto (graph when R*H*B is small), 0 never, 1 always
extern "C" int mxm_set_loop_graph(int32_t mode) {
    g_loop_graph = mode < 0 ? -1 : (mode > 0 ? 1 : 0);
    return 0;
}

static int enqueue_iterations(const double *M, int64_t ldm, const double *P, int64_t ldp, const double *w,
                              int64_t R, int32_t H, int32_t B, double *props_cur, double *props_new,
                              double *colsum, mxm_em_state *state, double tol, int32_t max_iter, int64_t n,
                              void *ws, size_t ws_bytes, hipStream_t s, bool p_is_f32) {
    for (int64_t i = 0; i < n; ++i) {
        int rc = p_is_f32 ? mxm_em_iter_f32(reinterpret_cast<const float *>(P), ldp, w, props_cur, R, H, B, state,
                                            colsum, ws, ws_bytes, s)
                          : mxm_em_iter(M, ldm, P, ldp, w, props_cur, R, H, B, state, colsum, ws, ws_bytes, s);
        if (rc != 0) return rc;
        rc = mxm_m_finalize(colsum, props_cur, props_new, H, B, tol, max_iter, state, s);
        if (rc != 0) return rc;
    }
    return 0;
}

static int em_loop_impl(const double *M, int64_t ldm, const double *P, int64_t ldp, const double *w,
                        int64_t R, int32_t H, int32_t B, double *props_cur, double *props_new,
                        double *colsum, mxm_em_state *state, double tol, int32_t max_iter,
                        int32_t check_every, void *ws, size_t ws_bytes, void *stream,
                        mxm_em_state *state_host, bool p_is_f32) {
    if (state_host == nullptr || state == nullptr) return fail(-1, "mxm_em_loop: state pointers required%s", "");
    if (check_every < 1) check_every = 1;
    hipStream_t caller = (hipStream_t)stream;
    (void)num_cu();                                    // device query outside any capture
    const bool want_graph = g_loop_graph == 1 ||
                            (g_loop_graph == -1 && (double)R * (double)H * (double)B < 6.4e7);

    // the loop runs on a private stream (the caller's may be the legacy default stream, which
    // cannot be captured); it is ordered after / before the caller's stream with events
    hipStream_t s = nullptr;
    hipEvent_t ev = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    int rc = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int64_t graph_iters = 0;
#define LOOP_TRY(expr)                                                                        \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) { rc = fail(-2, "HIP error: %s (line %lld)", hipGetErrorString(e_), __LINE__); goto done; } \
    } while (0)
    LOOP_TRY(hipEventRecord(ev, caller));
    LOOP_TRY(hipStreamWaitEvent(s, ev, 0));
    LOOP_TRY(hipMemcpyAsync(state_host, state, sizeof(mxm_em_state) * B, hipMemcpyDeviceToHost, s));
    LOOP_TRY(hipStreamSynchronize(s));
    {
        int64_t issued = 0;
        for (;;) {
            bool all_done = true;
            for (int b = 0; b < B; ++b) all_done = all_done && (state_host[b].done != 0);
            if (all_done || issued >= (int64_t)max_iter) break;
            int64_t n = (int64_t)max_iter - issued;
            if (n > check_every) n = check_every;
            bool launched = false;
            if (want_graph) {
                if (exec == nullptr || graph_iters != n) {
                    if (exec != nullptr) { (void)hipGraphExecDestroy(exec); exec = nullptr; }
                    if (graph != nullptr) { (void)hipGraphDestroy(graph); graph = nullptr; }
                    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                        const int crc = enqueue_iterations(M, ldm, P, ldp, w, R, H, B, props_cur, props_new, colsum,
                                                           state, tol, max_iter, n, ws, ws_bytes, s, p_is_f32);
                        const hipError_t ee = hipStreamEndCapture(s, &graph);
                        if (crc == 0 && ee == hipSuccess &&
                            hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                            graph_iters = n;
                        } else {
                            exec = nullptr;            // capture unavailable: plain launches below
                            (void)hipGetLastError();
                        }
                    }
                }
                if (exec != nullptr) {
                    LOOP_TRY(hipGraphLaunch(exec, s));
                    launched = true;
                }
            }
            if (!launched) {
                rc = enqueue_iterations(M, ldm, P, ldp, w, R, H, B, props_cur, props_new, colsum, state, tol,
                                        max_iter, n, ws, ws_bytes, s, p_is_f32);
                if (rc != 0) goto done;
            }
            issued += n;
            LOOP_TRY(hipMemcpyAsync(state_host, state, sizeof(mxm_em_state) * B, hipMemcpyDeviceToHost, s));
            LOOP_TRY(hipStreamSynchronize(s));
        }
    }
    LOOP_TRY(hipEventRecord(ev, s));
    LOOP_TRY(hipStreamWaitEvent(caller, ev, 0));
done:
#undef LOOP_TRY
    if (exec != nullptr) (void)hipGraphExecDestroy(exec);
    if (graph != nullptr) (void)hipGraphDestroy(graph);
    if (s != nullptr) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    if (ev != nullptr) (void)hipEventDestroy(ev);
    return rc;
}

extern "C" int mxm_em_loop(const double *M, int64_t ldm, const double *P, int64_t ldp, const double *w,
                           int64_t R, int32_t H, int32_t B, double *props_cur, double *props_new,
                           double *colsum, mxm_em_state *state, double tol, int32_t max_iter,
                           int32_t check_every, void *ws, size_t ws_bytes, void *stream,
                           mxm_em_state *state_host) {
    return em_loop_impl(M, ldm, P, ldp, w, R, H, B, props_cur, props_new, colsum, state, tol, max_iter,
                        check_every, ws, ws_bytes, stream, state_host, false);
}

extern "C" int mxm_em_loop_f32(const float *P, int64_t ldp, const double *w, int64_t R, int32_t H, int32_t B,
                               double *props_cur, double *props_new, double *colsum, mxm_em_state *state,
                               double tol, int32_t max_iter, int32_t check_every, void *ws, size_t ws_bytes,
                               void *stream, mxm_em_state *state_host) {
    return em_loop_impl(nullptr, 0, reinterpret_cast<const double *>(P), ldp, w, R, H, B, props_cur, props_new,
                        colsum, state, tol, max_iter, check_every, ws, ws_bytes, stream, state_host, true);
}

template <int NCH>
static void launch_estep_wide(const double *M, int64_t ldm, const double *w, const double *lnp, int64_t R, int H,
                              int grid, int64_t rows_per_wg, double *out, int64_t ldo, int mode, double *partial,
                              int64_t ldpart, hipStream_t s) {
    if (partial != nullptr)
        hipLaunchKernelGGL((estep_wide_kernel<NCH, true>), dim3(grid), dim3(256), 0, s, M, ldm, w, lnp, R, H,
                           rows_per_wg, out, ldo, mode, partial, ldpart);
    else
        hipLaunchKernelGGL((estep_wide_kernel<NCH, false>), dim3(grid), dim3(256), 0, s, M, ldm, w, lnp, R, H,
                           rows_per_wg, out, ldo, mode, partial, ldpart);
}

extern "C" int mxm_em_step(const double *M, int64_t ldm, const double *w, const double *ln_props, int64_t R,
                           int32_t H, double *out, int64_t ldo, int32_t mode, double *colsum, void *ws,
                           size_t ws_bytes, void *stream) {
    if (R <= 0 || H <= 0 || ldm < H) return fail(-1, "mxm_em_step: bad shape%s", "");
    if (out != nullptr && ldo < H) return fail(-1, "mxm_em_step: ldo < H%s", "");
    if (colsum != nullptr && (ws == nullptr || ws_bytes < mxm_workspace_bytes(R, H, 1)))
        return fail(-1, "mxm_em_step: workspace too small%s", "");
    const int64_t ldpart = part_ld(H);
    hipStream_t s = (hipStream_t)stream;
    double *partial = colsum ? (double *)ws : (double *)nullptr;
    int nwg;
    const bool aligned = ((H & 1) == 0) && ((ldm & 1) == 0) && ((reinterpret_cast<uintptr_t>(M) & 15) == 0) &&
                         (out == nullptr || (((ldo & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)));
    const int nch = (H / 2 + 255) / 256;
    // register budget of estep_wide_kernel (spill-free instances only): 13 column chunks per
    // thread without the M-step sums, 10 with them
    if (aligned && mxm_linear_supported(H) && nch <= (colsum != nullptr ? 10 : 13)) {
        // wide rows: one read + one write per cell, rows held in registers
        int cap = num_cu() * 2 < MXM_MAX_WG ? num_cu() * 2 : MXM_MAX_WG;
        nwg = clamp_grid((R + 1) / 2, cap);
        int64_t rows_per_wg = (R + nwg - 1) / nwg;
        rows_per_wg = (rows_per_wg + 1) / 2 * 2;
        nwg = (int)((R + rows_per_wg - 1) / rows_per_wg);
        switch (nch) {
#define EW_CASE(n) case n: launch_estep_wide<n>(M, ldm, w, ln_props, R, (int)H, nwg, rows_per_wg, out, ldo, (int)mode, partial, ldpart, s); break;
            EW_CASE(1) EW_CASE(2) EW_CASE(3) EW_CASE(4) EW_CASE(5) EW_CASE(6) EW_CASE(7) EW_CASE(8)
            EW_CASE(9) EW_CASE(10) EW_CASE(11) EW_CASE(12) EW_CASE(13) EW_CASE(14) EW_CASE(15) EW_CASE(16)
#undef EW_CASE
            default: return fail(-1, "mxm_em_step: H=%s%lld outside the wide kernel's range", "", H);
        }
    } else {
        const size_t lds = 2 * (size_t)H * sizeof(double);
        if (lds > 150 * 1024) return fail(-1, "mxm_em_step: H=%s%lld too large", "", H);
        nwg = clamp_grid(R, num_cu() * 2 < MXM_MAX_WG ? num_cu() * 2 : MXM_MAX_WG);
        hipLaunchKernelGGL((estep_log_kernel<false>), dim3(nwg), dim3(ROW_THREADS), lds, s, M, ldm, w, ln_props, R,
                           (int)H, out, ldo, (int)mode, partial, ldpart, (const mxm_em_state *)nullptr);
    }
    HIP_TRY(hipGetLastError());
    if (colsum != nullptr) {
        hipLaunchKernelGGL(colreduce_kernel, dim3((H + 63) / 64, 1), dim3(COLRED_THREADS), 0, s, (const double *)ws, ldpart, nwg,
                           1, (int)H, (const double *)nullptr, colsum, (const mxm_em_state *)nullptr);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

extern "C" int mxm_log_normalize(const double *colsum, int32_t H, double *ln_new, void *stream) {
    if (H <= 0) return fail(-1, "mxm_log_normalize: H <= 0%s", "");
    hipLaunchKernelGGL(log_normalize_kernel, dim3(1), dim3(FIN_THREADS), 0, (hipStream_t)stream, colsum, (int)H, ln_new);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_l1_exp_diff(const double *a, const double *b, int32_t H, double *l1_out, void *stream) {
    if (H <= 0) return fail(-1, "mxm_l1_exp_diff: H <= 0%s", "");
    hipLaunchKernelGGL(l1_exp_diff_kernel, dim3(1), dim3(FIN_THREADS), 0, (hipStream_t)stream, a, b, (int)H, l1_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_add_scalar(double *x, int64_t ld, int64_t R, int32_t H, double delta, void *stream) {
    if (R <= 0 || H <= 0 || ld < H) return fail(-1, "mxm_add_scalar: bad shape%s", "");
    hipLaunchKernelGGL(add_scalar_kernel, dim3(clamp_grid(R, num_cu() * 8)), dim3(256), 0, (hipStream_t)stream, x, ld, R, (int)H, delta);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_assign_reads(const double *X, int64_t ldx, const double *log_props, const int32_t *cols,
                                int32_t nC, int64_t R, int32_t H, double log_min_fold, int32_t *assigned,
                                void *stream) {
    if (R <= 0 || H <= 0 || nC <= 0 || ldx < H) return fail(-1, "mxm_assign_reads: bad shape%s", "");
    hipLaunchKernelGGL(assign_reads_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X,
                       ldx, log_props, cols, (int)nC, R, log_min_fold, assigned);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_diag_stream_read(const void *src, size_t bytes, int32_t wg_per_cu, int32_t blocked, void *sink,
                                    void *stream) {
    if (src == nullptr || sink == nullptr || bytes < 16 || wg_per_cu < 1) return fail(-1, "mxm_diag_stream_read: bad arguments%s", "");
    if (blocked)
        hipLaunchKernelGGL(diag_stream_read_kernel<true>, dim3(num_cu() * wg_per_cu), dim3(256), 0, (hipStream_t)stream,
                           (const uint4 *)src, (int64_t)(bytes / 16), (unsigned int *)sink);
    else
        hipLaunchKernelGGL(diag_stream_read_kernel<false>, dim3(num_cu() * wg_per_cu), dim3(256), 0, (hipStream_t)stream,
                           (const uint4 *)src, (int64_t)(bytes / 16), (unsigned int *)sink);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_row_argmax_votes(const double *X, int64_t ldx, const double *w, int64_t R, int32_t H,
                                    int32_t *best, double *votes, void *stream) {
    if (R <= 0 || H <= 0 || ldx < H) return fail(-1, "mxm_row_argmax_votes: bad shape%s", "");
    hipLaunchKernelGGL(row_argmax_votes_kernel, dim3(clamp_grid(R, num_cu() * 8)), dim3(ROW_THREADS), 0,
                       (hipStream_t)stream, X, ldx, w, R, (int)H, best, votes);
    HIP_TRY(hipGetLastError());
    return 0;
}
