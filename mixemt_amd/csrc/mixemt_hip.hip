// mixemt_hip.hip -- gfx950 (MI355X, CDNA4) kernels + C ABI for mixemt's EM hot path.
//
// Replaces, behind include/mixemt_hip.h:
//   preprocess.build_em_matrix   /root/reference/mixemt/preprocess.py:177-198
//   em.em_step / em.run_em loop  /root/reference/mixemt/em.py:57-91, :126-143
//   (+ consumers of the result: assemble.py:103-123, :284-334)
//
// Everything here is HBM-bound byte/fp64 streaming + reductions: 64-wide
// wavefronts, 16-byte coalesced loads, rows held in VGPRs between the row
// reduction and the column accumulation, deterministic two-stage column sums
// (no float atomics).  No MFMA: there is no contraction to feed it.
//
// One translation unit; the kernels live in headers beside this file
// (DESIGN.md section 4 has the roofline of each):
//   common.hpp          error plumbing, reductions
//   build_kernels.hpp   build_em_matrix_kernel (byte table, L2/MALL: any alphabet, any width)
//   build_lut_kernels.hpp  build_lut_kernel (hit/miss by LDS lookup)
//   build_sparse_kernels.hpp  build_sparse_kernel (the row from the haplogroups' markers: one in-order sum per distinct cell value)
//   em_kernels.hpp      linearize, em_iter_wide_kernel (THE hot kernel: R*H*8 B read per EM iteration,
//                       1-4 restarts per pass), em_iter_wide_f32_kernel (opt-in storage variant),
//                       colreduce_kernel, finalize_kernel
//   estep_kernels.hpp   estep_log_kernel (any H), estep_wide_kernel (register-resident posterior pass)
//   records_kernels.hpp posterior_argmax_kernel (row argmax of the folded posterior from records), votes_from_best_kernel
//   aux_kernels.hpp     log_normalize, l1_exp_diff, add_scalar, row_argmax_votes, assign_reads, gather, fold, diag_stream_read
//   fused_kernels.hpp   em_fused_loop_kernel (the whole EM loop of a cache-resident matrix in one persistent launch)
//   fused_cols_kernels.hpp  em_fused_cols_kernel (the same for up to 1536 rows, columns split over the workgroups, matrix in registers)
//   coded_kernels.hpp   encode_rows_kernel, em_iter_coded_kernel (row-dictionary storage: one byte per cell + the row's distinct values)
//   fused_coded_kernels.hpp  em_fused_coded_kernel (the whole EM loop over records in one persistent launch)
//   fused_narrow_kernels.hpp  em_fused_narrow_kernel (the same for the refinement EM's few columns: the matrix in registers)
//   aln_encode.hpp      HOST code: mxm_aln_encode, the batched alignment front end (process_reads + reduce_reads + row order)
// This file: the host side of the C ABI (shape checks, grid sizing, dispatch, the loop driver).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#include "mixemt_hip.h"
#include "mixemt_hip_tuning.h"

#include "common.hpp"
#include "aln_encode.hpp"
#include "bam_reader.hpp"
#include "coded_kernels.hpp"
#include "quad_kernels.hpp"
#include "quad_batched_kernels.hpp"
#include "build_kernels.hpp"
#include "build_lut_kernels.hpp"
#include "build_sparse_kernels.hpp"
#include "em_kernels.hpp"
#include "estep_kernels.hpp"
#include "aux_kernels.hpp"
#include "records_kernels.hpp"
#include "fused_kernels.hpp"
#include "fused_cols_kernels.hpp"
#include "fused_coded_kernels.hpp"
#include "fused_narrow_kernels.hpp"
#include "exchange.hpp"


// ------------------------------------------------------------------------------------------
// Tuning / measurement state (include/mixemt_hip_tuning.h).  ONE struct, guarded by a mutex: the setters
// lock it, and every entry point of the ABI takes a private snapshot into thread-local storage when
// it starts (MXM_ENTER), so a call sees one consistent set of knobs from beginning to end whatever
// other host threads (one per GPU) set in the meantime.  Results never depend on any of it beyond the
// rounding of a different summation order.
// ------------------------------------------------------------------------------------------
struct mxm_em_state;
typedef void (*mxm_progress_fn)(const mxm_em_state *state_host, int32_t B, void *user);
struct mxm_tuning {
    int sparse_maxd = -1;           // distinct non-zero masks a row of the marker build may have (-1: the kernel's limit)
    int min_rows_per_wg = 8;        // measured: 1000 x 5408 31 us/step (vs 39 at 2); no effect from 10^4 rows up
    int compact_restarts = 2;       // mxm_em_loop: 1 packs the running restarts, 2 also keeps ONE full tile of them iterating
    int max_bt = 4;                 // restarts per matrix pass (1..MXM_MAX_BT)
    int loop_graph = -1;            // -1 auto (graph when R*H*B is small), 0 never, 1 always
    int loop_fused = -1;            // -1 auto (by size), 0 never, 1 whenever the shape allows
    int fused_chunk = 0;            // iterations per launch of the one-launch loop (0 = run to the end)
    int fused_cols = 1;             // matrices of up to 1536 rows take the transposed form (columns split)
    int fused_coded_wg = 0;         // workgroups of the one-launch loop over records (0 = by size)
    int quad_left_wg = 0;           // workgroups of the leftover pass beside the quad pass (0 = by the rows' measured cost)
    int sparse_long_entries = 5120; // ... whose rows with more marker entries than this go to the fallback list (0: no limit)
    int quad_encoder = 1;           // the quad dictionary's encoder: 1 = a wave per row (quad_encode_wave_kernel), 0 = a workgroup per row
    int sparse_long = 1;            // the marker build's second launch for rows of 65 .. 128 observations (0: they go to the fallback list)
    int coded_bt = 3;               // restarts per pass over records beside a quad dictionary (1 = one per pass, 3 = the batched kernel)
    int fused_force_abort = 0;      // test hook: the one-launch loop starts with its abort flag raised (as if starved)
    double fused_cells = 1.0e8;     // ~18 000 rows at H = 5408: measured break-even is ~30 000 rows (profiles/r02/small_runs.txt)
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;     // timing hook around the dominant kernel
    mxm_progress_fn progress = nullptr;
    void *progress_user = nullptr;
    int progress_every = 10;
};
static std::mutex g_tune_mu;
static mxm_tuning g_tune;
static thread_local mxm_tuning T;   // the calling thread's snapshot for the entry point it is in
#define MXM_ENTER() do { std::lock_guard<std::mutex> lk_(g_tune_mu); T = g_tune; } while (0)
template <typename F>
static int tune_set(F f) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    f(g_tune);
    return 0;
}
extern "C" int mxm_reset_tuning(void) {
    return tune_set([](mxm_tuning &t) { t = mxm_tuning(); });
}

// ------------------------------------------------------------------------------------------
// host side of the C ABI
// ------------------------------------------------------------------------------------------
static inline int clamp_grid(int64_t want, int cap) {
    if (want < 1) want = 1;
    return (int)(want < cap ? want : cap);
}

static inline int64_t part_ld(int H) { return ((int64_t)H + 1) & ~(int64_t)1; }

// More than 64 KiB of dynamic LDS must be opted into per kernel (and device).  Re-issued on every
// call: the attribute is idempotent and costs a table write, and nothing is cached that two host
// threads (one per GPU) could race on.  A refusal is reported with the kernel's name and the size.
static hipError_t raise_dynamic_lds(const void *kernel, size_t bytes, const char *name) {
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess)
        snprintf(g_err, sizeof(g_err), "%s: cannot opt into %zu bytes of dynamic LDS: %s", name, bytes,
                 hipGetErrorString(e));
    return e;
}

extern "C" int mxm_version(void) { return MXM_VERSION; }
extern "C" const char *mxm_last_error(void) { return g_err; }

extern "C" int mxm_linear_supported(int32_t H) {
    return (H >= MXM_LINEAR_MIN_H && H <= 8192) ? 1 : 0;
}

// Scratch layouts of the one-launch loops (fused_kernels.hpp, fused_cols_kernels.hpp), all inside `ws`:
//   [sync block][exchange buffers of the form in use][... free ...][snapshot of the loop vectors at the tail]
static size_t fused_sync_bytes() { return (sizeof(fused_sync) + 255) & ~(size_t)255; }
// The loop vectors of the restarts in flight are saved at the TAIL of the workspace before every launch of a
// one-launch loop (3 x [B][H] doubles + B states), so that a launch that gives up (status -3) can be undone.
static size_t fused_snapshot_bytes(int H, int B) {
    return ((size_t)3 * B * H * sizeof(double) + (size_t)B * sizeof(mxm_em_state) + 255) & ~(size_t)255;
}
// rows split: T / ln T (2 rows) + one partial row per workgroup
static size_t fused_rows_bytes(int H, int nwg) { return (size_t)(nwg + 2) * part_ld(H) * sizeof(double); }
// columns split (R <= FCOLS_MAX_RPT * FCOLS_THREADS): z partials [nwg][ldz] + c [ldz] + L1 partials [2][nwg]
static size_t fused_cols_bytes(int64_t R, int nwg) {
    const size_t ldz = (size_t)((R + 1) & ~(int64_t)1);
    return ((size_t)(nwg + 1) * ldz + 2 * (size_t)nwg) * sizeof(double);
}

extern "C" size_t mxm_workspace_bytes(int64_t R, int32_t H, int32_t B) {
    // restart tiles are processed one after another over the same scratch: one partial row per workgroup and tile member
    const size_t tiles = (size_t)MXM_MAX_WG * MXM_MAX_BT * (size_t)part_ld(H) * sizeof(double);
    // the one-launch loops' blocks (the transposed form's z partials are R-sized, not H-sized, and
    // outgrow the tile scratch of a narrow matrix -- H = 66, R = 1500: 3.2 MB against 2.2 MB)
    size_t fused = fused_rows_bytes(H, MXM_MAX_WG);
    if (R > 0 && R <= (int64_t)FCOLS_MAX_RPT * FCOLS_THREADS) fused = std::max(fused, fused_cols_bytes(R, MXM_MAX_WG));
    fused += fused_sync_bytes() + fused_snapshot_bytes(H, B > 0 ? B : 1) + 512;
    return std::max(tiles, fused);
}

extern "C" int64_t mxm_encode_signatures(const char *text, const int64_t *off, int64_t R,
                                         const int32_t *site_of_pos, int64_t ref_len, int64_t *row_ptr,
                                         uint16_t *site, uint8_t *obs, int64_t cap) {
    int64_t n = 0;
    row_ptr[0] = 0;
    for (int64_t r = 0; r < R; ++r) {
        const char *p = text + off[r];
        const char *end = text + off[r + 1] - 1;           // the separator byte is not part of it
        if (p >= end) return -(r + 1);                     // '' -> int('') in the reference
        while (p < end) {
            int64_t pos = 0;
            int digits = 0;
            while (p < end && *p >= '0' && *p <= '9' && digits < 10) {
                pos = pos * 10 + (*p - '0');
                ++p;
                ++digits;
            }
            if (digits == 0 || digits >= 10 || p >= end || *p != ':') return -(r + 1);
            ++p;
            const char *b = p;
            while (p < end && *p != ',') {
                if (*p == ':') return -(r + 1);
                ++p;
            }
            if (pos >= ref_len || site_of_pos[pos] < 0 || n >= cap) return -(r + 1);
            site[n] = (uint16_t)site_of_pos[pos];
            obs[n] = (p - b == 1) ? (uint8_t)*b : (uint8_t)0;
            ++n;
            if (p < end) {
                ++p;                                       // ','
                if (p >= end) return -(r + 1);             // trailing ',': an empty item
            }
        }
        row_ptr[r + 1] = n;
    }
    return n;
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

static int launch_build_lut(int nt, int grid, hipStream_t s, const uint8_t *E, int64_t lde, int64_t e_bytes,
                            const double *lhit, const double *lmiss, const uint8_t *obsmap, const int64_t *row_ptr,
                            const uint16_t *site, const uint8_t *obs, const int64_t *order, int64_t R, int H, double *M,
                            int64_t ldm, int vec_ok, int compact = 0) {
    switch (nt) {
#define BL_CASE(n) case n: hipLaunchKernelGGL((build_lut_kernel<n, LUT_CPL>), dim3(grid), dim3(LUT_THREADS), 0, s, E, lde, e_bytes, lhit, lmiss, obsmap, row_ptr, site, obs, order, R, H, M, ldm, vec_ok, compact); break;
        BL_CASE(1) BL_CASE(2) BL_CASE(3) BL_CASE(4) BL_CASE(5) BL_CASE(6) BL_CASE(7) BL_CASE(8)
#undef BL_CASE
        default: return 1;
    }
    return 0;
}

extern "C" int mxm_build_em_matrix_lut(const uint8_t *Ecode, int64_t lde, const double *lhit, const double *lmiss,
                                       const uint8_t *obsmap, const int64_t *row_ptr, const uint16_t *site,
                                       const uint8_t *obs, const int64_t *order, int64_t R, int32_t H, int32_t S,
                                       double *M, int64_t ldm, void *stream) {
    if (R < 0 || H <= 0 || S <= 0) return fail(-1, "mxm_build_em_matrix_lut: bad shape R=%s%lld H=%lld", "", R, H);
    if (H > 8192) return fail(-1, "mxm_build_em_matrix_lut: more than 8192 haplogroups%s (H=%lld): use mxm_build_em_matrix", "", H);
    if (lde < (((int64_t)H + 7) & ~(int64_t)7) || (lde & 7) != 0 || (reinterpret_cast<uintptr_t>(Ecode) & 7) != 0)
        return fail(-1, "mxm_build_em_matrix_lut: Ecode must be 8-byte aligned with lde a multiple of 8 and >= H rounded up to 8%s (lde=%lld)", "", lde);
    if (ldm < H) return fail(-1, "mxm_build_em_matrix_lut: ldm < H%s", "");
    if (S > 65536 || (int64_t)S * lde >= ((int64_t)1 << 31))
        return fail(-1, "mxm_build_em_matrix_lut: table of %s%lld x %lld bytes exceeds one buffer descriptor", "", S, lde);
    if (R == 0) return 0;
    const int grid = clamp_grid(R, num_cu() * 8);
    const int nt = (H + LUT_THREADS * LUT_CPL - 1) / (LUT_THREADS * LUT_CPL);
    const int vec_ok = ((ldm & 1) == 0) && ((reinterpret_cast<uintptr_t>(M) & 15) == 0);
    if (launch_build_lut(nt, grid, (hipStream_t)stream, Ecode, lde, (int64_t)S * lde, lhit, lmiss, obsmap, row_ptr, site, obs,
                         order, R, (int)H, M, ldm, vec_ok) != 0)
        return fail(-1, "mxm_build_em_matrix_lut: H=%s%lld outside the kernel's range", "", H);
    HIP_TRY(hipGetLastError());
    return 0;
}

// the listed rows into a COMPACT matrix: row rows[i] of the input -> row i of M_out (build_em_records_device's slabs)
extern "C" int mxm_build_em_matrix_lut_rows(const uint8_t *Ecode, int64_t lde, const double *lhit, const double *lmiss,
                                            const uint8_t *obsmap, const int64_t *row_ptr, const uint16_t *site,
                                            const uint8_t *obs, const int64_t *rows, int64_t n_rows, int32_t H, int32_t S,
                                            double *M_out, int64_t ldm, void *stream) {
    if (n_rows < 0 || H <= 0 || S <= 0 || rows == nullptr) return fail(-1, "mxm_build_em_matrix_lut_rows: bad arguments%s", "");
    if (H > 8192) return fail(-1, "mxm_build_em_matrix_lut_rows: more than 8192 haplogroups%s (H=%lld)", "", H);
    if (lde < (((int64_t)H + 7) & ~(int64_t)7) || (lde & 7) != 0 || (reinterpret_cast<uintptr_t>(Ecode) & 7) != 0)
        return fail(-1, "mxm_build_em_matrix_lut_rows: Ecode must be 8-byte aligned with lde a multiple of 8 and >= H rounded up to 8%s (lde=%lld)", "", lde);
    if (ldm < H) return fail(-1, "mxm_build_em_matrix_lut_rows: ldm < H%s", "");
    if (S > 65536 || (int64_t)S * lde >= ((int64_t)1 << 31))
        return fail(-1, "mxm_build_em_matrix_lut_rows: table of %s%lld x %lld bytes exceeds one buffer descriptor", "", S, lde);
    if (n_rows == 0) return 0;
    const int grid = clamp_grid(n_rows, num_cu() * 8);
    const int nt = (H + LUT_THREADS * LUT_CPL - 1) / (LUT_THREADS * LUT_CPL);
    const int vec_ok = ((ldm & 1) == 0) && ((reinterpret_cast<uintptr_t>(M_out) & 15) == 0);
    if (launch_build_lut(nt, grid, (hipStream_t)stream, Ecode, lde, (int64_t)S * lde, lhit, lmiss, obsmap, row_ptr, site, obs, rows,
                         n_rows, (int)H, M_out, ldm, vec_ok, 1) != 0)
        return fail(-1, "mxm_build_em_matrix_lut_rows: H=%s%lld outside the kernel's range", "", H);
    HIP_TRY(hipGetLastError());
    return 0;
}

// The records of a slab of rows that were coded from their dense form (mxm_encode_rows over a compact side matrix) take
// their place in the matrix's record arrays: row rows[i] gets sub_off[i] + base, sub_nd[i], sub_rm[i] where sub_nd[i] > 0.
__global__ __launch_bounds__(256) void scatter_records_kernel(const int64_t *__restrict__ rows, int64_t n, const int64_t *__restrict__ sub_off,
                                                             const int32_t *__restrict__ sub_nd, const double *__restrict__ sub_rm,
                                                             int64_t base, int64_t *__restrict__ rec_off, int32_t *__restrict__ ndist,
                                                             double *__restrict__ rowmax) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || sub_nd[i] <= 0) return;
    const int64_t r = rows[i];
    rec_off[r] = sub_off[i] + base;
    ndist[r] = sub_nd[i];
    rowmax[r] = sub_rm[i];
}
extern "C" int mxm_scatter_records(const int64_t *rows, int64_t n, const int64_t *sub_off, const int32_t *sub_nd, const double *sub_rm,
                                   int64_t base, int64_t *rec_off, int32_t *ndist, double *rowmax, void *stream) {
    if (n < 0 || (n > 0 && (rows == nullptr || sub_off == nullptr || sub_nd == nullptr || sub_rm == nullptr || rec_off == nullptr ||
                            ndist == nullptr || rowmax == nullptr)))
        return fail(-1, "mxm_scatter_records: bad arguments%s", "");
    if (n == 0) return 0;
    hipLaunchKernelGGL(scatter_records_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, n, sub_off,
                       sub_nd, sub_rm, base, rec_off, ndist, rowmax);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Loads the library's code object onto the calling thread's device (HIP does that lazily, on the first launch: ~18 ms that
// would otherwise land in whatever stage launches first) and returns once it is there.
__global__ void preload_kernel() {}
extern "C" int mxm_preload(void) {
    hipLaunchKernelGGL(preload_kernel, dim3(1), dim3(64), 0, (hipStream_t)nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

extern "C" int mxm_set_sparse_max_distinct(int32_t n) {
    return tune_set([n](mxm_tuning &t) { t.sparse_maxd = n < 0 ? -1 : n; });
}

static inline int coded_ld(int H) { return (H + 7) & ~7; }

// marker build: dense rows (M != nullptr) and / or row-dictionary records (out != nullptr)
static int build_sparse_impl(const char *who, const uint8_t *maj, const double *lhit, const double *lmiss,
                             const int32_t *mk_ptr, const uint16_t *mk_hap, const uint8_t *mk_base,
                             const int64_t *row_ptr, const uint16_t *site, const uint8_t *obs, const int64_t *order,
                             int64_t R, int32_t H, int32_t S, double *M, int64_t ldm, const spb_records *out,
                             int64_t *fallback, int64_t *n_fallback, hipStream_t s) {
    if (R < 0 || H <= 0 || S <= 0) return fail(-1, "%s: bad shape R=%lld H=%lld", who, R, H);
    if (H > 8192) return fail(-1, "%s: more than 8192 haplogroups (H=%lld): use mxm_build_em_matrix", who, H);
    if (M != nullptr && ldm < H) return fail(-1, "%s: ldm < H", who);
    if (maj == nullptr || mk_ptr == nullptr || fallback == nullptr || n_fallback == nullptr)
        return fail(-1, "%s: marker tables and the fallback list are required", who);
    HIP_TRY(hipMemsetAsync(n_fallback, 0, sizeof(int64_t), s));
    if (out != nullptr) HIP_TRY(hipMemsetAsync(out->stats, 0, 2 * sizeof(int64_t), s));
    if (R == 0) return 0;
    const int hpad = (H + 1) & ~1;
    const int nch = (hpad / 2 + SPB_THREADS - 1) / SPB_THREADS;
    const int vec_ok = (M != nullptr) && ((ldm & 1) == 0) && ((reinterpret_cast<uintptr_t>(M) & 15) == 0);
#ifndef SPB_PASSES
#define SPB_PASSES 3
#endif
    // column ranges per row: the mask array is 1 / passes of a row, which (with the table) decides how many rows a CU holds.
    // 10^6 x 5408, kernel alone, since the marker entries are looked up once per row: 2 ranges 12.0-12.3 ms (4-5 rows per CU),
    // 3 ranges 10.5 ms (6 rows; 12.4 when compiled for 5), 4 ranges 11.2-12.4 ms (profiles/r03/build_kernel_experiments.txt)
    const int passes = SPB_PASSES;
    const int maxd = (T.sparse_maxd < 0 || T.sparse_maxd > SPB_MAXD) ? SPB_MAXD : T.sparse_maxd;
    const int maxd_long = (T.sparse_maxd < 0 || T.sparse_maxd > SPB_MAXD_LONG) ? SPB_MAXD_LONG : T.sparse_maxd;
    const int kpp = (nch + passes - 1) / passes;
    const size_t lds = (size_t)kpp * 2 * SPB_THREADS * 8 + SPB_SLOTS * 8 + 6 * 1024;      // mask array + table + the kernel's other LDS
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    const int grid = clamp_grid(R, num_cu() * per_cu * 2);
    const int long_too = T.sparse_long;                  // (0: rows beyond 64 observations go to the fallback list, as before round 6)
    const int grid_long = clamp_grid((R + SPB_LONG_CHUNK - 1) / SPB_LONG_CHUNK, num_cu() * 4 * 2);
    spb_records none = {};
    const spb_records rec = out != nullptr ? *out : none;
#define SPB_ARGS maj, lhit, lmiss, mk_ptr, mk_hap, mk_base, row_ptr, site, obs, order, R, (int)H, M, ldm, vec_ok, fallback, reinterpret_cast<unsigned long long *>(n_fallback), maxd, rec
#define SPB_LAUNCH(n, p) do { if (out != nullptr) hipLaunchKernelGGL((build_sparse_kernel<n, p, true, 1>), dim3(grid), dim3(SPB_THREADS), 0, s, SPB_ARGS, long_too, 0); \
                          else hipLaunchKernelGGL((build_sparse_kernel<n, p, false, 1>), dim3(grid), dim3(SPB_THREADS), 0, s, SPB_ARGS, long_too, 0); } while (0)
    // (only the instances of the column-range count this build runs with are compiled: the other three quarters of them
    // were a third of the library's build time)
#define SPB_CASE(n) case n: if constexpr (SPB_PASSES == 1) { if constexpr (n <= 7) SPB_LAUNCH(n, 1); else return fail(-1, "%s: one pass covers H <= 3584", who); } else if constexpr (SPB_PASSES == 2) SPB_LAUNCH(n, 2); else if constexpr (SPB_PASSES == 3) SPB_LAUNCH(n, 3); else SPB_LAUNCH(n, 4); break;
    switch (nch) {
        SPB_CASE(1) SPB_CASE(2) SPB_CASE(3) SPB_CASE(4) SPB_CASE(5) SPB_CASE(6) SPB_CASE(7) SPB_CASE(8)
        SPB_CASE(9) SPB_CASE(10) SPB_CASE(11) SPB_CASE(12) SPB_CASE(13) SPB_CASE(14) SPB_CASE(15) SPB_CASE(16)
        default: return fail(-1, "%s: H=%lld outside the kernel's range", who, H);
    }
    HIP_TRY(hipGetLastError());
    if (long_too != 0) {
        // the rows of 65 .. 128 observations (merged mates, long reads): 128-bit masks, one column range per 512 haplogroups
        // and a table of 1024 slots (build_sparse_kernels.hpp, W = 2); its grid looks at the rows 64 at a time
#define SPB_ARGS_LONG maj, lhit, lmiss, mk_ptr, mk_hap, mk_base, row_ptr, site, obs, order, R, (int)H, M, ldm, vec_ok, fallback, reinterpret_cast<unsigned long long *>(n_fallback), maxd_long, rec
#define SPB_LONG(n, p) case n: if (out != nullptr) hipLaunchKernelGGL((build_sparse_kernel<n, p, true, 2>), dim3(grid_long), dim3(SPB_THREADS), 0, s, SPB_ARGS_LONG, 0, T.sparse_long_entries); \
                               else hipLaunchKernelGGL((build_sparse_kernel<n, p, false, 2>), dim3(grid_long), dim3(SPB_THREADS), 0, s, SPB_ARGS_LONG, 0, T.sparse_long_entries); break;
        switch (nch) {
#ifndef SPB_LONG_KPP
#define SPB_LONG_KPP 2                // 512-haplogroup chunks per column range of the long rows' instance (1: 34.9, 2: 34.5, 4: 36.5 ms
                                      // per 10^6 paired-end fragments before the walk over a long row's further entries was fixed)
#endif
#define SPB_LP(n) SPB_LONG(n, ((n + SPB_LONG_KPP - 1) / SPB_LONG_KPP))
            SPB_LP(1) SPB_LP(2) SPB_LP(3) SPB_LP(4) SPB_LP(5) SPB_LP(6) SPB_LP(7) SPB_LP(8)
            SPB_LP(9) SPB_LP(10) SPB_LP(11) SPB_LP(12) SPB_LP(13) SPB_LP(14) SPB_LP(15) SPB_LP(16)
#undef SPB_LP
            default: break;
        }
#undef SPB_LONG
#undef SPB_ARGS_LONG
    }
#undef SPB_CASE
#undef SPB_LAUNCH
#undef SPB_ARGS
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_build_em_matrix_sparse(const uint8_t *maj, const double *lhit, const double *lmiss,
                                          const int32_t *mk_ptr, const uint16_t *mk_hap, const uint8_t *mk_base,
                                          const int64_t *row_ptr, const uint16_t *site, const uint8_t *obs,
                                          const int64_t *order, int64_t R, int32_t H, int32_t S, double *M, int64_t ldm,
                                          int64_t *fallback, int64_t *n_fallback, void *stream) {
    MXM_ENTER();
    if (M == nullptr) return fail(-1, "mxm_build_em_matrix_sparse: M required%s", "");
    return build_sparse_impl("mxm_build_em_matrix_sparse", maj, lhit, lmiss, mk_ptr, mk_hap, mk_base, row_ptr, site, obs,
                             order, R, H, S, M, ldm, nullptr, fallback, n_fallback, (hipStream_t)stream);
}

extern "C" size_t mxm_record_bytes(int64_t R, int32_t H) {
    if (R < 0 || H <= 0) return 0;
    // the larger of a full byte-coded record and a full wide one
    return (size_t)(R > 0 ? R : 1) * (2 * (size_t)coded_ld(H) + 16 * ENC_MAX_WIDE);
}

extern "C" int mxm_build_em_records(const uint8_t *maj, const double *lhit, const double *lmiss,
                                    const int32_t *mk_ptr, const uint16_t *mk_hap, const uint8_t *mk_base,
                                    const int64_t *row_ptr, const uint16_t *site, const uint8_t *obs,
                                    const int64_t *order, int64_t R, int32_t H, int32_t S, double *M, int64_t ldm,
                                    uint8_t *rec, size_t rec_bytes, int64_t *rec_off, int32_t *ndist, double *rowmax,
                                    int64_t *stats, int64_t *fallback, int64_t *n_fallback, void *stream) {
    MXM_ENTER();
    if (!mxm_linear_supported(H))
        return fail(-1, "mxm_build_em_records: records need H in [65, 8192]%s (H=%lld)", "", H);
    if (rec == nullptr || (reinterpret_cast<uintptr_t>(rec) & 15) || rec_bytes < (size_t)coded_ld(H) + 16 * ENC_MAX_CODES ||
        rec_off == nullptr || ndist == nullptr || rowmax == nullptr || stats == nullptr)
        return fail(-1, "mxm_build_em_records: record buffer (16-byte aligned, >= one record) and output arrays required%s", "");
    spb_records out;
    out.rec = rec;
    out.rec_cap = (long long)rec_bytes;
    out.rec_off = rec_off;
    out.ndist = ndist;
    out.rowmax = rowmax;
    out.stats = reinterpret_cast<unsigned long long *>(stats);
    out.ldc = coded_ld(H);
    return build_sparse_impl("mxm_build_em_records", maj, lhit, lmiss, mk_ptr, mk_hap, mk_base, row_ptr, site, obs, order,
                             R, H, S, M, ldm, &out, fallback, n_fallback, (hipStream_t)stream);
}

template <typename ST>
static int linearize_wide(const double *M, int64_t ldm, int64_t R, int H, ST *P, int64_t ldp, double *rowmax,
                          hipStream_t s) {
    const int nch = (H / 2 + 255) / 256;
    int cap = num_cu() * 2 < MXM_MAX_WG ? num_cu() * 2 : MXM_MAX_WG;
    const int nwg = clamp_grid((R + 1) / 2, cap);           // rows are dealt round-robin (row_deal)
    switch (nch) {
#define LW_CASE(n) case n: hipLaunchKernelGGL((linearize_wide_kernel<n, ST>), dim3(nwg), dim3(256), 0, s, M, ldm, R, H, P, ldp, rowmax); break;
        LW_CASE(1) LW_CASE(2) LW_CASE(3) LW_CASE(4) LW_CASE(5) LW_CASE(6) LW_CASE(7) LW_CASE(8)
        LW_CASE(9) LW_CASE(10) LW_CASE(11) LW_CASE(12) LW_CASE(13) LW_CASE(14) LW_CASE(15) LW_CASE(16)
#undef LW_CASE
        default: return 1;
    }
    return 0;
}

static inline bool wide_rows_ok(const void *M, int64_t ldm, int H) {
    return ((H & 1) == 0) && ((ldm & 1) == 0) && ((reinterpret_cast<uintptr_t>(M) & 15) == 0) && H >= 64 && H <= 8192;
}

extern "C" int mxm_linearize(const double *M, int64_t ldm, int64_t R, int32_t H, double *P, int64_t ldp,
                             double *rowmax, void *stream) {
    if (R < 0 || H <= 0) return fail(-1, "mxm_linearize: bad shape%s", "");
    if (ldm < H || ldp < H || (ldp & 1)) return fail(-1, "mxm_linearize: ldp must be even and >= H%s (ldp=%lld)", "", ldp);
    if (R == 0) return 0;
    if (wide_rows_ok(M, ldm, H) && (reinterpret_cast<uintptr_t>(P) & 15) == 0 &&
        linearize_wide<double>(M, ldm, R, (int)H, P, ldp, rowmax, (hipStream_t)stream) == 0) {
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const int grid = clamp_grid(R, num_cu() * 8);
    hipLaunchKernelGGL(linearize_kernel, dim3(grid), dim3(ROW_THREADS), 0, (hipStream_t)stream, M, ldm, R,
                       (int)H, P, ldp, rowmax);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- optional timing hook (bench.py): events recorded right around the dominant kernel --------
extern "C" int mxm_set_timing_events(void *ev_start, void *ev_stop) {
    return tune_set([=](mxm_tuning &t) { t.ev_start = (hipEvent_t)ev_start; t.ev_stop = (hipEvent_t)ev_stop; });
}

extern "C" int mxm_set_min_rows_per_wg(int32_t n) {
    if (n < 1) return fail(-1, "mxm_set_min_rows_per_wg: n < 1%s", "");
    return tune_set([n](mxm_tuning &t) { t.min_rows_per_wg = n; });
}

extern "C" int mxm_set_compact_restarts(int32_t mode) {
    return tune_set([mode](mxm_tuning &t) { t.compact_restarts = mode < 0 ? 0 : (mode > 2 ? 2 : mode); });
}

extern "C" int mxm_set_batch_tile(int32_t bt) {
    if (bt < 1 || bt > 4) return fail(-1, "mxm_set_batch_tile: tile must be 1..4%s", "");
    return tune_set([bt](mxm_tuning &t) { t.max_bt = bt; });
}

extern "C" int mxm_set_fused_coded_grid(int32_t nwg) {
    return tune_set([nwg](mxm_tuning &t) { t.fused_coded_wg = nwg > 0 ? nwg : 0; });
}

extern "C" int mxm_set_sparse_long_rows(int32_t on) {
    return tune_set([on](mxm_tuning &t) { t.sparse_long = on ? 1 : 0; });
}
extern "C" int mxm_set_sparse_long_entries(int32_t n) {
    return tune_set([n](mxm_tuning &t) { t.sparse_long_entries = n > 0 ? n : 0; });
}
extern "C" int mxm_set_coded_batch_tile(int32_t bt) {
    if (bt != 1 && bt != 3) return fail(-1, "mxm_set_coded_batch_tile: 1 or 3, got %s%lld", "", bt);
    return tune_set([bt](mxm_tuning &t) { t.coded_bt = bt; });
}
extern "C" int mxm_set_quad_encoder(int32_t kind) {
    if (kind != 0 && kind != 1) return fail(-1, "mxm_set_quad_encoder: 0 or 1, got %s%lld", "", kind);
    return tune_set([kind](mxm_tuning &t) { t.quad_encoder = kind; });
}
extern "C" int mxm_set_quad_left_grid(int32_t nwg) {
    return tune_set([nwg](mxm_tuning &t) { t.quad_left_wg = nwg > 0 ? nwg : 0; });
}

extern "C" int mxm_diag_fused_force_abort(int32_t on) {
    return tune_set([on](mxm_tuning &t) { t.fused_force_abort = on ? 1 : 0; });
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
#define MXM_V2_NBUF 2                 // with dealt rows: 6.42 ms vs 6.85 ms (ring 3) per pass at 1M x 5408 (profiles/r01/row_mapping.txt)
#endif
#ifndef MXM_V3_THREADS
#define MXM_V3_THREADS 512
#endif
#ifndef MXM_V3_NBUF
#define MXM_V3_NBUF 2
#endif
#ifndef MXM_V4_THREADS
#define MXM_V4_THREADS 512
#endif
#ifndef MXM_V4_NBUF
#define MXM_V4_NBUF 2
#endif
// restarts of a batch whose proportions stay in VGPRs (the rest sit in LDS: 160 KiB hold three
// vectors of H = 5408, so four restarts per pass need one of them in registers)
#ifndef MXM_V2_PREG
#define MXM_V2_PREG 1
#endif
#ifndef MXM_V3_PREG
#define MXM_V3_PREG 2
#endif
#ifndef MXM_V4_PREG
#define MXM_V4_PREG 2
#endif
// the single-restart shape of the dense matrix (512 threads, one workgroup per CU, ring 3: in-process A/B 6.43 vs 6.50 ms
// median, profiles/r01/tune_sweep.txt); the 256-thread shape above runs the dense leftover rows of a row-dictionary plan
#ifndef MXM_V1B_THREADS
#define MXM_V1B_THREADS 512
#endif
#ifndef MXM_V1B_NBUF
#define MXM_V1B_NBUF 3
#endif
#define MXM_MAX_COL2 4096             // column pairs per row the register tiling covers (H <= 8192)
#define MXM_LDS_BUDGET (156 * 1024)   // of the CU's 160 KiB, leaving room for the exchange buffers

static inline int variant_threads(int nb) {
    return nb == 1 ? MXM_V1_THREADS : (nb == 2 ? MXM_V2_THREADS : (nb == 3 ? MXM_V3_THREADS : MXM_V4_THREADS));
}
static inline int variant_preg(int nb) {
    return nb == 1 ? 1 : (nb == 2 ? MXM_V2_PREG : (nb == 3 ? MXM_V3_PREG : MXM_V4_PREG));
}

static size_t batch_lds_bytes(int H, int nb) {
    const int threads = variant_threads(nb);
    const int nch = ((H + 1) / 2 + threads - 1) / threads;
    return (size_t)(nb - variant_preg(nb)) * nch * threads * 16;
}

// A tile of nb restarts needs its LDS-resident proportions to fit and a spill-free kernel
// instance (scratch in the row loop costs 2-5x): column chunks per thread up to which the default
// shapes compile without scratch (-Rpass-analysis=kernel-resource-usage), per tile size.
static bool batch_fits(int H, int nb) {
    if (nb <= 1) return true;
    static const int max_nch[MXM_MAX_BT + 1] = {0, 16, 8, 7, 6};
    const int threads = variant_threads(nb);
    const int nch = ((H + 1) / 2 + threads - 1) / threads;
    return nch <= max_nch[nb] && batch_lds_bytes(H, nb) <= MXM_LDS_BUDGET;
}

template <int THREADS, int NCH, int BT, int NBUF, int PREG>
static int launch_wide(const double *P, int64_t ldp, const double *w, const double *props, int64_t R,
                       int H, int grid, double *partial, int64_t ldpart,
                       const mxm_em_state *state, mxm_slots slots, hipStream_t stream) {
    if constexpr (NCH * THREADS > MXM_MAX_COL2) {
        return fail(-1, "mxm_em_iter: H=%s%lld outside the linear kernel's range", "", H);
    } else {
        const size_t lds = (size_t)(BT - PREG) * NCH * THREADS * sizeof(d2);
        if (lds > 60 * 1024 &&
            raise_dynamic_lds(reinterpret_cast<const void *>(&em_iter_wide_kernel<THREADS, NCH, BT, NBUF, PREG>), lds,
                              "em_iter_wide_kernel") != hipSuccess)
            return -2;
        hipLaunchKernelGGL((em_iter_wide_kernel<THREADS, NCH, BT, NBUF, PREG>), dim3(grid), dim3(THREADS), lds,
                           stream, P, ldp, w, props, R, H, partial, ldpart, state, slots);
        return 0;
    }
}

template <int THREADS, int BT, int NBUF, int PREG>
static int dispatch_wide(int nch, const double *P, int64_t ldp, const double *w, const double *props,
                         int64_t R, int H, int grid, double *partial,
                         int64_t ldpart, const mxm_em_state *state, mxm_slots slots, hipStream_t stream) {
    switch (nch) {
#define WIDE_CASE(n) case n: return launch_wide<THREADS, n, BT, NBUF, PREG>(P, ldp, w, props, R, H, grid, partial, ldpart, state, slots, stream);
        WIDE_CASE(1) WIDE_CASE(2) WIDE_CASE(3) WIDE_CASE(4) WIDE_CASE(5) WIDE_CASE(6) WIDE_CASE(7) WIDE_CASE(8)
        WIDE_CASE(9) WIDE_CASE(10) WIDE_CASE(11) WIDE_CASE(12) WIDE_CASE(13) WIDE_CASE(14) WIDE_CASE(15) WIDE_CASE(16)
#undef WIDE_CASE
        default: break;
    }
    return fail(-1, "mxm_em_iter: H=%s%lld outside the linear kernel's range", "", H);
}

static inline mxm_slots slots_from(int first) {
    mxm_slots sl;
    for (int i = 0; i < MXM_MAX_BT; ++i) sl.s[i] = first + i;
    return sl;
}

// One tile of nb (<= MXM_MAX_BT) restarts over the linear matrix: streaming kernel + column reduce.
// props / state / colsum are the BASES of the loop vectors; the tile's members are slots.s[0 .. nb).
// stream_linear_tile: the streaming kernel alone, partial rows [0, *nwg_out) of `partial`.
static int stream_linear_tile(const double *P, int64_t ldp, const double *w, const double *props, int64_t R,
                              int H, int nb, mxm_slots &slots, const mxm_em_state *state,
                              double *partial, hipStream_t stream, bool timed, int *nwg_out, int max_wg = MXM_MAX_WG,
                              int v1_shape = -1) {
    const int64_t ldpart = part_ld(H);
    const int ncol2 = (H + 1) / 2;
    if (v1_shape < 0) v1_shape = 1;
    const bool alt = (nb == 1) && (v1_shape == 1);          // the second single-restart shape
    const int threads = alt ? MXM_V1B_THREADS : variant_threads(nb);
    const int wg_per_cu = alt ? 1 : ((nb == 1) ? MXM_V1_WG_PER_CU : 1);
    const int nch = (ncol2 + threads - 1) / threads;
    int cap = num_cu() * wg_per_cu;
    if (cap > max_wg) cap = max_wg;
    // rows are dealt round-robin over the workgroups (row_deal, common.hpp).  Small matrices:
    // fewer workgroups with more rows each (every workgroup pays 2 x H x 8 bytes of proportion
    // loads and partial stores, which colreduce then reads back)
    const int nwg = clamp_grid((R + T.min_rows_per_wg - 1) / T.min_rows_per_wg, cap);
    for (int i = nb; i < MXM_MAX_BT; ++i) slots.s[i] = slots.s[0];          // unused entries stay in range
    if (timed && T.ev_start != nullptr) HIP_TRY(hipEventRecord(T.ev_start, stream));
    int rc;
    if (alt)
        rc = dispatch_wide<MXM_V1B_THREADS, 1, MXM_V1B_NBUF, 1>(nch, P, ldp, w, props, R, H, nwg, partial, ldpart, state, slots, stream);
    else if (nb == 1)
        rc = dispatch_wide<MXM_V1_THREADS, 1, MXM_V1_NBUF, 1>(nch, P, ldp, w, props, R, H, nwg, partial, ldpart, state, slots, stream);
    else if (nb == 2)
        rc = dispatch_wide<MXM_V2_THREADS, 2, MXM_V2_NBUF, MXM_V2_PREG>(nch, P, ldp, w, props, R, H, nwg, partial, ldpart, state, slots, stream);
    else if (nb == 3)
        rc = dispatch_wide<MXM_V3_THREADS, 3, MXM_V3_NBUF, MXM_V3_PREG>(nch, P, ldp, w, props, R, H, nwg, partial, ldpart, state, slots, stream);
    else
        rc = dispatch_wide<MXM_V4_THREADS, 4, MXM_V4_NBUF, MXM_V4_PREG>(nch, P, ldp, w, props, R, H, nwg, partial, ldpart, state, slots, stream);
    if (rc != 0) return rc;
    HIP_TRY(hipGetLastError());
    if (timed && T.ev_stop != nullptr) HIP_TRY(hipEventRecord(T.ev_stop, stream));
    *nwg_out = nwg;
    return 0;
}

// The template instance stream_linear_tile launches for a tile of nb restarts at width H, as a profiler prints it
// ("em_iter_wide_kernel<512, 6, 1, 3, 1>"): bench.py matches its committed counter files by this name.
extern "C" int mxm_describe_stream_kernel(int32_t H, int32_t nb, char *buf, size_t len) {
    if (buf == nullptr || len == 0 || H <= 0 || nb < 1 || nb > MXM_MAX_BT) return fail(-1, "mxm_describe_stream_kernel: bad arguments%s", "");
    const int ncol2 = (H + 1) / 2;
    const bool alt = nb == 1;
    const int threads = alt ? MXM_V1B_THREADS : variant_threads(nb);
    const int nbuf = alt ? MXM_V1B_NBUF : (nb == 2 ? MXM_V2_NBUF : (nb == 3 ? MXM_V3_NBUF : MXM_V4_NBUF));
    snprintf(buf, len, "em_iter_wide_kernel<%d, %d, %d, %d, %d>", threads, (ncol2 + threads - 1) / threads, nb, nbuf,
             variant_preg(nb));
    return 0;
}

// What the loop driver hands down when the finalize may ride on the column reduce's launch (no collective in between).
struct fin_args {
    double *ln_cur, *ln_new, *props_cur;
    mxm_em_state *state;
    double tol;
    int max_iter;
};

// column reduce of a tile's partial rows [0, nwg); with `fin` the finalize follows in the same launch
static int reduce_tile(const double *partial, int nwg, int nb, int H, double *colsum, const mxm_em_state *state,
                       const mxm_slots &slots, const fin_args *fin, hipStream_t stream,
                       wide_check wc = wide_check{nullptr, 0, 0}) {
    if (fin != nullptr && H <= 64 * FIN_MAX_BLOCKS)
        hipLaunchKernelGGL(colreduce_finalize_kernel, dim3((H + 63) / 64, nb), dim3(COLRED_THREADS), 0, stream, partial,
                           part_ld(H), nwg, nb, H, colsum, fin->ln_cur, fin->ln_new, fin->props_cur, fin->tol, fin->max_iter,
                           fin->state, slots, wc);
    else
        hipLaunchKernelGGL(colreduce_kernel, dim3((H + 63) / 64, nb), dim3(COLRED_THREADS), 0, stream, partial, part_ld(H), nwg,
                           nb, H, (const double *)nullptr, colsum, const_cast<mxm_em_state *>(state), slots, wc);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int em_iter_linear_tile(const double *P, int64_t ldp, const double *w, const double *props, int64_t R,
                               int H, int nb, mxm_slots slots, const mxm_em_state *state, double *colsum,
                               double *partial, hipStream_t stream, bool timed, const fin_args *fin = nullptr) {
    int nwg = 0;
    const int rc = stream_linear_tile(P, ldp, w, props, R, H, nb, slots, state, partial, stream, timed, &nwg);
    if (rc != 0) return rc;
    return reduce_tile(partial, nwg, nb, H, colsum, state, slots, fin, stream);
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
                           int grid, double *partial, int64_t ldpart,
                           const mxm_em_state *state, hipStream_t stream) {
    if constexpr (NCH * MXM_F32_THREADS > 2048) {
        return fail(-1, "mxm_em_iter_f32: H=%s%lld outside the kernel's range", "", H);
    } else {
        hipLaunchKernelGGL((em_iter_wide_f32_kernel<MXM_F32_THREADS, NCH, MXM_F32_NBUF>), dim3(grid),
                           dim3(MXM_F32_THREADS), 0, stream, P, ldp, w, props, R, H, partial, ldpart,
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
    const int nwg = clamp_grid((R + nbuf - 1) / nbuf, cap);
    if (timed && T.ev_start != nullptr) HIP_TRY(hipEventRecord(T.ev_start, stream));
    switch (nch) {
#define F32_CASE(n) case n: { const int lrc = launch_wide_f32<n>(P, ldp, w, props, R, H, nwg, partial, ldpart, state, stream); if (lrc != 0) return lrc; } break;
        F32_CASE(1) F32_CASE(2) F32_CASE(3) F32_CASE(4) F32_CASE(5) F32_CASE(6) F32_CASE(7) F32_CASE(8)
#undef F32_CASE
        default: return fail(-1, "mxm_em_iter_f32: H=%s%lld outside the kernel's range", "", H);
    }
    HIP_TRY(hipGetLastError());
    if (timed && T.ev_stop != nullptr) HIP_TRY(hipEventRecord(T.ev_stop, stream));
    hipLaunchKernelGGL(colreduce_kernel, dim3((H + 63) / 64, 1), dim3(COLRED_THREADS), 0, stream, partial, ldpart, nwg,
                       1, H, (const double *)nullptr, colsum, const_cast<mxm_em_state *>(state), slots_from(0), wide_check{nullptr, 0, 0});
    HIP_TRY(hipGetLastError());
    return 0;
}

// Narrow matrices (the refinement EM's contributor columns): one thread per row up to 32 columns,
// a lane pair per row up to 64.
#define MXM_NARROW_MAX_H 64
template <bool ITER>
static int launch_narrow(const double *M, int64_t ldm, const double *w, const double *ln_props, int64_t R, int H,
                         double *out, int64_t ldo, int mode, double *partial, int64_t ldpart,
                         const mxm_em_state *state, hipStream_t stream, int *nwg_out) {
    const int cap = num_cu() * 4 < MXM_MAX_WG ? num_cu() * 4 : MXM_MAX_WG;
    const int rows_per_wg = (H <= 32) ? 256 : 128;
    const int nwg = clamp_grid((R + rows_per_wg - 1) / rows_per_wg, cap);
    *nwg_out = nwg;
#define NARROW_LAUNCH(hmax, lpr)                                                                             \
    hipLaunchKernelGGL((estep_narrow_kernel<hmax, ITER, lpr>), dim3(nwg), dim3(256), 0, stream, M, ldm, w, ln_props, R, H, \
                       out, ldo, mode, partial, ldpart, state)
    if (H <= 4) NARROW_LAUNCH(4, 1);
    else if (H <= 8) NARROW_LAUNCH(8, 1);
    else if (H <= 16) NARROW_LAUNCH(16, 1);
    else if (H <= 32) NARROW_LAUNCH(32, 1);
    else NARROW_LAUNCH(32, 2);
#undef NARROW_LAUNCH
    HIP_TRY(hipGetLastError());
    return 0;
}

static int em_iter_log_one(const double *M, int64_t ldm, const double *w, const double *ln_props, int64_t R, int H,
                           const mxm_em_state *state, double *colsum, double *partial, hipStream_t stream) {
    if (M == nullptr) return fail(-1, "mxm_em_iter: M is NULL and the linear path does not apply%s", "");
    if (ln_props == nullptr) return fail(-1, "mxm_em_iter: ln_props is NULL and the log-space path needs it%s", "");
    const size_t lds = 2 * (size_t)H * sizeof(double);
    const int64_t ldpart = part_ld(H);
    int nwg;
    if (lds > 150 * 1024) {
        // wider than the log-space kernel's LDS vectors (9600 columns): the any-width form (vectors in global memory)
        nwg = clamp_grid(R, num_cu() * 2 < MXM_MAX_WG ? num_cu() * 2 : MXM_MAX_WG);
        hipLaunchKernelGGL((estep_global_kernel<true>), dim3(nwg), dim3(ROW_THREADS), 0, stream, M, ldm, w, ln_props, R, H,
                           (double *)nullptr, (int64_t)0, 0, partial, ldpart, state, (const int64_t *)nullptr);
        HIP_TRY(hipGetLastError());
    } else if (H <= MXM_NARROW_MAX_H) {
        const int rc = launch_narrow<true>(M, ldm, w, ln_props, R, H, (double *)nullptr, (int64_t)0, 0, partial, ldpart,
                                           state, stream, &nwg);
        if (rc != 0) return rc;
    } else {
        nwg = clamp_grid(R, num_cu() * 2 < MXM_MAX_WG ? num_cu() * 2 : MXM_MAX_WG);
        if (lds > 60 * 1024 && raise_dynamic_lds(reinterpret_cast<const void *>(&estep_log_kernel<true>), lds, "estep_log_kernel") != hipSuccess)
            return -2;
        hipLaunchKernelGGL((estep_log_kernel<true>), dim3(nwg), dim3(ROW_THREADS), lds, stream, M, ldm, w, ln_props,
                           R, H, (double *)nullptr, (int64_t)0, 0, partial, ldpart, state);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(colreduce_kernel, dim3((H + 63) / 64, 1), dim3(COLRED_THREADS), 0, stream, partial, ldpart, nwg, 1,
                       H, (const double *)nullptr, colsum, const_cast<mxm_em_state *>(state), slots_from(0), wide_check{nullptr, 0, 0});
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_em_iter(const double *M, int64_t ldm, const double *P, int64_t ldp, const double *w,
                           const double *props, const double *ln_props, int64_t R, int32_t H, int32_t B,
                           const mxm_em_state *state, double *colsum, void *ws, size_t ws_bytes, void *stream) {
    MXM_ENTER();
    if (R <= 0 || H <= 0 || B <= 0) return fail(-1, "mxm_em_iter: bad shape R=%s%lld H=%lld", "", R, H);
    if (ws == nullptr || ws_bytes < mxm_workspace_bytes(R, H, B)) return fail(-1, "mxm_em_iter: workspace too small%s", "");
    if (P != nullptr && ((ldp & 1) || ldp < H)) return fail(-1, "mxm_em_iter: ldp must be even and >= H%s", "");
    if (M != nullptr && ldm < H) return fail(-1, "mxm_em_iter: ldm < H%s", "");
    const bool linear = (P != nullptr) && mxm_linear_supported(H);
    // restarts are taken in tiles of up to T.max_bt that share one pass over the matrix; the
    // scratch is reused tile after tile (same stream, so the passes are ordered)
    // a batch keeps (most of) its proportion vectors in LDS: the largest tile that fits
    int max_bt = linear ? T.max_bt : 1;
    while (max_bt > 1 && !batch_fits((int)H, max_bt)) --max_bt;
    for (int b = 0; b < B;) {
        // the fewest passes that cover what is left, restarts spread evenly over them (a pass
        // costs nearly the same for 1..4 restarts, a little more the fuller it is):
        // 10 -> 4 + 3 + 3, 5 -> 3 + 2, 8 -> 4 + 4
        const int left = B - b;
        const int passes = (left + max_bt - 1) / max_bt;
        const int nb = (left + passes - 1) / passes;
        const mxm_em_state *st = state ? state + b : nullptr;
        int rc;
        if (linear)
            rc = em_iter_linear_tile(P, ldp, w, props, R, (int)H, nb, slots_from(b), state, colsum, (double *)ws,
                                     (hipStream_t)stream, b == 0);
        else
            rc = em_iter_log_one(M, ldm, w, ln_props ? ln_props + (int64_t)b * H : nullptr, R, (int)H, st,
                                 colsum + (int64_t)b * H, (double *)ws, (hipStream_t)stream);
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
    if (wide_rows_ok(M, ldm, H) && (reinterpret_cast<uintptr_t>(P) & 7) == 0 && (ldp & 1) == 0 &&
        linearize_wide<float>(M, ldm, R, (int)H, P, ldp, rowmax, (hipStream_t)stream) == 0) {
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const int grid = clamp_grid(R, num_cu() * 8);
    hipLaunchKernelGGL(linearize_f32_kernel, dim3(grid), dim3(ROW_THREADS), 0, (hipStream_t)stream, M, ldm, R, (int)H,
                       P, ldp, rowmax);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_em_iter_f32(const float *P, int64_t ldp, const double *w, const double *props, int64_t R,
                               int32_t H, int32_t B, const mxm_em_state *state, double *colsum, void *ws,
                               size_t ws_bytes, void *stream) {
    MXM_ENTER();
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

// ---- row-dictionary storage (coded_kernels.hpp) ------------------------------------------------
extern "C" size_t mxm_coded_bytes(int64_t R, int32_t H) { return mxm_record_bytes(R, H); }

extern "C" int mxm_encode_rows(const double *M, int64_t ldm, int64_t R, int32_t H, uint8_t *rec, size_t rec_bytes,
                               int64_t *rec_off, int32_t *ndist, double *rowmax, int64_t *stats, void *stream) {
    if (R <= 0 || H <= 0 || ldm < H) return fail(-1, "mxm_encode_rows: bad shape R=%s%lld H=%lld", "", R, H);
    // (round 5: any H in the linear kernels' range and any row stride -- a row's loads go through a descriptor of exactly
    // H doubles and need the doubles' own 8-byte alignment only; an odd H or an odd stride used to send the matrix back to
    // the 7 x larger dense form)
    if (!mxm_linear_supported(H) || (reinterpret_cast<uintptr_t>(M) & 7) != 0)
        return fail(-1, "mxm_encode_rows: needs H in [65, 8192]%s (H=%lld ldm=%lld)", "", H, ldm);
    if (rec == nullptr || (reinterpret_cast<uintptr_t>(rec) & 15) || rec_bytes < (size_t)coded_ld(H) + 16 * ENC_MAX_CODES)
        return fail(-1, "mxm_encode_rows: record buffer missing, unaligned or smaller than one record%s", "");
    if (rec_off == nullptr || ndist == nullptr || rowmax == nullptr || stats == nullptr)
        return fail(-1, "mxm_encode_rows: output arrays required%s", "");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(stats, 0, 2 * sizeof(int64_t), s));
    const int ldc = coded_ld(H);
    const int nch = (ldc / 4 + ENC_THREADS - 1) / ENC_THREADS;
    const int grid = clamp_grid(R, num_cu() * 4);
    // first pass: every row, byte codes; second pass: the rows it left without a record, 16-bit codes
    const int grid_w = clamp_grid((R + 31) / 32, num_cu() * 3);
    switch (nch) {
#define ENC_CASE(n) case n: hipLaunchKernelGGL((encode_rows_kernel<n>), dim3(grid), dim3(ENC_THREADS), 0, s, M, ldm, R, (int)H, ldc, rec, (int64_t)rec_bytes, rec_off, ndist, rowmax, reinterpret_cast<unsigned long long *>(stats)); \
                    hipLaunchKernelGGL((encode_wide_rows_kernel<n>), dim3(grid_w), dim3(ENC_THREADS), 0, s, M, ldm, R, (int)H, ldc, rec, (int64_t)rec_bytes, rec_off, ndist, rowmax, reinterpret_cast<unsigned long long *>(stats)); break;
        ENC_CASE(1) ENC_CASE(2) ENC_CASE(3) ENC_CASE(4) ENC_CASE(5) ENC_CASE(6) ENC_CASE(7) ENC_CASE(8)
#undef ENC_CASE
        default: return fail(-1, "mxm_encode_rows: H=%s%lld outside the kernel's range", "", H);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

static int coded_check(const mxm_coded *c, int32_t H, const char *who) {
    if (c == nullptr || c->R <= 0 || H <= 0 || !mxm_linear_supported(H))
        return fail(-1, "%s: bad coded matrix (rows %lld, H %lld)", who, c ? c->R : 0, H);
    if (c->rec == nullptr || c->rec_off == nullptr || c->ndist == nullptr) return fail(-1, "%s: coded matrix arrays missing", who);
    if (c->R_rest < 0 || (c->R_rest > 0 && (c->P_rest == nullptr || c->ldp_rest < H || (c->ldp_rest & 1) ||
                                            (reinterpret_cast<uintptr_t>(c->P_rest) & 15))))
        return fail(-1, "%s: the dense rest needs 16-byte aligned rows with an even ld >= H", who);
    if (c->n_wide < 0 || (c->n_wide > 0 && c->wide_rows == nullptr)) return fail(-1, "%s: n_wide > 0 needs wide_rows", who);
    if (c->qrec != nullptr) {                            // a quad dictionary beside the records
        if (c->qoff == nullptr || c->nquad == nullptr || c->n_quad_rows < 0 || c->n_byte_rows < 0 ||
            (c->n_quad_rows > 0 && c->quad_rows == nullptr) || (c->n_byte_rows > 0 && c->byte_rows == nullptr) ||
            (reinterpret_cast<uintptr_t>(c->qrec) & 31))
            return fail(-1, "%s: quad dictionary arrays missing (or qrec not 32-byte aligned)", who);
        if (c->n_quad_rows + c->n_byte_rows + c->n_wide + c->R_rest != c->R)
            return fail(-1, "%s: quad_rows + byte_rows + wide_rows + the dense rest must be all %lld rows", who, c->R);
    }
    return 0;
}

extern "C" size_t mxm_quad_bytes(int64_t R, int32_t H) {
    (void)H;
    // the encoders hand `qrec` out in pieces of QUAD_CHUNK bytes: a piece holds at least six records of the largest size
    // (2048 + 256 x 32 bytes) and at most 5120 pieces are open -- reserved, partly unused -- when the call ends
    // (R x the largest record alone, the figure until round 6, left the pieces' tails out)
    constexpr size_t largest = QUAD_CODE_BYTES + QUAD_MAX * 32, per_piece = (size_t)QUAD_CHUNK / largest;
    return R > 0 ? (((size_t)R + per_piece - 1) / per_piece + 5120) * (size_t)QUAD_CHUNK : 0;
}

extern "C" int mxm_expand_tables(const uint8_t *maj, const int32_t *mk_ptr, const uint16_t *mk_hap, const uint8_t *mk_base,
                                 const uint8_t *map256, int32_t S, int32_t H, int64_t lde, uint8_t *out, void *stream) {
    if (maj == nullptr || mk_ptr == nullptr || mk_hap == nullptr || mk_base == nullptr || out == nullptr || S <= 0 || H <= 0 || lde < H)
        return fail(-1, "mxm_expand_tables: bad arguments%s", "");
    hipLaunchKernelGGL(expand_tables_kernel, dim3(clamp_grid(S, num_cu() * 8)), dim3(256), 0, (hipStream_t)stream, maj, mk_ptr, mk_hap,
                       mk_base, map256, (int)S, (int)H, lde, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" size_t mxm_quad_lists_scratch_bytes(int64_t R) {
    return R > 0 ? (size_t)((R + QLIST_CHUNK - 1) / QLIST_CHUNK) * 2 * sizeof(long long) : 0;
}

extern "C" int mxm_quad_lists(const int32_t *ndist, const int32_t *nquad, int64_t R, int64_t *quad_rows, int64_t *byte_rows,
                              int64_t *counts, void *scratch, size_t scratch_bytes, void *stream) {
    MXM_ENTER();
    if (ndist == nullptr || nquad == nullptr || R <= 0 || quad_rows == nullptr || byte_rows == nullptr || counts == nullptr)
        return fail(-1, "mxm_quad_lists: bad arguments%s", "");
    if (scratch == nullptr || scratch_bytes < mxm_quad_lists_scratch_bytes(R)) return fail(-1, "mxm_quad_lists: scratch too small%s", "");
    const int64_t nchunk = (R + QLIST_CHUNK - 1) / QLIST_CHUNK;
    if (nchunk > 0x7fffffff) return fail(-1, "mxm_quad_lists: too many rows%s", "");
    hipStream_t s = (hipStream_t)stream;
    long long *cc = static_cast<long long *>(scratch);
    hipLaunchKernelGGL(quad_list_count_kernel, dim3((unsigned)nchunk), dim3(QLIST_THREADS), 0, s, ndist, nquad, R, cc);
    hipLaunchKernelGGL(quad_list_scan_kernel, dim3(1), dim3(QLIST_THREADS), 0, s, cc, nchunk, reinterpret_cast<long long *>(counts));
    hipLaunchKernelGGL(quad_list_fill_kernel, dim3((unsigned)nchunk), dim3(QLIST_THREADS), 0, s, ndist, nquad, R, cc, quad_rows, byte_rows);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_build_quads(const mxm_coded *c, int32_t H, uint8_t *qrec, size_t qrec_bytes, int64_t *qoff, int32_t *nquad,
                               uint64_t *stats, void *stream) {
    MXM_ENTER();
    if (c == nullptr || c->R <= 0 || c->rec == nullptr || c->rec_off == nullptr || c->ndist == nullptr || !mxm_linear_supported(H))
        return fail(-1, "mxm_build_quads: bad coded matrix (rows %s%lld, H %lld)", "", c ? c->R : 0, H);
    if (qoff == nullptr || nquad == nullptr || stats == nullptr || (qrec == nullptr && qrec_bytes > 0) ||
        (reinterpret_cast<uintptr_t>(qrec) & 31))
        return fail(-1, "mxm_build_quads: qoff, nquad, stats and a 32-byte aligned qrec required%s", "");
    const int ldc = coded_ld(H);
    if (ldc / 4 > 8 * QUAD_THREADS) return fail(-1, "mxm_build_quads: H=%s%lld beyond eight quads per thread", "", H);
    HIP_TRY(hipMemsetAsync(stats, 0, 2 * sizeof(uint64_t), (hipStream_t)stream));
    // (either way at most 5120 pieces of QUAD_CHUNK bytes are open at the end: what the caller's first guess leaves room for)
    if (T.quad_encoder == 1 && ldc / 4 <= 64 * QW_K)
        hipLaunchKernelGGL(quad_encode_wave_kernel, dim3(clamp_grid((c->R + QW_WAVES - 1) / QW_WAVES, std::min(num_cu() * QW_MIN_WAVES, 5120 / QW_WAVES))), dim3(QUAD_THREADS),
                           0, (hipStream_t)stream, c->rec, c->rec_off, c->ndist, ldc, (int)H, c->R, qrec,
                           (unsigned long long)qrec_bytes, qoff, nquad, reinterpret_cast<unsigned long long *>(stats));
    else
        hipLaunchKernelGGL(quad_encode_kernel, dim3(clamp_grid(c->R, num_cu() * 8)), dim3(QUAD_THREADS), 0, (hipStream_t)stream,
                           c->rec, c->rec_off, c->ndist, ldc, (int)H, c->R, qrec, (unsigned long long)qrec_bytes, qoff, nquad,
                           reinterpret_cast<unsigned long long *>(stats));
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_decode_rows(const mxm_coded *c, int32_t H, double *P, int64_t ldp, void *stream) {
    const int rc = coded_check(c, H, "mxm_decode_rows");
    if (rc != 0) return rc;
    if (P == nullptr || ldp < H) return fail(-1, "mxm_decode_rows: ldp < H%s", "");
    hipLaunchKernelGGL(decode_rows_kernel, dim3(clamp_grid(c->R, num_cu() * 8)), dim3(256), 0, (hipStream_t)stream, c->rec,
                       c->rec_off, c->ndist, coded_ld(H), c->R, (int)H, P, ldp);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_row_argmax_votes_coded(const mxm_coded *c, int32_t H, int32_t n_runs, const double *ln_props,
                                          const double *props, const double *rowmax, const double *M_rest, int64_t ldm_rest,
                                          const int64_t *rest_rows, int64_t n_rest, const double *w, int32_t *best,
                                          double *votes, void *ws, size_t ws_bytes, void *stream) {
    const int rc = coded_check(c, H, "mxm_row_argmax_votes_coded");
    if (rc != 0) return rc;
    if (ln_props == nullptr || best == nullptr || n_runs < 1 || n_runs > RECK_MAX_RUNS)
        return fail(-1, "mxm_row_argmax_votes_coded: ln_props, best and 1..%s%lld runs required", "", (long long)RECK_MAX_RUNS);
    if (n_runs > 1 && (props == nullptr || rowmax == nullptr))
        return fail(-1, "mxm_row_argmax_votes_coded: several runs need props and rowmax (each run's row normaliser)%s", "");
    if (n_rest < 0 || (n_rest > 0 && (M_rest == nullptr || rest_rows == nullptr || ldm_rest < H)))
        return fail(-1, "mxm_row_argmax_votes_coded: the rows without a record need M_rest, rest_rows and ldm_rest >= H%s", "");
    if (votes != nullptr && (ws == nullptr || ws_bytes < mxm_workspace_bytes(c->R, H, 1)))
        return fail(-1, "mxm_row_argmax_votes_coded: workspace too small%s", "");
    hipStream_t s = (hipStream_t)stream;
    const size_t lse_lds = (size_t)n_runs * sizeof(double);
    const int ldc_a = coded_ld(H);
    const int nch_a = (ldc_a / 4 + 255) / 256;
    bool fast = n_runs == 1 && nch_a <= 8;
    if (fast) {
        // one run: the normaliser drops out -- the register-resident form (records_argmax_kernel)
        switch (nch_a) {
            // one wave of workgroups: as many as are resident at once (a second, partial wave of equally long workgroups
            // would double the kernel's time)
#define RA_CASE(n) case n: { int per_cu = 2; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, records_argmax_narrow_kernel<n>, 256, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 2; } \
                hipLaunchKernelGGL((records_argmax_narrow_kernel<n>), dim3(clamp_grid(c->R, num_cu() * per_cu)), dim3(256), 0, s, c->rec, c->rec_off, c->ndist, ldc_a, c->R, (int)H, ln_props, best); \
                /* the wide rows, found from ndist itself (no list: the vote must not depend on one) */ \
                hipLaunchKernelGGL((records_argmax_kernel<n, true>), dim3(clamp_grid((c->R + 255) / 256, num_cu() * 2)), dim3(256), 0, s, c->rec, c->rec_off, c->ndist, ldc_a, c->R, (int)H, ln_props, best, (const int64_t *)nullptr); } break;
            RA_CASE(1) RA_CASE(2) RA_CASE(3) RA_CASE(4) RA_CASE(5) RA_CASE(6) RA_CASE(7) RA_CASE(8)
#undef RA_CASE
            default: fast = false;
        }
    }
    if (!fast)
        hipLaunchKernelGGL(posterior_argmax_kernel<true>, dim3(clamp_grid(c->R, num_cu() * 8)), dim3(256), lse_lds, s, c->rec, c->rec_off,
                           c->ndist, coded_ld(H), (const double *)nullptr, (int64_t)0, (const int64_t *)nullptr, c->R, (int)H,
                           (int)n_runs, ln_props, props, rowmax, best);
    HIP_TRY(hipGetLastError());
    if (n_rest > 0) {
        hipLaunchKernelGGL(posterior_argmax_kernel<false>, dim3(clamp_grid(n_rest, num_cu() * 8)), dim3(256), lse_lds, s,
                           (const uint8_t *)nullptr, (const int64_t *)nullptr, (const int32_t *)nullptr, 0, M_rest, ldm_rest,
                           rest_rows, n_rest, (int)H, (int)n_runs, ln_props, props, rowmax, best);
        HIP_TRY(hipGetLastError());
    }
    if (votes != nullptr) {
        const int64_t ldpart = part_ld(H);
        const int nwg = clamp_grid((c->R + 255) / 256, num_cu() * 2 < MXM_MAX_WG ? num_cu() * 2 : MXM_MAX_WG);
        const size_t vlds = (size_t)H * sizeof(double);
        if (vlds > 150 * 1024) return fail(-1, "mxm_row_argmax_votes_coded: H=%s%lld too wide for the vote kernel", "", H);
        if (vlds > 60 * 1024 && raise_dynamic_lds(reinterpret_cast<const void *>(&votes_from_best_kernel), vlds, "votes_from_best_kernel") != hipSuccess)
            return -2;
        hipLaunchKernelGGL(votes_from_best_kernel, dim3(nwg), dim3(256), vlds, s, best, w, c->R, (int)H, (double *)ws, ldpart);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(colreduce_kernel, dim3((H + 63) / 64, 1), dim3(COLRED_THREADS), 0, s, (const double *)ws, ldpart, nwg,
                           1, (int)H, (const double *)nullptr, votes, (mxm_em_state *)nullptr, slots_from(0), wide_check{nullptr, 0, 0});
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

static int rest_check(const char *who, int32_t H, const double *M_rest, int64_t ldm_rest, const int64_t *rest_rows, int64_t n_rest) {
    if (n_rest < 0 || (n_rest > 0 && (M_rest == nullptr || rest_rows == nullptr || ldm_rest < H)))
        return fail(-1, "%s: the rows without a record need M_rest, rest_rows and ldm_rest >= H", who);
    return 0;
}

extern "C" int mxm_em_step_coded(const mxm_coded *c, int32_t H, const double *ln_props, const double *props,
                                 const double *rowmax, const double *M_rest, int64_t ldm_rest, const int64_t *rest_rows,
                                 int64_t n_rest, double *out, int64_t ldo, int32_t mode, void *stream) {
    int rc = coded_check(c, H, "mxm_em_step_coded");
    if (rc != 0) return rc;
    if (ln_props == nullptr || props == nullptr || rowmax == nullptr || out == nullptr || ldo < H)
        return fail(-1, "mxm_em_step_coded: bad arguments%s", "");
    if ((rc = rest_check("mxm_em_step_coded", H, M_rest, ldm_rest, rest_rows, n_rest)) != 0) return rc;
    hipLaunchKernelGGL(coded_posterior_kernel, dim3(clamp_grid(c->R, num_cu() * 8)), dim3(256), 0, (hipStream_t)stream, c->rec,
                       c->rec_off, c->ndist, coded_ld(H), c->R, (int)H, ln_props, props, rowmax, out, ldo, (int)mode);
    HIP_TRY(hipGetLastError());
    if (n_rest > 0) {
        // the rows without a record: the reference's E-step in log space on their dense copies, each written to its
        // own row of `out`
        const size_t lds = 2 * (size_t)H * sizeof(double);
        if (lds > 150 * 1024) return fail(-1, "mxm_em_step_coded: H=%s%lld too large", "", H);
        if (lds > 60 * 1024 && raise_dynamic_lds(reinterpret_cast<const void *>(&estep_log_kernel<false>), lds, "estep_log_kernel") != hipSuccess)
            return -2;
        hipLaunchKernelGGL((estep_log_kernel<false>), dim3(clamp_grid(n_rest, num_cu() * 2)), dim3(ROW_THREADS), lds,
                           (hipStream_t)stream, M_rest, ldm_rest, (const double *)nullptr, ln_props, n_rest, (int)H, out, ldo,
                           (int)mode, (double *)nullptr, (int64_t)0, (const mxm_em_state *)nullptr, rest_rows);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

extern "C" int mxm_gather_columns_coded(const mxm_coded *c, int32_t H, const int32_t *cols, int32_t nC, const double *M_rest,
                                        int64_t ldm_rest, const int64_t *rest_rows, int64_t n_rest, double *out,
                                        int64_t ldo, void *stream) {
    int rc = coded_check(c, H, "mxm_gather_columns_coded");
    if (rc != 0) return rc;
    if (cols == nullptr || out == nullptr || nC <= 0 || ldo < nC) return fail(-1, "mxm_gather_columns_coded: bad arguments%s", "");
    if ((rc = rest_check("mxm_gather_columns_coded", H, M_rest, ldm_rest, rest_rows, n_rest)) != 0) return rc;
    const int64_t blocks = (c->R * nC + 255) / 256;
    hipLaunchKernelGGL(coded_gather_columns_kernel, dim3(clamp_grid(blocks, num_cu() * 16)), dim3(256), 0, (hipStream_t)stream,
                       c->rec, c->rec_off, c->ndist, coded_ld(H), c->R, cols, (int)nC, out, ldo);
    HIP_TRY(hipGetLastError());
    if (n_rest > 0) {
        hipLaunchKernelGGL(gather_columns_kernel, dim3(clamp_grid((n_rest * nC + 255) / 256, num_cu() * 16)), dim3(256), 0,
                           (hipStream_t)stream, M_rest, ldm_rest, n_rest, cols, (int)nC, out, ldo, rest_rows);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// Shape of em_iter_coded_kernel {threads, rows in flight, workgroups per CU}: 256 x 4 x 2 measured best of five
// (profiles/r02/coded_shapes.txt)
struct coded_shape { int threads, nbuf, wg_per_cu; };
static const coded_shape g_coded_shapes[] = {{256, 4, 2}};
static const int g_coded_shape = 0;
template <int THREADS, int NBUF, int MINWG>
static int launch_coded(int nch, int nwg, hipStream_t stream, const mxm_coded *c, int ldc, const double *w, const double *props,
                        int H, double *partial, int64_t ldpart, const mxm_em_state *state, int run,
                        const int64_t *row_list = nullptr, int64_t n_list = 0, int part_row0 = 0) {
    switch (nch) {
#define COD_CASE(n) case n: if constexpr (n * THREADS <= 2048) { hipLaunchKernelGGL((em_iter_coded_kernel<THREADS, n, NBUF, MINWG>), dim3(nwg), dim3(THREADS), 0, stream, c->rec, c->rec_off, c->ndist, ldc, w, c->wide_rows, c->n_wide, props, c->R, H, partial, ldpart, state, run, row_list, n_list, c->nquad, part_row0); return 0; } break;
        COD_CASE(1) COD_CASE(2) COD_CASE(3) COD_CASE(4) COD_CASE(5) COD_CASE(6) COD_CASE(7) COD_CASE(8)
#undef COD_CASE
        default: break;
    }
    return fail(-1, "mxm_em_iter_coded: H=%s%lld outside the kernel's range", "", H);
}
#define QUAD_MAX_NCH 6              // H <= 6144: the instances of the quad pass that compile without scratch
// The leftover pass's share of a grid of `slots` workgroups it shares with the quad pass, by the measured cost of a row in
// each pass (us per row and workgroup: quad 0.67; byte 1.0 and wide 1.5 where only a few workgroups run that pass -- a
// sweep of the share at 10^6 rows, profiles/r05/quads_product_1m.txt: 22 workgroups 1.56 ms, 32 1.33, 48 1.36, 64 1.40,
// 128 1.62), so that both finish together; 0 when there is nothing left over.
static int quad_left_share(const mxm_coded *c, int slots) {
    if (c->n_byte_rows <= 0 && c->n_wide <= 0) return 0;
    const double left_cost = 1.0 * (double)c->n_byte_rows + 1.5 * (double)c->n_wide;
    const double quad_cost = 0.67 * (double)c->n_quad_rows;
    int n = (int)((double)slots * left_cost / (left_cost + quad_cost) + 0.5);
    if (T.quad_left_wg > 0) n = T.quad_left_wg;
    if (n < 4) n = 4;
    if (n > slots / 2) n = slots / 2;
    const int64_t left = c->n_byte_rows > c->n_wide ? c->n_byte_rows : c->n_wide;
    n = clamp_grid(left, n);
    return n < 1 ? 1 : n;
}
static int launch_quad(int nch, int nwg, hipStream_t stream, const mxm_coded *c, const double *w, const double *props, int H,
                       double *partial, int64_t ldpart, const mxm_em_state *state, int run) {
    switch (nch) {
#define QUAD_CASE(n) case n: hipLaunchKernelGGL((em_iter_quad_kernel<n, 4>), dim3(nwg), dim3(QUAD_THREADS), 0, stream, c->qrec, c->qoff, c->nquad, c->quad_rows, c->n_quad_rows, c->R, w, props, H, partial, ldpart, 0, state, run); return 0;
        QUAD_CASE(1) QUAD_CASE(2) QUAD_CASE(3) QUAD_CASE(4) QUAD_CASE(5) QUAD_CASE(6)      // (7, 8: 35 registers spilled)
#undef QUAD_CASE
        default: break;
    }
    return fail(-1, "mxm_em_iter_coded: H=%s%lld outside the quad kernel's range", "", H);
}
static int launch_quad_coded(int nch, int nwg, int nwg_left, hipStream_t stream, const mxm_coded *c, int ldc, const double *w,
                             const double *props, int H, double *partial, int64_t ldpart, const mxm_em_state *state, int run) {
    switch (nch) {
#define QC_CASE(n) case n: hipLaunchKernelGGL((em_iter_quad_coded_kernel<n, 4>), dim3(nwg), dim3(QUAD_THREADS), 0, stream, c->rec, c->rec_off, c->ndist, ldc, c->wide_rows, c->n_wide, c->byte_rows != nullptr ? c->byte_rows : c->quad_rows /* an EMPTY list is still a list: a null pointer would mean "every row" to the pass */, c->n_byte_rows, c->qrec, c->qoff, c->nquad, c->quad_rows, c->n_quad_rows, c->R, w, props, H, partial, ldpart, nwg_left, state, run); return 0;
        QC_CASE(1) QC_CASE(2) QC_CASE(3) QC_CASE(4) QC_CASE(5) QC_CASE(6)
#undef QC_CASE
        default: break;
    }
    return fail(-1, "mxm_em_iter_coded: H=%s%lld outside the quad kernel's range", "", H);
}

// One restart's pass over a coded matrix: dictionary rows through em_iter_coded_kernel, the dense rest
// through em_iter_wide_kernel into the partial rows behind, one column reduce over both.
static int em_iter_coded_one(const mxm_coded *c, const double *w, const double *props, int H, int run,
                             const mxm_em_state *state, double *colsum, double *partial, hipStream_t stream, bool timed,
                             const fin_args *fin = nullptr) {
    const int64_t ldpart = part_ld(H);
    const int ldc = coded_ld(H);
    const coded_shape sh = g_coded_shapes[g_coded_shape];
    const int nch = (ldc / 4 + sh.threads - 1) / sh.threads;
    int cap = num_cu() * sh.wg_per_cu;
    if (cap > MXM_MAX_WG - num_cu()) cap = MXM_MAX_WG - num_cu();             // the dense rest's rows come behind
    if (cap < 1) cap = 1;
    const bool quads = c->qrec != nullptr && c->n_quad_rows > 0 && nch <= QUAD_MAX_NCH;   // (wider: the records alone)
    // beside a quad dictionary: the quad rows' pass (partial rows 0 .. nwg_q) and the pass over the byte-coded rows without
    // quads and the wide rows (rows nwg_q .. nwg) share ONE grid (em_iter_quad_coded_kernel)
    int nwg_q = 0, nwg = 0;
    if (quads) {
        const int slots = cap < 2 * num_cu() ? cap : 2 * num_cu();
        const int nwg_b = quad_left_share(c, slots);
        nwg_q = clamp_grid((c->n_quad_rows + sh.nbuf - 1) / sh.nbuf, slots - nwg_b < 1 ? 1 : slots - nwg_b);
        nwg = nwg_q + nwg_b;
    } else {
        nwg = clamp_grid((c->R + sh.nbuf - 1) / sh.nbuf, cap);
    }
    if (timed && T.ev_start != nullptr) HIP_TRY(hipEventRecord(T.ev_start, stream));
    // {wide rows met, list fault} per workgroup: the kernel leaves them behind the partial rows this path can use
    // (coded_row_pass, CHECK; em_iter_coded_kernel forms the same address)
    int *chk = reinterpret_cast<int *>(partial + (int64_t)MXM_MAX_WG * ldpart);
    int lrc;
    if (quads) {
        // one grid: the leftover pass on its first nwg - nwg_q workgroups beside the quad pass on the others
        // (em_iter_quad_coded_kernel); no leftover rows at all: the quad kernel alone
        if (nwg > nwg_q) lrc = launch_quad_coded(nch, nwg, nwg - nwg_q, stream, c, ldc, w, props, H, partial, ldpart, state, run);
        else lrc = launch_quad(nch, nwg_q, stream, c, w, props, H, partial, ldpart, state, run);
        if (lrc != 0) return lrc;
    } else {
        lrc = launch_coded<256, 4, 2>(nch, nwg, stream, c, ldc, w, props, H, partial, ldpart, state, run);
        if (lrc != 0) return lrc;
    }
    HIP_TRY(hipGetLastError());
    if (timed && T.ev_stop != nullptr) HIP_TRY(hipEventRecord(T.ev_stop, stream));
    int nwg_rest = 0;
    mxm_slots sl;
    for (int i = 0; i < MXM_MAX_BT; ++i) sl.s[i] = run;
    if (c->R_rest > 0) {
        const int rc = stream_linear_tile(c->P_rest, c->ldp_rest, c->w_rest, props, c->R_rest, H, 1, sl, state,
                                          partial + (int64_t)nwg * ldpart, stream, false, &nwg_rest, MXM_MAX_WG - nwg,
                                          0 /* the 256-thread instance: the few dense rows show up under a kernel
                                               name of their own in a trace, apart from the dense matrix's passes */);
        if (rc != 0) return rc;
    }
    // (with a row list the byte pass meets no wide row in its main loop: the lists are checked entry by entry instead)
    return reduce_tile(partial, nwg + nwg_rest, 1, H, colsum, state, sl, fin, stream, wide_check{chk, nwg, quads ? 0ll : (long long)c->n_wide});
}

// ---- QB_BT restarts through one pass over the records beside a quad dictionary (quad_batched_kernels.hpp) ----------
#define QB_BT 3
// Can a tile of QB_BT restarts share one pass over this coded matrix?  It takes a quad dictionary (the kernel's first
// and largest class of rows), a width the quad pass covers, and dense leftover rows (random matrices), if any, that
// the dense kernel takes three restarts at a time too.
static bool coded_batch_ok(const mxm_coded *c, int H) {
    if (T.coded_bt < QB_BT || c == nullptr || c->qrec == nullptr || c->n_quad_rows <= 0) return false;
    if (c->R_rest > 0 && !batch_fits(H, QB_BT)) return false;                // (the dense leftover rows' tile of three)
    const int nch256 = (coded_ld(H) / 4 + QUAD_THREADS - 1) / QUAD_THREADS;
    return nch256 <= QUAD_MAX_NCH && mxm_linear_supported(H);
}
static int em_iter_coded_batched(const mxm_coded *c, const double *w, const double *props, int H, const mxm_slots &tile,
                                 const mxm_em_state *state, double *colsum, double *partial, hipStream_t stream, bool timed,
                                 const fin_args *fin = nullptr) {
    const int64_t ldpart = part_ld(H);
    const int ldc = coded_ld(H);
    const int nch = ((ldc / 4 + QUAD_THREADS - 1) / QUAD_THREADS + 1) / 2;      // code bytes per thread of 512
    int cap = num_cu();
    if (cap > MXM_MAX_WG / QB_BT - num_cu() / QB_BT) cap = MXM_MAX_WG / QB_BT - num_cu() / QB_BT;   // (the dense rest's rows come behind)
    if (cap < 1) cap = 1;
    const int nwg = clamp_grid((c->n_quad_rows + 2) / 3, cap);
    mxm_slots sl = tile;
    for (int i = QB_BT; i < MXM_MAX_BT; ++i) sl.s[i] = sl.s[0];
    if (timed && T.ev_start != nullptr) HIP_TRY(hipEventRecord(T.ev_start, stream));
    const int64_t *byte_rows = c->byte_rows != nullptr ? c->byte_rows : c->quad_rows;      // (an empty list is still a list)
    const int64_t *wide_rows = c->wide_rows != nullptr ? c->wide_rows : c->quad_rows;
    // two launches on the same grid: the quad rows start the partial sums, the leftover rows (byte-coded without quads,
    // wide) are added to them (quad_batched_kernels.hpp: as one kernel the three row loops spill into the quad rows' loop)
    const bool left = c->n_byte_rows > 0 || c->n_wide > 0;
    switch (nch) {
#define QB_ARGS c->rec, c->rec_off, c->ndist, ldc, wide_rows, c->n_wide, byte_rows, c->n_byte_rows, c->qrec, c->qoff, c->nquad, c->quad_rows, c->n_quad_rows, c->R, w, props, H, partial, ldpart, state, sl
#define QB_CASE(n) case n: hipLaunchKernelGGL((em_iter_quad_batched_kernel<QB_BT, n, 1>), dim3(nwg), dim3(QB_THREADS), 0, stream, QB_ARGS); if (left) hipLaunchKernelGGL((em_iter_quad_batched_kernel<QB_BT, n, 6>), dim3(nwg), dim3(QB_THREADS), 0, stream, QB_ARGS); break;
        QB_CASE(1) QB_CASE(2) QB_CASE(3)
#undef QB_CASE
#undef QB_ARGS
        default: return fail(-1, "mxm_em_iter_coded: H=%s%lld outside the batched quad kernel's range", "", H);
    }
    HIP_TRY(hipGetLastError());
    if (timed && T.ev_stop != nullptr) HIP_TRY(hipEventRecord(T.ev_stop, stream));
    int nwg_rest = 0;
    if (c->R_rest > 0) {
        const int rc = stream_linear_tile(c->P_rest, c->ldp_rest, c->w_rest, props, c->R_rest, H, QB_BT, sl, state,
                                          partial + (int64_t)nwg * QB_BT * ldpart, stream, false, &nwg_rest,
                                          (MXM_MAX_WG - nwg * QB_BT) / QB_BT);
        if (rc != 0) return rc;
    }
    int *chk = reinterpret_cast<int *>(partial + (int64_t)MXM_MAX_WG * ldpart);
    return reduce_tile(partial, nwg + nwg_rest, QB_BT, H, colsum, state, sl, fin, stream, wide_check{chk, nwg, 0ll});
}

extern "C" int mxm_em_iter_coded(const mxm_coded *c, const double *w, const double *props, int32_t H, int32_t B,
                                 mxm_em_state *state, double *colsum, void *ws, size_t ws_bytes, void *stream) {
    MXM_ENTER();
    const int rc = coded_check(c, H, "mxm_em_iter_coded");
    if (rc != 0) return rc;
    if (B <= 0 || props == nullptr || colsum == nullptr) return fail(-1, "mxm_em_iter_coded: bad arguments%s", "");
    if (ws == nullptr || ws_bytes < mxm_workspace_bytes(c->R, H, B)) return fail(-1, "mxm_em_iter_coded: workspace too small%s", "");
    // beside a quad dictionary full tiles of QB_BT restarts share a pass (the remainder: one per pass)
    int b = 0;
    if (coded_batch_ok(c, (int)H)) {
        for (; b + QB_BT <= B; b += QB_BT) {
            const int irc = em_iter_coded_batched(c, w, props, (int)H, slots_from(b), state, colsum, (double *)ws, (hipStream_t)stream, b == 0);
            if (irc != 0) return irc;
        }
    }
    for (; b < B; ++b) {
        const int irc = em_iter_coded_one(c, w, props, (int)H, b, state, colsum, (double *)ws, (hipStream_t)stream, b == 0);
        if (irc != 0) return irc;
    }
    return 0;
}

extern "C" int mxm_restart_tile_coded(const mxm_coded *c, int32_t H) {
    MXM_ENTER();
    return coded_batch_ok(c, (int)H) ? QB_BT : 1;
}

extern "C" int mxm_restart_tile(int32_t H) {
    MXM_ENTER();
    if (!mxm_linear_supported(H)) return 1;
    int bt = T.max_bt;
    while (bt > 1 && !batch_fits((int)H, bt)) --bt;
    return bt;
}

extern "C" int mxm_m_finalize(const double *colsum, double *ln_cur, double *ln_new, double *props_cur, int32_t H,
                              int32_t B, double tol, int32_t max_iter, mxm_em_state *state, void *stream) {
    if (H <= 0 || B <= 0 || state == nullptr) return fail(-1, "mxm_m_finalize: bad arguments%s", "");
    if (H > 64 * FIN_MAX_BLOCKS) {                        // any width: one workgroup per restart
        hipLaunchKernelGGL(finalize_any_kernel, dim3(1, B), dim3(FIN_THREADS), 0, (hipStream_t)stream, colsum, ln_cur, ln_new,
                           props_cur, (int)H, tol, (int)max_iter, state, 0, 0, slots_from(0));
        HIP_TRY(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(finalize_kernel, dim3((H + FIN_THREADS - 1) / FIN_THREADS, B), dim3(FIN_THREADS), 0, (hipStream_t)stream,
                       colsum, ln_cur, ln_new, props_cur, (int)H, tol, (int)max_iter, state, 0, 0, slots_from(0));
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- the run_em inner loop (em.py:126-143) as a native driver -----------------------------------
// Iterations are enqueued in chunks; for small matrices the chunk is captured once into a
// hipGraph and replayed (the loop is launch-bound there: 3 kernels of a few microseconds each),
// for large ones plain launches already run ahead of the GPU.  Either way the kernels of a
// finished restart are no-ops, so the state freezes on the iteration the reference stops on.
extern "C" int mxm_set_loop_graph(int32_t mode) {
    return tune_set([mode](mxm_tuning &t) { t.loop_graph = mode < 0 ? -1 : (mode > 0 ? 1 : 0); });
}

// ---- progress hook: called on the host thread inside mxm_em_loop after every state read-back ------
extern "C" int mxm_set_progress_callback(mxm_progress_fn fn, void *user, int32_t every) {
    return tune_set([=](mxm_tuning &t) { t.progress = fn; t.progress_user = user; t.progress_every = every > 0 ? every : 10; });
}

// ---- one-launch loop for cache-resident matrices (fused_kernels.hpp) ---------------------------
extern "C" int mxm_set_loop_fused(int32_t mode, int32_t chunk) {
    return tune_set([=](mxm_tuning &t) {
        t.loop_fused = mode < 0 ? -1 : (mode > 0 ? 1 : 0);
        t.fused_cols = (mode == 2) ? 0 : 1;            // mode 2: one launch, rows split (A/B against the transposed form)
        t.fused_chunk = chunk > 0 ? chunk : 0;
    });
}

// The transposed one-launch loop (fused_cols_kernels.hpp): up to 1536 rows, a 256-CU grid.
static bool fused_cols_eligible(int64_t R, int H, int nwg) {
    if (!T.fused_cols || nwg != 256) return false;          // the Z reduce is laid out for 256 partials per row
    if (R > (int64_t)FCOLS_MAX_RPT * FCOLS_THREADS || (R + nwg - 1) / nwg > 2 * FCOLS_NQ) return false;
    const int cp = (H + nwg - 1) / nwg;
    if (cp > FCOLS_MAX_CP) return false;
    // 24 columns x 3 rows per thread do not fit the register file without scratch (22 x 3 -- Build 17's width
    // on 256 CUs -- do): the widest matrices up to 1024 rows
    return cp <= 22 || R <= 2 * (int64_t)FCOLS_THREADS;
}
// ... and its exchange buffers must fit between the sync block and the snapshot (a caller may hand in less than
// mxm_workspace_bytes asks for today: the rows-split form or the per-iteration kernels then run instead)
static bool fused_cols_fits(int64_t R, int H, int B, int nwg, size_t ws_bytes) {
    return ws_bytes >= fused_sync_bytes() + fused_cols_bytes(R, nwg) + fused_snapshot_bytes(H, B) + 256;
}

// `running`: restarts that still have iterations to do.  Automatic mode takes the one-launch loop for ONE
// restart only: it runs restarts one after another (17 / 26 / 80 us per restart-iteration at 600 / 2400 /
// 10 000 rows), while the per-iteration kernels share each pass between up to four of them (11 / 15 / 27 us
// from three restarts on; profiles/r02/small_runs_restarts.txt).
static bool fused_eligible(const double *P, int64_t ldp, int64_t R, int H, int B, size_t ws_bytes, bool p_is_f32, int running) {
    if (T.loop_fused == 0 || p_is_f32 || P == nullptr || !mxm_linear_supported(H)) return false;
    if ((ldp & 1) || (reinterpret_cast<uintptr_t>(P) & 15)) return false;
    int nwg = num_cu() < MXM_MAX_WG ? num_cu() : MXM_MAX_WG;
    const int ncol2 = (H + 1) / 2;
    if ((ncol2 + nwg - 1) / nwg > 16 * FUSED_MAX_M) return false;           // slice wider than the column reduce covers
    if ((ncol2 + FUSED_THREADS - 1) / FUSED_THREADS > FUSED_MAX_NCH) return false;   // spill-free instances only
    if ((int64_t)nwg * part_ld(H) * 8 >= ((int64_t)1 << 31)) return false;  // one buffer descriptor over the partials
    if (ws_bytes < fused_sync_bytes() + fused_rows_bytes(H, nwg) + fused_snapshot_bytes(H, B) + 256) return false;
    if (T.loop_fused == 1) return true;
    // the transposed form runs a restart-iteration in 12 us at 600 rows whatever the number of restarts; the
    // batched kernels need 18 / 13 / 11 us with 2 / 3 / 4 restarts per pass: up to three restarts stay here
    if (running <= 3 && fused_cols_eligible(R, H, nwg) && fused_cols_fits(R, H, B, nwg, ws_bytes)) return true;
    return running <= 1 && (double)R * (double)H <= T.fused_cells;
}

// The one-launch loop over records (fused_coded_kernels.hpp): any number of rows, restarts one after another (the
// per-iteration coded kernels take one restart per pass too, so nothing is lost), no dense leftover rows.
static int fused_coded_grid(int64_t R) {
    int nwg = num_cu() * 2 < MXM_MAX_WG ? num_cu() * 2 : MXM_MAX_WG;         // two workgroups of 256 per CU
    if (T.fused_coded_wg > 0) {
        if (T.fused_coded_wg < nwg) nwg = T.fused_coded_wg;
    } else {
        // an iteration costs ~0.7 us per row of a workgroup plus ~0.026 us per workgroup (barriers, one partial row
        // each way through the fabric): the minimum of 0.7 R / n + 0.026 n is at n = 5.2 sqrt(R) (measured sweep:
        // profiles/r04/small_runs_coded_grids.txt -- 600 rows: 128 workgroups 18.8 us, 512 30.1; 10^4 rows: level
        // from 384 up), in whole eights (the XCDs)
        const int best = ((int)(5.2 * sqrt((double)R)) + 7) & ~7;
        if (best < nwg) nwg = best;
    }
    if ((int64_t)nwg > R) nwg = (int)R;
    return nwg < 1 ? 1 : nwg;
}
#define QUAD_PER_ITER_MIN_ROWS 300000
#define QUAD_BATCH_MIN_ROWS 15000     // ... with three or more restarts running: tiles of three share a pass, which beats the one-launch
                                      // loop (restarts one after another) from the smallest record plans on: 45 against 55 us per
                                      // restart-iteration at 2*10^4 rows, 172 against 244 at 1.5*10^5 (profiles/r06/multi_restart_routes.txt)
// the floor a binding's "auto" rule must use too: a dictionary of fewer quad rows is never looked at by mxm_em_loop_coded
extern "C" int64_t mxm_quad_loop_min_rows(int32_t B) { return B >= 3 ? QUAD_BATCH_MIN_ROWS : QUAD_PER_ITER_MIN_ROWS; }
static bool fused_coded_eligible(const mxm_coded *c, int H, int B, size_t ws_bytes, int running = 1) {
    if (T.loop_fused == 0 || c == nullptr || c->R_rest > 0 || !mxm_linear_supported(H) || (H & 1)) return false;
    // beside a quad dictionary the per-iteration kernels are the faster loop from a few 10^5 rows (1.35 against 1.45 ms per
    // iteration at 10^6: the quad pass saves 0.13 ms of the row pass, the one-launch loop ~0.04 ms of launches and tail;
    // a one-launch loop with the quad pass inside ran 1.41): the one-launch loop, which reads the records only, keeps
    // the small matrices (mxm_set_loop_fused(1, ...) still forces it)
    if (c->qrec != nullptr && c->n_quad_rows >= QUAD_PER_ITER_MIN_ROWS && T.loop_fused != 1 &&
        (coded_ld(H) / 4 + QUAD_THREADS - 1) / QUAD_THREADS <= QUAD_MAX_NCH)
        return false;
    // ... and with a full tile of restarts still running, from far fewer rows (round 6)
    if (running >= 3 && T.loop_fused != 1 && c->n_quad_rows >= QUAD_BATCH_MIN_ROWS && coded_batch_ok(c, H)) return false;
    const int nwg = fused_coded_grid(c->R);
    const int ncol2 = H / 2;
    if ((ncol2 + nwg - 1) / nwg > 16 * FCODED_MAX_M) return false;           // slice wider than the column reduce covers
    if ((coded_ld(H) / 4 + FCODED_THREADS - 1) / FCODED_THREADS > 8) return false;
    if ((int64_t)nwg * part_ld(H) * 8 >= ((int64_t)1 << 31)) return false;   // one buffer descriptor over the partials
    return ws_bytes >= fused_sync_bytes() + fused_rows_bytes(H, nwg) + fused_snapshot_bytes(H, B) + 256;
}

// Diagnostic (-DFUSED_STAMPS builds): the per-phase clock sums of the last one-launch loop, from the
// sync block at the start of `ws`; all zeros in the shipped library.
extern "C" int mxm_diag_fused_stamps(const void *ws, unsigned long long *out_host) {
    if (ws == nullptr || out_host == nullptr) return fail(-1, "mxm_diag_fused_stamps: bad arguments%s", "");
    const fused_sync *sync = reinterpret_cast<const fused_sync *>(ws);
    HIP_TRY(hipMemcpy(out_host, sync->stamps, sizeof(sync->stamps), hipMemcpyDeviceToHost));
    return 0;
}

// A persistent grid with grid barriers must be co-resident: ask the runtime how many workgroups of the instance fit a
// CU before launching (returns false -> the caller takes the per-iteration kernels instead).
template <typename K>
static bool grid_fits(K kernel, int threads, int nwg) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return (long long)per_cu * num_cu() >= nwg;
}

template <int CP, int RPT>
static bool launch_fused_cols(int nwg, hipStream_t s, const double *P, int64_t ldp, const double *w, int64_t R, int H,
                              int B, double *ln_cur, double *ln_new, double *props_cur, mxm_em_state *state, double tol,
                              int max_iter, int chunk, double *zpart, int64_t ldz, double *cbuf, double *l1part,
                              fused_sync *sync) {
    if (!grid_fits(em_fused_cols_kernel<CP, RPT>, FCOLS_THREADS, nwg)) return false;
    hipLaunchKernelGGL((em_fused_cols_kernel<CP, RPT>), dim3(nwg), dim3(FCOLS_THREADS), 0, s, P, ldp, w, R, H, B, ln_cur,
                       ln_new, props_cur, state, tol, max_iter, chunk, zpart, ldz, cbuf, l1part, sync);
    return true;
}

template <int NCH>
static bool launch_fused(int nwg, hipStream_t s, const double *P, int64_t ldp, const double *w, int64_t R, int H, int B,
                         double *ln_cur, double *ln_new, double *props_cur, mxm_em_state *state, double tol,
                         int max_iter, int chunk, double *partial, int64_t ldpart, double *tbuf, fused_sync *sync) {
    if (!grid_fits(em_fused_loop_kernel<NCH, FUSED_NBUF>, FUSED_THREADS, nwg)) return false;
    hipLaunchKernelGGL((em_fused_loop_kernel<NCH, FUSED_NBUF>), dim3(nwg), dim3(FUSED_THREADS), 0, s, P, ldp, w, R, H, B, ln_cur,
                       ln_new, props_cur, state, tol, max_iter, chunk, partial, ldpart, tbuf, sync);
    return true;
}

template <int NCH>
static bool launch_fused_coded(int nwg, bool resident, hipStream_t s, const mxm_coded *c, int ldc, const double *w, int H, int B,
                               double *ln_cur, double *ln_new, double *props_cur, mxm_em_state *state, double tol,
                               int max_iter, int chunk, double *partial, int64_t ldpart, double *tbuf, fused_sync *sync) {
    if constexpr (NCH * FCODED_THREADS > 2048) {
        return false;
    } else {
        fcoded_args a;
        a.partial = partial; a.tbuf = tbuf; a.ln_cur = ln_cur; a.ln_new = ln_new; a.props_cur = props_cur;
        a.state = state; a.sync = sync; a.ldpart = ldpart; a.tol = tol; a.max_iter = max_iter; a.H = H;
        if (resident) {
            if (!grid_fits(em_fused_coded_kernel<NCH, 4, true>, FCODED_THREADS, nwg)) return false;
            hipLaunchKernelGGL((em_fused_coded_kernel<NCH, 4, true>), dim3(nwg), dim3(FCODED_THREADS), 0, s, c->rec, c->rec_off,
                               c->ndist, ldc, w, c->wide_rows, c->n_wide, c->R, B, chunk, a);
        } else {
            if (!grid_fits(em_fused_coded_kernel<NCH, 4, false>, FCODED_THREADS, nwg)) return false;
            hipLaunchKernelGGL((em_fused_coded_kernel<NCH, 4, false>), dim3(nwg), dim3(FCODED_THREADS), 0, s, c->rec, c->rec_off,
                               c->ndist, ldc, w, c->wide_rows, c->n_wide, c->R, B, chunk, a);
        }
        return true;
    }
}

// The one-launch loop of a narrow matrix (fused_narrow_kernels.hpp): grid and rows per thread, or 0 if it does not apply
// (more than 32 columns, more rows than the grid's registers hold).
static int fused_narrow_grid(int64_t R, int H, int *rpt_out) {
    if (H < 1 || H > 16 || R < 1) return 0;
    const int hmax = H <= 4 ? 4 : (H <= 8 ? 8 : 16);
    int cap = num_cu() < FNARROW_MAX_WG ? num_cu() : FNARROW_MAX_WG;          // one workgroup of 512 per CU
    const int nwg = clamp_grid((R + FNARROW_THREADS - 1) / FNARROW_THREADS, cap);
    const int64_t per = (R + (int64_t)nwg * FNARROW_THREADS - 1) / ((int64_t)nwg * FNARROW_THREADS);
    int rpt = 1;
    while (rpt < per) rpt *= 2;
    if (rpt * hmax > FNARROW_MAX_CELLS || (hmax == 16 && rpt > 1)) return 0;      // the instances that compile without scratch
    *rpt_out = rpt;
    return nwg;
}
static size_t fused_narrow_bytes(int H, int nwg) { return (size_t)2 * H * ((nwg + 7) & ~7) * sizeof(double); }

template <int HMAX, int RPT>
static bool launch_fused_narrow_one(int nwg, hipStream_t s, const double *M, int64_t ldm, const double *w, int64_t R, int H, int B,
                                    double *ln_cur, double *ln_new, double *props_cur, mxm_em_state *state, double tol,
                                    int max_iter, int chunk, double *partial, fused_sync *sync) {
    if constexpr (HMAX * RPT > FNARROW_MAX_CELLS || (HMAX == 16 && RPT > 1)) {
        return false;
    } else {
        if (!grid_fits(em_fused_narrow_kernel<HMAX, RPT>, FNARROW_THREADS, nwg)) return false;
        hipLaunchKernelGGL((em_fused_narrow_kernel<HMAX, RPT>), dim3(nwg), dim3(FNARROW_THREADS), 0, s, M, ldm, w, R, H, B, ln_cur,
                           ln_new, props_cur, state, tol, max_iter, chunk, partial, (nwg + 7) & ~7, sync);
        return true;
    }
}
static bool launch_fused_narrow(int nwg, int rpt, hipStream_t s, const double *M, int64_t ldm, const double *w, int64_t R, int H,
                                int B, double *ln_cur, double *ln_new, double *props_cur, mxm_em_state *state, double tol,
                                int max_iter, int chunk, double *partial, fused_sync *sync) {
#define FN_ARGS nwg, s, M, ldm, w, R, H, B, ln_cur, ln_new, props_cur, state, tol, max_iter, chunk, partial, sync
#define FN_RPT(hm) (rpt == 1 ? launch_fused_narrow_one<hm, 1>(FN_ARGS) : rpt == 2 ? launch_fused_narrow_one<hm, 2>(FN_ARGS) \
                   : rpt == 4 ? launch_fused_narrow_one<hm, 4>(FN_ARGS) : rpt == 8 ? launch_fused_narrow_one<hm, 8>(FN_ARGS) : false)
    if (H <= 4) return FN_RPT(4);
    if (H <= 8) return FN_RPT(8);
    return FN_RPT(16);
#undef FN_RPT
#undef FN_ARGS
}

// The loop of every restart in [0, B) that is not done yet, in launches of at most `chunk` iterations
// per restart (one launch unless the caller wants to look at the state in between).
// Returns 0 when every restart has stopped; MXM_FUSED_GAVE_UP when a launch could not run to its end -- the grid does
// not fit the device, or its bounded grid barrier timed out because not all workgroups became resident (another
// process or stream holding CUs): the loop vectors and states are then back at their values from before that launch
// (snapshot at the tail of the workspace), so the caller can continue with the per-iteration kernels.
#define MXM_FUSED_GAVE_UP 1
static int em_loop_fused(const double *P, int64_t ldp, const double *w, int64_t R, int32_t H, int32_t B,
                         double *props_cur, double *ln_cur, double *ln_new, mxm_em_state *state, double tol,
                         int32_t max_iter, int32_t chunk, void *ws, size_t ws_bytes, hipStream_t s,
                         mxm_em_state *state_host, const mxm_coded *coded = nullptr, const double *M_narrow = nullptr,
                         int64_t ldm_narrow = 0) {
    // The persistent grid needs every workgroup resident, one per CU.  Two such grids in flight on one
    // device (two host threads, two streams) can each hold a part of the CUs and wait for the rest for
    // ever -- the bounded spins would end both after seconds.  Inside one process the launches are
    // therefore serialised here; across processes sharing a GPU nothing can (see the header).
    static std::mutex one_loop_at_a_time;
    std::lock_guard<std::mutex> guard(one_loop_at_a_time);
    int narrow_rpt = 1;
    const int nwg = M_narrow != nullptr ? fused_narrow_grid(R, (int)H, &narrow_rpt)
                    : coded != nullptr  ? fused_coded_grid(coded->R)
                                        : (num_cu() < MXM_MAX_WG ? num_cu() : MXM_MAX_WG);
    const int64_t ldpart = part_ld(H);
    char *base = static_cast<char *>(ws);
    fused_sync *sync = reinterpret_cast<fused_sync *>(base);
    double *tbuf = reinterpret_cast<double *>(base + fused_sync_bytes());
    double *partial = tbuf + 2 * ldpart;
    const int nch = ((H + 1) / 2 + FUSED_THREADS - 1) / FUSED_THREADS;
    const size_t vec = (size_t)B * H * sizeof(double);
    char *snap = base + ((ws_bytes - fused_snapshot_bytes(H, B)) & ~(size_t)255);
    auto snapshot = [&](bool save) -> hipError_t {
        double *vecs[3] = {props_cur, ln_cur, ln_new};
        for (int i = 0; i < 3; ++i) {
            const hipError_t e = save ? hipMemcpyAsync(snap + i * vec, vecs[i], vec, hipMemcpyDeviceToDevice, s)
                                      : hipMemcpyAsync(vecs[i], snap + i * vec, vec, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return e;
        }
        return save ? hipMemcpyAsync(snap + 3 * vec, state, sizeof(mxm_em_state) * B, hipMemcpyDeviceToDevice, s)
                    : hipMemcpyAsync(state, snap + 3 * vec, sizeof(mxm_em_state) * B, hipMemcpyDeviceToDevice, s);
    };
    if (chunk < 1) chunk = 1;
    bool launched_once = false;
    for (;;) {
        HIP_TRY(hipMemcpyAsync(state_host, state, sizeof(mxm_em_state) * B, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        bool all_done = true, gave_up = false;
        for (int b = 0; b < B; ++b) {
            gave_up = gave_up || state_host[b].done < 0;
            all_done = all_done && state_host[b].done != 0;
        }
        if (gave_up) {
            // the grid barrier timed out: undo the launch (its vectors are mid-iteration) and hand back
            HIP_TRY(snapshot(false));
            HIP_TRY(hipMemcpyAsync(state_host, state, sizeof(mxm_em_state) * B, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            fail(-3, "mxm_em_loop: the one-launch loop's grid barrier timed out: not all %s%lld workgroups became resident "
                     "(is another process or stream using the GPU?)", "", nwg);
            return MXM_FUSED_GAVE_UP;
        }
        if (launched_once && T.progress != nullptr) T.progress(state_host, B, T.progress_user);
        if (all_done) return 0;
        HIP_TRY(snapshot(true));
        HIP_TRY(hipMemsetAsync(sync, 0, fused_sync_bytes(), s));       // every polled word, before EVERY launch
        if (T.fused_force_abort)                                       // test hook: as if a workgroup had given up
            HIP_TRY(hipMemsetAsync(&sync->abort_[0], 1, sizeof(unsigned), s));
        bool fits = true;
        if (M_narrow != nullptr) {
            // narrow matrix: [sync][partial sums 2 x H x grid]; the rows live in registers for the whole launch
            fits = launch_fused_narrow(nwg, narrow_rpt, s, M_narrow, ldm_narrow, w, R, (int)H, (int)B, ln_cur, ln_new, props_cur,
                                       state, tol, (int)max_iter, (int)chunk, reinterpret_cast<double *>(base + fused_sync_bytes()), sync);
        } else if (coded != nullptr) {
            // records: [sync][T (1 row)][partial rows]; the metadata of a workgroup's rows stays in LDS for the whole
            // launch when they fit its blocks (256 rows and 256 wide rows per workgroup)
            const int ldc = coded_ld(H);
            const int cnch = (ldc / 4 + FCODED_THREADS - 1) / FCODED_THREADS;
            const bool resident = (coded->R + nwg - 1) / nwg <= FCODED_THREADS && (coded->n_wide + nwg - 1) / nwg <= FCODED_THREADS;
            switch (cnch) {
#define FC_CASE(n) case n: fits = launch_fused_coded<n>(nwg, resident, s, coded, ldc, w, (int)H, (int)B, ln_cur, ln_new, props_cur, state, tol, (int)max_iter, (int)chunk, partial, ldpart, tbuf, sync); break;
                FC_CASE(1) FC_CASE(2) FC_CASE(3) FC_CASE(4) FC_CASE(5) FC_CASE(6) FC_CASE(7) FC_CASE(8)
#undef FC_CASE
                default: return fail(-1, "mxm_em_loop_coded: H=%s%lld outside the one-launch loop's range", "", H);
            }
        } else if (fused_cols_eligible(R, (int)H, nwg) && fused_cols_fits(R, (int)H, (int)B, nwg, ws_bytes)) {
            // smallest matrices: columns split over the workgroups, the matrix in registers (workspace:
            // [sync][z partials nwg x ldz][c ldz][l1 partials 2 x nwg]; checked against ws_bytes above)
            const int64_t ldz = (R + 1) & ~(int64_t)1;
            double *zpart = reinterpret_cast<double *>(base + fused_sync_bytes());
            double *cbuf = zpart + (int64_t)nwg * ldz;
            double *l1part = cbuf + ldz;
            const int cp = ((int)H + nwg - 1) / nwg;
            const int rpt = (int)((R + FCOLS_THREADS - 1) / FCOLS_THREADS);
#define FC_ARGS nwg, s, P, ldp, w, R, (int)H, (int)B, ln_cur, ln_new, props_cur, state, tol, (int)max_iter, (int)chunk, zpart, ldz, cbuf, l1part, sync
            if (cp <= 12) {
                if (rpt <= 1) fits = launch_fused_cols<12, 1>(FC_ARGS);
                else if (rpt == 2) fits = launch_fused_cols<12, 2>(FC_ARGS);
                else fits = launch_fused_cols<12, 3>(FC_ARGS);
            } else if (cp <= 22) {
                if (rpt <= 1) fits = launch_fused_cols<22, 1>(FC_ARGS);
                else if (rpt == 2) fits = launch_fused_cols<22, 2>(FC_ARGS);
                else fits = launch_fused_cols<22, 3>(FC_ARGS);
            } else {
                if (rpt <= 1) fits = launch_fused_cols<24, 1>(FC_ARGS);
                else fits = launch_fused_cols<24, 2>(FC_ARGS);
            }
#undef FC_ARGS
        } else {
            switch (nch) {
#define FU_CASE(n) case n: fits = launch_fused<n>(nwg, s, P, ldp, w, R, (int)H, (int)B, ln_cur, ln_new, props_cur, state, tol, (int)max_iter, (int)chunk, partial, ldpart, tbuf, sync); break;
                FU_CASE(1) FU_CASE(2) FU_CASE(3)
#if FUSED_MAX_NCH > 3
                FU_CASE(4) FU_CASE(5) FU_CASE(6)
#endif
#undef FU_CASE
                default: return fail(-1, "mxm_em_loop: H=%s%lld outside the one-launch loop's range", "", H);
            }
        }
        if (!fits) {
            fail(-3, "mxm_em_loop: the one-launch loop's %s%lld workgroups cannot be co-resident on this device", "", nwg);
            return MXM_FUSED_GAVE_UP;                                  // nothing was launched, nothing to undo
        }
        HIP_TRY(hipGetLastError());
        launched_once = true;
    }
}

// One iteration of the restarts named by `tile` (nb <= the restart tile): E+M pass, column reduce,
// finalize.  Linear fp64 matrices share the pass; the other paths (fp32 storage, log-space kernels)
// run one restart per pass through pointers offset to that restart.
static int enqueue_tile_iteration(const double *M, int64_t ldm, const double *P, int64_t ldp, const double *w,
                                  int64_t R, int32_t H, const mxm_slots &tile, int nb, double *props_cur,
                                  double *ln_cur, double *ln_new, double *colsum, mxm_em_state *state, double tol,
                                  int32_t max_iter, void *ws, size_t ws_bytes, hipStream_t s, bool p_is_f32,
                                  bool timed, const mxm_coded *coded) {
    const bool linear = !p_is_f32 && P != nullptr && mxm_linear_supported(H);
    // records and the linear fp64 matrix: the finalize rides on the column reduce's launch (one ticket per restart)
    fin_args fin = {ln_cur, ln_new, props_cur, state, tol, (int)max_iter};
    if (coded != nullptr) {
        if (nb == QB_BT && coded_batch_ok(coded, (int)H))
            return em_iter_coded_batched(coded, w, props_cur, (int)H, tile, state, colsum, (double *)ws, s, timed, &fin);
        for (int i = 0; i < nb; ++i) {
            const int rc = em_iter_coded_one(coded, w, props_cur, (int)H, tile.s[i], state, colsum, (double *)ws, s, timed && i == 0, &fin);
            if (rc != 0) return rc;
        }
        return 0;
    } else if (linear) {
        return em_iter_linear_tile(P, ldp, w, props_cur, R, (int)H, nb, tile, state, colsum, (double *)ws, s, timed, &fin);
    } else {
        for (int i = 0; i < nb; ++i) {
            const int64_t off = (int64_t)tile.s[i] * H;
            const int rc = p_is_f32 ? mxm_em_iter_f32(reinterpret_cast<const float *>(P), ldp, w, props_cur + off, R, H, 1,
                                                      state + tile.s[i], colsum + off, ws, ws_bytes, s)
                                    : mxm_em_iter(M, ldm, P, ldp, w, props_cur + off, ln_cur + off, R, H, 1,
                                                  state + tile.s[i], colsum + off, ws, ws_bytes, s);
            if (rc != 0) return rc;
        }
    }
    if (H > 64 * FIN_MAX_BLOCKS)
        hipLaunchKernelGGL(finalize_any_kernel, dim3(1, nb), dim3(FIN_THREADS), 0, s, colsum, ln_cur, ln_new, props_cur, (int)H, tol,
                           (int)max_iter, state, 0, 1, tile);
    else
        hipLaunchKernelGGL(finalize_kernel, dim3((H + FIN_THREADS - 1) / FIN_THREADS, nb), dim3(FIN_THREADS), 0, s, colsum, ln_cur,
                           ln_new, props_cur, (int)H, tol, (int)max_iter, state, 0, 1, tile);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int em_loop_impl(const double *M, int64_t ldm, const double *P, int64_t ldp, const double *w,
                        int64_t R, int32_t H, int32_t B, double *props_cur, double *ln_cur, double *ln_new,
                        double *colsum, mxm_em_state *state, double tol, int32_t max_iter,
                        int32_t check_every, void *ws, size_t ws_bytes, void *stream,
                        mxm_em_state *state_host, bool p_is_f32, const mxm_coded *coded = nullptr) {
    if (state_host == nullptr || state == nullptr) return fail(-1, "mxm_em_loop: state pointers required%s", "");
    if (R <= 0 || H <= 0 || B <= 0) return fail(-1, "mxm_em_loop: bad shape R=%s%lld H=%lld", "", R, H);
    if (ws == nullptr || ws_bytes < mxm_workspace_bytes(R, H, B)) return fail(-1, "mxm_em_loop: workspace too small%s", "");
    if (check_every < 1) check_every = 1;
    if (T.progress != nullptr && check_every > T.progress_every) check_every = T.progress_every;
    hipStream_t caller = (hipStream_t)stream;
    (void)num_cu();                                    // device query outside any capture
    HIP_TRY(hipMemcpyAsync(state_host, state, sizeof(mxm_em_state) * B, hipMemcpyDeviceToHost, caller));
    HIP_TRY(hipStreamSynchronize(caller));
    int running = 0;
    for (int b = 0; b < B; ++b) running += (state_host[b].done == 0) ? 1 : 0;
    if (coded != nullptr && running > 0 && max_iter > 0 && fused_coded_eligible(coded, (int)H, (int)B, ws_bytes, running)) {
        // records: the whole loop in persistent launches on the caller's stream, restarts one after another.  A launch
        // is kept to about half a second (a 10^7-row matrix takes 15 ms per iteration): `chunk` iterations each.
        int chunk = T.fused_chunk > 0 ? T.fused_chunk : max_iter;
        if (T.progress != nullptr && chunk > T.progress_every) chunk = T.progress_every;
        const double est_us = (double)R * ((double)coded_ld(H) + 400.0) / 4.0e6 + 20.0;
        const int cap = (int)std::max(8.0, 5.0e5 / est_us);
        if (chunk > cap) chunk = cap;
        const int frc = em_loop_fused(nullptr, 0, w, R, H, B, props_cur, ln_cur, ln_new, state, tol, max_iter, chunk, ws, ws_bytes,
                                      caller, state_host, coded);
        if (frc != MXM_FUSED_GAVE_UP) return frc;
        if (T.loop_fused == 1) return -3;
    } else if (int nrpt = 0; !p_is_f32 && coded == nullptr && M != nullptr && T.loop_fused != 0 && running > 0 && max_iter > 0 &&
                            !(P != nullptr && mxm_linear_supported(H)) && fused_narrow_grid(R, (int)H, &nrpt) > 0 &&
                            ws_bytes >= fused_sync_bytes() + fused_narrow_bytes((int)H, FNARROW_MAX_WG) + fused_snapshot_bytes(H, B) + 256) {
        // narrow matrix (the refinement EM on the contributors' columns): the whole loop in one persistent launch,
        // restarts one after another, the rows in registers
        int chunk = T.fused_chunk > 0 ? T.fused_chunk : max_iter;
        if (T.progress != nullptr && chunk > T.progress_every) chunk = T.progress_every;
        const int frc = em_loop_fused(nullptr, 0, w, R, H, B, props_cur, ln_cur, ln_new, state, tol, max_iter, chunk, ws, ws_bytes,
                                      caller, state_host, nullptr, M, ldm);
        if (frc != MXM_FUSED_GAVE_UP) return frc;
        if (T.loop_fused == 1) return -3;
    } else if (fused_eligible(P, ldp, R, (int)H, (int)B, ws_bytes, p_is_f32, running)) {
        // cache-resident matrix: the whole loop in one persistent launch on the caller's stream
        // (the host only waits for it; nothing is decided between iterations)
        int chunk = T.fused_chunk > 0 ? T.fused_chunk : max_iter;
        if (T.progress != nullptr && chunk > T.progress_every) chunk = T.progress_every;     // someone is watching
        const int frc = em_loop_fused(P, ldp, w, R, H, B, props_cur, ln_cur, ln_new, state, tol, max_iter, chunk, ws, ws_bytes,
                                      caller, state_host);
        // Gave up (grid not co-resident / starved): only mxm_set_loop_fused(1, ...) -- "whenever the shape allows" --
        // makes that an error; by default the same call goes on through the per-iteration kernels from the restored
        // state (another summation order: rounding-level differences, same stopping rule).
        if (frc != MXM_FUSED_GAVE_UP) return frc;
        if (T.loop_fused == 1) return -3;
    }
    const bool want_graph = T.loop_graph == 1 ||
                            (T.loop_graph == -1 && (double)R * (double)H * (double)B < 6.4e7);
    int window = T.max_bt;                             // the largest restart tile that fits (mxm_em_iter)
    if (p_is_f32 || P == nullptr || !mxm_linear_supported(H)) window = 1;
    while (window > 1 && !batch_fits((int)H, window)) --window;
    if (coded != nullptr && coded_batch_ok(coded, (int)H)) window = QB_BT;     // records beside a quad dictionary: tiles of three

    // the loop runs on a private stream (the caller's may be the legacy default stream, which
    // cannot be captured); it is ordered after / before the caller's stream with events
    hipStream_t s = nullptr;
    hipEvent_t ev = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    int rc = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<int> graph_key;
    // Restarts stop on different iterations, and a pass over the matrix costs nearly the same for one
    // restart as for a full tile.  Schedules (mxm_set_compact_restarts):
    //   0  every restart in every iteration, static tiles over all B (a tile idles once all its members stopped)
    //   1  tiles over the restarts still running, spread evenly (10 -> 4 + 3 + 3)
    //   2  ONE full tile per iteration, dealt round-robin over the running restarts chunk by chunk: every
    //      pass carries a full tile until fewer than a tile's worth are left, and all restarts advance at
    //      the same rate, so they also finish together (no half-empty tail generation).
    // A tile names its restarts by index (mxm_slots, by value): nothing is moved in memory.  Each restart
    // counts its own iterations (finalize_kernel), so when it is scheduled changes nothing in its result.
    std::vector<int> order;                            // unfinished restarts, round-robin order
#define LOOP_TRY(expr)                                                                        \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) { rc = fail(-2, "HIP error: %s (line %lld)", hipGetErrorString(e_), __LINE__); goto done; } \
    } while (0)
    LOOP_TRY(hipEventRecord(ev, caller));
    LOOP_TRY(hipStreamWaitEvent(s, ev, 0));
    LOOP_TRY(hipMemcpyAsync(state_host, state, sizeof(mxm_em_state) * B, hipMemcpyDeviceToHost, s));
    LOOP_TRY(hipStreamSynchronize(s));
    for (int b = 0; b < B; ++b)
        if (state_host[b].done == 0) order.push_back(b);
    if (max_iter <= 0) order.clear();
    while (!order.empty()) {
        // the tiles of this chunk: flat list of restart indices + tile sizes
        std::vector<int> members, sizes;
        // (with one restart per pass every schedule costs the same: no rotation, the tile list then only
        // changes when a restart stops and a captured graph stays valid in between)
        const bool rotating = T.compact_restarts == 2 && window > 1 && (int)order.size() > window;
        if (rotating) {
            members.assign(order.begin(), order.begin() + window);
            sizes.push_back(window);
            std::rotate(order.begin(), order.begin() + window, order.end());       // they go to the back of the queue
        } else {
            std::vector<int> all;
            if (T.compact_restarts == 0) { for (int b = 0; b < B; ++b) all.push_back(b); }
            else all = order;
            for (size_t at = 0; at < all.size();) {
                // the fewest passes that cover what is left, restarts spread evenly over them
                const int left = (int)(all.size() - at);
                const int passes = (left + window - 1) / window;
                // (records: only a FULL tile shares a pass -- 4 left are 3 + 1, not 2 + 2)
                const int nb = coded != nullptr ? (left < window ? left : window) : (left + passes - 1) / passes;
                members.insert(members.end(), all.begin() + at, all.begin() + at + nb);
                sizes.push_back(nb);
                at += nb;
            }
        }
        // finalize stops each restart at its own max_iter and kernels of a stopped restart are no-ops -- but a pass that
        // several restarts share costs the same with a stopped member in it: a chunk ends where its first member reaches
        // max_iter (what converges on the way cannot be known beforehand)
        int64_t n = check_every;
        for (int b : members) {
            const int64_t left = (int64_t)max_iter - (int64_t)state_host[b].iters;
            if (left > 0 && left < n) n = left;
        }
        auto enqueue_chunk = [&]() -> int {
            for (int64_t it = 0; it < n; ++it) {
                size_t at = 0;
                for (size_t ti = 0; ti < sizes.size(); ++ti) {
                    mxm_slots tile;
                    for (int i = 0; i < MXM_MAX_BT; ++i) tile.s[i] = members[at + (i < sizes[ti] ? i : 0)];
                    const int erc = enqueue_tile_iteration(M, ldm, P, ldp, w, R, H, tile, sizes[ti], props_cur, ln_cur,
                                                           ln_new, colsum, state, tol, max_iter, ws, ws_bytes, s,
                                                           p_is_f32, ti == 0, coded);
                    if (erc != 0) return erc;
                    at += sizes[ti];
                }
            }
            return 0;
        };
        bool launched = false;
        if (want_graph && !rotating) {                 // a rotating tile list would be re-captured every chunk
            std::vector<int> key(members);
            key.insert(key.end(), sizes.begin(), sizes.end());
            key.push_back((int)n);
            if (exec == nullptr || key != graph_key) {
                if (exec != nullptr) { (void)hipGraphExecDestroy(exec); exec = nullptr; }
                if (graph != nullptr) { (void)hipGraphDestroy(graph); graph = nullptr; }
                if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                    const int crc = enqueue_chunk();
                    const hipError_t ee = hipStreamEndCapture(s, &graph);
                    if (crc == 0 && ee == hipSuccess && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                        graph_key = key;
                    } else {
                        exec = nullptr;                // capture unavailable: plain launches below
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
            rc = enqueue_chunk();
            if (rc != 0) goto done;
        }
        LOOP_TRY(hipMemcpyAsync(state_host, state, sizeof(mxm_em_state) * B, hipMemcpyDeviceToHost, s));
        LOOP_TRY(hipStreamSynchronize(s));
        for (int b = 0; b < B; ++b)
            if (state_host[b].error != 0) {
                rc = fail(-1, "mxm_em_loop: the records' wide_rows list does not match the rows with more than 256 values%s", "");
                goto done;
            }
        if (T.progress != nullptr) T.progress(state_host, B, T.progress_user);
        std::vector<int> still;
        for (int b : order)
            if (state_host[b].done == 0) still.push_back(b);
        order.swap(still);
    }
    LOOP_TRY(hipGetLastError());
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
                           int64_t R, int32_t H, int32_t B, double *props_cur, double *ln_cur, double *ln_new,
                           double *colsum, mxm_em_state *state, double tol, int32_t max_iter,
                           int32_t check_every, void *ws, size_t ws_bytes, void *stream,
                           mxm_em_state *state_host) {
    MXM_ENTER();
    return em_loop_impl(M, ldm, P, ldp, w, R, H, B, props_cur, ln_cur, ln_new, colsum, state, tol, max_iter,
                        check_every, ws, ws_bytes, stream, state_host, false);
}

extern "C" int mxm_em_loop_coded(const mxm_coded *c, const double *w, int32_t H, int32_t B, double *props_cur,
                                 double *ln_cur, double *ln_new, double *colsum, mxm_em_state *state, double tol,
                                 int32_t max_iter, int32_t check_every, void *ws, size_t ws_bytes, void *stream,
                                 mxm_em_state *state_host) {
    MXM_ENTER();
    const int rc = coded_check(c, H, "mxm_em_loop_coded");
    if (rc != 0) return rc;
    if (ws == nullptr || ws_bytes < mxm_workspace_bytes(c->R, H, B)) return fail(-1, "mxm_em_loop_coded: workspace too small%s", "");
    {
        // The loop skips the rows with 16-bit codes in its main pass and takes them from `wide_rows`: the list must be
        // exactly those rows.  This call blocks anyway, so it is checked once, here (the one-launch loop has no room for
        // the in-kernel check the per-iteration kernel carries).
        unsigned long long *chk = static_cast<unsigned long long *>(ws), host[2] = {0, 0};
        hipStream_t s = (hipStream_t)stream;
        HIP_TRY(hipMemsetAsync(chk, 0, 2 * sizeof(unsigned long long), s));
        hipLaunchKernelGGL(coded_validate_kernel, dim3(clamp_grid((c->R + 255) / 256, num_cu() * 4)), dim3(256), 0, s, c->ndist, c->R,
                           c->wide_rows, c->n_wide, chk);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(host, chk, sizeof(host), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (c->qrec != nullptr && host[1] == 0 && (long long)host[0] == (long long)c->n_wide) {   // ... and the quad dictionary's lists
            unsigned long long qhost[2] = {0, 0};
            HIP_TRY(hipMemsetAsync(chk, 0, 2 * sizeof(unsigned long long), s));
            hipLaunchKernelGGL(quad_validate_kernel, dim3(clamp_grid((c->R + 255) / 256, num_cu() * 4)), dim3(256), 0, s, c->ndist,
                               c->nquad, c->R, c->quad_rows, c->n_quad_rows, c->byte_rows, c->n_byte_rows, chk);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(qhost, chk, sizeof(qhost), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            if (qhost[1] != 0)
                return fail(-1, "mxm_em_loop_coded: quad_rows / byte_rows must list the rows with / without quads, ascending%s", "");
        }
        if (host[1] != 0 || (long long)host[0] != (long long)c->n_wide)
            return fail(-1, "mxm_em_loop_coded: wide_rows must list exactly the rows with more than 256 values, ascending "
                            "%s(%lld such rows, n_wide = %lld)", "", (long long)host[0], (long long)c->n_wide);
    }
    return em_loop_impl(nullptr, 0, nullptr, 0, w, c->R, H, B, props_cur, ln_cur, ln_new, colsum, state, tol, max_iter,
                        check_every, ws, ws_bytes, stream, state_host, false, c);
}

extern "C" int mxm_em_loop_f32(const float *P, int64_t ldp, const double *w, int64_t R, int32_t H, int32_t B,
                               double *props_cur, double *ln_cur, double *ln_new, double *colsum,
                               mxm_em_state *state, double tol, int32_t max_iter, int32_t check_every, void *ws,
                               size_t ws_bytes, void *stream, mxm_em_state *state_host) {
    MXM_ENTER();
    return em_loop_impl(nullptr, 0, reinterpret_cast<const double *>(P), ldp, w, R, H, B, props_cur, ln_cur, ln_new,
                        colsum, state, tol, max_iter, check_every, ws, ws_bytes, stream, state_host, true);
}

template <int NCH>
static void launch_estep_wide(const double *M, int64_t ldm, const double *w, const double *lnp, int64_t R, int H,
                              int grid, double *out, int64_t ldo, int mode, double *partial,
                              int64_t ldpart, hipStream_t s) {
    if (partial != nullptr)
        hipLaunchKernelGGL((estep_wide_kernel<NCH, true>), dim3(grid), dim3(256), 0, s, M, ldm, w, lnp, R, H,
                           out, ldo, mode, partial, ldpart);
    else
        hipLaunchKernelGGL((estep_wide_kernel<NCH, false>), dim3(grid), dim3(256), 0, s, M, ldm, w, lnp, R, H,
                           out, ldo, mode, partial, ldpart);
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
    const int wide_cap = num_cu() * 2 < MXM_MAX_WG ? num_cu() * 2 : MXM_MAX_WG;
    if (aligned && mxm_linear_supported(H) && nch <= (colsum != nullptr ? 10 : 13)) {
        // wide rows: one read + one write per cell, rows held in registers
        nwg = clamp_grid((R + 1) / 2, wide_cap);            // rows are dealt round-robin (row_deal)
        switch (nch) {
#define EW_CASE(n) case n: launch_estep_wide<n>(M, ldm, w, ln_props, R, (int)H, nwg, out, ldo, (int)mode, partial, ldpart, s); break;
            EW_CASE(1) EW_CASE(2) EW_CASE(3) EW_CASE(4) EW_CASE(5) EW_CASE(6) EW_CASE(7) EW_CASE(8)
            EW_CASE(9) EW_CASE(10) EW_CASE(11) EW_CASE(12) EW_CASE(13) EW_CASE(14) EW_CASE(15) EW_CASE(16)
#undef EW_CASE
            default: return fail(-1, "mxm_em_step: H=%s%lld outside the wide kernel's range", "", H);
        }
    } else if (H <= MXM_NARROW_MAX_H) {
        const int rc = launch_narrow<false>(M, ldm, w, ln_props, R, (int)H, out, ldo, (int)mode, partial, ldpart,
                                            (const mxm_em_state *)nullptr, s, &nwg);
        if (rc != 0) return rc;
    } else {
        const size_t lds = 2 * (size_t)H * sizeof(double);
        nwg = clamp_grid(R, num_cu() * 2 < MXM_MAX_WG ? num_cu() * 2 : MXM_MAX_WG);
        if (lds > 150 * 1024) {                            // any width: the vectors in global memory
            hipLaunchKernelGGL((estep_global_kernel<false>), dim3(nwg), dim3(ROW_THREADS), 0, s, M, ldm, w, ln_props, R, (int)H, out,
                               ldo, (int)mode, partial, ldpart, (const mxm_em_state *)nullptr, (const int64_t *)nullptr);
        } else {
            if (lds > 60 * 1024 && raise_dynamic_lds(reinterpret_cast<const void *>(&estep_log_kernel<false>), lds, "estep_log_kernel") != hipSuccess)
                return -2;
            hipLaunchKernelGGL((estep_log_kernel<false>), dim3(nwg), dim3(ROW_THREADS), lds, s, M, ldm, w, ln_props, R,
                               (int)H, out, ldo, (int)mode, partial, ldpart, (const mxm_em_state *)nullptr, (const int64_t *)nullptr);
        }
    }
    HIP_TRY(hipGetLastError());
    if (colsum != nullptr) {
        hipLaunchKernelGGL(colreduce_kernel, dim3((H + 63) / 64, 1), dim3(COLRED_THREADS), 0, s, (const double *)ws, ldpart, nwg,
                           1, (int)H, (const double *)nullptr, colsum, (mxm_em_state *)nullptr, slots_from(0), wide_check{nullptr, 0, 0});
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
    if (blocked == 3) {
        // em_iter_coded_kernel's pattern: records of 5408 code bytes + 27 doubles (5624 bytes), two workgroups per CU
        const int code_bytes = 5408, tbl = 27, rec_bytes = code_bytes + tbl * 8;
        const int64_t recs = (int64_t)bytes / rec_bytes;
        if (recs < 1) return fail(-1, "mxm_diag_stream_read: buffer smaller than one record%s", "");
        hipLaunchKernelGGL(diag_stream_records_kernel<6>, dim3(clamp_grid(recs, num_cu() * wg_per_cu)), dim3(256), 0,
                           (hipStream_t)stream, (const uint8_t *)src, recs, rec_bytes, code_bytes, tbl, (unsigned int *)sink);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (blocked == 2) {
        // the EM kernel's pattern: rows of 5408 doubles (2704 x 16 B) dealt over one workgroup per CU
        const int row16 = 2704;
        const int64_t rows = (int64_t)(bytes / 16) / row16;
        if (rows < 1) return fail(-1, "mxm_diag_stream_read: buffer smaller than one row%s", "");
        hipLaunchKernelGGL(diag_stream_dealt_kernel<6>, dim3(clamp_grid(rows, num_cu() * wg_per_cu)), dim3(512), 0,
                           (hipStream_t)stream, (const double *)src, rows, row16, (unsigned int *)sink);
    } else if (blocked)
        hipLaunchKernelGGL(diag_stream_read_kernel<true>, dim3(num_cu() * wg_per_cu), dim3(256), 0, (hipStream_t)stream,
                           (const uint4 *)src, (int64_t)(bytes / 16), (unsigned int *)sink);
    else
        hipLaunchKernelGGL(diag_stream_read_kernel<false>, dim3(num_cu() * wg_per_cu), dim3(256), 0, (hipStream_t)stream,
                           (const uint4 *)src, (int64_t)(bytes / 16), (unsigned int *)sink);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_diag_stream_coded(const mxm_coded *c, int32_t H, int32_t wg_per_cu, void *sink, void *stream) {
    const int rc = coded_check(c, H, "mxm_diag_stream_coded");
    if (rc != 0) return rc;
    if (sink == nullptr || wg_per_cu < 1) return fail(-1, "mxm_diag_stream_coded: bad arguments%s", "");
    const int ldc = coded_ld(H);
    if ((ldc / 4 + 255) / 256 != 6) return fail(-1, "mxm_diag_stream_coded: built for H in (5120, 6144]%s", "");
    hipLaunchKernelGGL(diag_stream_coded_kernel<6>, dim3(clamp_grid(c->R, num_cu() * wg_per_cu)), dim3(256), 0, (hipStream_t)stream,
                       c->rec, c->rec_off, c->ndist, ldc, c->R, (unsigned int *)sink);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_diag_stream_quads(const mxm_coded *c, int32_t H, int32_t wg_per_cu, void *sink, void *stream) {
    const int rc = coded_check(c, H, "mxm_diag_stream_quads");
    if (rc != 0) return rc;
    if (sink == nullptr || wg_per_cu < 1 || c->qrec == nullptr || c->n_quad_rows <= 0)
        return fail(-1, "mxm_diag_stream_quads: a coded matrix with a quad dictionary required%s", "");
    hipLaunchKernelGGL(diag_stream_quads_kernel, dim3(clamp_grid(c->n_quad_rows, num_cu() * wg_per_cu)), dim3(QUAD_THREADS), 0,
                       (hipStream_t)stream, c->qrec, c->qoff, c->nquad, c->quad_rows, c->n_quad_rows, c->R, (unsigned int *)sink);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_row_argmax_votes(const double *X, int64_t ldx, const double *w, int64_t R, int32_t H,
                                    int32_t *best, double *votes, void *ws, size_t ws_bytes, void *stream) {
    if (R <= 0 || H <= 0 || ldx < H) return fail(-1, "mxm_row_argmax_votes: bad shape%s", "");
    if (votes != nullptr && (ws == nullptr || ws_bytes < mxm_workspace_bytes(R, H, 1)))
        return fail(-1, "mxm_row_argmax_votes: workspace too small%s", "");
    // votes: every workgroup fills its own row of the scratch (one thread adds in row order), the
    // rows are summed in fixed order -- no float atomics, so fractional weights reproduce bit for bit
    double *part = votes != nullptr ? (double *)ws : (double *)nullptr;
    const int64_t ldpart = part_ld(H);
    int nwg;
    if (wide_rows_ok(X, ldx, H)) {
        const int nch = (H / 2 + 255) / 256;
        const int cap = num_cu() * 2 < MXM_MAX_WG ? num_cu() * 2 : MXM_MAX_WG;
        nwg = clamp_grid((R + 1) / 2, cap);
        const size_t lds = votes != nullptr ? (size_t)H * sizeof(double) : 0;
        switch (nch) {
#define AW_CASE(n) case n:                                                                                   \
        if (lds > 60 * 1024 && /* with the static exchange buffers this passes the 64 KiB default */          \
            raise_dynamic_lds(reinterpret_cast<const void *>(&row_argmax_wide_kernel<n>), lds,                \
                              "row_argmax_wide_kernel") != hipSuccess)                                        \
            return -2;                                                                                        \
        hipLaunchKernelGGL((row_argmax_wide_kernel<n>), dim3(nwg), dim3(256), lds, (hipStream_t)stream, X, ldx, \
                           w, R, (int)H, best, part, ldpart);                                                  \
        break;
            AW_CASE(1) AW_CASE(2) AW_CASE(3) AW_CASE(4) AW_CASE(5) AW_CASE(6) AW_CASE(7) AW_CASE(8)
            AW_CASE(9) AW_CASE(10) AW_CASE(11) AW_CASE(12) AW_CASE(13) AW_CASE(14) AW_CASE(15) AW_CASE(16)
#undef AW_CASE
            default: return fail(-1, "mxm_row_argmax_votes: H=%s%lld outside the wide kernel's range", "", H);
        }
    } else {
        nwg = clamp_grid(R, num_cu() * 4 < MXM_MAX_WG ? num_cu() * 4 : MXM_MAX_WG);
        hipLaunchKernelGGL(row_argmax_votes_kernel, dim3(nwg), dim3(ROW_THREADS), 0, (hipStream_t)stream, X, ldx, w, R,
                           (int)H, best, part, ldpart);
    }
    HIP_TRY(hipGetLastError());
    if (votes != nullptr) {
        hipLaunchKernelGGL(colreduce_kernel, dim3((H + 63) / 64, 1), dim3(COLRED_THREADS), 0, (hipStream_t)stream,
                           (const double *)part, ldpart, nwg, 1, (int)H, (const double *)nullptr, votes,
                           (mxm_em_state *)nullptr, slots_from(0), wide_check{nullptr, 0, 0});
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

extern "C" int mxm_first_seen(const int32_t *best, int64_t R, int32_t H, int64_t *first, void *stream) {
    if (best == nullptr || first == nullptr || R < 0 || H <= 0) return fail(-1, "mxm_first_seen: bad arguments%s", "");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(fill_u64_kernel, dim3(clamp_grid((H + 255) / 256, 64)), dim3(256), 0, s,
                       reinterpret_cast<unsigned long long *>(first), (int64_t)H, (unsigned long long)R);
    if (R > 0) {
        // ranges of at least 1024 rows and below 2^32 (32-bit offsets in LDS)
        int grid = clamp_grid((R + 1023) / 1024, num_cu() * 4);
        while ((R + grid - 1) / grid >= ((int64_t)1 << 32)) grid *= 2;
        for (int h0 = 0; h0 < H; h0 += FSEEN_MAX_H)          // (one launch up to 8192 haplogroups)
            hipLaunchKernelGGL(first_seen_kernel, dim3(grid), dim3(256), 0, s, best, R, h0,
                               (int)(H - h0 < FSEEN_MAX_H ? H - h0 : FSEEN_MAX_H), reinterpret_cast<unsigned long long *>(first));
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_gather_columns(const double *M, int64_t ldm, int64_t R, int32_t H, const int32_t *cols, int32_t nC,
                                  double *out, int64_t ldo, void *stream) {
    if (R < 0 || H <= 0 || nC <= 0 || ldm < H || ldo < nC) return fail(-1, "mxm_gather_columns: bad shape%s", "");
    if (R == 0) return 0;
    const int64_t want = (R * (int64_t)nC + 255) / 256;
    hipLaunchKernelGGL(gather_columns_kernel, dim3(clamp_grid(want, num_cu() * 16)), dim3(256), 0, (hipStream_t)stream,
                       M, ldm, R, cols, (int)nC, out, ldo, (const int64_t *)nullptr);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_fold_logaddexp(double *acc, int64_t lda, const double *const *in_host, const int64_t *ld_in_host,
                                  int32_t n_in, int64_t R, int32_t H, double delta, void *stream) {
    if (R < 0 || H <= 0 || lda < H || n_in < 0 || n_in > FOLD_MAX_IN)
        return fail(-1, "mxm_fold_logaddexp: bad shape (at most %s%lld inputs per call)", "", (long long)FOLD_MAX_IN);
    if (R == 0) return 0;
    fold_inputs in;
    bool vec = ((H & 1) == 0) && ((lda & 1) == 0) && ((reinterpret_cast<uintptr_t>(acc) & 15) == 0);
    for (int k = 0; k < FOLD_MAX_IN; ++k) {
        in.ptr[k] = k < n_in ? in_host[k] : nullptr;
        in.ld[k] = k < n_in ? ld_in_host[k] : 0;
        if (k < n_in) {
            if (in.ptr[k] == nullptr || in.ld[k] < H) return fail(-1, "mxm_fold_logaddexp: bad input %s%lld", "", k);
            vec = vec && ((in.ld[k] & 1) == 0) && ((reinterpret_cast<uintptr_t>(in.ptr[k]) & 15) == 0);
        }
    }
    const int grid = clamp_grid(R, num_cu() * 8);
    if (vec)
        hipLaunchKernelGGL(fold_logaddexp_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, acc, lda, in,
                           (int)n_in, R, (int)H, delta);
    else
        hipLaunchKernelGGL(fold_logaddexp_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, acc, lda, in,
                           (int)n_in, R, (int)H, delta);
    HIP_TRY(hipGetLastError());
    return 0;
}
