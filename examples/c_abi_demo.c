/*
 * c_abi_demo.c -- the drop-in boundary used from plain C: no Python, no PyTorch, only the HIP
 * runtime for memory and libmixemt_hip.so (include/mixemt_hip.h) for the hot path.
 *
 *   build_em_matrix  (preprocess.py:177-198)  ->  mxm_build_em_matrix
 *   run_em loop      (em.py:126-143)          ->  mxm_linearize + mxm_em_loop
 *   posterior        (em.py:80-83)            ->  mxm_em_step
 * With a third argument "records" the loop and the posterior run over row-dictionary records instead of the dense
 * matrix:  mxm_encode_rows -> mxm_em_loop_coded -> mxm_em_step_coded.
 *
 * Reads a problem file (flat tables + CSR observations + initial proportions, written by
 * tests/test_gpu_c_demo.py), runs the path on GPU 0 and writes proportions, iteration count and
 * the posterior matrix to a result file.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_demo.c \
 *       -Lmixemt_amd/lib -lmixemt_hip -L/opt/rocm/lib -lamdhip64 -lm -o c_abi_demo
 *   LD_LIBRARY_PATH=mixemt_amd/lib:/opt/rocm/lib ./c_abi_demo problem.bin result.bin [records]
 *
 * problem.bin: int64 header {R, H, S, nnz, lde, max_iter} ; double tol ;
 *              uint8 E[S*lde] ; double lhit[S] ; double lmiss[S] ; int64 row_ptr[R+1] ;
 *              uint16 site[nnz] ; uint8 obs[nnz] ; double weights[R] ; double init[H]
 * result.bin:  int64 {iters, done} ; double props[H] ; double M[R*H] ; double read_mix[R*H]
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "mixemt_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP: %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define MXM_OK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "mixemt_hip: %s (%d) at line %d\n", mxm_last_error(), rc_, __LINE__); return 3; } } while (0)

static void *slurp(FILE *f, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (p == NULL || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(4); }
    return p;
}

static void *to_device(const void *host, size_t bytes) {
    void *d = NULL;
    if (hipMalloc(&d, bytes ? bytes : 1) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(5); }
    if (bytes && hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "hipMemcpy failed\n"); exit(5); }
    return d;
}

int main(int argc, char **argv) {
    if (argc != 3 && argc != 4) { fprintf(stderr, "usage: %s problem.bin result.bin [records]\n", argv[0]); return 1; }
    const int records = (argc == 4);
    FILE *in = fopen(argv[1], "rb");
    if (in == NULL) { perror(argv[1]); return 1; }
    int64_t hdr[6];
    double tol;
    if (fread(hdr, sizeof(int64_t), 6, in) != 6 || fread(&tol, sizeof(double), 1, in) != 1) return 4;
    const int64_t R = hdr[0], H = hdr[1], S = hdr[2], nnz = hdr[3], lde = hdr[4], max_iter = hdr[5];
    uint8_t *E = (uint8_t *)slurp(in, (size_t)(S * lde));
    double *lhit = (double *)slurp(in, (size_t)S * 8), *lmiss = (double *)slurp(in, (size_t)S * 8);
    int64_t *row_ptr = (int64_t *)slurp(in, (size_t)(R + 1) * 8);
    uint16_t *site = (uint16_t *)slurp(in, (size_t)nnz * 2);
    uint8_t *obs = (uint8_t *)slurp(in, (size_t)nnz);
    double *wts = (double *)slurp(in, (size_t)R * 8), *init = (double *)slurp(in, (size_t)H * 8);
    fclose(in);
    printf("libmixemt_hip version %d: %lld reads x %lld haplogroups, %lld sites\n", mxm_version(),
           (long long)R, (long long)H, (long long)S);

    HIP_OK(hipSetDevice(0));
    void *dE = to_device(E, (size_t)(S * lde)), *dhit = to_device(lhit, (size_t)S * 8);
    void *dmiss = to_device(lmiss, (size_t)S * 8), *dptr = to_device(row_ptr, (size_t)(R + 1) * 8);
    void *dsite = to_device(site, (size_t)nnz * 2), *dobs = to_device(obs, (size_t)nnz);
    void *dw = to_device(wts, (size_t)R * 8);
    const int64_t ldp = (H + 1) / 2 * 2;
    double *dM, *dP, *drowmax, *dmix, *dcolsum, *dprops, *dlncur, *dlnnew;
    mxm_em_state *dstate, hstate;
    HIP_OK(hipMalloc((void **)&dM, (size_t)(R * H) * 8));
    HIP_OK(hipMalloc((void **)&dP, (size_t)(R * ldp) * 8));
    HIP_OK(hipMalloc((void **)&drowmax, (size_t)R * 8));
    HIP_OK(hipMalloc((void **)&dmix, (size_t)(R * H) * 8));
    HIP_OK(hipMalloc((void **)&dcolsum, (size_t)H * 8));
    HIP_OK(hipMalloc((void **)&dprops, (size_t)H * 8));
    HIP_OK(hipMalloc((void **)&dlncur, (size_t)H * 8));
    HIP_OK(hipMalloc((void **)&dlnnew, (size_t)H * 8));
    HIP_OK(hipMalloc((void **)&dstate, sizeof(mxm_em_state)));
    HIP_OK(hipMemset(dstate, 0, sizeof(mxm_em_state)));
    const size_t ws_bytes = mxm_workspace_bytes(R, (int32_t)H, 1);
    void *ws;
    HIP_OK(hipMalloc(&ws, ws_bytes));

    /* preprocess.build_em_matrix */
    MXM_OK(mxm_build_em_matrix((const uint8_t *)dE, lde, (const double *)dhit, (const double *)dmiss,
                               (const int64_t *)dptr, (const uint16_t *)dsite, (const uint8_t *)dobs, R,
                               (int32_t)H, (int32_t)S, dM, H, NULL));

    /* em.run_em, one run: log proportions are the loop's state (em.py:123-124) */
    double *ln0 = (double *)malloc((size_t)H * 8), *p0 = (double *)malloc((size_t)H * 8);
    for (int64_t h = 0; h < H; ++h) { ln0[h] = log(init[h]); p0[h] = exp(ln0[h]); }
    HIP_OK(hipMemcpy(dlncur, ln0, (size_t)H * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dlnnew, ln0, (size_t)H * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dprops, p0, (size_t)H * 8, hipMemcpyHostToDevice));
    const int linear = mxm_linear_supported((int32_t)H);
    if (!records) {
        if (linear) MXM_OK(mxm_linearize(dM, H, R, (int32_t)H, dP, ldp, drowmax, NULL));
        MXM_OK(mxm_em_loop(dM, H, linear ? dP : NULL, linear ? ldp : 0, (const double *)dw, R, (int32_t)H, 1,
                           dprops, dlncur, dlnnew, dcolsum, dstate, tol, (int32_t)max_iter, 16, ws, ws_bytes,
                           NULL, &hstate));
        /* the reference returns the E-step under theta_k with theta_{k+1} (em.py:130-143) */
        MXM_OK(mxm_em_step(dM, H, NULL, dlncur, R, (int32_t)H, dmix, H, 0, NULL, NULL, 0, NULL));
    } else {
        /* the same loop over row-dictionary records: one code per cell + the row's distinct values (lossless) */
        if (!linear || (H & 1)) { fprintf(stderr, "records need an even H in [66, 8192]\n"); return 6; }
        const size_t rec_bytes = mxm_coded_bytes(R, (int32_t)H);
        uint8_t *drec;
        int64_t *drec_off, *dstats, stats[2];
        int32_t *dndist, *ndist = (int32_t *)malloc((size_t)R * 4);
        HIP_OK(hipMalloc((void **)&drec, rec_bytes));
        HIP_OK(hipMalloc((void **)&drec_off, (size_t)R * 8));
        HIP_OK(hipMalloc((void **)&dndist, (size_t)R * 4));
        HIP_OK(hipMalloc((void **)&dstats, 16));
        MXM_OK(mxm_encode_rows(dM, H, R, (int32_t)H, drec, rec_bytes, drec_off, dndist, drowmax, dstats, NULL));
        HIP_OK(hipMemcpy(stats, dstats, 16, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(ndist, dndist, (size_t)R * 4, hipMemcpyDeviceToHost));
        if (stats[1] != 0) { fprintf(stderr, "%lld rows with more than 1024 distinct values: keep them dense (mxm_coded.P_rest)\n", (long long)stats[1]); return 6; }
        /* the rows with 16-bit codes, in row order, are part of the descriptor */
        int64_t *wide = (int64_t *)malloc((size_t)R * 8), n_wide = 0;
        for (int64_t r = 0; r < R; ++r) if (ndist[r] > 256) wide[n_wide++] = r;
        void *dwide = n_wide ? to_device(wide, (size_t)n_wide * 8) : NULL;
        mxm_coded c = {drec, drec_off, dndist, R, NULL, 0, NULL, 0, (const int64_t *)dwide, n_wide};
        printf("records: %.1f KB for a %.1f KB matrix, %lld rows with 16-bit codes\n", stats[0] / 1024.0,
               (double)(R * H) * 8 / 1024.0, (long long)n_wide);
        MXM_OK(mxm_em_loop_coded(&c, (const double *)dw, (int32_t)H, 1, dprops, dlncur, dlnnew, dcolsum, dstate, tol,
                                 (int32_t)max_iter, 16, ws, ws_bytes, NULL, &hstate));
        /* posterior under theta_k from the records' log tables; it wants exp(ln theta_k) beside the logs */
        double *lnk = (double *)malloc((size_t)H * 8);
        HIP_OK(hipMemcpy(lnk, dlncur, (size_t)H * 8, hipMemcpyDeviceToHost));
        for (int64_t h = 0; h < H; ++h) lnk[h] = exp(lnk[h]);
        void *dpk = to_device(lnk, (size_t)H * 8);
        MXM_OK(mxm_em_step_coded(&c, (int32_t)H, dlncur, (const double *)dpk, drowmax, NULL, 0, NULL, 0, dmix, H, 0, NULL));
    }
    HIP_OK(hipDeviceSynchronize());

    double *lnnew = (double *)malloc((size_t)H * 8);
    double *M = (double *)malloc((size_t)(R * H) * 8), *mix = (double *)malloc((size_t)(R * H) * 8);
    HIP_OK(hipMemcpy(lnnew, dlnnew, (size_t)H * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(M, dM, (size_t)(R * H) * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(mix, dmix, (size_t)(R * H) * 8, hipMemcpyDeviceToHost));
    FILE *out = fopen(argv[2], "wb");
    if (out == NULL) { perror(argv[2]); return 1; }
    int64_t tail[2] = {hstate.iters, hstate.done};
    fwrite(tail, sizeof(int64_t), 2, out);
    for (int64_t h = 0; h < H; ++h) lnnew[h] = exp(lnnew[h]);          /* em.py:163 */
    fwrite(lnnew, 8, (size_t)H, out);
    fwrite(M, 8, (size_t)(R * H), out);
    fwrite(mix, 8, (size_t)(R * H), out);
    fclose(out);
    printf("%s after %d iterations (l1 = %.3g)\n", hstate.done == 1 ? "converged" : "stopped", hstate.iters, hstate.l1);
    return 0;
}
