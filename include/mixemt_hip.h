/*
 * mixemt_hip.h -- C ABI of libmixemt_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for mixemt's EM hot path.  The reference has no native
 * code and no plugin registry: the boundary it offers is four Python functions
 *   preprocess.build_em_matrix(refseq, phylo, reads, haplogroups, args)   /root/reference/mixemt/preprocess.py:177
 *   em.em_step(read_hap_mat, weights, ln_props, read_mix_mat)             /root/reference/mixemt/em.py:57
 *   em.converged(prop, last_prop, tolerance)                              /root/reference/mixemt/em.py:39
 *   em.run_em(read_hap_mat, weights, args)                                /root/reference/mixemt/em.py:94
 * Each entry point below names the reference lines whose arithmetic it
 * replaces.  A Python maintainer binds them with ctypes (INTEGRATION.md shows
 * the stub); mixemt_amd/{preprocess,em}.py are that binding, re-exposing the
 * four signatures above.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     every call only enqueues work on it unless stated otherwise;
 *   - matrices are row-major with an explicit leading dimension (in elements);
 *   - return value: 0 = ok, negative = error (text via mxm_last_error());
 *   - no allocation inside: the caller owns every buffer including `ws`
 *     (size from mxm_workspace_bytes), so calls are hipGraph-capturable;
 *   - work goes to the calling thread's current device; one process (or thread) per GPU.
 *     mxm_last_error() is per thread.  Tuning / measurement knobs are NOT part of this
 *     boundary: they live in mixemt_hip_tuning.h.
 */
#ifndef MIXEMT_HIP_H
#define MIXEMT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 0.5.4.  Bumped whenever a signature or a documented default of this header changes; mxm_version() returns
 * the value the loaded library was built with, and a binding must refuse a library whose value differs from the
 * header it was written against (mixemt_amd/_lib.py does).  History: 100 rounds 1-2 (mxm_row_argmax_votes gained
 * ws / ws_bytes and mxm_set_compact_restarts became 0/1/2 inside that number -- the reason for this rule);
 * 300 round 3: mxm_build_em_matrix_packed / mxm_build_packed_lds_bytes removed, mxm_build_em_matrix_lut lost its
 * P / ldp / rowmax outputs, mxm_row_argmax_votes_coded replaces mxm_row_argmax_coded,
 * mxm_em_step_coded / mxm_gather_columns_coded cover the rows without a record;
 * 400 round 4: mxm_coded gained wide_rows / n_wide (records with 16-bit codes for rows of 257..1024 distinct values),
 * mxm_record_bytes / mxm_coded_bytes grew with them, mxm_workspace_bytes covers the one-launch loops' layouts,
 * mxm_em_state gained `ticket` (24 bytes: mxm_m_finalize runs on several workgroups, the last arriver finishes);
 * 500 round 5: mxm_aln_* (the batched alignment front end), mxm_preload, mxm_build_em_matrix_lut_rows and
 * mxm_scatter_records added; mxm_em_state.reserved_ became .error and
 * mxm_em_iter_coded's state lost its const (the wide-rows list is checked where it is used);
 * 501: mxm_bam_* (a BAM file into the front end's columns; the library now links zlib);
 * 502: mxm_coded gained the quad dictionary's fields (qrec .. n_byte_rows), mxm_build_quads, mxm_quad_bytes;
 * 503: mxm_quad_lists, mxm_quad_lists_scratch_bytes;
 * 504: mxm_exchange_* (the optional one-shot exchange of a row-sharded loop); mxm_em_state.error may be 2;
 * 505: mxm_exchange_reduce; 506: mxm_expand_tables;
 * 600 round 6: mxm_restart_tile_coded (three restarts share a pass over records beside a quad dictionary:
 * mxm_em_iter_coded / mxm_em_loop_coded take full tiles of three through em_iter_quad_batched_kernel),
 * mxm_quad_loop_min_rows. */
#define MXM_VERSION 600

/* per-restart loop state, written by mxm_m_finalize (24 bytes); allocate it ZEROED */
typedef struct mxm_em_state {
    int32_t done;                /* 0 running, 1 converged, 2 max_iter reached */
    int32_t iters;               /* EM steps executed so far ("Converged! (n)", em.py:135) */
    double  l1;                  /* last sum_h |p_new - p_cur|  (em.py:53-54) */
    uint32_t ticket;             /* scratch of mxm_m_finalize (arrival count of its workgroups): 0 between calls */
    uint32_t error;              /* 0 = fine; 2 = mxm_exchange_pull timed out waiting for a rank; 1 = mxm_em_iter_coded found that mxm_coded.wide_rows is not exactly the set of
                                    rows with more than 256 values (colsum is then NaN throughout); mxm_em_loop_coded
                                    returns -1 instead of iterating on */
} mxm_em_state;

int         mxm_version(void);
const char *mxm_last_error(void);
/* HIP loads a library's code object lazily, on its first kernel launch (~18 ms for this one); mxm_preload() does it now, on
 * the calling thread's current device, and returns when it is loaded (blocking).  Optional: a binding calls it when it loads
 * the library so that the cost does not land in the first stage that happens to launch. */
int         mxm_preload(void);

/* Largest haplogroup count the linear-space streaming kernel accepts; above it
 * (and below MXM_LINEAR_MIN_H) mxm_em_iter runs the generic log-space kernel. */
int         mxm_linear_supported(int32_t H);

/* How many restarts share one pass over the matrix in mxm_em_iter at this width (1..4: the
 * per-restart on-chip state -- accumulators, proportions -- must fit a CU beside the row buffers);
 * B restarts take ceil(B / tile) passes.  mxm_em_loop keeps one full tile iterating. */
int         mxm_restart_tile(int32_t H);

/* Bytes of scratch mxm_em_iter / mxm_em_step / mxm_em_loop need. */
size_t      mxm_workspace_bytes(int64_t R, int32_t H, int32_t B);

/*
 * build_em_matrix -- preprocess.py:177-198 (with :69-96 inlined).
 *   M[r][h] = sum over the row's observations k, IN ORDER, of
 *             (E[site_k][h] == obs_k ? lhit[site_k] : lmiss[site_k])
 * E[S][lde]  uint8: the base (ASCII) haplogroup h is expected to show at site s
 *            (marker if it carries one there, else the reference base; :60-66, :75-83)
 * lhit/lmiss[S]  log(1-mu_s), log(mu_s/3) computed on the host (:48-51, :80-84)
 * row_ptr[R+1], site[nnz] (index into the sorted site list), obs[nnz] (ASCII):
 *            CSR form of the read signatures (:151-160)
 * lde must be a multiple of 8 and >= H rounded up to 8; E 8-byte aligned (pad bytes 0).
 */
int mxm_build_em_matrix(const uint8_t *E, int64_t lde,
                        const double *lhit, const double *lmiss,
                        const int64_t *row_ptr, const uint16_t *site,
                        const uint8_t *obs,
                        int64_t R, int32_t H, int32_t S,
                        double *M, int64_t ldm, void *stream);

/*
 * The dense tables of the two cell-by-cell build kernels from the marker form, on the device (nothing of the reference: its
 * HapVarBaseMatrix is a dict of dicts, preprocess.py:39-67): out[s][h] = map256[maj[s]] for every haplogroup h < H, then
 * out[s][mk_hap[j]] = map256[mk_base[j]] for the markers j of site s; columns H .. lde are 0.  map256 NULL = identity:
 * `E` of mxm_build_em_matrix; map256 = the 4-bit code table (<< 3): `Ecode` of mxm_build_em_matrix_lut.
 * maj / mk_ptr / mk_hap / mk_base as for mxm_build_em_matrix_sparse.
 */
int mxm_expand_tables(const uint8_t *maj, const int32_t *mk_ptr, const uint16_t *mk_hap, const uint8_t *mk_base,
                      const uint8_t *map256, int32_t S, int32_t H, int64_t lde, uint8_t *out, void *stream);

/*
 * build_em_matrix, lookup-table fast path -- same arithmetic, same order, same bits as
 * mxm_build_em_matrix (preprocess.py:177-198 with :69-96 inlined); the hit / miss choice of _prob
 * (:75-84) is a 16-entry LDS lookup instead of two selects.  The host pre-encodes
 *   Ecode[S][lde]  uint8: 4-bit code of the expected base, SHIFTED LEFT BY 3 (code 1..14 names the
 *                  alphabet of the table, at most 14 letters; pad bytes 0); 8-byte aligned, lde a
 *                  multiple of 8 and >= H rounded up to 8
 *   obsmap[256]    uint8: observation byte -> its code << 3; 15 << 3 for a byte that equals no expected base
 * order[R] (nullable): a permutation of the rows -- the order in which the grid takes them (rows that
 *   start at nearby positions share table rows; dealing them out together keeps those in L2).
 *   Results do not depend on it.
 * H <= 8192, S * lde < 2^31; otherwise call mxm_build_em_matrix.
 */
int mxm_build_em_matrix_lut(const uint8_t *Ecode, int64_t lde, const double *lhit, const double *lmiss,
                            const uint8_t *obsmap, const int64_t *row_ptr, const uint16_t *site,
                            const uint8_t *obs, const int64_t *order, int64_t R, int32_t H, int32_t S,
                            double *M, int64_t ldm, void *stream);

/* The same for a LIST of rows into a compact matrix: row rows[i] of the CSR -> row i of M_out[n_rows][ldm] (the rows the
 * marker kernel hands back, built a slab at a time instead of inside the whole matrix). */
int mxm_build_em_matrix_lut_rows(const uint8_t *Ecode, int64_t lde, const double *lhit, const double *lmiss,
                                 const uint8_t *obsmap, const int64_t *row_ptr, const uint16_t *site,
                                 const uint8_t *obs, const int64_t *rows, int64_t n_rows, int32_t H, int32_t S,
                                 double *M_out, int64_t ldm, void *stream);

/*
 * build_em_matrix from the haplogroups' MARKERS -- preprocess.py:177-198, the same bits as the kernels
 * above, formed once per distinct cell value of a row instead of once per cell:
 *   maj[S]            the base most haplogroups expect at site s
 *   mk_ptr[S+1], mk_hap[], mk_base[]   CSR over sites of the (haplogroup, expected base) pairs that
 *                     differ from maj[s]  (Build 17: 113 027 entries)
 *   obs[]             the observation's byte itself (it hits where it equals the expected base's byte)
 * A haplogroup's cell is decided by the set of the row's sites where its term differs from the
 * majority's (a mask OR-ed together from the marker lists: 64 bits for rows of up to 64 observations, 128 bits --
 * round 6, a second launch over the same rows -- for rows of 65 .. 128: the reference merges mates into one
 * fragment, preprocess.py:118-138, and two thirds of 2 x 150 paired-end fragments observe more than 64 sites);
 * the row's distinct masks are deduplicated and each one's sum is formed in signature order from 0.0, as
 * prob_for_vars does (preprocess.py:86-96).  Rows with more than 128 observations, more than 352 (long rows:
 * 704) distinct values, or -- long rows -- more than 5120 marker entries are NOT written: their indices are
 * appended to fallback[] (device int64[R], *n_fallback = how many, device) and the caller builds them with
 * mxm_build_em_matrix_lut (order = fallback, R = *n_fallback; any order of the list; every row at most once).
 */
int mxm_build_em_matrix_sparse(const uint8_t *maj, const double *lhit, const double *lmiss,
                               const int32_t *mk_ptr, const uint16_t *mk_hap, const uint8_t *mk_base,
                               const int64_t *row_ptr, const uint16_t *site, const uint8_t *obs,
                               const int64_t *order, int64_t R, int32_t H, int32_t S,
                               double *M, int64_t ldm, int64_t *fallback, int64_t *n_fallback, void *stream);

/*
 * The same marker build, leaving each row as a ROW-DICTIONARY RECORD (mxm_coded below) -- the kernel holds
 * the row's distinct values and every haplogroup's index into them, so no pass over a dense matrix
 * (mxm_encode_rows) is needed, and with M == NULL no dense matrix is written at all:
 *     record = codes[ldc] ++ P table[ndist] ++ table of the log sums themselves [ndist]
 * (P = exp(sum - rowmax[r]); code 0 = the value of a haplogroup without a deviating marker in the window).
 * Rows of 257 .. 705 distinct values get a record with 16-bit codes ("wide", round 6; 2 * ldc bytes of codes).
 * Rows that get no record have ndist[r] = 0: the rows on the fallback list (as above), which the caller builds
 * densely; with M given their dense row is NOT written either (the list says which).  stats[0] = bytes used,
 * stats[1] = rows without a record (device int64[2]); rec_bytes >= mxm_record_bytes(R, H) never overflows.
 */
size_t mxm_record_bytes(int64_t R, int32_t H);
int mxm_build_em_records(const uint8_t *maj, const double *lhit, const double *lmiss,
                         const int32_t *mk_ptr, const uint16_t *mk_hap, const uint8_t *mk_base,
                         const int64_t *row_ptr, const uint16_t *site, const uint8_t *obs,
                         const int64_t *order, int64_t R, int32_t H, int32_t S, double *M, int64_t ldm,
                         uint8_t *rec, size_t rec_bytes, int64_t *rec_off, int32_t *ndist, double *rowmax,
                         int64_t *stats, int64_t *fallback, int64_t *n_fallback, void *stream);

/*
 * One-time change of variables for the streaming loop:
 *   rowmax[r] = max_h M[r][h]   (0 if not finite),  P[r][h] = exp(M[r][h] - rowmax[r])
 * ldp must be even and >= H; pad columns [H, ldp) are written as 0.
 * (em.py:80-83 evaluated once instead of every iteration; see DESIGN.md.)
 */
int mxm_linearize(const double *M, int64_t ldm, int64_t R, int32_t H,
                  double *P, int64_t ldp, double *rowmax, void *stream);

/*
 * em_step, E+M fused, this rank's row shard -- em.py:80-88.
 *   colsum[b][h] = T_bh = sum_r w[r] * P[r][h] / Z_b[r],   Z_b[r] = sum_h props[b][h] * P[r][h]
 * i.e. the M-step sums WITHOUT their factor props[b][h]:  sum_r w[r] * posterior_b[r][h] =
 * props[b][h] * T_bh.  The factor is left out so that a proportion that underflows in linear
 * space still gets an exact log update in mxm_m_finalize (the reference keeps log proportions).
 * P != NULL and mxm_linear_supported(H): streams P (no transcendental), reads `props`;
 * otherwise streams M in log space and reads `ln_props` (must then be non-NULL).
 * Restarts with state[b].done != 0 are skipped (their colsum is left untouched).
 * A row with Z_b[r] == 0 (-inf in every column) and w[r] != 0 makes every T_bh NaN, as the
 * reference's -inf - (-inf) does (em.py:81-83, :87); with w[r] == 0 it is dropped like scipy drops it.
 * LIMIT of the linear form: Z_b[r] is also 0 when every haplogroup that supports row r (P > 0) has a
 * proportion whose exp() underflowed (ln p < -745) although the row is not empty; the reference's
 * log-space E-step stays finite there, this kernel poisons the restart with NaN (it then runs to
 * max_iter and reports NaN proportions -- loud, not silent).  Matrices from build_em_matrix cannot
 * get there: every cell is finite and >= k * log(0.0033), so the best-supported haplogroup of a row
 * keeps P = 1 and would need ln p < -745 while the row's weight pulls it up every iteration.
 * props[B][H] = exp(ln_props[B][H]) (theta_k), w[R] fp64 weights (NULL = all 1).
 * With several ranks the caller all-reduces (SUM) colsum before mxm_m_finalize.
 */
int mxm_em_iter(const double *M, int64_t ldm, const double *P, int64_t ldp,
                const double *w, const double *props, const double *ln_props,
                int64_t R, int32_t H, int32_t B,
                const mxm_em_state *state, double *colsum,
                void *ws, size_t ws_bytes, void *stream);

/*
 * M-step normalisation + convergence test -- em.py:87-89, :39-54, :133-143, in the reference's
 * own variable, the LOG proportions:
 *   ln_new = ln_cur + log(colsum) - log(sum_h props_cur * colsum)
 *   l1 = sum_h |exp(ln_new) - props_cur| ;  iters += 1
 *   l1 < tol -> done = 1 ; else iters >= max_iter -> done = 2 ;
 *   else ln_cur := ln_new, props_cur := exp(ln_new)
 * After the loop stops: ln_cur = log theta_k, ln_new = log theta_{k+1} -- the pair the
 * reference returns (posterior one step behind the proportions).  props_cur = exp(ln_cur).
 */
int mxm_m_finalize(const double *colsum, double *ln_cur, double *ln_new,
                   double *props_cur, int32_t H, int32_t B, double tol,
                   int32_t max_iter, mxm_em_state *state, void *stream);

/*
 * fp32-STORAGE variant of the loop (opt-in; NOT the reference's arithmetic type for the stored
 * matrix): P is kept as float -- half the HBM bytes per iteration -- while every product, row sum
 * and column sum stays fp64.  One restart per pass.  ldp (floats) a multiple of 4, pad columns 0.
 * Everything else (w, props, colsum, state, finalize, posterior from the fp64 M) is unchanged.
 */
int mxm_linearize_f32(const double *M, int64_t ldm, int64_t R, int32_t H,
                      float *P, int64_t ldp, double *rowmax, void *stream);
int mxm_em_iter_f32(const float *P, int64_t ldp, const double *w, const double *props,
                    int64_t R, int32_t H, int32_t B, const mxm_em_state *state,
                    double *colsum, void *ws, size_t ws_bytes, void *stream);
int mxm_em_loop_f32(const float *P, int64_t ldp, const double *w, int64_t R, int32_t H,
                    int32_t B, double *props_cur, double *ln_cur, double *ln_new,
                    double *colsum, mxm_em_state *state, double tol, int32_t max_iter,
                    int32_t check_every, void *ws, size_t ws_bytes, void *stream,
                    mxm_em_state *state_host);

/*
 * ROW-DICTIONARY storage of the linearised matrix (lossless; opt-in like the fp32 variant, but it keeps
 * every bit).  A row of build_em_matrix's output (preprocess.py:177-198) is a sum of per-site terms with
 * two possible values each, so its H cells hold few DISTINCT doubles.  Row r is stored as one record
 *     codes[ldc] (one per column, ldc = H rounded up to 8)  ++  table[ndist[r]] (doubles)  ++  mtable[ndist[r]],
 *     P[r][h] = exp(M[r][h] - rowmax[r]) = table[codes[h]]          -- the bits mxm_linearize writes
 *     M[r][h] = mtable[codes[h]]                                    -- the log value itself
 * at rec + rec_off[r]: ~5.9 KB instead of 43 KB at H = 5408.  Codes are uint8 for rows of at most 256 distinct
 * values and uint16 for 257..1024 ("wide" records, 2 * ldc bytes of codes; round 4 -- these rows used to stay
 * dense).  Rows with more than 1024 distinct values get ndist[r] = 0; the caller keeps those dense (P_rest / w_rest:
 * their mxm_linearize rows and weights, in any fixed order) and both parts are summed by one column reduce.
 *   mxm_coded_bytes(R, H)   record buffer size that can never overflow (R * (2 ldc + 16384)).  A smaller buffer is
 *                           allowed: rows that no longer fit get no record, and stats[0] > rec_bytes afterwards says so
 *                           (and is the size that would have sufficed) -- the caller repeats the call with that much
 *   mxm_encode_rows         M -> records (byte-coded pass over all rows, then a 16-bit pass over what it left);
 *                           stats[0] = bytes asked for, stats[1] = rows left dense (device int64[2]);
 *                           needs an even H in [66, 8192], even ldm, 16-byte aligned M and rec
 *   mxm_decode_rows         P[r][:] = row r decoded, coded rows only (tests; posterior passes)
 *   mxm_em_iter_coded / mxm_em_loop_coded   = mxm_em_iter / mxm_em_loop over a coded matrix
 *                           (w[R] indexes all rows, coded or not; one restart per pass -- beside a quad dictionary
 *                           full tiles of mxm_restart_tile_coded() = 3 restarts share a pass, em.py:117-161 runs them
 *                           one after another over the same matrix)
 */
typedef struct mxm_coded {
    const uint8_t *rec;          /* records */
    const int64_t *rec_off;      /* [R] byte offset of row r's record */
    const int32_t *ndist;        /* [R] table entries of row r: 1..256 byte codes, 257..1024 16-bit codes ("wide"),
                                    0 = row r is one of the dense rest */
    int64_t        R;            /* rows, coded or not */
    const double  *P_rest;       /* [R_rest][ldp_rest] linearised rows that did not code (NULL if none) */
    int64_t        ldp_rest;
    const double  *w_rest;       /* their weights (NULL = 1) */
    int64_t        R_rest;
    const int64_t *wide_rows;    /* [n_wide] EXACTLY the rows with ndist > 256, ASCENDING (torch.nonzero(ndist > 256)).  The
                                    EM iteration skips those rows in its main pass and takes them from this list, so
                                    its sums depend on it: mxm_em_loop_coded checks the list on entry (-1 if it is
                                    not that set), mxm_em_iter_coded checks it on the device in every pass (a fault:
                                    colsum all NaN and state[b].error = 1; faulty entries are never dereferenced).
                                    Nothing else reads it: the vote, the posterior, the gathers and the decoder find
                                    the wide rows from ndist themselves. */
    int64_t        n_wide;
    /* QUAD dictionary beside the records (round 5, optional: all NULL / 0 = none) -- an acceleration structure for
     * mxm_em_iter_coded / mxm_em_loop_coded alone, made by mxm_build_quads; every other consumer reads the records,
     * which stay complete.  A quad record names FOUR consecutive columns' values with one code byte (256 x 8 code
     * bytes, thread-contiguous, then nquad[r] x 32 bytes of table: csrc/quad_kernels.hpp): 1.27 against 1.41 ms per
     * pass at 10^6 x 5408.  With quads the iteration takes the rows of `quad_rows` from `qrec`, the rows of `byte_rows`
     * and `wide_rows` from `rec`, the dense rest from P_rest; the lists are checked where they are used (in range,
     * ascending, of their class: a fault poisons colsum with NaN and raises state[b].error, as for wide_rows), and
     * n_quad_rows + n_byte_rows + n_wide + R_rest must be R. */
    const uint8_t *qrec;         /* quad records */
    const int64_t *qoff;         /* [R] byte offset of row r's quad record */
    const int32_t *nquad;        /* [R] distinct quads of row r: 1..256, 0 = the row has no quad record */
    const int64_t *quad_rows;    /* [n_quad_rows] the rows with nquad > 0, ASCENDING */
    int64_t        n_quad_rows;
    const int64_t *byte_rows;    /* [n_byte_rows] the rows with 0 < ndist <= 256 and nquad == 0, ASCENDING */
    int64_t        n_byte_rows;
} mxm_coded;
size_t mxm_coded_bytes(int64_t R, int32_t H);
/*
 * mxm_build_quads    records -> quad records (device work on `stream`, no allocation): qoff[R] / nquad[R] for EVERY row
 *                    (0 for rows without a byte-coded record and for rows with more than 256 distinct quads); the
 *                    records go to qrec[0 .. qrec_bytes) by a bump allocator (64 KB reserved at a time by each of at most
 *                    5120 waves: up to 5120 x 64 KB are open, i.e. reserved and partly unused, when the call ends),
 *                    stats[0] = bytes reserved (a record that no longer fits is not written and its row keeps
 *                    nquad = 0: repeat with stats[0] bytes and some room to spare),
 *                    stats[1] = byte-coded rows left without quads (device uint64[2], zeroed by the call).  qrec must be
 *                    32-byte aligned.  The caller forms quad_rows / byte_rows from nquad and ndist.
 * mxm_quad_bytes     a buffer size that can never overflow: 64 KB pieces of at least six largest records (2048 + 256 * 32
 *                    bytes) each, + the 5120 pieces that may be open at the end
 */
size_t mxm_quad_bytes(int64_t R, int32_t H);
/* mxm_quad_lists     the two row lists from ndist / nquad, on the device, ASCENDING: quad_rows[R] / byte_rows[R] (room for
 *                    every row), counts[2] = {n_quad_rows, n_byte_rows} (device int64); scratch of
 *                    mxm_quad_lists_scratch_bytes(R) bytes.  (Forming them on the host is as valid; this saves the round
 *                    trip: 4 bytes per row down, 8 bytes per row up.) */
size_t mxm_quad_lists_scratch_bytes(int64_t R);
int mxm_quad_lists(const int32_t *ndist, const int32_t *nquad, int64_t R, int64_t *quad_rows, int64_t *byte_rows,
                   int64_t *counts, void *scratch, size_t scratch_bytes, void *stream);
int mxm_build_quads(const mxm_coded *c, int32_t H, uint8_t *qrec, size_t qrec_bytes, int64_t *qoff, int32_t *nquad,
                    uint64_t *stats, void *stream);
int mxm_encode_rows(const double *M, int64_t ldm, int64_t R, int32_t H, uint8_t *rec, size_t rec_bytes,
                    int64_t *rec_off, int32_t *ndist, double *rowmax, int64_t *stats, void *stream);
int mxm_decode_rows(const mxm_coded *c, int32_t H, double *P, int64_t ldp, void *stream);
/* Records that mxm_encode_rows made from a compact side matrix of the rows rows[0 .. n) take their place in the matrix's
 * record arrays: where sub_nd[i] > 0, rec_off[rows[i]] = sub_off[i] + base, ndist[rows[i]] = sub_nd[i], rowmax[rows[i]] = sub_rm[i]. */
int mxm_scatter_records(const int64_t *rows, int64_t n, const int64_t *sub_off, const int32_t *sub_nd, const double *sub_rm,
                        int64_t base, int64_t *rec_off, int32_t *ndist, double *rowmax, void *stream);
/* The contributor vote straight from records -- assemble.py:115-123 / stats.py:39-40 over run_em's returned posterior
 * (em.py:145-161), for ALL rows, without a dense matrix or a posterior matrix:
 *   best[r] = first index of max_h  fold_k logaddexp( ln_props[k][h] + M[r][h] - lse_k[r] )     k = 0 .. n_runs-1 in run order
 *   votes[h] = sum of w[r] over the rows with best[r] == h   (NULL = not wanted; no float atomics: reproducible)
 * ln_props / props [n_runs][H]: each run's log theta_k and exp of it; rowmax[R] from the encoder.  With one run the
 * row normaliser drops out (props / rowmax may be NULL); with several, each run's normaliser weighs that run's columns.
 * Rows without a record (ndist[r] == 0) are read from M_rest[n_rest][ldm_rest] -- their LOG values, row i being row
 * rest_rows[i] of the matrix; a row without a record that is not among them gets best[r] = -1 and no vote.
 * n_runs <= 4096.  ws / ws_bytes as for mxm_row_argmax_votes (only when votes != NULL). */
int mxm_row_argmax_votes_coded(const mxm_coded *c, int32_t H, int32_t n_runs, const double *ln_props,
                               const double *props, const double *rowmax, const double *M_rest, int64_t ldm_rest,
                               const int64_t *rest_rows, int64_t n_rest, const double *w, int32_t *best,
                               double *votes, void *ws, size_t ws_bytes, void *stream);
/* mxm_em_step for coded rows -- em.py:80-83 (mode 0 store) and :156 (mode 1 logaddexp fold):
 *   out[r][h] = (ln_props[h] + M[r][h]) - (rowmax[r] + log(sum_h props[h] * P[r][h]))
 * the row's log-sum-exp taken in the loop's own linear variables (props = exp(ln_props), rowmax from the encoder);
 * a row whose linear sum is 0 or not finite (every supported haplogroup underflowed) is redone in log space with a
 * max shift, so the result is finite exactly where mxm_em_step's is.
 * The rows without a record (ndist[r] == 0) come from M_rest[n_rest][ldm_rest], their LOG values, row i of it being
 * row rest_rows[i] of the matrix (n_rest = 0: none; they are then left untouched): the reference's E-step in log
 * space (mxm_em_step's generic kernel) / a plain gather, written to their own rows of `out`. */
int mxm_em_step_coded(const mxm_coded *c, int32_t H, const double *ln_props, const double *props,
                      const double *rowmax, const double *M_rest, int64_t ldm_rest, const int64_t *rest_rows,
                      int64_t n_rest, double *out, int64_t ldo, int32_t mode, void *stream);
int mxm_gather_columns_coded(const mxm_coded *c, int32_t H, const int32_t *cols, int32_t nC,
                             const double *M_rest, int64_t ldm_rest, const int64_t *rest_rows, int64_t n_rest,
                             double *out, int64_t ldo, void *stream);
int mxm_em_iter_coded(const mxm_coded *c, const double *w, const double *props, int32_t H, int32_t B,
                      mxm_em_state *state /* nullable; only .error is ever written */, double *colsum, void *ws,
                      size_t ws_bytes, void *stream);
/* Rows WITH quads from which mxm_em_loop_coded iterates with the per-iteration kernels over the quad dictionary instead of
 * the one-launch loop over the records alone: a binding that attaches a dictionary "where it pays" must test the same
 * quantity (n_quad_rows after mxm_build_quads), or it builds one the loop never looks at. */
int64_t mxm_quad_loop_min_rows(int32_t B /* restarts of the run: from three on, tiles of three share a pass and the floor is far lower */);
/* Restarts that share one pass over THIS coded matrix in mxm_em_iter_coded / mxm_em_loop_coded: 3 beside a quad dictionary
 * (width within the quad pass's range, no dense leftover rows), else 1.  B restarts take floor(B / 3) shared passes and
 * B mod 3 single ones; per-restart sums differ from the one-per-pass kernel's by the rounding of another order only. */
int mxm_restart_tile_coded(const mxm_coded *c, int32_t H);
int mxm_em_loop_coded(const mxm_coded *c, const double *w, int32_t H, int32_t B,
                      double *props_cur, double *ln_cur, double *ln_new, double *colsum,
                      mxm_em_state *state, double tol, int32_t max_iter, int32_t check_every,
                      void *ws, size_t ws_bytes, void *stream, mxm_em_state *state_host);

/*
 * The run_em inner loop for ONE rank -- em.py:126-143: repeats
 * {mxm_em_iter; mxm_m_finalize} on `stream` until every restart is done.
 * A single restart on a matrix of up to 1e8 cells (H <= 6144) runs its whole loop in ONE persistent launch instead (one
 * workgroup per CU, grid barriers; same results up to rounding of the summation order).  That grid needs
 * the device to itself while it runs: the library checks with the runtime that the grid can be co-resident,
 * serialises calls from several threads of one process, and bounds every spin; if a second PROCESS holding CUs
 * starves the grid it gives up after a few seconds, the launch is undone (loop vectors restored from a snapshot in
 * `ws`) and the same call finishes through the per-iteration kernels (never a hang, never a wrong result).
 * Processes that share a GPU avoid the wait with mxm_set_loop_fused(0) (mixemt_hip_tuning.h); only
 * mxm_set_loop_fused(1) turns giving up into an error (-3).
 * Iterations are enqueued in chunks of `check_every`; kernels of a finished
 * restart are no-ops, so the state freezes on exactly the iteration the
 * reference would stop on.  Blocks the calling thread (stream sync per chunk).
 * state_host[B] receives the final states.
 */
int mxm_em_loop(const double *M, int64_t ldm, const double *P, int64_t ldp,
                const double *w, int64_t R, int32_t H, int32_t B,
                double *props_cur, double *ln_cur, double *ln_new, double *colsum,
                mxm_em_state *state, double tol, int32_t max_iter,
                int32_t check_every, void *ws, size_t ws_bytes, void *stream,
                mxm_em_state *state_host);

/*
 * E-step with the reference's semantics, writing the posterior -- em.py:80-83:
 *   out[r][h] = (ln_props[h] + M[r][h]) - logsumexp_h(ln_props + M[r])
 * mode 0: store; mode 1: out = logaddexp(out, value) (multi-run fold, em.py:156).
 * colsum (nullable): also the M-step sums sum_r w[r]*exp(out) (em.py:87-88).
 */
int mxm_em_step(const double *M, int64_t ldm, const double *w,
                const double *ln_props, int64_t R, int32_t H,
                double *out, int64_t ldo, int32_t mode, double *colsum,
                void *ws, size_t ws_bytes, void *stream);

/* ln_new[h] = log(colsum[h]) - log(sum_h colsum[h])   (em.py:87-89) */
int mxm_log_normalize(const double *colsum, int32_t H, double *ln_new,
                      void *stream);

/* l1_out[0] = sum_h |exp(a[h]) - exp(b[h])|   (em.py:53-54) */
int mxm_l1_exp_diff(const double *a, const double *b, int32_t H,
                    double *l1_out, void *stream);

/* x[i] += delta over an [R][ld] matrix's first H columns (em.py:161) */
int mxm_add_scalar(double *x, int64_t ld, int64_t R, int32_t H, double delta,
                   void *stream);

/*
 * Row argmax + weighted votes -- assemble.py:115-123 / stats.py:39-40:
 *   best[r] = first index of max_h X[r][h] ;  votes[h] = sum_{r: best[r]==h} w[r]
 * votes[H] is overwritten (NULL = not wanted); w NULL = 1.  The votes are summed without float
 * atomics (per-workgroup rows of `ws`, reduced in fixed order), so fractional weights give the
 * same bits on every run; ws / ws_bytes as for mxm_em_iter (mxm_workspace_bytes(R, H, 1)),
 * only needed when votes != NULL.
 */
int mxm_row_argmax_votes(const double *X, int64_t ldx, const double *w,
                         int64_t R, int32_t H, int32_t *best, double *votes,
                         void *ws, size_t ws_bytes, void *stream);

/*
 * Insertion order of the reference's vote table -- assemble.py:116-119 fills a dict row by row, and the contributors
 * come out in the order their haplogroup FIRST won a row:
 *   first[h] = the smallest r with best[r] == h, R if no row voted for h   (device int64[H])
 * so that only H values, not best[R], have to reach the host.  Entries of best outside [0, H) are ignored.
 */
int mxm_first_seen(const int32_t *best, int64_t R, int32_t H, int64_t *first, void *stream);

/*
 * Read -> contributor assignment -- assemble.py:284-334 (_find_best_n_for_read :267-281 inlined):
 *   v_c = X[r][c] - log_props[c] for the nC contributor columns cols[];
 *   assigned[r] = ordinal (0..nC-1) of the largest v_c if it beats the runner-up by at least
 *   log_min_fold, else -1 ("unassigned").  nC == 1 is the caller's trivial case.
 */
int mxm_assign_reads(const double *X, int64_t ldx, const double *log_props,
                     const int32_t *cols, int32_t nC, int64_t R, int32_t H,
                     double log_min_fold, int32_t *assigned, void *stream);

/*
 * Column subset for the refinement EM -- preprocess.py:247-251 (em_mat[:, indexes]):
 *   out[r][i] = M[r][cols[i]],  i < nC;  cols[] device int32, each in [0, H).
 */
int mxm_gather_columns(const double *M, int64_t ldm, int64_t R, int32_t H,
                       const int32_t *cols, int32_t nC, double *out, int64_t ldo,
                       void *stream);

/*
 * Fold of log-posterior blocks across runs / ranks -- em.py:156 and :161:
 *   acc[r][h] = logaddexp(... logaddexp(logaddexp(acc, in[0]), in[1]) ..., in[n_in-1]) + delta
 * in that fixed order, in log space (entries below exp(-745) keep their finite logs).
 * in_host[n_in] / ld_in_host[n_in] are HOST arrays of device pointers / leading dimensions,
 * n_in <= 8 per call; delta = -log(n_multi) on the last call, 0 otherwise.
 */
int mxm_fold_logaddexp(double *acc, int64_t lda, const double *const *in_host,
                       const int64_t *ld_in_host, int32_t n_in, int64_t R, int32_t H,
                       double delta, void *stream);

/*
 * HOST function (no device work): the read signatures 'pos:base,pos:base,...' that
 * preprocess.build_em_matrix receives (reference: preprocess.py:142-160, parsed per cell at
 * :151-160) -> the CSR observations mxm_build_em_matrix takes.
 *   text     all signatures back to back; signature r is text[off[r] .. off[r+1] - 1)
 *            (one separator byte follows each signature; its value is not looked at)
 *   site_of_pos[ref_len]  0-based position -> index into the sorted variant sites, -1 if none
 *   row_ptr[R+1], site[cap], obs[cap]  outputs; obs is the ASCII base, 0 for an observation that
 *            is not exactly one character (it can never equal a base)
 * Returns the number of observations written (>= 0), or -(r + 1) when signature r needs the
 * caller's slow path: empty signature, a position that is not plain decimal digits, not a
 * variant site or out of range, a stray ':' or more than `cap` observations.  The Python host
 * then re-parses with the reference's own expressions so that the exception raised
 * (ValueError / KeyError) is the reference's.
 */
int64_t mxm_encode_signatures(const char *text, const int64_t *off, int64_t R,
                              const int32_t *site_of_pos, int64_t ref_len, int64_t *row_ptr,
                              uint16_t *site, uint8_t *obs, int64_t cap);

/*
 * HOST functions (no device work): the alignment front end, batched -- what preprocess.process_reads
 * (preprocess.py:99-139), read_signature (:142-148), reduce_reads (:163-174) and the row order / weights / id lists
 * of build_em_input (:218-220, :225) produce together, from alignments held as COLUMNS instead of one Python object
 * per read and one interpreter step per aligned base:
 *   ref_start[i]   pysam reference_start (0-based);  mapq[i]  mapping_quality;  frag[i] in [0, n_frag): the
 *                  alignment's fragment (query_name; mates share one)
 *   cigar[cig_ptr[i] .. cig_ptr[i+1])   BAM encoding, length << 4 | op  (0 M, 1 I, 2 D, 3 N, 4 S, 5 H, 6 P, 7 =, 8 X)
 *   seq / qual [seq_ptr[i] .. seq_ptr[i+1])   query_sequence (ASCII) and query_qualities (phred, same offsets);
 *                  qual == NULL: no alignment has qualities; has_qual[i] == 0: this one has none (None in pysam)
 * An observation is made at every variant site under an M / = / X operation of an alignment with mapq >= min_mq
 * whose base quality is >= min_bq (or absent); the base is upper-cased; a (fragment, site) seen with two different
 * bases, or as 'N', is dropped (:126-138).  Fragments with equal observation lists share one row; rows come in
 * the order of Python's sorted() over the signature strings 'pos:base,pos:base,...' (pos = site_pos[site]).
 *   site_of_pos[ref_len]  position -> site index or -1;  site_pos[n_sites]  its inverse (ascending)
 *   n_threads     host threads to use (<= 0: the hardware's, at most 16)
 * mxm_aln_encode allocates the result (host memory owned by the library); sizes via mxm_aln_sizes_of, copies via
 * mxm_aln_fetch* (NULL = not wanted), released by mxm_aln_free.  Returns 0, -1 (bad arguments) or -4: an alignment
 * whose CIGAR runs past its sequence or holds an unknown operation -- the caller then takes the object-by-object
 * path, which raises what the reference raises.
 *   rows:       row_ptr[n_rows+1], site[nnz], obs[nnz]  (the CSR mxm_build_em_matrix* take), weights[n_rows]
 *               (fragments per signature), group_ptr[n_rows+1] / group_frag[n_grouped] (each row's fragments in the
 *               reference's list order), text / text_off[n_rows+1] (the signatures, '\n' after each)
 *   dropped[n_dropped]  fragments left with NO site (signature '': the reference dies on it in int(''), :156-160)
 *   fragments:  process_reads' own result in its dict order: frag_id[n_frag_seen], frag_ptr[n_frag_seen+1], site, obs
 */
typedef struct mxm_aln_columns {
    int64_t         n_aln, n_frag;
    const int64_t  *ref_start;
    const int32_t  *mapq;
    const int64_t  *frag;
    const int64_t  *cig_ptr;
    const uint32_t *cigar;
    const int64_t  *seq_ptr;
    const uint8_t  *seq;
    const uint8_t  *qual;
    const uint8_t  *has_qual;
} mxm_aln_columns;
typedef struct mxm_aln_sizes {
    int64_t n_rows, nnz, n_grouped, n_dropped, text_bytes, n_frag_seen, frag_nnz;
} mxm_aln_sizes;
typedef struct mxm_aln_enc mxm_aln_enc;
int  mxm_aln_encode(const mxm_aln_columns *cols, const int32_t *site_of_pos, int64_t ref_len, const int64_t *site_pos,
                    int32_t n_sites, int32_t min_mq, int32_t min_bq, int32_t n_threads, mxm_aln_enc **out);
int  mxm_aln_sizes_of(const mxm_aln_enc *enc, mxm_aln_sizes *sizes);
int  mxm_aln_fetch(const mxm_aln_enc *enc, int64_t *row_ptr, uint16_t *site, uint8_t *obs, int64_t *weights,
                   int64_t *group_ptr, int64_t *group_frag, int64_t *dropped, char *text, int64_t *text_off);
int  mxm_aln_fetch_fragments(const mxm_aln_enc *enc, int64_t *frag_id, int64_t *frag_ptr, uint16_t *site, uint8_t *obs);
void mxm_aln_free(mxm_aln_enc *enc);

/*
 * HOST functions (no device work): a BAM file straight into those columns -- what the reference gets from pysam
 * (bin/mixemt:139-147 opens the file; preprocess.py:209 iterates bamfile.fetch(); :118-132 read mapping_quality,
 * query_name, query_sequence, query_qualities and the aligned pairs of every AlignedSegment).  The BGZF members are
 * inflated on `n_threads` host threads (zlib), the records decoded into columns, equal read names given one fragment
 * index by first appearance.  Like fetch() without a region: every record placed on a reference (refID >= 0) in file
 * order, no flag filtering (the reference filters by mapping quality only, :119); records without a reference are
 * skipped and counted.
 *   mxm_bam_read      0, -1 (bad arguments), -6 (cannot open / read), -4 (not BGZF / BAM, truncated, or a CIGAR kept in
 *                     a CG tag: more than 65535 operations), -5 (out of memory)
 *   mxm_bam_columns   fills `cols` with pointers INTO the handle (valid until mxm_bam_free): mxm_aln_encode reads them
 *                     in place
 *   mxm_bam_fetch_names  names[names_bytes] / name_off[n_frag+1]: the fragments' read names back to back;
 *                     ref_id[n_aln], flag[n_aln]: the records' reference index and FLAG (not used by the encoder);
 *                     NULL = not wanted
 */
typedef struct mxm_bam_sizes {
    int64_t n_aln, n_frag, n_cigar, n_bases, names_bytes, n_ref, n_records_total, n_skipped_unplaced;
} mxm_bam_sizes;
typedef struct mxm_bam mxm_bam;
int  mxm_bam_read(const char *path, int32_t n_threads, mxm_bam **out);
int  mxm_bam_sizes_of(const mxm_bam *bam, mxm_bam_sizes *sizes);
int  mxm_bam_columns(const mxm_bam *bam, mxm_aln_columns *cols);
int  mxm_bam_fetch_names(const mxm_bam *bam, char *names, int64_t *name_off, int32_t *ref_id, uint16_t *flag);
void mxm_bam_free(mxm_bam *bam);

/*
 * One-shot exchange of the M-step sums between the ranks of a row-sharded loop -- OPTIONAL, instead of the all-reduce
 * between mxm_em_iter and mxm_m_finalize (SURVEY.md section 8 e: "each GPU writes its 43 KB to its peers, sum in fixed
 * rank order"; the reference has no counterpart: it is single-process).  Every rank owns one device buffer that all ranks
 * map (hipIpc); mxm_exchange_push writes the rank's sums into its slot of EVERY rank's buffer and then raises that
 * rank's flag there; mxm_exchange_pull waits (bounded: on a timeout colsum is NaN and state[0 .. nb).error = 2) until
 * all ranks' flags of this exchange are up in its own buffer and forms colsum[i] = sum over the ranks' slots in RANK
 * order -- the same bits on every rank.  Both only enqueue kernels on `stream` (the exchange count lives on the device:
 * a burst of iterations is capturable in a hipGraph).  Exercised with several processes on ONE GPU; over xGMI UNMEASURED
 * (csrc/exchange.hpp states the visibility rules it relies on).  mixemt_amd.dist.sharded_em_loop(exchange="oneshot").
 *   mxm_exchange_create   HOST, blocking: allocates this rank's buffer for exchanges of at most n_doubles doubles on the
 *                         current device and writes its IPC handle (mxm_exchange_handle_bytes() bytes) to handle_out
 *   mxm_exchange_connect  HOST: handles[world][handle_bytes] of ALL ranks in rank order (the caller all-gathers them)
 */
typedef struct mxm_exchange mxm_exchange;
size_t mxm_exchange_handle_bytes(void);
int  mxm_exchange_create(int32_t world, int32_t rank, int64_t n_doubles, mxm_exchange **out, void *handle_out);
int  mxm_exchange_connect(mxm_exchange *x, const void *handles);
int  mxm_exchange_push(mxm_exchange *x, const double *colsum, int64_t n, void *stream);
int  mxm_exchange_pull(mxm_exchange *x, double *colsum, int64_t n, mxm_em_state *state, int32_t nb, void *stream);
/* push + pull in ONE launch (in place: colsum in, the ranks' sum out) -- what the loop uses */
int  mxm_exchange_reduce(mxm_exchange *x, double *colsum, int64_t n, mxm_em_state *state, int32_t nb, void *stream);
int  mxm_exchange_info(const mxm_exchange *x, int32_t *fine_grained, int64_t *bytes);
void mxm_exchange_destroy(mxm_exchange *x);

#ifdef __cplusplus
}
#endif
#endif /* MIXEMT_HIP_H */
