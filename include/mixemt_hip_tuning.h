/*
 * mixemt_hip_tuning.h -- measurement hooks and kernel-shape knobs of libmixemt_hip.so.
 *
 * NOT part of the drop-in boundary (include/mixemt_hip.h): nothing here replaces a line of
 * the reference.  bench.py, tools/ and the parity tests use these to time the dominant kernel
 * without a profiler and to A/B kernel shapes inside one process.  The state they set is PROCESS-WIDE
 * but guarded: setters take a lock, and every entry point of mixemt_hip.h works on a snapshot taken when it
 * starts, so host threads (one per GPU) never see a half-changed set.  mxm_reset_tuning() restores every
 * default below (the test suite calls it after each test).  Results never depend on any of this beyond the
 * rounding of a different summation order.
 */
#ifndef MIXEMT_HIP_TUNING_H
#define MIXEMT_HIP_TUNING_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct mxm_em_state;             /* include/mixemt_hip.h */

/* Every knob below back to its default (and the timing / progress hooks off). */
int mxm_reset_tuning(void);

/*
 * Measurement hook (bench.py): when both handles are non-NULL, mxm_em_iter
 * records hipEvent_t `ev_start` / `ev_stop` on its stream immediately before /
 * after the streaming kernel (the dominant one), so its device time can be read
 * without a profiler.  Pass NULLs to switch it off.
 */
int mxm_set_timing_events(void *ev_start, void *ev_stop);

/*
 * The streaming kernel's template instance for a tile of nb restarts at width H, spelled the way rocprofv3 prints it
 * ("em_iter_wide_kernel<512, 6, 1, 3, 1>"); bench.py selects its committed counter files (profiles/) by this name.
 */
int mxm_describe_stream_kernel(int32_t H, int32_t nb, char *buf, size_t len);

/*
 * Diagnostic, not part of the reference boundary: a bare streaming read of `bytes` bytes
 * (16 B per lane, 8 loads in flight per lane, `wg_per_cu` workgroups of 256 per CU; blocked = 0:
 * grid-stride plain loads, 1: one contiguous block per workgroup, non-temporal loads, 2: the
 * streaming EM kernel's own pattern without its arithmetic -- 43 264-byte rows dealt over
 * workgroups of 512, per-row buffer descriptors, non-temporal loads, a ring of 3 rows; 3: em_iter_coded_kernel's
 * pattern -- records of 5408 code bytes read 4 B per lane + a 27-entry table read 8 B per lane, workgroups of 256:
 * the counter calibration for that access width, tools/pmc_summary.py) to measure
 * the practical HBM read ceiling on the device at hand (tools/stream_ceiling.py).
 */
int mxm_diag_stream_read(const void *src, size_t bytes, int32_t wg_per_cu, int32_t blocked,
                         void *sink, void *stream);

/*
 * Diagnostic: exactly the loads em_iter_coded_kernel issues over the records of `c` (code words, P tables, the wide
 * rows' 16-bit codes and tables; H in (5120, 6144]) and nothing else -- the bare-read time of a record buffer, and the
 * counter calibration for that kernel: FETCH_SIZE of this launch is what the counter makes of the bytes the EM
 * iteration has to read (tools/pmc_calibrate_coded.py, tools/pmc_summary.py).
 */
struct mxm_coded;
int mxm_diag_stream_coded(const struct mxm_coded *c, int32_t H, int32_t wg_per_cu, void *sink, void *stream);

/*
 * Diagnostic: exactly the loads of em_iter_quad_coded_kernel's quad pass over the quad dictionary of `c` (2048 code
 * bytes and the 32-byte table entries of every row of c->quad_rows) and nothing else: the counter calibration for that
 * kernel (tools/pmc_calibrate_coded.py --quads). -1 when `c` carries no quad dictionary.
 */
int mxm_diag_stream_quads(const struct mxm_coded *c, int32_t H, int32_t wg_per_cu, void *sink, void *stream);

/*
 * Progress hook for mxm_em_loop (the reference prints a dot every 10 iterations while it runs,
 * em.py:127-135): fn(state_host, B, user) is called on the calling host thread after every read-back
 * of the loop state, which then happens at least every `every` iterations.  NULL switches it off.
 * Host-side only; results do not depend on it.
 */
int mxm_set_progress_callback(void (*fn)(const struct mxm_em_state *state_host, int32_t B, void *user),
                              void *user, int32_t every);

/*
 * mxm_em_loop replays its chunk of iterations from a hipGraph when that pays
 * (launch-bound sizes): mode -1 = automatic (R*H*B < 6.4e7 cells), 0 = never,
 * 1 = always.  Results are identical either way.
 */
int mxm_set_loop_graph(int32_t mode);

/*
 * mxm_em_loop runs the whole loop of ONE restart (up to three on matrices of at most 1536 rows) on a small matrix (R * H <= 1e8 cells = 800 MB of fp64;
 * the kernels' own break-even is ~1.6e8; several restarts share the per-iteration kernels' passes, which
 * is faster from two restarts on) in ONE persistent launch (em_fused_loop_kernel: grid barriers instead of kernel
 * boundaries): mode -1 = automatic by size, 0 = never (per-iteration kernels), 1 = whenever the
 * shape allows (a launch that cannot run to its end -- grid not co-resident, grid barrier timed out -- is then an error, -3;
 * in automatic mode the call undoes that launch and goes on through the per-iteration kernels), 2 = as 1 but always with the rows split over the workgroups (matrices of up to 1536
 * rows normally take the transposed form, em_fused_cols_kernel: columns split, matrix in registers).  chunk > 0 splits the loop into launches of that many iterations per restart
 * (same bits: a resumed restart continues from its saved proportions); 0 = one launch.
 * Against the per-iteration kernels the results differ by rounding only (another summation
 * order; the linear proportions are carried as p T / tot instead of exp(ln p')).
 */
int mxm_set_loop_fused(int32_t mode, int32_t chunk);

/*
 * Workgroups of the one-launch loop over records (em_fused_coded_kernel; at most two per CU): 0 = chosen by size.
 * Fewer workgroups make the per-iteration exchange (one partial row of H doubles per workgroup) cheaper and the row
 * pass longer.  Results differ by the rounding of another summation order only.
 */
int mxm_set_fused_coded_grid(int32_t nwg);

/*
 * Workgroups of the leftover pass (byte-coded rows without quads, wide rows) inside the grid it shares with the quad pass
 * (em_iter_quad_coded_kernel; at most half the grid): 0 = by the rows' measured cost, so that both passes finish together.
 * Results differ by the rounding of another summation order only.
 */
int mxm_set_quad_left_grid(int32_t nwg);

/*
 * The marker build's launch for rows of 65 .. 128 observations (mxm_build_em_matrix_sparse / mxm_build_em_records; round 6):
 * 1 (default) = they are built by the 128-bit instance of the marker kernel, 0 = they go to the fallback list like every
 * row beyond 64 observations did before.  The same bits either way.
 */
int mxm_set_sparse_long_rows(int32_t on);
/* ... and the most marker entries (sum of the marker-list lengths of its sites) a row of that launch may have; rows beyond go
 * to the fallback list (0 = no limit).  The same bits either way. */
int mxm_set_sparse_long_entries(int32_t n);

/*
 * The quad dictionary's encoder (mxm_build_quads): 1 (default) = a wave per row (quad_encode_wave_kernel, round 6),
 * 0 = a workgroup per row (quad_encode_kernel).  The same record bytes for every row either way.
 */
int mxm_set_quad_encoder(int32_t kind);

/*
 * Restarts per pass over records beside a quad dictionary: 3 (default; em_iter_quad_batched_kernel takes full tiles of
 * three) or 1 (every restart its own pass, as before round 6).  Results differ by the rounding of another summation
 * order only.
 */
int mxm_set_coded_batch_tile(int32_t bt);

/*
 * Test hook: the next one-launch loops start with their abort flag already raised, i.e. behave as if a workgroup had
 * waited in vain at the first grid barrier (the situation a second process holding CUs creates).  mxm_em_loop must then
 * undo the launch and finish through the per-iteration kernels (mode -1), or return -3 (mode 1).
 */
int mxm_diag_fused_force_abort(int32_t on);

/*
 * Diagnostic: per-phase clock sums (100 MHz ticks) of the last one-launch loop that used workspace
 * `ws`, workgroup 0: [0] row pass, [1] barrier 1, [2] slice reduce, [3] barrier 2, [4] normalise + test,
 * [5] iterations.  Filled only by a library built with -DFUSED_STAMPS (a diagnostic build whose stamps
 * serialise the phases); zeros otherwise.  out_host[8].
 */
int mxm_diag_fused_stamps(const void *ws, unsigned long long *out_host);

/*
 * How many restarts at most share one pass over the matrix in mxm_em_iter (1..4,
 * default 4; B restarts take ceil(B / tile) passes with the restarts spread evenly:
 * 10 -> 4 + 3 + 3).  1 reproduces the unbatched schedule (B passes per iteration).
 * Results do not depend on it beyond rounding of the reduction order.
 */
int mxm_set_batch_tile(int32_t bt);

/*
 * Restart schedule of mxm_em_loop.  0: every restart in every iteration, static tiles over all B
 * (ceil(B / tile) passes until each tile's last member stops).  1: tiles over the restarts still
 * running, ceil(running / tile) passes per iteration.  2 (default): ONE full tile per iteration,
 * dealt round-robin over the running restarts chunk by chunk, so every pass over the matrix carries a
 * full tile and all restarts advance at the same rate.  Tiles name their restarts by index; nothing
 * moves in memory.  Results are the same in all modes: each restart counts its own iterations and is
 * frozen once it has stopped.
 */
int mxm_set_compact_restarts(int32_t mode);

/* Tuning knob: rows a workgroup of the streaming kernel handles at least (grid = min(cap, R / n)). */
int mxm_set_min_rows_per_wg(int32_t n);

/* Test knob: distinct non-zero masks a row of the marker build kernel may have before it goes to the
 * fallback list (negative = the kernel's own limit, the default); lowering it drives ordinary rows through the fallback path. */
int mxm_set_sparse_max_distinct(int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* MIXEMT_HIP_TUNING_H */
