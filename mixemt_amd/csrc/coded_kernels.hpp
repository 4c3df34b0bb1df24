// coded_kernels.hpp -- part of libmixemt_hip.so (gfx950); included by mixemt_hip.hip only.
// Row-dictionary storage of the linearised matrix and the EM iteration that streams it.
#ifndef MIXEMT_CODED_KERNELS_HPP
#define MIXEMT_CODED_KERNELS_HPP

// ------------------------------------------------------------------------------------------
// A row of build_em_matrix's output (preprocess.py:177-198) is a sum of per-site terms that take one
// of two values each, so the 5408 cells of a row hold only a few DISTINCT doubles (one per pattern of
// mismatching sites among the haplogroups: median 25, 98 % of synth-v1 rows at most 256, all at most ~900).  The
// dictionary form of row r of P = exp(M - rowmax) (mxm_linearize's output) is
//     record(r) = codes[ldc] (one code per column, pad columns 0)  ++  table[D_r] (doubles)  ++  mtable[D_r]
//     P[r][h]   = table[codes[h]]                       -- the SAME bits as the dense P
//     M[r][h]   = mtable[codes[h]]                      -- the log matrix itself (posterior / argmax / column gathers)
// with BYTE codes for D_r <= 256 (5.4 KB + 16 D bytes instead of 43 KB per row) and 16-BIT codes for
// 256 < D_r <= 1024 ("wide" records, round 4: 10.8 KB + 16 D; 2 % of the rows, which used to stay dense and cost a
// launch of their own every iteration).  Rows with more than 1024 distinct values (random matrices; none from
// build_em_matrix) stay dense (ndist[r] = 0) and go through em_iter_wide_kernel; the kernels' column partials are
// summed by one colreduce.  Every cell still gets its own two FMAs: nothing is skipped, only the bytes shrink.
// ------------------------------------------------------------------------------------------
#define ENC_THREADS 256
#define ENC_SLOTS 1024                 // hash slots per row (>= 4 x the 256 codes a row may use)
#define ENC_MAX_CODES 256              // distinct values of a row with byte codes
#define ENC_MAX_WIDE 1024              // ... with 16-bit codes
#define ENC_WIDE_SLOTS 4096            // hash slots of the wide encoder (>= 4 x ENC_MAX_WIDE)
#define ENC_EMPTY 0xFFFFFFFFFFFFFFFFull

// geometry of a record: bytes of its code array (the tables follow), and the code of column h
__host__ __device__ __forceinline__ int rec_code_bytes(int nd, int ldc) { return nd > ENC_MAX_CODES ? 2 * ldc : ldc; }
__device__ __forceinline__ int rec_code_at(const uint8_t *codes, int h, bool wide) {
    return wide ? (int)reinterpret_cast<const uint16_t *>(codes)[h] : (int)codes[h];
}

// K7  encode_rows: one workgroup per row.  Thread t holds the columns 4 (t + 256 k) .. + 3.
//   1. row maximum (mxm_linearize's shift);
//   2. the row's distinct bit patterns go into an LDS hash table (64-bit compare-and-swap, linear
//      probing); a thread skips the probe when its column repeats the previous one's value (85 % of a
//      row is one value), the lanes that are left mostly hold different keys and probe side by side;
//   3. occupied slots are numbered by a workgroup scan -> codes; more than MAXC -> the row gets no record here;
//   4. the record is bump-allocated (one atomic per row; the order of records is irrelevant, every row
//      carries its offset) and written: table[code] = exp(key - shift), one code per column.
// WIDE = false: the first pass over ALL rows (strided), byte codes, at most 256 values, 1024 slots.
// WIDE = true: the second pass over the rows the first left without a record (ndist[r] == 0, found by a scan of
// ndist in chunks of 256): 16-bit codes, at most 1024 values, 4096 slots; stats[1] (rows without a record) goes
// down by one for every row it codes.
template <int NCH, bool WIDE>
__device__ __forceinline__ void encode_one_row(int64_t r, const double *__restrict__ M, int64_t ldm, int H, int ldc,
                                               uint8_t *__restrict__ rec, int64_t rec_cap, int64_t *__restrict__ rec_off,
                                               int32_t *__restrict__ ndist, double *__restrict__ rowmax,
                                               unsigned long long *__restrict__ stats) {
    constexpr int NW = ENC_THREADS / 64;
    constexpr int SLOTS = WIDE ? ENC_WIDE_SLOTS : ENC_SLOTS;
    constexpr int MAXC = WIDE ? ENC_MAX_WIDE : ENC_MAX_CODES;
    constexpr int SPT = SLOTS / ENC_THREADS;                 // slots numbered per thread
    __shared__ unsigned long long s_key[SLOTS];
    __shared__ unsigned short s_code[SLOTS];
    __shared__ double s_red[NW];
    __shared__ int s_wcnt[NW];
    __shared__ int s_flag, s_n;
    __shared__ long long s_off;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));

    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(M + r * ldm), 0, H * 8, 0x00020000);
    double x[NCH][4];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int off = (t + k * ENC_THREADS) * 32;     // past the row: the descriptor returns 0 (masked below)
        const d2 a = __builtin_bit_cast(d2, (u4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 2));
        const d2 b = __builtin_bit_cast(d2, (u4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 16, 2));
        x[k][0] = a.x;
        x[k][1] = a.y;
        x[k][2] = b.x;
        x[k][3] = b.y;
    }
    if (H & 1) {
        // odd width (round 5): the row's last double is the first half of a 16-byte load whose second half lies past the
        // descriptor.  Whether such a load keeps its in-range half is not something to depend on: its owner fetches the
        // column again by itself.
        const double tail = M[r * ldm + (H - 1)];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (4 * (t + k * ENC_THREADS) == H - 1) x[k][0] = tail;
            if (4 * (t + k * ENC_THREADS) + 2 == H - 1) x[k][2] = tail;
        }
    }
#pragma unroll
    for (int q = 0; q < SPT; ++q) s_key[t + q * ENC_THREADS] = ENC_EMPTY;
    if (t == 0) {
        s_flag = 0;
        s_n = 0;
    }
    double m = -INFINITY;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * (t + k * ENC_THREADS) + e < H) m = fmax(m, x[k][e]);
    m = wave_max(m);
    if (lane == 0) s_red[wv] = m;
    __syncthreads();                                    // table cleared, wave maxima in place
    m = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
    const double shift = isfinite(m) ? m : 0.0;

    int slot[NCH][4];
    unsigned long long prev_key = ENC_EMPTY;
    int prev_slot = 0;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const bool give_up = *(volatile int *)&s_flag != 0;   // the row is already known to get no record (looked at once per four columns)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool valid = 4 * (t + k * ENC_THREADS) + e < H;
            const unsigned long long key = (unsigned long long)__double_as_longlong(x[k][e]);
            int sl = 0;
            bool need = valid;
            if (valid && key == ENC_EMPTY) {            // the one pattern the table cannot hold
                s_flag = 1;
                need = false;
            }
            if (need && key == prev_key) {              // same value as the thread's previous column
                sl = prev_slot;
                need = false;
            }
            if (give_up) need = false;
            // every lane that still needs a slot probes for itself: after the same-as-previous-column filter the
            // lanes of a wave mostly hold DIFFERENT keys, whose compare-and-swaps go through the LDS side by side
            // (a leader lane serving one distinct key per round -- the first form -- took 27.6 ms at 10^6 rows, this 16.8)
            if (need) {
                unsigned int h = (unsigned int)((key ^ (key >> 29)) * 0x9E3779B97F4A7C15ull >> 40) & (SLOTS - 1);
                for (int probes = 0;; ++probes) {
                    const unsigned long long old = atomicCAS(&s_key[h], ENC_EMPTY, key);
                    if (old == ENC_EMPTY) {
                        if (atomicAdd(&s_n, 1) >= MAXC) s_flag = 1;
                        break;
                    }
                    if (old == key) break;
                    h = (h + 1) & (SLOTS - 1);
                    if (probes >= SLOTS) {
                        s_flag = 1;
                        break;
                    }
                }
                sl = (int)h;
            }
            slot[k][e] = sl;
            if (valid) {
                prev_key = key;
                prev_slot = sl;
            }
        }
    }
    __syncthreads();                                    // every key is in the table
    const bool dense = (s_flag != 0);

    // number the occupied slots: thread t scans slots SPT t .. SPT t + SPT - 1
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < SPT; ++j) cnt += (s_key[SPT * t + j] != ENC_EMPTY) ? 1 : 0;
    const int incl = wave_inclusive_scan_i32(cnt);      // DPP: no ds_bpermute round trips on the LDS pipe
    if (lane == 63) s_wcnt[wv] = incl;
    __syncthreads();                                    // wave totals in place; s_flag read by everyone
    int base = 0, D = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        if (q < wv) base += s_wcnt[q];
        D += s_wcnt[q];
    }
    const int cbytes = WIDE ? 2 * ldc : ldc;
    const int64_t bytes = (int64_t)cbytes + 16 * (int64_t)D;        // codes ++ table of P ++ table of the log values
    // the wide pass leaves a row that fits byte codes alone (the first pass gave it up for another reason -- it cannot
    // happen today, and must not produce a 16-bit record that readers would take for a byte one)
    const bool coded = !dense && D <= MAXC && (!WIDE || D > ENC_MAX_CODES);
    if (t == 0) {
        long long off = -1;
        if (coded) {
            off = (long long)atomicAdd(&stats[0], (unsigned long long)bytes);
            if (off + bytes > rec_cap) off = -1;         // cannot happen with mxm_coded_bytes(R, H)
        }
        s_off = off;
    }
    int code = base + incl - cnt;
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
        if (s_key[SPT * t + j] != ENC_EMPTY) s_code[SPT * t + j] = (unsigned short)code++;
    }
    __syncthreads();                                    // codes of the slots and the record offset in place
    const long long off = s_off;
    if (off >= 0) {
        double *tbl = reinterpret_cast<double *>(rec + off + cbytes);
        code = base + incl - cnt;
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const unsigned long long kk = s_key[SPT * t + j];
            if (kk != ENC_EMPTY) {
                const double mval = __longlong_as_double((long long)kk);
                tbl[code] = exp(mval - shift);
                tbl[D + code] = mval;                   // the log value itself: posterior / argmax passes read it
                ++code;
            }
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c0 = 4 * (t + k * ENC_THREADS);
            if (c0 < ldc) {
                unsigned int cd[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) cd[e] = (c0 + e < H) ? (unsigned int)s_code[slot[k][e]] : 0u;
                if constexpr (WIDE) {
                    u2 word;
                    word.x = cd[0] | (cd[1] << 16);
                    word.y = cd[2] | (cd[3] << 16);
                    reinterpret_cast<u2 *>(rec + off)[t + k * ENC_THREADS] = word;
                } else {
                    reinterpret_cast<unsigned int *>(rec + off)[t + k * ENC_THREADS] =
                        cd[0] | (cd[1] << 8) | (cd[2] << 16) | (cd[3] << 24);
                }
            }
        }
    }
    if (t == 0) {
        if constexpr (WIDE) {
            if (off >= 0) {                              // rowmax[r] is the first pass's (same shift)
                ndist[r] = D;
                rec_off[r] = off;
                atomicAdd(&stats[1], ~0ull);             // one row less without a record
            }
        } else {
            rowmax[r] = shift;
            ndist[r] = (off >= 0) ? D : 0;
            rec_off[r] = (off >= 0) ? off : 0;
            if (off < 0) atomicAdd(&stats[1], 1ull);
        }
    }
    __syncthreads();                                    // the next row clears the table
}

template <int NCH>
__global__ __launch_bounds__(ENC_THREADS) void encode_rows_kernel(
    const double *__restrict__ M, int64_t ldm, int64_t R, int H, int ldc, uint8_t *__restrict__ rec, int64_t rec_cap,
    int64_t *__restrict__ rec_off, int32_t *__restrict__ ndist, double *__restrict__ rowmax,
    unsigned long long *__restrict__ stats) {
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x)
        encode_one_row<NCH, false>(r, M, ldm, H, ldc, rec, rec_cap, rec_off, ndist, rowmax, stats);
}

template <int NCH>
__global__ __launch_bounds__(ENC_THREADS) void encode_wide_rows_kernel(
    const double *__restrict__ M, int64_t ldm, int64_t R, int H, int ldc, uint8_t *__restrict__ rec, int64_t rec_cap,
    int64_t *__restrict__ rec_off, int32_t *__restrict__ ndist, double *__restrict__ rowmax,
    unsigned long long *__restrict__ stats) {
    constexpr int CH = 32;                               // rows per chunk (2 % of them are candidates: ~0.6 per chunk)
    __shared__ int s_list[CH];
    __shared__ int s_nlist;
    const int t = threadIdx.x;
    const int64_t nchunk = (R + CH - 1) / CH;
    for (int64_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
        if (t == 0) s_nlist = 0;
        __syncthreads();
        const int64_t r = c * CH + t;
        if (t < CH && r < R && ndist[r] == 0) s_list[atomicAdd(&s_nlist, 1)] = t;
        __syncthreads();
        const int n = s_nlist;                           // uniform
        for (int i = 0; i < n; ++i)
            encode_one_row<NCH, true>(c * CH + s_list[i], M, ldm, H, ldc, rec, rec_cap, rec_off, ndist, rowmax, stats);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// K3c  em_iter_coded: em_iter_wide_kernel's step over dictionary rows.
//
// Same decomposition (rows dealt round-robin, thread t owns fixed columns, the row waits in VGPRs
// between its dot product and the accumulation, column partials in registers for the whole launch),
// but a row arrives as 4 code bytes per thread and chunk plus its table (<= 2 KB), which the first 256
// threads park in LDS one row ahead (double buffered; the same barrier that publishes the wave sums
// publishes it).  Per cell: one shift-by-3 of a code byte, one ds_read_b64 (issued one row ahead), two fp64 FMAs.
// 8x fewer HBM bytes per row make the per-row chain (reduce -> barrier -> divide) the bound, so two or
// three workgroups share a CU and overlap each other's chains.
// ------------------------------------------------------------------------------------------
// Shapes (A/B with mxm_set_coded_shape): threads per workgroup, rows in flight (NBUF - 1 rows of codes
// are on their way while one is processed; the row's table follows one step behind), workgroups per CU.
// Measured at 10^6 x 5408 (profiles/r02/coded_shapes.txt; iteration = this kernel + 0.14 ms dense rest + reduce):
//   256 threads x 2 per CU: 1.75-1.77 ms with 3, 4 or 6 rows in flight      512 x 2 per CU: 1.96-2.20 ms
//   256 x 3 per CU without the kept row values (second LDS lookup in the accumulation): 2.37-2.46 ms;
//   256 x 3 per CU with them (168 VGPRs, 47 spilled into the row loop): 4.2 ms;
//   384 threads x 2 per CU (six waves per row, 152 VGPRs, 3 waves per SIMD without spills): 3.2-3.4 ms
// i.e. neither more rows in flight nor more waves help: the kernel is bound by its VALU + LDS instruction
// streams (per wave and row ~70 instructions of reduction / exchange / division beside 3.25 per cell; SQ
// counters: 50 % of the wave cycles issuing at 2 waves per SIMD, profiles/r02/coded_pmc_sq_summary.txt),
// which is why fewer, fatter waves per row win over more, thinner ones.  Two attempts on the fixed part
// (profiles/r02/coded_reduce_division_variants.txt): the wave sum as an xor butterfly on the LDS crossbar (6 VALU
// ops instead of 22, but six dependent ds_bpermute round trips) 2.07 ms; reciprocal + two Newton steps instead of
// the IEEE division 1.73 ms (-2 %, not taken: it would change the quotient's last bit against the dense kernel).
// Round 3 (profiles/r03/coded_variants.txt; same column sums bit for bit): timing builds with one stage knocked out
// each put 0.11 ms on the DPP ladder, 0.09 on the barrier, 0.16 on what follows it (partials + division), 0.01 on the
// LDS lookups once they are issued a step ahead, and 1.17 ms on everything else -- the step is a latency chain, not
// an instruction count.  Kept: lookups one step ahead, interleaved with the accumulation FMAs (-4 %); the wave sums
// after the barrier as two broadcast 16-byte LDS reads instead of a read and two DPP steps; s_setprio 1 from the
// ladder to the coefficient, so the chain does not queue behind the other workgroup's bulk FMAs (-6 %); the next
// step's metadata read from LDS ahead of the lookups (-3 %): 1.63 -> 1.45-1.48 ms.  Lost: the wave sum on the matrix pipe (two v_mfma_f64_16x16x4 with B = 1 + three adds instead of the
// 22-instruction ladder: +8 %, the MFMA's result latency is longer than the ladder it replaces), a branch-free
// quotient, 5 or 6 rows in flight, 16-byte code loads (a thread owning 24 consecutive bytes), other cache policies
// on the record loads: all within noise (round 5 adds: ONE s_waitcnt for the row's 24 lookups instead of the compiler's
// countdown in front of the dot product's FMAs -- 11 instructions fewer per step, 1.4435 against 1.4500 ms:
// profiles/r05/ab_onewait.txt).  With the chain knocked out the kernel takes 1.19 ms, of which the record
// loads are 0.40 (0.79 without them) and the 48 FMAs per thread 0.08: what is left is neither arithmetic nor HBM
// bandwidth (5 of 8 TB/s) but two waves per SIMD overlapping their loads, LDS traffic and issue imperfectly.
// The row pass is a device function so that the per-iteration kernel (em_iter_coded_kernel) and the one-launch loop
// (fused_coded_kernels.hpp) run the very same code: acc[k][e] += (w_r / Z_r) P[r][c] over this workgroup's dealt rows,
// Z_r = sum_c p[c] P[r][c], thread t owning the columns c = 4 (t + THREADS k) + e.
//   NT        cache policy of the record loads: true = non-temporal (a pass over records that do not fit the caches),
//             false = default (the one-launch loop re-reads the same rows every iteration: L2 / Infinity Cache)
//   RESIDENT  the one-launch loop on a matrix whose per-workgroup row count fits the LDS metadata blocks
//             (nq <= THREADS, wide rows per workgroup <= THREADS): the metadata is fetched by the FIRST pass of a
//             launch only (meta_ready says whether it is there) instead of once per pass -- two dependent gathers
//             (~2 us) that a 15 us iteration would otherwise pay every time.
// WIDE rows (16-bit codes, 256 < D <= 1024; `wide_rows` lists them) are skipped by the main loop (weight 0, empty
// table) and taken by a second loop over the list, dealt round-robin like the rows: codes as 8 bytes per thread and
// chunk, the table (<= 8 KB) through LDS, one row in flight ahead of the one being reduced.  They are 2 % of the rows
// of a build_em_matrix matrix; as dense rows they cost a kernel launch of their own per iteration (0.147 ms at 10^6
// rows, 9 % of the step).
// CHECK (the per-iteration kernel): the sums are only right if `wide_rows` lists EVERY row with more than 256
// values -- the main loop skips those.  The pass therefore counts the wide rows its main loop meets (fetch_meta has
// their ndist in hand: one LDS atomic per wide row, no register across the row loop) and looks at every list entry it
// takes (in range, ascending, really wide; a faulty entry is neutralised instead of dereferenced) and leaves
// {wide rows met, list fault} in chk_result[0..1] (uniform; the kernel stores it beside its partial row, at an address
// formed from values that are live at the end anyway -- a pointer parameter carried across the row loop cost two
// registers the kernel does not have); the column reduce compares the total with n_wide and poisons
// the sums (NaN) + raises mxm_em_state.error when the list is not exactly the set of wide rows.  The one-launch loop is
// instantiated without it (its register budget; mxm_em_loop_coded validates the list once on entry instead).
template <int THREADS, int NCH, int NBUF, bool NT, bool RESIDENT, bool WIDE_PREFETCH, bool CHECK = false>
__device__ __forceinline__ void coded_row_pass(const uint8_t *__restrict__ rec, const int64_t *__restrict__ rec_off,
                                               const int32_t *__restrict__ ndist, int ldc, const double *__restrict__ w,
                                               const int64_t *__restrict__ wide_rows, int64_t n_wide, int64_t R,
                                               const double (&p)[NCH][4], double (&acc)[NCH][4], bool &meta_ready,
                                               int *chk_result = nullptr, const int64_t *__restrict__ row_list = nullptr,
                                               int64_t n_list = 0, const int32_t *__restrict__ nquad = nullptr, int vbid = -1,
                                               int vgrid = 0) {
    static_assert(NBUF >= 3, "codes NBUF - 1 rows ahead, tables NBUF - 2");
    constexpr int NW = THREADS / 64;
    constexpr int AUX = NT ? 2 : 0;
    __shared__ double s_tbl[NBUF][ENC_MAX_CODES];
    __shared__ __attribute__((aligned(16))) double red[NBUF][NW];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int nword = ldc >> 2;
    // row_list (the per-iteration kernel beside a quad dictionary, quad_kernels.hpp): the main loop takes ONLY the listed
    // rows -- the byte-coded rows that have no quads, ascending -- instead of every row
    // (vbid / vgrid: the workgroup's place among the workgroups that do THIS pass, where a kernel's grid is shared with
    // another pass -- em_iter_quad_coded_kernel; otherwise the block index and the grid's size)
    const int64_t bid = vbid >= 0 ? (int64_t)vbid : (int64_t)blockIdx.x, grid = vbid >= 0 ? (int64_t)vgrid : (int64_t)gridDim.x;
    const row_deal deal(row_list != nullptr ? n_list : R, bid, grid);
    const int voff = t * 4;
    int last_w = t + (NCH - 1) * THREADS;               // words past the row are clamped: their columns have p = 0
    if (last_w > nword - 1) last_w = nword - 1;
    const int voff_last = last_w * 4;
    const bool tbl_thread = (THREADS <= ENC_MAX_CODES) || t < ENC_MAX_CODES;
    const int tslot = t & (ENC_MAX_CODES - 1);
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    __shared__ int s_chk[2];                              // CHECK: {wide rows the main loop met, a list entry was faulty}
    if constexpr (CHECK) {
        if (t == 0) s_chk[0] = s_chk[1] = 0;
        __syncthreads();
    }
    // entry e of the wide rows' list; CHECK: -1 (and the fault noted) unless it is in range and above its predecessor
    auto list_row = [&](int64_t e) -> int64_t {
        int64_t r = wide_rows[e];
        if constexpr (CHECK) {
            if (r < 0 || r >= R || (e > 0 && wide_rows[e - 1] >= r)) {
                s_chk[1] = 1;
                r = -1;
            }
        }
        return r;
    };
    // table entries of a listed row; CHECK: a row that is not wide at all must not be read as 16-bit codes
    auto list_nd = [&](int64_t r) -> int {
        if (r < 0) return 0;
        int nd = ndist[r];
        if constexpr (CHECK) {
            if (nd <= ENC_MAX_CODES) {
                s_chk[1] = 1;
                nd = 0;
            }
        }
        return nd;
    };

    // Per-step metadata {record offset, table entries, weight} of THREADS steps at a time in LDS (double
    // buffered): fetched by one thread per step, read back at a uniform address.  As scalar loads inside the
    // step they were three dependent L2 round trips per row -- the whole step time.
    __shared__ long long s_off[2][THREADS];
    __shared__ double s_wr[2][THREADS];
    __shared__ int s_nd[2][THREADS];
    auto fetch_meta = [&](int half, int64_t q0) {       // steps q0 .. q0 + THREADS - 1, thread t takes step q0 + t
        const int64_t q = q0 + t;
        int64_t r = deal.row(q);
        bool struck = false;
        if (row_list != nullptr) {                       // (uniform) entry r of the list; CHECK: in range, ascending
            const int64_t e = r;
            r = row_list[e];
            if constexpr (CHECK) {
                if (r < 0 || r >= R || (e > 0 && row_list[e - 1] >= r)) {
                    s_chk[1] = 1;
                    r = 0;                               // (some row; the entry gets an empty table and weight 0)
                    struck = true;
                }
            }
        }
        int nd = ndist[r];
        if (nd > ENC_MAX_CODES) {                        // a wide row: the second loop's (empty table here, weight 0)
            if constexpr (CHECK) {
                if (row_list != nullptr) s_chk[1] = 1;   // a listed row must be byte-coded
                else if (deal.live(q)) atomicAdd(&s_chk[0], 1);
            }
            nd = 0;
        }
        if constexpr (CHECK) {                           // a listed row is byte-coded and has no quads (or the quad pass
            if (row_list != nullptr && !struck && (nd <= 0 || nquad[r] != 0)) {     // takes it too)
                s_chk[1] = 1;
                struck = true;
            }
        }
        if (struck) nd = 0;
        s_off[half][t] = rec_off[r];
        s_nd[half][t] = nd;
        s_wr[half][t] = (deal.live(q) && nd > 0) ? (w != nullptr ? w[r] : 1.0) : 0.0;   // dense rows are not ours
    };

    unsigned int cw[NBUF][NCH];
    double tring[NBUF];
    // The metadata a step needs -- the record of the row whose loads it issues, the weight of the row it reduces --
    // is read from LDS during the step BEFORE, after that step's barrier and ahead of the table lookups: LDS returns in
    // order, so read at the top of the step it waited for the 24 lookups queued in front of it, with the wave's global
    // loads and dot product stuck behind that wait (-3 %).
    int pre_off_lo, pre_off_hi, pre_nd;                  // uniform (SGPRs)
    double pre_wr;
    auto read_meta = [&](int64_t q_load, int64_t q_weight) {
        const int half = (int)((q_load / THREADS) & 1), idx = (int)(q_load % THREADS);
        const long long off = s_off[half][idx];
        pre_nd = __builtin_amdgcn_readfirstlane(s_nd[half][idx]);
        pre_off_hi = __builtin_amdgcn_readfirstlane((int)(off >> 32));
        pre_off_lo = __builtin_amdgcn_readfirstlane((int)off);
        pre_wr = s_wr[(q_weight / THREADS) & 1][q_weight % THREADS];
    };
    // a row's record: codes into cws, entry t of its table into tbl_entry (lanes past the table read 0)
    auto load_row = [&](unsigned int(&cws)[NCH], double &tbl_entry) {
        const int nd = pre_nd;
        const uint8_t *base = rec + (((long long)pre_off_hi << 32) | (unsigned int)pre_off_lo);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, ldc, 0x00020000);
#pragma unroll
        for (int k = 0; k < NCH - 1; ++k) cws[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, k * THREADS * 4, AUX);
        cws[NCH - 1] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff_last, 0, AUX);
        if (tbl_thread) {
            const auto rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base + ldc), 0, nd * 8, 0x00020000);
            const u2v v = __builtin_amdgcn_raw_buffer_load_b64(rt, tslot * 8, 0, AUX);
            tbl_entry = __hiloint2double((int)v.y, (int)v.x);
        }
    };

    // byte e of a code word x 8 = the byte offset of its table entry.  One SDWA shift per cell; written out because
    // the compiler takes byte 0 as shift + and.
    const unsigned int three = 3;
    auto entry_off = [&](unsigned int word, auto E) -> unsigned int {
        constexpr int e = decltype(E)::value;
        unsigned int r;
        if constexpr (e == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(three), "v"(word));
        else if constexpr (e == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(three), "v"(word));
        else if constexpr (e == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(three), "v"(word));
        else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(three), "v"(word));
        return r;
    };
    auto lookup = [&](const char *tb, unsigned int word, auto E) -> double {
        return *reinterpret_cast<const double *>(tb + entry_off(word, E));
    };
    using E0 = std::integral_constant<int, 0>;
    using E1 = std::integral_constant<int, 1>;
    using E2 = std::integral_constant<int, 2>;
    using E3 = std::integral_constant<int, 3>;

    // v = the row's values, looked up ONE STEP AHEAD: row q + 1's lookups are issued between the accumulation FMAs
    // of row q (each into the register that FMA just released), so their LDS latency runs under those FMAs instead
    // of in front of the next dot product.
    double v[NCH][4];
    auto lookup_row = [&](const char *tb, const unsigned int(&cws)[NCH]) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            v[k][0] = lookup(tb, cws[k], E0{});
            v[k][1] = lookup(tb, cws[k], E1{});
            v[k][2] = lookup(tb, cws[k], E2{});
            v[k][3] = lookup(tb, cws[k], E3{});
        }
    };

    auto step = [&](auto J, int64_t q) {
        constexpr int j = decltype(J)::value;
        constexpr int jn = (j + 1) % NBUF, jl = (j + NBUF - 1) % NBUF;
        if constexpr (!RESIDENT) {
            if ((q % THREADS) == 0) fetch_meta((int)((q / THREADS + 1) & 1), q + THREADS);   // the block after this one
        }
        load_row(cw[jl], tring[jl]);                     // row q + NBUF - 1; slot jl held row q - 1: consumed
        const double wr = pre_wr;
        double s4[4] = {0.0, 0.0, 0.0, 0.0};             // four independent chains (a dependent fp64 FMA stalls its wave)
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) s4[e] = fma(v[k][e], p[k][e], s4[e]);
        double s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        // From here to the row's coefficient the wave runs a chain of dependent instructions (DPP ladder, LDS exchange,
        // barrier, division) that the whole workgroup waits for; the other workgroup's wave on this SIMD is mostly in
        // its FMA / lookup bulk.  Issue priority for the chain: 1.58 -> 1.47 ms (profiles/r03/coded_variants.txt).
        __builtin_amdgcn_s_setprio(1);
        s = wave_sum_lane63(s);
        if (lane == 63) red[j][wv] = s;
        if (tbl_thread) s_tbl[jn][tslot] = tring[jn];    // row q + 1's table, published by the same barrier
        __syncthreads();
        read_meta(q + NBUF, q + 1);                      // for the next step; returns under the division
        // the four wave sums at a uniform address (two broadcast 16-byte reads) and two additions, in the order of the
        // DPP form it replaces, (r0 + r1) + (r2 + r3): no cross-lane step after the barrier
        static_assert(NW == 4, "four wave sums");
        typedef double d2v __attribute__((ext_vector_type(2)));
        const d2v ra = *reinterpret_cast<const d2v *>(&red[j][0]), rb = *reinterpret_cast<const d2v *>(&red[j][2]);
        const double cf = readlane_f64(weight_over_norm(wr, (ra.x + ra.y) + (rb.x + rb.y)), 0);
        __builtin_amdgcn_s_setprio(0);
        const char *tbn = reinterpret_cast<const char *>(&s_tbl[jn][0]);
        auto upd = [&](int k, auto E) {
            constexpr int e = decltype(E)::value;
            acc[k][e] = fma(cf, v[k][e], acc[k][e]);
            // pin the update here: left alone, the compiler sinks all NBUF steps' updates to the end of
            // the unrolled loop body and keeps (spills) every step's row values until then
            asm volatile("" : "+v"(acc[k][e]));
            v[k][e] = lookup(tbn, cw[jn][k], E);
        };
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            upd(k, E0{});
            upd(k, E1{});
            upd(k, E2{});
            upd(k, E3{});
        }
    };

    // The wide rows' metadata (second loop below) is asked for HERE, ahead of the main loop, and parked in registers:
    // two dependent gathers (list -> record offset) that cost ~3 us when they were issued after the main loop --
    // 27 us of a 238 us pass at 125 000 rows, more than the dense launch they replace.
    const int64_t nq_w = (n_wide > bid) ? (n_wide - bid + grid - 1) / grid : 0;
    // (WIDE_PREFETCH: the per-iteration kernel only -- the one-launch loop has no five registers to spare across its
    // row loop, and its metadata is mostly resident anyway)
    const bool wide_fetch = WIDE_PREFETCH && nq_w > 0 && (!RESIDENT || !meta_ready);
    int64_t w_row = -1;
    if (wide_fetch && t < nq_w) w_row = list_row(bid + (int64_t)t * grid);
    long long w_off = 0;
    int w_nd = CHECK ? -1 : 0;
    double w_wr = 0.0;

    if (deal.nq > 0) {                                   // (a one-launch grid may be larger than a tiny matrix)
        if (!RESIDENT || !meta_ready) {
            __syncthreads();                             // the blocks may still be read by a slower wave of the pass before
            fetch_meta(0, 0);
            if constexpr (RESIDENT) fetch_meta(1, THREADS);  // steps past the last row read it too (clamped rows)
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NBUF - 1; ++j) {
            read_meta(j, 0);
            load_row(cw[j], tring[j]);
        }
        if (tbl_thread) s_tbl[0][tslot] = tring[0];
        __syncthreads();
        read_meta(NBUF - 1, 0);
        lookup_row(reinterpret_cast<const char *>(&s_tbl[0][0]), cw[0]);
        if (w_row >= 0) {                                // second level of the wide rows' metadata: in flight under the main loop
            w_off = rec_off[w_row];                      // (CHECK looks at w_nd where it is consumed, not here: a use
            w_nd = ndist[w_row];                         //  would wait for the loads instead of leaving them in flight)
            w_wr = (w != nullptr) ? w[w_row] : 1.0;
        }
        for (int64_t q = 0; q < deal.nq; q += NBUF) {
            step(std::integral_constant<int, 0>{}, q);
            step(std::integral_constant<int, 1>{}, q + 1);
            step(std::integral_constant<int, 2>{}, q + 2);
            if constexpr (NBUF > 3) step(std::integral_constant<int, 3>{}, q + 3);
            if constexpr (NBUF > 4) step(std::integral_constant<int, 4>{}, q + 4);
            if constexpr (NBUF > 5) step(std::integral_constant<int, 5>{}, q + 5);
            static_assert(NBUF <= 6, "unrolled by hand");
        }
    }

    else if (w_row >= 0) {                               // (no byte-coded row here at all)
        w_off = rec_off[w_row];
        w_nd = ndist[w_row];
        w_wr = (w != nullptr) ? w[w_row] : 1.0;
    }

    // ---- the wide rows of this workgroup: wide_rows[bid + i * grid] ----------------------------------------
    if (nq_w > 0) {                                      // workgroup uniform
        __shared__ double s_wide[ENC_MAX_WIDE];
        __shared__ long long s_woff[THREADS];
        __shared__ double s_wwr[THREADS];
        __shared__ int s_wnd[THREADS];
        constexpr int TPT = ENC_MAX_WIDE / THREADS;      // table entries per thread
        const int voff8 = t * 8;
        const int voff8_last = last_w * 8;
        for (int64_t q0 = 0; q0 < nq_w; q0 += THREADS) {
            if (q0 == 0 && wide_fetch) {
                __syncthreads();
                if constexpr (CHECK) {                   // a listed row that is not wide must not be read as 16-bit codes
                    if (w_nd >= 0 && w_nd <= ENC_MAX_CODES) s_chk[1] = 1;      // (-1: no entry, or one list_row struck)
                    if (w_nd <= ENC_MAX_CODES) {
                        w_nd = 0;
                        w_wr = 0.0;
                    }
                }
                s_woff[t] = w_off;                       // asked for before the main loop
                s_wnd[t] = w_nd;
                s_wwr[t] = w_wr;
            } else if (q0 > 0 || !RESIDENT || !meta_ready || nq_w > THREADS) {
                __syncthreads();                         // the previous batch's entries have been read
                const int64_t q = q0 + t;
                if (q < nq_w) {
                    const int64_t r = list_row(bid + q * grid);
                    const int nd = list_nd(r);
                    s_woff[t] = (r >= 0) ? rec_off[r] : 0;
                    s_wnd[t] = nd;
                    s_wwr[t] = (nd > 0) ? ((w != nullptr) ? w[r] : 1.0) : 0.0;
                }
            }
            __syncthreads();
            const int n_here = (int)((nq_w - q0) < THREADS ? (nq_w - q0) : THREADS);
            // THREE rows in flight (codes: four 16-bit codes per chunk; this thread's entries of the table): with one,
            // a row took the memory latency -- 2.4 us against 0.7 us for a byte-coded row (tools/time_coded_parts.py)
            u2v cwa[NCH], cwb[NCH], cwc[NCH];
            double tna[TPT], tnb[TPT], tnc[TPT];
            auto fetch_wide = [&](int i, u2v(&cwn)[NCH], double(&tn)[TPT]) {
                const long long off = s_woff[i];
                const int nd = __builtin_amdgcn_readfirstlane(s_wnd[i]);
                const uint8_t *base = rec + (((long long)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) |
                                             (unsigned int)__builtin_amdgcn_readfirstlane((int)off));
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, 2 * ldc, 0x00020000);
#pragma unroll
                for (int k = 0; k < NCH - 1; ++k) cwn[k] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff8, k * THREADS * 8, AUX);
                cwn[NCH - 1] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff8_last, 0, AUX);
                const auto rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base + 2 * ldc), 0, nd * 8, 0x00020000);
#pragma unroll
                for (int j = 0; j < TPT; ++j) {
                    const u2v x = __builtin_amdgcn_raw_buffer_load_b64(rt, (t + j * THREADS) * 8, 0, AUX);
                    tn[j] = __hiloint2double((int)x.y, (int)x.x);
                }
            };
            auto process_wide = [&](int i, const u2v(&cwn)[NCH], const double(&tn)[TPT]) {
                const double wr = s_wwr[i];
#pragma unroll
                for (int j = 0; j < TPT; ++j) s_wide[t + j * THREADS] = tn[j];
                __syncthreads();                         // the table is in LDS (and red[0] of the row before is read)
                const char *tb = reinterpret_cast<const char *>(&s_wide[0]);
                double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    v[k][0] = *reinterpret_cast<const double *>(tb + ((cwn[k].x & 0xffffu) << 3));
                    v[k][1] = *reinterpret_cast<const double *>(tb + ((cwn[k].x >> 16) << 3));
                    v[k][2] = *reinterpret_cast<const double *>(tb + ((cwn[k].y & 0xffffu) << 3));
                    v[k][3] = *reinterpret_cast<const double *>(tb + ((cwn[k].y >> 16) << 3));
                }
#pragma unroll
                for (int k = 0; k < NCH; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s4[e] = fma(v[k][e], p[k][e], s4[e]);
                double s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
                s = wave_sum_lane63(s);
                if (lane == 63) red[0][wv] = s;
                __syncthreads();                         // wave sums in place; every lookup of s_wide is done
                static_assert(NW == 4, "four wave sums");
                const double cf = readlane_f64(weight_over_norm(wr, (red[0][0] + red[0][1]) + (red[0][2] + red[0][3])), 0);
#pragma unroll
                for (int k = 0; k < NCH; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[k][e] = fma(cf, v[k][e], acc[k][e]);
            };
            fetch_wide(0, cwa, tna);
            if (n_here > 1) fetch_wide(1, cwb, tnb);
            for (int i = 0; i < n_here; i += 3) {        // every branch below is workgroup uniform
                if (i + 2 < n_here) fetch_wide(i + 2, cwc, tnc);
                process_wide(i, cwa, tna);
                if (i + 1 >= n_here) break;
                if (i + 3 < n_here) fetch_wide(i + 3, cwa, tna);
                process_wide(i + 1, cwb, tnb);
                if (i + 2 >= n_here) break;
                if (i + 4 < n_here) fetch_wide(i + 4, cwb, tnb);
                process_wide(i + 2, cwc, tnc);
            }
        }
    }
    meta_ready = true;
    if constexpr (CHECK) {
        __syncthreads();                                 // every fault and every count is in
        chk_result[0] = s_chk[0];
        chk_result[1] = s_chk[1];
    }
}

template <int THREADS, int NCH, int NBUF, int MINWG>
__global__ __launch_bounds__(THREADS, MINWG *THREADS / 256) void em_iter_coded_kernel(
    const uint8_t *__restrict__ rec, const int64_t *__restrict__ rec_off, const int32_t *__restrict__ ndist, int ldc,
    const double *__restrict__ w, const int64_t *__restrict__ wide_rows, int64_t n_wide,
    const double *__restrict__ props, int64_t R, int H, double *__restrict__ partial,
    int64_t ldpart, const mxm_em_state *__restrict__ state, int run, const int64_t *__restrict__ row_list, int64_t n_list,
    const int32_t *__restrict__ nquad, int part_row0) {
    if (state != nullptr && state[run].done != 0) return;
    const int t = threadIdx.x;
    props += (int64_t)run * H;

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
    bool meta_ready = false;
    int chk[2] = {(int)n_wide, 0};                       // (MXM_CODED_CHECK=0, A/B builds only: "the list is right")
#ifndef MXM_CODED_CHECK
#define MXM_CODED_CHECK 1
#endif
    coded_row_pass<THREADS, NCH, NBUF, true, false, true, MXM_CODED_CHECK != 0>(rec, rec_off, ndist, ldc, w, wide_rows, n_wide, R, p, acc,
                                                                                 meta_ready, chk, row_list, n_list, nquad);
    const int64_t prow = (int64_t)part_row0 + blockIdx.x;   // (beside a quad dictionary the quad kernel's rows come first)
    if (t == 0) {                                        // {wide rows met, list fault}: behind the partial rows this path can use
        int *out = reinterpret_cast<int *>(partial + (int64_t)MXM_MAX_WG * ldpart) + 2 * prow;
        out[0] = chk[0];
        out[1] = chk[1];
    }

    double *dst = partial + prow * ldpart;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = 4 * (t + k * THREADS);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < H) dst[c + e] = acc[k][e];
    }
}

// The wide rows' list against ndist, for the blocking entry point (mxm_em_loop_coded validates once, on entry):
// out[0] += rows with more than 256 values, out[1] = 1 if a list entry is out of range, not ascending or not wide.
__global__ __launch_bounds__(256) void coded_validate_kernel(const int32_t *__restrict__ ndist, int64_t R,
                                                            const int64_t *__restrict__ wide_rows, int64_t n_wide,
                                                            unsigned long long *__restrict__ out) {
    __shared__ unsigned long long s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    unsigned long long n = 0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < R; r += stride) n += (ndist[r] > ENC_MAX_CODES) ? 1 : 0;
    bool bad = false;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_wide; e += stride) {
        const int64_t r = wide_rows[e];
        bad = bad || r < 0 || r >= R || (e > 0 && wide_rows[e - 1] >= r) || ndist[r < 0 || r >= R ? 0 : r] <= ENC_MAX_CODES;
    }
    if (n) atomicAdd(&s_n, n);
    if (bad) out[1] = 1ull;
    __syncthreads();
    if (threadIdx.x == 0 && s_n) atomicAdd(&out[0], s_n);
}

// decode (tests, posterior pass from records): P[r][h] = table[codes[h]] for coded rows; others untouched
__global__ __launch_bounds__(256) void decode_rows_kernel(const uint8_t *__restrict__ rec,
                                                          const int64_t *__restrict__ rec_off,
                                                          const int32_t *__restrict__ ndist, int ldc, int64_t R, int H,
                                                          double *__restrict__ P, int64_t ldp) {
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
        const int nd = ndist[r];
        if (nd <= 0) continue;
        const bool wide = nd > ENC_MAX_CODES;
        const uint8_t *codes = rec + rec_off[r];
        const double *tbl = reinterpret_cast<const double *>(codes + rec_code_bytes(nd, ldc));
        for (int h = threadIdx.x; h < H; h += 256) P[r * ldp + h] = tbl[rec_code_at(codes, h, wide)];
    }
}

// ------------------------------------------------------------------------------------------
// Consumers of a coded matrix that need the LOG values (the record's second table):
//   (the row argmax of the posterior lives in records_kernels.hpp: posterior_argmax_kernel)
//   out[r][i] = M[r][cols[i]]                                 -- preprocess.py:247-251 (em_mat[:, indexes])
// One workgroup per row; rows without a record are left untouched (the caller has them dense).
// ------------------------------------------------------------------------------------------
// Log-sum-exp of a coded row under one proportion vector, lse_r = logsumexp_h(ln_props[h] + M[r][h]) (em.py:82), for a
// workgroup of 256 that holds the row's tables in LDS (s_p = P table, s_m = log table).  Fast form, in the loop's own
// variables: rowmax_r + log(sum_h props[h] P[r][h]) -- one lookup and one FMA per cell, one logarithm per row.  Where
// that sum is 0 or not finite (every supported haplogroup's proportion or P underflowed, or a NaN proportion) the
// row is redone in log space with a max shift, as the dense pass (mxm_em_step) and the reference do, so the records
// posterior stays finite exactly where theirs does.  Uniform result; contains barriers.
__device__ __forceinline__ double coded_row_lse(const uint8_t *codes, bool wide, const double *s_p, const double *s_m,
                                                const double *__restrict__ props, const double *__restrict__ ln_props,
                                                double rowmax_r, int H, double *s_red) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    double z = 0.0;
    for (int h = t; h < H; h += 256) z = fma(props[h], s_p[rec_code_at(codes, h, wide)], z);
    z = wave_sum(z);
    __syncthreads();                                        // s_red free
    if (lane == 0) s_red[wv] = z;
    __syncthreads();
    z = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    if (z > 0.0 && z < INFINITY) return rowmax_r + log(z);  // uniform branch
    double m = -INFINITY;
    for (int h = t; h < H; h += 256) m = fmax(m, ln_props[h] + s_m[rec_code_at(codes, h, wide)]);
    m = wave_max(m);
    __syncthreads();
    if (lane == 0) s_red[wv] = m;
    __syncthreads();
    m = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
    const double shift = (m > -INFINITY && m < INFINITY) ? m : 0.0;         // as the dense pass (estep_kernels.hpp)
    double sacc = 0.0;
    for (int h = t; h < H; h += 256) sacc += exp((ln_props[h] + s_m[rec_code_at(codes, h, wide)]) - shift);
    sacc = wave_sum(sacc);
    __syncthreads();
    if (lane == 0) s_red[wv] = sacc;
    __syncthreads();
    sacc = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    return log(sacc) + m;                                   // m, not shift: -inf rows stay -inf, as scipy does
}

// Posterior pass from records (em.py:80-83, :156): out[r][h] = ln_props[h] + M[r][h] - lse_r (coded_row_lse).
// mode 1 folds with logaddexp (multi-run).
__global__ __launch_bounds__(256) void coded_posterior_kernel(const uint8_t *__restrict__ rec,
                                                             const int64_t *__restrict__ rec_off,
                                                             const int32_t *__restrict__ ndist, int ldc, int64_t R, int H,
                                                             const double *__restrict__ ln_props,
                                                             const double *__restrict__ props,
                                                             const double *__restrict__ rowmax,
                                                             double *__restrict__ out, int64_t ldo, int mode) {
    __shared__ double s_p[ENC_MAX_WIDE], s_m[ENC_MAX_WIDE];
    __shared__ double s_red[4];
    const int t = threadIdx.x;
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
        const int nd = ndist[r];
        if (nd <= 0) continue;                               // uniform
        const bool wide = nd > ENC_MAX_CODES;
        const uint8_t *codes = rec + rec_off[r];
        const double *ptab = reinterpret_cast<const double *>(codes + rec_code_bytes(nd, ldc));
        for (int i = t; i < nd; i += 256) {
            s_p[i] = ptab[i];
            s_m[i] = ptab[nd + i];
        }
        __syncthreads();
        const double lse = coded_row_lse(codes, wide, s_p, s_m, props, ln_props, rowmax[r], H, s_red);
        double *dst = out + r * ldo;
        for (int h = t; h < H; h += 256) {
            const double v = (ln_props[h] + s_m[rec_code_at(codes, h, wide)]) - lse;
            dst[h] = (mode == 1) ? logaddexp_f64(dst[h], v) : v;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void coded_gather_columns_kernel(const uint8_t *__restrict__ rec,
                                                                  const int64_t *__restrict__ rec_off,
                                                                  const int32_t *__restrict__ ndist, int ldc, int64_t R,
                                                                  const int32_t *__restrict__ cols, int nC,
                                                                  double *__restrict__ out, int64_t ldo) {
    const int64_t total = R * nC;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / nC;
        const int i = (int)(e - r * nC);
        const int nd = ndist[r];
        if (nd <= 0) continue;
        const uint8_t *codes = rec + rec_off[r];
        const double *mtab = reinterpret_cast<const double *>(codes + rec_code_bytes(nd, ldc)) + nd;
        out[r * ldo + i] = mtab[rec_code_at(codes, cols[i], nd > ENC_MAX_CODES)];
    }
}

#endif  // MIXEMT_CODED_KERNELS_HPP
