// quad_kernels.hpp -- part of libmixemt_hip.so (gfx950); included by mixemt_hip.hip only.
// A QUAD dictionary beside the row-dictionary records, for the EM iteration alone (round 5).
#ifndef MIXEMT_QUAD_KERNELS_HPP
#define MIXEMT_QUAD_KERNELS_HPP

// ------------------------------------------------------------------------------------------
// Why.  The records pass (coded_row_pass) spends, per cell, a shift (code byte -> table offset), a ds_read_b64 and two
// fp64 FMAs, and runs at 0.65 of what its loads alone reach: VALU cycles, the LDS pipe and HBM are each a bit over half
// used by two waves per SIMD that wait on each other's barrier (profiles/r05/experiments.md, 1b-1e).  A row holds few
// distinct values (median 25) and they come in runs, so the row's 1352 aligned GROUPS OF FOUR columns hold few distinct
// value quadruples too: median 93, at most 256 for 98.8 % of the byte-coded rows of a build_em_matrix matrix.  A quad
// record names four columns' values with ONE code byte:
//     codes   QUAD_CODE_BYTES = 256 threads x 8 bytes, thread-contiguous: byte j of thread t is the code of the quad
//             t + 256 j (columns 4 (t + 256 j) .. + 3) -- the columns thread t of the pass owns; one 8-byte load per
//             thread and row instead of six dword loads
//     table   nq x 32 bytes: the four values (the record's own P table entries, the same bits) of each distinct quad
// Per four cells: one shift and two ds_read_b128 instead of four shifts and four ds_read_b64; 4.8 KB per row instead of
// 5.7.  Measured as an experiment on the real records first (tools/experiments/quad_experiment.hip, quad_1m.txt): 1.27 ms
// per pass over 10^6 rows against 1.41-1.45, column sums within 2.5e-16 (same values, same thread-to-column map, same
// order of the per-thread sums).
// The quads are an ACCELERATION STRUCTURE for mxm_em_iter_coded / mxm_em_loop_coded: every other consumer (votes,
// posterior, gathers, decode, the save files) keeps reading the records, which stay complete.  Rows without quads
// (more than 256 distinct quads: 1.2 %; wide rows: 2 %) are taken from the records by the byte pass through a row list.
// ------------------------------------------------------------------------------------------
#define QUAD_MAX 256
#define QUAD_CODE_BYTES 2048
#define QUAD_HASH 1024
#define QUAD_THREADS 256
#define QUAD_CHUNK 65536ull               // bytes a workgroup of the encoder reserves at a time

typedef unsigned int quad_u2 __attribute__((ext_vector_type(2)));
typedef unsigned int quad_u4 __attribute__((ext_vector_type(4)));
typedef double quad_d2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------
// K3q-enc  quad_encode_kernel: one workgroup per byte-coded row.  The row's quads (4 code bytes = one dword; code bytes
// of columns past H are cleared so that equal quads compare equal) are de-duplicated in an LDS hash table (64-bit slots:
// a tag bit + the quad, so that the quad 0xFFFFFFFF is an ordinary key), the distinct ones ranked by value (each counts
// the smaller ones; QUAD_RANK_BY_COUNT: by occurrences first -- built and measured in round 6, no gain) -- the codes are
// those ranks, so a record's bytes do not depend on which thread won a slot -- and the
// record goes to a bump-allocated place in `qrec` (a workgroup reserves 64 KB at a time: stats[0] = bytes RESERVED, an upper
// bound of what is in use; a record that does not fit any more is not written and the row keeps nquad = 0: the caller sees
// stats[0] > capacity and may repeat with that much and a chunk per workgroup to spare).
// stats[1] counts the byte-coded rows left without quads.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(QUAD_THREADS) void quad_encode_kernel(const uint8_t *__restrict__ rec,
                                                                   const int64_t *__restrict__ rec_off,
                                                                   const int32_t *__restrict__ ndist, int ldc, int H, int64_t R,
                                                                   uint8_t *__restrict__ qrec, unsigned long long qcap,
                                                                   int64_t *__restrict__ qoff, int32_t *__restrict__ nquad,
                                                                   unsigned long long *__restrict__ stats) {
#ifndef QUAD_RANK_BY_COUNT
#define QUAD_RANK_BY_COUNT 0              // 1: codes = ranks by (occurrences descending, quad ascending) -- the round-6 A/B build:
#endif                                    // the pass's LDS bank conflicts (24.9 %) and its time did not move (profiles/r06/ab_quad_ranking_1m.txt)
    __shared__ unsigned long long s_tab[QUAD_HASH];
    __shared__ int s_cnt[QUAD_RANK_BY_COUNT ? QUAD_HASH : 1];       // occurrences of the quad in slot i among the row's quads
    __shared__ int s_kcnt[QUAD_RANK_BY_COUNT ? QUAD_MAX : 1];
    __shared__ unsigned int s_keys[QUAD_MAX], s_sorted[QUAD_MAX];
    __shared__ unsigned short s_slot[QUAD_MAX];
    __shared__ unsigned char s_rank[QUAD_HASH];
    __shared__ int s_n, s_n2;
    __shared__ double s_ptab[ENC_MAX_CODES];
    __shared__ unsigned long long s_base, s_chunk_at, s_chunk_left;
    const int t = threadIdx.x;
    const int nqc = ldc >> 2;                              // quads per row
    if (t == 0) s_chunk_at = s_chunk_left = 0ull;          // (ordered before its first use by the row loop's barriers)
    // The row loop is a chain of dependent steps (six barriers, ~17 us per row and workgroup, SQ: parked 74 %): the next row's
    // record offset / size are asked for at the top of a row and its code words once they are back, after the inserts'
    // barrier, so that neither load's latency sits in front of the next row's first step (round 6).
    const auto codes_ok = [](int nd) { return nd > 0 && nd <= ENC_MAX_CODES; };
    auto load_codes = [&](long long off, int nd, unsigned int(&dst)[8]) {
        const unsigned int *codes = reinterpret_cast<const unsigned int *>(rec + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = t + QUAD_THREADS * j;
            dst[j] = (codes_ok(nd) && idx < nqc) ? codes[idx] : 0u;
        }
    };
    int64_t r = blockIdx.x;
    int nd_nx = r < R ? ndist[r] : 0;
    long long off_nx = r < R ? rec_off[r] : 0;
    unsigned int q_nx[8];
    load_codes(off_nx, nd_nx, q_nx);
    for (; r < R; r += gridDim.x) {
        const int nd = nd_nx;
        const long long off_r = off_nx;
        unsigned int q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = q_nx[j];
        const int64_t r2 = r + gridDim.x;
        nd_nx = r2 < R ? ndist[r2] : 0;                   // (in flight under this row's first steps)
        off_nx = r2 < R ? rec_off[r2] : 0;
        if (!codes_ok(nd)) {                               // uniform: wide rows and rows without a record have no quads
            if (t == 0) {
                nquad[r] = 0;
                qoff[r] = 0;
            }
            load_codes(off_nx, nd_nx, q_nx);
            continue;
        }
        for (int i = t; i < QUAD_HASH; i += QUAD_THREADS) {
            s_tab[i] = 0ull;
            if constexpr (QUAD_RANK_BY_COUNT != 0) s_cnt[i] = 0;
        }
        if (t == 0) s_n = s_n2 = 0;
        __syncthreads();
        const uint8_t *base = rec + off_r;
        // this thread's entry of the record's P table: asked for now, parked in LDS behind the ranking's barrier, so that the
        // four values of a table entry are LDS reads at the end of the row instead of dependent global loads
        const double ptab_mine = (t < nd) ? reinterpret_cast<const double *>(base + ldc)[t] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = t + QUAD_THREADS * j;
            unsigned int key = 0;
            if (idx < nqc) {
                key = q[j];
                const int c0 = idx * 4;                    // columns past H (ldc - H <= 7 of them): code 0
                if (c0 + 3 >= H) {
                    if (c0 + 0 >= H) key &= ~0x000000ffu;
                    if (c0 + 1 >= H) key &= ~0x0000ff00u;
                    if (c0 + 2 >= H) key &= ~0x00ff0000u;
                    if (c0 + 3 >= H) key &= ~0xff000000u;
                }
                const unsigned long long tagged = (1ull << 32) | key;
                unsigned int slot = (key * 2654435761u) >> 22;               // 10 bits
                bool placed = false;
                for (int probe = 0; probe < QUAD_HASH && s_n <= QUAD_MAX; ++probe) {
                    // most of a row's quads are ONE quad (the row's majority value four times): look before the atomic, or
                    // a thousand compare-and-swaps queue up on one LDS word
                    unsigned long long old = __hip_atomic_load(&s_tab[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (a ds_read: a volatile access went the flat way)
                    if (old == 0ull) old = atomicCAS(&s_tab[slot], 0ull, tagged);
                    if (old == 0ull) {
                        atomicAdd(&s_n, 1);
                        placed = true;
                        break;
                    }
                    if (old == tagged) {
                        placed = true;
                        break;
                    }
                    slot = (slot + 1) & (QUAD_HASH - 1);
                }
                if (!placed) atomicAdd(&s_n, QUAD_MAX + 1);                  // too many distinct quads (or a full table)
#if QUAD_RANK_BY_COUNT
                // how often the quad occurs: the wave's lanes that hold the same quad add their number ONCE (a row is
                // mostly one quad: an atomic per lane would queue a thousand adds on one LDS word)
                if (placed) {
                    unsigned long long todo = __ballot(1);
                    while (todo != 0ull) {
                        const int lead = __ffsll((long long)todo) - 1;
                        const unsigned int k0 = (unsigned int)__builtin_amdgcn_readlane((int)key, lead);
                        const unsigned long long same = __ballot(key == k0) & todo;
                        if ((t & 63) == lead) atomicAdd(&s_cnt[slot], __popcll(same));
                        todo &= ~same;
                    }
                }
#endif
                key = slot;                                // from here on the thread only needs where its quad sits
            }
            q[j] = key;
        }
        load_codes(off_nx, nd_nx, q_nx);                   // the next row's code words: back by the time this row is written
        __syncthreads();
        const int n = s_n;                                 // uniform
        if (n > QUAD_MAX) {
            if (t == 0) {
                nquad[r] = 0;
                qoff[r] = 0;
                atomicAdd(&stats[1], 1ull);
            }
            __syncthreads();
            continue;
        }
        // the distinct quads, numbered by RANK (ascending quad): compact them, count for each how many are smaller (n
        // broadcast reads), and leave the rank in the quad's hash slot for the threads' own quads to pick up -- no sort,
        // four barriers per row
        for (int i = t; i < QUAD_HASH; i += QUAD_THREADS) {   // (four rounds, every thread in each: the ballots are whole)
            const unsigned long long v = s_tab[i];
            // one reservation per wave and round instead of one atomic per occupied slot (a row's ~93 on one LDS word)
            const unsigned long long votes = __ballot(v != 0ull);
            int first = 0;
            if ((t & 63) == 0 && votes != 0ull) first = atomicAdd(&s_n2, __popcll(votes));
            first = __builtin_amdgcn_readfirstlane(first);
            if (v != 0ull) {
                const int at_i = first + __popcll(votes & ((1ull << (t & 63)) - 1ull));
                s_keys[at_i] = (unsigned int)v;
                s_slot[at_i] = (unsigned short)i;
                if constexpr (QUAD_RANK_BY_COUNT != 0) s_kcnt[at_i] = s_cnt[i];
            }
        }
        __syncthreads();
        s_ptab[t] = ptab_mine;                              // (read after the next barrier)
        unsigned int my_key = 0;
        if (t < n) {
            my_key = s_keys[t];
            int rank = 0;
#if QUAD_RANK_BY_COUNT
            // by occurrences (descending), then by quad: the row's most frequent entries get the codes 0, 1, 2, ... -- their
            // 32-byte table entries then sit in DIFFERENT bank groups of the LDS table the pass looks them up in (ranked by
            // value alone, which groups two hot entries shared was left to chance: 24.8 % of the quad pass's LDS cycles
            // were bank conflicts, profiles/r05/coded_pmc_sq_summary.txt)
            const int my_cnt = s_kcnt[t];
            for (int j = 0; j < n; ++j) {
                const int cj = s_kcnt[j];
                rank += (cj > my_cnt || (cj == my_cnt && s_keys[j] < my_key)) ? 1 : 0;
            }
#else
            for (int j = 0; j < n; ++j) rank += (s_keys[j] < my_key) ? 1 : 0;
#endif
            s_rank[s_slot[t]] = (unsigned char)rank;
            s_sorted[rank] = my_key;
        }
        __syncthreads();
        unsigned long long word = 0ull;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = t + QUAD_THREADS * j;
            if (idx < nqc) word |= (unsigned long long)s_rank[q[j]] << (8 * j);      // (q[j]: the quad's SLOT, kept from its insert)
        }
        const unsigned long long bytes = QUAD_CODE_BYTES + 32ull * (unsigned long long)n;
#ifdef QUAD_FIXED_SLOTS                                     // (timing experiment: no shared bump pointer)
        if (t == 0) s_base = (unsigned long long)r * (QUAD_CODE_BYTES + 32ull * QUAD_MAX);
#else
        // The bump pointer is shared by every workgroup of the grid: one atomic add per ROW on one address was 6 of the
        // kernel's 14.6 ms at 10^6 rows (a fixed-slot timing build ran 8.4).  A workgroup therefore reserves QUAD_CHUNK
        // bytes at a time and hands its rows out of that: one global atomic per dozen rows; what a chunk's tail cannot hold
        // is left unused (< one record per chunk, + at most one chunk per workgroup at the end: ~5 % at 10^6 rows).
        if (t == 0) {
            if (s_chunk_left < bytes) {
                const unsigned long long grab = bytes > QUAD_CHUNK ? bytes : QUAD_CHUNK;
                s_chunk_at = atomicAdd(&stats[0], grab);
                s_chunk_left = grab;
            }
            s_base = s_chunk_at;
            s_chunk_at += bytes;
            s_chunk_left -= bytes;
        }
#endif
        __syncthreads();
        const unsigned long long at = s_base;
        const bool fits = at + bytes <= qcap;              // uniform
        if (fits) {
            *reinterpret_cast<unsigned long long *>(qrec + at + (unsigned long long)t * 8ull) = word;
            if (t < n) {
                const double *tbl = s_ptab;
                const unsigned int key = s_sorted[t];
                quad_d2 *dst = reinterpret_cast<quad_d2 *>(qrec + at + QUAD_CODE_BYTES + (unsigned long long)t * 32ull);
                dst[0] = quad_d2{tbl[key & 255u], tbl[(key >> 8) & 255u]};
                dst[1] = quad_d2{tbl[(key >> 16) & 255u], tbl[key >> 24]};
            }
        }
        if (t == 0) {
            nquad[r] = fits ? n : 0;
            qoff[r] = fits ? (int64_t)at : 0;
            if (!fits) atomicAdd(&stats[1], 1ull);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// K3q-encw  quad_encode_wave_kernel: the same records, one WAVE per row (round 6).  The workgroup-per-row encoder above is
// a chain of six barriers per row with five or six quads per thread between them (SQ: parked 74 %, issuing 17 %); a wave
// that owns its row needs no barrier at all -- a wave's LDS instructions are carried out in the order it issued them, so a
// compiler fence is all that stands between the steps -- and a CU holds sixteen rows instead of eight.  Lane l owns the
// quads l + 64 k (k < 24), i.e. the code words of the threads l, l + 64, l + 128, l + 192 of the pass; the distinct quads
// are counted by ballots as they are inserted (no shared counter), the hash has 512 slots (a round adds at most 64 to at
// most 256: it never fills), the bump pointer's chunk is the wave's own, in registers.  The bytes written for a row are
// those of the kernel above (codes = ranks by value); where the record lies in `qrec` differs from run to run in both.
// ------------------------------------------------------------------------------------------
#define QW_HASH 512
#define QW_K 24
#define QW_WAVES (QUAD_THREADS / 64)
#ifndef QW_MIN_WAVES
#define QW_MIN_WAVES 5                    // waves per SIMD the encoder is compiled for (20 rows per CU; LDS: 26 KB per workgroup)
#endif

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// body(integral_constant<int, 0>) ... body(integral_constant<int, N - 1>): a loop whose index is a constant where the
// source is parsed.  (Arrays indexed inside `#pragma unroll` loops become ONE 24-wide vector value once they are carried
// round the row loop -- spilled and reloaded as a block; indexed by constants from the start they are 24 registers.)
template <class F, int... K>
__device__ __forceinline__ void quad_static_for_impl(F &&body, std::integer_sequence<int, K...>) {
    (body(std::integral_constant<int, K>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void quad_static_for(F &&body) {
    quad_static_for_impl(body, std::make_integer_sequence<int, N>{});
}

__global__ __launch_bounds__(QUAD_THREADS, QW_MIN_WAVES) void quad_encode_wave_kernel(const uint8_t *__restrict__ rec,
                                                                        const int64_t *__restrict__ rec_off,
                                                                        const int32_t *__restrict__ ndist, int ldc, int H, int64_t R,
                                                                        uint8_t *__restrict__ qrec, unsigned long long qcap,
                                                                        int64_t *__restrict__ qoff, int32_t *__restrict__ nquad,
                                                                        unsigned long long *__restrict__ stats) {
    __shared__ __attribute__((aligned(16))) unsigned long long s_tab_all[QW_WAVES][QW_HASH];
    __shared__ unsigned char s_rank_all[QW_WAVES][QW_HASH];
    __shared__ double s_ptab_all[QW_WAVES][ENC_MAX_CODES];
    int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned long long *s_tab = s_tab_all[wv];
    // once the table has been read out (into registers), its 4 KB hold the distinct quads in slot order (+ 4: read four at a
    // time), the same in rank order, and their slots
    unsigned int *s_keys = reinterpret_cast<unsigned int *>(s_tab), *s_sorted = s_keys + QUAD_MAX + 4;
    unsigned short *s_slot = reinterpret_cast<unsigned short *>(s_sorted + QUAD_MAX);
    static_assert((2 * QUAD_MAX + 4) * 4 + QUAD_MAX * 2 <= QW_HASH * 8, "the lists live in the table's LDS");
    unsigned char *s_rank = s_rank_all[wv];
    double *s_ptab = s_ptab_all[wv];
    const int nqc = ldc >> 2;                              // quads per row (<= 64 QW_K, checked by the host)
    unsigned long long chunk_at = 0ull, chunk_left = 0ull; // the wave's piece of `qrec` (uniform)
    const auto codes_ok = [](int nd) { return nd > 0 && nd <= ENC_MAX_CODES; };
    // a row's code words and its record's P table (lane l: entries l, l + 64, ...), asked for during the row before
    auto load_row = [&](long long off, int nd, unsigned int(&dst)[QW_K], double(&tab)[4]) {
        const unsigned int *codes = reinterpret_cast<const unsigned int *>(rec + off);
        const double *ptab = reinterpret_cast<const double *>(rec + off + ldc);
        const bool ok = codes_ok(nd);
        quad_static_for<QW_K>([&](auto K) {
            constexpr int k = decltype(K)::value;
            dst[k] = (ok && lane + 64 * k < nqc) ? codes[lane + 64 * k] : 0u;
        });
        quad_static_for<4>([&](auto M) {
            constexpr int m = decltype(M)::value;
            tab[m] = (ok && lane + 64 * m < nd) ? ptab[lane + 64 * m] : 0.0;
        });
    };
    // a wave takes a run of consecutive rows: their sizes and offsets come 64 rows at a time, one per lane (a row's own
    // scalar loads would put a trip to memory in front of every row: they share the LDS instructions' counter), and what
    // the wave found goes back the same way
    const int64_t waves = (int64_t)gridDim.x * QW_WAVES;
    const int64_t per = (R + waves - 1) / waves;
    const int64_t row_lo = ((int64_t)blockIdx.x * QW_WAVES + wv) * per;
    const int64_t row_hi = row_lo + per < R ? row_lo + per : R;
    if (row_lo >= row_hi) return;
    const auto meta_nd = [&](int64_t first) { return first + lane < row_hi ? ndist[first + lane] : 0; };
    const auto meta_off = [&](int64_t first) { return first + lane < row_hi ? (long long)rec_off[first + lane] : 0ll; };
    const auto lane_i32 = [](int v, int i) { return __builtin_amdgcn_readlane(v, i); };
    const auto lane_i64 = [](long long v, int i) {
        const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, i);
        const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)((unsigned long long)v >> 32), i);
        return (long long)(((unsigned long long)hi << 32) | lo);
    };
    int nd_v = meta_nd(row_lo), nd_v2 = 0;
    long long off_v = meta_off(row_lo), off_v2 = 0;
    unsigned int q[QW_K];
    double pt[4];
    load_row(lane_i64(off_v, 0), lane_i32(nd_v, 0), q, pt);
    int nq_v = 0;                                          // lane i: what row first + i got
    long long at_v = 0;
    for (int64_t first = row_lo; first < row_hi; first += 64) {
        nd_v2 = meta_nd(first + 64);                       // (back long before the chunk's last row asks for them)
        off_v2 = meta_off(first + 64);
        const int rows_here = row_hi - first < 64 ? (int)(row_hi - first) : 64;
        for (int i = 0; i < rows_here; ++i) {
            const int nd = lane_i32(nd_v, i);
            const int nd_nx = i + 1 < 64 ? lane_i32(nd_v, (i + 1) & 63) : lane_i32(nd_v2, 0);     // (0 past the wave's last row)
            const long long off_nx = i + 1 < 64 ? lane_i64(off_v, (i + 1) & 63) : lane_i64(off_v2, 0);
            int n_out = 0;
            long long at_out = 0;
            // (the lane number made opaque once per row: what is derived from it -- 24 indices, 24 predicates, 24 addresses
            // -- is otherwise kept in registers across the whole row loop)
            asm volatile("" : "+v"(lane));
            int h4 = H >> 2, nqc_r = nqc;                  // (the same for the 24 rounds' uniform conditions)
            asm volatile("" : "+s"(h4), "+s"(nqc_r));
            do {                                           // (one pass: `break` = the row keeps no quads)
                if (!codes_ok(nd)) {                       // uniform: wide rows and rows without a record have no quads
                    load_row(off_nx, nd_nx, q, pt);
                    break;
                }
                wave_lds_fence();                          // (the row before is done with the arrays)
                quad_static_for<QW_HASH / 64>([&](auto Z) { s_tab[lane + 64 * decltype(Z)::value] = 0ull; });
                quad_static_for<4>([&](auto M) { s_ptab[lane + 64 * decltype(M)::value] = pt[decltype(M)::value]; });
                wave_lds_fence();
                int n = 0;                                 // distinct quads so far (uniform)
                quad_static_for<QW_K>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    if (64 * k < nqc_r && n <= QUAD_MAX) { // uniform
                        const int idx = lane + 64 * k;
                        bool fresh = false;
                        if (idx < nqc) {
                            unsigned int key = q[k];
                            const int c0 = idx * 4;        // columns past H (ldc - H <= 7 of them): code 0
                            if (64 * k + 64 >= h4) {       // uniform, false for all rounds but the last
                                asm volatile("");          // (a branch, not twelve instructions with no lane enabled)
                                if (c0 + 0 >= H) key &= ~0x000000ffu;
                                if (c0 + 1 >= H) key &= ~0x0000ff00u;
                                if (c0 + 2 >= H) key &= ~0x00ff0000u;
                                if (c0 + 3 >= H) key &= ~0xff000000u;
                            }
                            const unsigned long long tagged = (1ull << 32) | key;
                            unsigned int slot = (key * 2654435761u) >> 23;       // 9 bits
                            // the first look straight-line: most of a row's quads are ONE quad (look before the atomic), and
                            // a table a fifth full rarely needs a second probe -- the loop below is skipped by most rounds
                            unsigned long long old = __hip_atomic_load(&s_tab[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (old == 0ull) old = atomicCAS(&s_tab[slot], 0ull, tagged);
                            fresh = old == 0ull;
                            bool more = !fresh && old != tagged;
                            while (more) {                 // (ends: the table holds at most 256 + 64 of 512)
                                slot = (slot + 1) & (QW_HASH - 1);
                                old = __hip_atomic_load(&s_tab[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                if (old == 0ull) old = atomicCAS(&s_tab[slot], 0ull, tagged);
                                fresh = old == 0ull;
                                more = !fresh && old != tagged;
                            }
                            q[k] = slot;                   // from here on the lane only needs where its quad sits
                        }
                        n += __popcll(__ballot(fresh));
                    }
                });
                // where the lane's quads sit, two to a register, so that the code words of the NEXT row can be on their
                // way while this row is ranked and written
                unsigned int where[QW_K / 2];
                quad_static_for<QW_K / 2>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    where[k] = q[2 * k] | (q[2 * k + 1] << 16);
                });
                asm volatile("" : "+v"(lane));
                load_row(off_nx, nd_nx, q, pt);
                asm volatile("" : "+v"(lane));
                if (n > QUAD_MAX) {
                    if (lane == 0) atomicAdd(&stats[1], 1ull);
                    break;
                }
                wave_lds_fence();
                // the distinct quads in slot order, then each counts the smaller ones: its rank is its code
                int cnt = 0;
                unsigned long long held[QW_HASH / 64];
                quad_static_for<QW_HASH / 64>([&](auto Z) { held[decltype(Z)::value] = s_tab[lane + 64 * decltype(Z)::value]; });
                wave_lds_fence();                          // (the lists below are written where the table was)
                quad_static_for<QW_HASH / 64>([&](auto Z) {
                    constexpr int z = decltype(Z)::value;
                    const unsigned long long votes = __ballot(held[z] != 0ull);
                    if (held[z] != 0ull) {
                        // (mbcnt: the votes of the lanes below this one)
                        const int at_i = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(votes >> 32),
                                                                              __builtin_amdgcn_mbcnt_lo((unsigned int)votes, 0u));
                        s_keys[at_i] = (unsigned int)held[z];
                        s_slot[at_i] = (unsigned short)(lane + 64 * z);
                    }
                    cnt += __popcll(votes);
                });
                if (lane < 4) s_keys[cnt + lane] = 0xffffffffu;        // (smaller than nothing: the ranking reads four keys at a time)
                wave_lds_fence();
                unsigned int mine0 = (lane < n) ? s_keys[lane] : 0u, mine1 = (lane + 64 < n) ? s_keys[lane + 64] : 0u;
                unsigned int mine2 = (lane + 128 < n) ? s_keys[lane + 128] : 0u, mine3 = (lane + 192 < n) ? s_keys[lane + 192] : 0u;
                int rank0 = 0, rank1 = 0, rank2 = 0, rank3 = 0;
                const quad_u4 *keys4 = reinterpret_cast<const quad_u4 *>(s_keys);
                const int n4 = (n + 3) >> 2;
                const auto smaller = [](quad_u4 kk, unsigned int me) {
                    return ((kk.x < me) ? 1 : 0) + ((kk.y < me) ? 1 : 0) + ((kk.z < me) ? 1 : 0) + ((kk.w < me) ? 1 : 0);
                };
                if (n <= 64) {
                    for (int j = 0; j < n4; ++j) rank0 += smaller(keys4[j], mine0);
                } else if (n <= 128) {
                    for (int j = 0; j < n4; ++j) {
                        const quad_u4 kk = keys4[j];
                        rank0 += smaller(kk, mine0);
                        rank1 += smaller(kk, mine1);
                    }
                } else {
                    for (int j = 0; j < n4; ++j) {
                        const quad_u4 kk = keys4[j];
                        rank0 += smaller(kk, mine0);
                        rank1 += smaller(kk, mine1);
                        rank2 += smaller(kk, mine2);
                        rank3 += smaller(kk, mine3);
                    }
                }
                const auto place = [&](int e, unsigned int me, int rk) {
                    if (e < n) {
                        s_rank[s_slot[e]] = (unsigned char)rk;
                        s_sorted[rk] = me;
                    }
                };
                place(lane, mine0, rank0);
                place(lane + 64, mine1, rank1);
                place(lane + 128, mine2, rank2);
                place(lane + 192, mine3, rank3);
                wave_lds_fence();
                unsigned long long word[4] = {0ull, 0ull, 0ull, 0ull};
                quad_static_for<QW_K>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    const unsigned int slot = (k & 1) ? (where[k >> 1] >> 16) : (where[k >> 1] & 0xffffu);
                    if (lane + 64 * k < nqc) word[k & 3] |= (unsigned long long)s_rank[slot] << (8 * (k >> 2));
                });
                const unsigned long long bytes = QUAD_CODE_BYTES + 32ull * (unsigned long long)n;
                if (chunk_left < bytes) {                  // uniform: one global atomic per QUAD_CHUNK bytes and wave
                    const unsigned long long grab = bytes > QUAD_CHUNK ? bytes : QUAD_CHUNK;
                    unsigned long long got = 0ull;
                    if (lane == 0) got = atomicAdd(&stats[0], grab);
                    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)got);
                    const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(got >> 32));
                    chunk_at = ((unsigned long long)hi << 32) | lo;
                    chunk_left = grab;
                }
                const unsigned long long at = chunk_at;
                chunk_at += bytes;
                chunk_left -= bytes;
                if (at + bytes > qcap) {                   // uniform: no room left in `qrec` (the caller sees stats[0] > capacity)
                    if (lane == 0) atomicAdd(&stats[1], 1ull);
                    break;
                }
                quad_static_for<4>([&](auto M) {
                    constexpr int m = decltype(M)::value;
                    *reinterpret_cast<unsigned long long *>(qrec + at + (unsigned long long)(lane + 64 * m) * 8ull) = word[m];
                });
                quad_static_for<4>([&](auto M) {
                    constexpr int m = decltype(M)::value;
                    if (lane + 64 * m < n) {
                        const unsigned int key = s_sorted[lane + 64 * m];
                        quad_d2 *dst = reinterpret_cast<quad_d2 *>(qrec + at + QUAD_CODE_BYTES + (unsigned long long)(lane + 64 * m) * 32ull);
                        dst[0] = quad_d2{s_ptab[key & 255u], s_ptab[(key >> 8) & 255u]};
                        dst[1] = quad_d2{s_ptab[(key >> 16) & 255u], s_ptab[key >> 24]};
                    }
                });
                n_out = n;
                at_out = (long long)at;
            } while (false);
            if (lane == i) {
                nq_v = n_out;
                at_v = at_out;
            }
        }
        if (lane < rows_here) {
            nquad[first + lane] = nq_v;
            qoff[first + lane] = at_v;
        }
        nd_v = nd_v2;
        off_v = off_v2;
    }
}

// The two row lists of a quad dictionary against ndist / nquad, for the blocking entry point (mxm_em_loop_coded validates
// once, on entry; the per-iteration kernels check their entries where they use them): out[1] = 1 if an entry is out of
// range, not above its predecessor or not of its class (quad_rows: 1 <= nquad <= 256; byte_rows: 0 < ndist <= 256 and
// nquad == 0).  With the counts adding up to R (coded_check) that makes the lists the partition they must be.
__global__ __launch_bounds__(256) void quad_validate_kernel(const int32_t *__restrict__ ndist, const int32_t *__restrict__ nquad,
                                                           int64_t R, const int64_t *__restrict__ quad_rows, int64_t n_quad,
                                                           const int64_t *__restrict__ byte_rows, int64_t n_byte,
                                                           unsigned long long *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    bool bad = false;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_quad; e += stride) {
        const int64_t r = quad_rows[e];
        if (r < 0 || r >= R || (e > 0 && quad_rows[e - 1] >= r)) bad = true;
        else bad = bad || nquad[r] < 1 || nquad[r] > QUAD_MAX || ndist[r] < 1 || ndist[r] > ENC_MAX_CODES;
    }
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_byte; e += stride) {
        const int64_t r = byte_rows[e];
        if (r < 0 || r >= R || (e > 0 && byte_rows[e - 1] >= r)) bad = true;
        else bad = bad || nquad[r] != 0 || ndist[r] < 1 || ndist[r] > ENC_MAX_CODES;
    }
    if (bad) out[1] = 1ull;
}

// ------------------------------------------------------------------------------------------
// The row lists of a quad dictionary, formed ON THE DEVICE in ascending order (mxm_quad_lists): rows with quads ->
// quad_rows, byte-coded rows without -> byte_rows.  Three small launches -- per-chunk counts, one workgroup's scan over
// the chunks, per-chunk ordered scatter -- instead of a round trip through the host (4 bytes per row down, numpy, 8
// bytes per row up: a process's first uploads from fresh pageable memory cost 20-30 ms each, more than the encoder).
// Class of row r: 1 = quads (nquad > 0), 2 = byte-coded without quads (0 < ndist <= 256), 0 = neither.
// ------------------------------------------------------------------------------------------
#define QLIST_THREADS 256
#define QLIST_PER_THREAD 16
#define QLIST_CHUNK (QLIST_THREADS * QLIST_PER_THREAD)
__device__ __forceinline__ int quad_row_class(const int32_t *__restrict__ ndist, const int32_t *__restrict__ nquad, int64_t r) {
    if (nquad[r] > 0) return 1;
    const int nd = ndist[r];
    return (nd > 0 && nd <= ENC_MAX_CODES) ? 2 : 0;
}
__global__ __launch_bounds__(QLIST_THREADS) void quad_list_count_kernel(const int32_t *__restrict__ ndist,
                                                                       const int32_t *__restrict__ nquad, int64_t R,
                                                                       long long *__restrict__ chunk_counts /* [nchunk][2] */) {
    __shared__ int s_c[2];
    if (threadIdx.x < 2) s_c[threadIdx.x] = 0;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * QLIST_CHUNK + (int64_t)threadIdx.x * QLIST_PER_THREAD;
    int nq = 0, nb = 0;
    for (int i = 0; i < QLIST_PER_THREAD; ++i) {
        const int64_t r = r0 + i;
        if (r < R) {
            const int c = quad_row_class(ndist, nquad, r);
            nq += (c == 1);
            nb += (c == 2);
        }
    }
    if (nq) atomicAdd(&s_c[0], nq);
    if (nb) atomicAdd(&s_c[1], nb);
    __syncthreads();
    if (threadIdx.x < 2) chunk_counts[2 * blockIdx.x + threadIdx.x] = s_c[threadIdx.x];
}
// exclusive scan of the chunk counts in place (one workgroup; the chunks number R / 4096); totals -> counts[0..1]
__global__ __launch_bounds__(QLIST_THREADS) void quad_list_scan_kernel(long long *__restrict__ chunk_counts, int64_t nchunk,
                                                                      long long *__restrict__ counts) {
    __shared__ long long s_sum[2][QLIST_THREADS];
    const int t = threadIdx.x;
    const int64_t per = (nchunk + QLIST_THREADS - 1) / QLIST_THREADS;
    const int64_t lo = (int64_t)t * per, hi = (lo + per < nchunk) ? lo + per : nchunk;
    long long a = 0, b = 0;
    for (int64_t i = lo; i < hi; ++i) {
        a += chunk_counts[2 * i];
        b += chunk_counts[2 * i + 1];
    }
    s_sum[0][t] = a;
    s_sum[1][t] = b;
    __syncthreads();
    if (t == 0) {                                        // 256 partial sums: serial, in order
        long long ra = 0, rb = 0;
        for (int i = 0; i < QLIST_THREADS; ++i) {
            const long long xa = s_sum[0][i], xb = s_sum[1][i];
            s_sum[0][i] = ra;
            s_sum[1][i] = rb;
            ra += xa;
            rb += xb;
        }
        counts[0] = ra;
        counts[1] = rb;
    }
    __syncthreads();
    a = s_sum[0][t];
    b = s_sum[1][t];
    for (int64_t i = lo; i < hi; ++i) {
        const long long xa = chunk_counts[2 * i], xb = chunk_counts[2 * i + 1];
        chunk_counts[2 * i] = a;
        chunk_counts[2 * i + 1] = b;
        a += xa;
        b += xb;
    }
}
__global__ __launch_bounds__(QLIST_THREADS) void quad_list_fill_kernel(const int32_t *__restrict__ ndist,
                                                                      const int32_t *__restrict__ nquad, int64_t R,
                                                                      const long long *__restrict__ chunk_base,
                                                                      int64_t *__restrict__ quad_rows,
                                                                      int64_t *__restrict__ byte_rows) {
    __shared__ int s_q[QLIST_THREADS], s_b[QLIST_THREADS];
    const int t = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * QLIST_CHUNK + (int64_t)t * QLIST_PER_THREAD;
    int cls[QLIST_PER_THREAD];
    int nq = 0, nb = 0;
#pragma unroll
    for (int i = 0; i < QLIST_PER_THREAD; ++i) {
        const int64_t r = r0 + i;
        cls[i] = (r < R) ? quad_row_class(ndist, nquad, r) : 0;
        nq += (cls[i] == 1);
        nb += (cls[i] == 2);
    }
    s_q[t] = nq;
    s_b[t] = nb;
    __syncthreads();
    if (t == 0) {                                        // exclusive scan over the 256 threads, in order
        int rq = 0, rb = 0;
        for (int i = 0; i < QLIST_THREADS; ++i) {
            const int xq = s_q[i], xb = s_b[i];
            s_q[i] = rq;
            s_b[i] = rb;
            rq += xq;
            rb += xb;
        }
    }
    __syncthreads();
    long long oq = chunk_base[2 * blockIdx.x] + s_q[t], ob = chunk_base[2 * blockIdx.x + 1] + s_b[t];
#pragma unroll
    for (int i = 0; i < QLIST_PER_THREAD; ++i) {
        if (cls[i] == 1) quad_rows[oq++] = r0 + i;
        else if (cls[i] == 2) byte_rows[ob++] = r0 + i;
    }
}

// code byte B of a word x 16 = the byte offset of the quad's entry in EACH of the two half tables in LDS: one SDWA shift
template <int B>
__device__ __forceinline__ unsigned int quad_byte_x16(unsigned int word) {
    const unsigned int four = 4;
    unsigned int r;
    if constexpr (B == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(four), "v"(word));
    else if constexpr (B == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(four), "v"(word));
    else if constexpr (B == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(four), "v"(word));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(four), "v"(word));
    return r;
}
// (x 32: the entry's offset in the record's own table -- QUAD_SPLIT_TABLE = 0 builds look it up in that layout)
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

// ------------------------------------------------------------------------------------------
// K3q  quad_row_pass: coded_row_pass's main loop over quad records -- the same decomposition (the rows of `quad_rows`
// dealt round-robin, thread t owns the columns 4 (t + 256 k) + e, the row's values wait in VGPRs between the dot product
// and the accumulation, lookups one row ahead, the next step's metadata read under the division), so that the per-thread
// sums are formed from the same values in the same order as the byte pass forms them.
//   quad_rows[n_rows]  the rows that have quads, ASCENDING; qoff / nquad / w are indexed by ROW
//   fault              set to 1 when a list entry is out of range, not above its predecessor or has no quads (the entry
//                      is then skipped, never dereferenced): the column reduce poisons the sums, as for the wide list
// ------------------------------------------------------------------------------------------
template <int NCH, int NBUF, bool NT>
__device__ __forceinline__ void quad_row_pass(const uint8_t *__restrict__ qrec, const int64_t *__restrict__ qoff,
                                              const int32_t *__restrict__ nquad, const int64_t *__restrict__ quad_rows,
                                              int64_t n_rows, int64_t R, const double *__restrict__ w,
                                              const double (&p)[NCH][4], double (&acc)[NCH][4], int *fault, int vbid = -1,
                                              int vgrid = 0) {
    static_assert(NBUF >= 3 && NBUF <= 6, "codes NBUF - 1 rows ahead; unrolled by hand");
    static_assert(NCH >= 1 && NCH <= 8, "eight code bytes per thread");
    constexpr int THREADS = QUAD_THREADS, NW = THREADS / 64, AUX = NT ? 2 : 0;
    __shared__ __attribute__((aligned(16))) double s_tbl[NBUF][QUAD_MAX * 4];
    __shared__ __attribute__((aligned(16))) double red[NBUF][NW];
    __shared__ long long s_off[2][THREADS];
    __shared__ double s_wr[2][THREADS];
    __shared__ int s_nd[2][THREADS];
    __shared__ int s_fault;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const row_deal deal(n_rows, vbid >= 0 ? (int64_t)vbid : (int64_t)blockIdx.x, vbid >= 0 ? (int64_t)vgrid : (int64_t)gridDim.x);
    if (t == 0) s_fault = 0;

    auto fetch_meta = [&](int half, int64_t q0) {       // steps q0 .. q0 + THREADS - 1, thread t takes step q0 + t
        const int64_t q = q0 + t;
        const int64_t e = deal.row(q);                   // list index
        int64_t r = quad_rows[e];
        int nd = 0;
        if (r < 0 || r >= R || (e > 0 && quad_rows[e - 1] >= r)) {
            s_fault = 1;
            r = -1;
        } else {
            nd = nquad[r];
            if (nd <= 0 || nd > QUAD_MAX) {
                s_fault = 1;
                nd = 0;
            }
        }
        s_off[half][t] = (nd > 0) ? qoff[r] : 0;
        s_nd[half][t] = nd;
        s_wr[half][t] = (deal.live(q) && nd > 0) ? (w != nullptr ? w[r] : 1.0) : 0.0;
    };
    quad_u2 cw[NBUF];
    double tring[NBUF][4];
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
    // a row's record: this thread's eight code bytes, entry t of its table (lanes past the table read 0; a row with
    // nd = 0 -- a faulty entry, a step past the end -- reads nothing: descriptors of 0 bytes)
    auto load_row = [&](auto SLOT) {
        constexpr int slot = decltype(SLOT)::value;
        const int nd = pre_nd;
        const uint8_t *base = qrec + (((long long)pre_off_hi << 32) | (unsigned int)pre_off_lo);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, nd > 0 ? QUAD_CODE_BYTES : 0, 0x00020000);
        cw[slot] = __builtin_amdgcn_raw_buffer_load_b64(rs, t * 8, 0, AUX);
        const auto rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base + QUAD_CODE_BYTES), 0, nd * 32, 0x00020000);
        const quad_u4 a = __builtin_amdgcn_raw_buffer_load_b128(rt, t * 32, 0, AUX);
        const quad_u4 b = __builtin_amdgcn_raw_buffer_load_b128(rt, t * 32, 16, AUX);
        tring[slot][0] = __hiloint2double((int)a.y, (int)a.x);
        tring[slot][1] = __hiloint2double((int)a.w, (int)a.z);
        tring[slot][2] = __hiloint2double((int)b.y, (int)b.x);
        tring[slot][3] = __hiloint2double((int)b.w, (int)b.z);
    };
    // In LDS a row's table is kept as TWO arrays of 16-byte elements (the entries' first halves, then their second halves)
    // instead of the record's 32-byte entries: a ds_read_b128 per half at a 32-byte stride used only every other bank
    // group, so lanes on different entries collided twice as often and the publishing writes always did -- 24.9 % of the
    // pass's LDS cycles were bank conflicts, 7.3 % with the split (profiles/r06/ab_quad_split_1m.txt: step 1.33 -> 1.30 ms).
#ifndef QUAD_SPLIT_TABLE
#define QUAD_SPLIT_TABLE 1
#endif
    auto publish = [&](auto SLOT) {
        constexpr int slot = decltype(SLOT)::value;
#if QUAD_SPLIT_TABLE
        *reinterpret_cast<quad_d2 *>(&s_tbl[slot][t * 2]) = quad_d2{tring[slot][0], tring[slot][1]};
        *reinterpret_cast<quad_d2 *>(&s_tbl[slot][QUAD_MAX * 2 + t * 2]) = quad_d2{tring[slot][2], tring[slot][3]};
#else
        quad_d2 *dst = reinterpret_cast<quad_d2 *>(&s_tbl[slot][t * 4]);
        dst[0] = quad_d2{tring[slot][0], tring[slot][1]};
        dst[1] = quad_d2{tring[slot][2], tring[slot][3]};
#endif
    };
    double v[NCH][4];
    auto lookup_quad = [&](const char *tb, auto K, const quad_u2 &c) {
        constexpr int k = decltype(K)::value;
        unsigned int off;
#if QUAD_SPLIT_TABLE
        if constexpr (k < 4) off = quad_byte_x16<k>(c.x);
        else off = quad_byte_x16<k - 4>(c.y);
        const quad_d2 a = *reinterpret_cast<const quad_d2 *>(tb + off), b = *reinterpret_cast<const quad_d2 *>(tb + off + QUAD_MAX * 16);
#else
        if constexpr (k < 4) off = quad_byte_x32<k>(c.x);
        else off = quad_byte_x32<k - 4>(c.y);
        const quad_d2 a = *reinterpret_cast<const quad_d2 *>(tb + off), b = *reinterpret_cast<const quad_d2 *>(tb + off + 16);
#endif
        v[k][0] = a.x;
        v[k][1] = a.y;
        v[k][2] = b.x;
        v[k][3] = b.y;
    };
    auto lookup_row = [&](const char *tb, const quad_u2 &c) {
        lookup_quad(tb, std::integral_constant<int, 0>{}, c);
        if constexpr (NCH > 1) lookup_quad(tb, std::integral_constant<int, 1>{}, c);
        if constexpr (NCH > 2) lookup_quad(tb, std::integral_constant<int, 2>{}, c);
        if constexpr (NCH > 3) lookup_quad(tb, std::integral_constant<int, 3>{}, c);
        if constexpr (NCH > 4) lookup_quad(tb, std::integral_constant<int, 4>{}, c);
        if constexpr (NCH > 5) lookup_quad(tb, std::integral_constant<int, 5>{}, c);
        if constexpr (NCH > 6) lookup_quad(tb, std::integral_constant<int, 6>{}, c);
        if constexpr (NCH > 7) lookup_quad(tb, std::integral_constant<int, 7>{}, c);
    };

    auto step = [&](auto J, int64_t q) {
        constexpr int j = decltype(J)::value;
        constexpr int jn = (j + 1) % NBUF, jl = (j + NBUF - 1) % NBUF;
        if ((q % THREADS) == 0) fetch_meta((int)((q / THREADS + 1) & 1), q + THREADS);   // the block after this one
        load_row(std::integral_constant<int, jl>{});     // row q + NBUF - 1; slot jl held row q - 1: consumed
        const double wr = pre_wr;
        double s4[4] = {0.0, 0.0, 0.0, 0.0};             // four independent chains, as the byte pass
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) s4[e] = fma(v[k][e], p[k][e], s4[e]);
        double s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        __builtin_amdgcn_s_setprio(1);
        s = wave_sum_lane63(s);
        if (lane == 63) red[j][wv] = s;
        publish(std::integral_constant<int, jn>{});      // row q + 1's table, published by the same barrier
        __syncthreads();
        read_meta(q + NBUF, q + 1);                      // for the next step; returns under the division
        static_assert(NW == 4, "four wave sums");
        const quad_d2 ra = *reinterpret_cast<const quad_d2 *>(&red[j][0]), rb = *reinterpret_cast<const quad_d2 *>(&red[j][2]);
        const double cf = readlane_f64(weight_over_norm(wr, (ra.x + ra.y) + (rb.x + rb.y)), 0);
        __builtin_amdgcn_s_setprio(0);
        const char *tbn = reinterpret_cast<const char *>(&s_tbl[jn][0]);
        auto upd = [&](auto K) {
            constexpr int k = decltype(K)::value;
            if constexpr (k < NCH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[k][e] = fma(cf, v[k][e], acc[k][e]);
                    asm volatile("" : "+v"(acc[k][e]));  // (pinned, as in the byte pass)
                }
                lookup_quad(tbn, K, cw[jn]);
            }
        };
        upd(std::integral_constant<int, 0>{});
        upd(std::integral_constant<int, 1>{});
        upd(std::integral_constant<int, 2>{});
        upd(std::integral_constant<int, 3>{});
        upd(std::integral_constant<int, 4>{});
        upd(std::integral_constant<int, 5>{});
        upd(std::integral_constant<int, 6>{});
        upd(std::integral_constant<int, 7>{});
    };

    if (deal.nq > 0) {
        __syncthreads();
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
        __syncthreads();
        read_meta(NBUF - 1, 0);
        lookup_row(reinterpret_cast<const char *>(&s_tbl[0][0]), cw[0]);
        for (int64_t q = 0; q < deal.nq; q += NBUF) {
            step(std::integral_constant<int, 0>{}, q);
            step(std::integral_constant<int, 1>{}, q + 1);
            step(std::integral_constant<int, 2>{}, q + 2);
            if constexpr (NBUF > 3) step(std::integral_constant<int, 3>{}, q + 3);
            if constexpr (NBUF > 4) step(std::integral_constant<int, 4>{}, q + 4);
            if constexpr (NBUF > 5) step(std::integral_constant<int, 5>{}, q + 5);
        }
    }
    __syncthreads();
    *fault = s_fault;
}

// K3q  em_iter_quad_kernel: the quad rows' share of one restart's pass; partial row part_row0 + blockIdx.x, and
// {0, list fault} where the byte pass leaves {wide rows met, list fault} (coded_kernels.hpp, CHECK) for the column reduce.
template <int NCH, int NBUF>
__global__ __launch_bounds__(QUAD_THREADS, 2) void em_iter_quad_kernel(
    const uint8_t *__restrict__ qrec, const int64_t *__restrict__ qoff, const int32_t *__restrict__ nquad,
    const int64_t *__restrict__ quad_rows, int64_t n_rows, int64_t R, const double *__restrict__ w,
    const double *__restrict__ props, int H, double *__restrict__ partial, int64_t ldpart, int part_row0,
    const mxm_em_state *__restrict__ state, int run) {
    if (state != nullptr && state[run].done != 0) return;
    const int t = threadIdx.x;
    props += (int64_t)run * H;
    double p[NCH][4], acc[NCH][4];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * (t + k * QUAD_THREADS) + e;
            p[k][e] = (c < H) ? props[c] : 0.0;
            acc[k][e] = 0.0;
        }
    }
    int fault = 0;
    quad_row_pass<NCH, NBUF, true>(qrec, qoff, nquad, quad_rows, n_rows, R, w, p, acc, &fault);
    const int64_t row = (int64_t)part_row0 + blockIdx.x;
    if (t == 0) {
        int *out = reinterpret_cast<int *>(partial + (int64_t)MXM_MAX_WG * ldpart) + 2 * row;
        out[0] = 0;
        out[1] = fault;
    }
    double *dst = partial + row * ldpart;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = 4 * (t + k * QUAD_THREADS);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < H) dst[c + e] = acc[k][e];
    }
}

// K3q+  em_iter_quad_coded_kernel: BOTH row passes of an iteration beside a quad dictionary in one grid.  One after the
// other the leftover pass (the byte-coded rows without quads, the wide rows: 3 % of the rows) is a tail of its own --
// a few thousand rows cannot fill a launch, and a launch that is sized to run beside the quad pass takes as long as it
// (measured: 1.22 + 0.45 ms, and 2.6 ms per iteration where a second stream did not overlap at all).  Here the first
// `nwg_left` workgroups of the grid do the leftover pass and the others the quad pass, each dealt its rows among its
// own kind; the host sizes the two shares by the measured cost of a row so that they finish together, and the whole
// grid is resident at once (two workgroups per CU: 76 KB of LDS each, the two passes' buffers side by side).
// Partial rows: quad workgroup i -> i, leftover workgroup i -> nwg_quad + i (the column reduce's order is the same as
// with two launches).
template <int NCH, int NBUF>
__global__ __launch_bounds__(QUAD_THREADS, 2) void em_iter_quad_coded_kernel(
    const uint8_t *__restrict__ rec, const int64_t *__restrict__ rec_off, const int32_t *__restrict__ ndist, int ldc,
    const int64_t *__restrict__ wide_rows, int64_t n_wide, const int64_t *__restrict__ byte_rows, int64_t n_byte_rows,
    const uint8_t *__restrict__ qrec, const int64_t *__restrict__ qoff, const int32_t *__restrict__ nquad,
    const int64_t *__restrict__ quad_rows, int64_t n_quad_rows, int64_t R, const double *__restrict__ w,
    const double *__restrict__ props, int H, double *__restrict__ partial, int64_t ldpart, int nwg_left,
    const mxm_em_state *__restrict__ state, int run) {
    if (state != nullptr && state[run].done != 0) return;
    const int t = threadIdx.x;
    props += (int64_t)run * H;
    double p[NCH][4], acc[NCH][4];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * (t + k * QUAD_THREADS) + e;
            p[k][e] = (c < H) ? props[c] : 0.0;
            acc[k][e] = 0.0;
        }
    }
    const int nwg_quad = (int)gridDim.x - nwg_left;
    int chk[2] = {0, 0};
    int64_t prow;
    if ((int)blockIdx.x < nwg_left) {                    // workgroup uniform
        bool meta_ready = false;
        coded_row_pass<QUAD_THREADS, NCH, NBUF, true, false, true, true>(rec, rec_off, ndist, ldc, w, wide_rows, n_wide, R, p, acc,
                                                                         meta_ready, chk, byte_rows, n_byte_rows, nquad,
                                                                         (int)blockIdx.x, nwg_left);
        prow = (int64_t)nwg_quad + blockIdx.x;
    } else {
        quad_row_pass<NCH, NBUF, true>(qrec, qoff, nquad, quad_rows, n_quad_rows, R, w, p, acc, &chk[1],
                                       (int)blockIdx.x - nwg_left, nwg_quad);
        prow = (int64_t)blockIdx.x - nwg_left;
    }
    if (t == 0) {
        int *out = reinterpret_cast<int *>(partial + (int64_t)MXM_MAX_WG * ldpart) + 2 * prow;
        out[0] = chk[0];
        out[1] = chk[1];
    }
    double *dst = partial + prow * ldpart;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = 4 * (t + k * QUAD_THREADS);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < H) dst[c + e] = acc[k][e];
    }
}

// Diagnostic (tools/pmc_calibrate_coded.py --quads): the quad pass's loads and nothing else -- a thread's 8 code bytes and
// its 32 bytes of the table, row after row of `quad_rows`, three rows ahead, the rows' offsets and sizes fetched 256 steps
// at a time through LDS as the pass does -- so that a counter pass can state the quad kernel's traffic against a reader of
// exactly its bytes through exactly its access widths, and its time against what those loads reach alone.
__global__ __launch_bounds__(QUAD_THREADS, 2) void diag_stream_quads_kernel(const uint8_t *__restrict__ qrec,
                                                                           const int64_t *__restrict__ qoff,
                                                                           const int32_t *__restrict__ nquad,
                                                                           const int64_t *__restrict__ quad_rows, int64_t n_rows,
                                                                           int64_t R, unsigned int *__restrict__ sink) {
    constexpr int NBUF = 4, THREADS = QUAD_THREADS;
    static_assert(THREADS % NBUF == 0, "a block of steps is a whole number of rounds");
    __shared__ long long s_off[2][THREADS];
    __shared__ int s_nd[2][THREADS];
    const int t = threadIdx.x;
    const row_deal deal(n_rows);
    if (deal.nq <= 0) return;
    quad_u2 x[NBUF];
    quad_u4 ya[NBUF], yb[NBUF];
    unsigned int acc = 0;
    auto fetch_meta = [&](int half, int64_t q0) {
        const int64_t q = q0 + t;
        const int64_t r = quad_rows[deal.row(q)];
        const bool ok = deal.live(q) && r >= 0 && r < R;   // a step past the end, an entry that names no row: nothing is read
        const int nd = ok ? nquad[r] : 0;
        s_off[half][t] = nd > 0 && nd <= QUAD_MAX ? qoff[r] : 0;
        s_nd[half][t] = nd > 0 && nd <= QUAD_MAX ? nd : 0;
    };
    auto load_rec = [&](int slot, int64_t q) {
        const int half = (int)((q / THREADS) & 1), idx = (int)(q % THREADS);
        const long long off = s_off[half][idx];
        const int nd = __builtin_amdgcn_readfirstlane(s_nd[half][idx]);
        const int off_hi = __builtin_amdgcn_readfirstlane((int)(off >> 32)), off_lo = __builtin_amdgcn_readfirstlane((int)off);
        const uint8_t *base = qrec + (((long long)off_hi << 32) | (unsigned int)off_lo);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, nd > 0 ? QUAD_CODE_BYTES : 0, 0x00020000);
        x[slot] = __builtin_amdgcn_raw_buffer_load_b64(rs, t * 8, 0, 2);
        const auto rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base + QUAD_CODE_BYTES), 0, nd * 32, 0x00020000);
        ya[slot] = __builtin_amdgcn_raw_buffer_load_b128(rt, t * 32, 0, 2);
        yb[slot] = __builtin_amdgcn_raw_buffer_load_b128(rt, t * 32, 16, 2);
    };
    fetch_meta(0, 0);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NBUF - 1; ++j) load_rec(j, j);
    const int64_t nblk = (deal.nq + THREADS - 1) / THREADS;
    for (int64_t b = 0; b < nblk; ++b) {
        __syncthreads();                                   // (the half written next was read until the block before this one ended)
        fetch_meta((int)((b + 1) & 1), (b + 1) * THREADS);
        __syncthreads();
        for (int i = 0; i < THREADS; i += NBUF) {
#pragma unroll
            for (int j = 0; j < NBUF; ++j) {
                load_rec((j + NBUF - 1) % NBUF, b * THREADS + i + j + NBUF - 1);
                acc ^= x[j].x ^ x[j].y ^ ya[j].x ^ ya[j].w ^ yb[j].y ^ yb[j].z;
            }
        }
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;                 // (keeps the loads)
}

#endif  // MIXEMT_QUAD_KERNELS_HPP
