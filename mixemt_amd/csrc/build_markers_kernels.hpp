// build_markers_kernels.hpp -- part of libmixemt_hip.so (gfx950); included by mixemt_hip.hip only.
// build_em_matrix (preprocess.py:177-198) from the haplogroups' MARKERS instead of cell by cell.
#ifndef MIXEMT_BUILD_MARKERS_KERNELS_HPP
#define MIXEMT_BUILD_MARKERS_KERNELS_HPP

// ------------------------------------------------------------------------------------------
// K1d build_markers: the same sums, the same order, the same bits as the cell-by-cell kernels (K1 / K1c) --
// formed once per DISTINCT cell value of a row instead of once per cell.
//
// M[r][h] = sum over the row's sites j, in signature order, of  (obs_j == expected(h, s_j)) ? lhit : lmiss
// (prob_for_vars, preprocess.py:86-96).  At a site almost every haplogroup expects the same base, so
//   * the row's "majority" term list tref[j] is what a haplogroup WITHOUT a deviating marker in the read's
//     window adds up -- 85 % of the row's cells;
//   * a haplogroup's cell is decided by the set of sites where its term flips against that list: a 64-bit mask;
//   * the row holds a few dozen DISTINCT masks (median 25): they are deduplicated in an LDS hash table and each
//     one's sum is formed exactly as the reference forms it -- start at 0.0, add the n terms in order --
//     so every cell carries the bits the cell-by-cell kernels produce;
//   * the row is written from that table: what is left is the 43 KB store per row.
//
// Round 3 form (the round-2 kernel ran at 15 ms per 10^6 x 5408 against the ~7 ms its store allows: four rows in
// flight per CU because the per-haplogroup mask array of a whole row sat in LDS, nine workgroup barriers per
// row, ~1000 LDS atomics per row, five dependent global round trips per row):
//   * where the flips come from is split by site.  LIGHT sites (at most `heavy_threshold` deviating
//     haplogroups: most of the 4070) keep their (haplogroup, base) lists, CSR over sites -- a row has ~230 such
//     entries instead of ~1000.  HEAVY sites (a whole clade deviates: up to 2503 haplogroups) have BITMAPS over
//     the haplogroups instead, one per observed base -- the lanes that own a haplogroup test its bit, no scatter;
//   * a WAVE owns a contiguous quarter of the haplogroups for the whole row and takes it in sub-passes of KPS x 128
//     haplogroups: its mask array is private and tiny (KPS KB), scatter -> read needs no workgroup barrier (a
//     wave's LDS operations execute in order), and a CU holds 6-8 rows instead of 4;
//   * the row's light entries are gathered from the tables ONCE into an LDS list that every wave then scans;
//   * the next row's observations and table entries are loaded while the current row is worked on;
//   * four workgroup barriers per row (list ready / masks deduplicated / sums ready / values read).
// Rows with more than 64 observations, more than MKB_LIST light entries, or more than MKB_MAXD distinct non-zero
// masks are appended to `fallback` for the cell-by-cell kernel (3 % of synth-v1 rows: the long ones).
// ------------------------------------------------------------------------------------------
#define MKB_THREADS 256
#ifndef MKB_WAVES
#define MKB_WAVES 6                   // waves per SIMD the dense variant is compiled for (= rows in flight per CU)
#endif
#define MKB_MAXN 64
#define MKB_SLOTS 512                 // hash slots for the row's distinct non-zero masks (8 bytes of LDS each)
#define MKB_MAXD 352                  // ... of which at most this many may fill up (load factor < 0.7); beyond: fallback
                                      // (rows of up to 64 observations: median 25 distinct values, 0.2 % above 256, none above 352 in 4000)
#define MKB_HVCAP 8                   // heavy-site observations of a row whose bitmap words are staged in LDS (96 % of rows have
                                      // at most 8; the rest of a row's are fetched word by word)
#define MKB_LIST 1024                 // light marker entries of a row (gathered in passes of 256); beyond: fallback
#define MKB_LCAP 384                  // ... staged in LDS per wave (the quarter of the haplogroups it owns; 4 bytes each); beyond: fallback
#define SPB_MAXD MKB_MAXD

// EMIT: the row also leaves as a row-dictionary record (coded_kernels.hpp: codes ++ table of P = exp(sum - rowmax),
// here followed by the table of the sums themselves) -- the kernel has the row's distinct values and every
// haplogroup's index into them in hand, so mxm_encode_rows' pass over the dense matrix is not needed; with
// M == nullptr the dense row is not written at all (rows that do not code then go to the fallback list).
struct spb_records {
    uint8_t *rec;
    long long rec_cap;
    int64_t *rec_off;
    int32_t *ndist;
    double *rowmax;
    unsigned long long *stats;      // [0] bytes used, [1] rows without a record
    int ldc;
};

// The marker tables (host: preprocess.HapVarTables.markers()).
struct mkb_tables {
    const uint8_t *maj;             // [S] the base most haplogroups expect at the site
    const double *lhit, *lmiss;     // [S]
    const int32_t *mk_ptr;          // [S+1] CSR over sites of the LIGHT sites' deviating (haplogroup, base) pairs
    const uint16_t *mk_hap;
    const uint8_t *mk_base;
    const int32_t *heavy_id;        // [S] -1, or the site's index into the heavy tables
    const uint8_t *heavy_alt;       // [n_heavy][4]: [1..3] the bases deviating haplogroups expect there (0 = unused)
    const uint32_t *heavy_bits;     // [n_heavy][4][ldw]: bit h of [0] = haplogroup h deviates; of [a] = it expects heavy_alt[a]
    int ldw;                        // words per bitmap, >= 16 * ceil(H / 512)
    int n_heavy;
};

// NCH rounds of 128 haplogroups per wave (ceil(H / 512)); KPS rounds per sub-pass.
// Registers: 80 VGPRs without scratch for the dense variant = 6 waves per SIMD = 6 rows per CU (its 21.5 KB of LDS would
// allow 7); the record variant's exponentials and compaction need more: 4 waves per SIMD as in round 2.
template <int NCH, int KPS, bool EMIT>
__global__ __launch_bounds__(MKB_THREADS, (EMIT ? 4 : MKB_WAVES)) void build_markers_kernel(
    mkb_tables tb, const int64_t *__restrict__ row_ptr, const uint16_t *__restrict__ site, const uint8_t *__restrict__ obs,
    const int64_t *__restrict__ order, int64_t R, int H, double *__restrict__ M, int64_t ldm, int vec_ok,
    int64_t *__restrict__ fallback, unsigned long long *__restrict__ n_fallback, int max_distinct, spb_records out) {
    constexpr int NW = MKB_THREADS / 64, SPT = MKB_SLOTS / MKB_THREADS;
    constexpr int SUBP = (NCH + KPS - 1) / KPS;
    constexpr int SPAN = KPS * 128;                         // haplogroups per wave and sub-pass
    constexpr int NQ = (MKB_MAXD + MKB_THREADS - 1) / MKB_THREADS;
    __shared__ unsigned long long s_mask[NW][SPAN];         // flip masks of the sub-pass's haplogroups (zero in between)
    __shared__ unsigned long long s_key[MKB_SLOTS];         // distinct masks, then their sums (zero between rows)
    __shared__ unsigned int s_list[NW][MKB_LCAP];           // the row's light flips by the wave that owns the haplogroup: haplogroup | observation << 16
    __shared__ int s_lcnt[NW];                              // entries per wave (zero between rows)
    __shared__ unsigned int s_hvq[NW][SUBP];                // per wave and sub-pass: which staged heavy observations have a bit set there
    __shared__ double s_tref[MKB_MAXN], s_talt[MKB_MAXN];
    __shared__ double s_sum0;
    __shared__ int s_flag, s_count;
    __shared__ unsigned short s_clist[MKB_MAXD];            // the occupied slots in the order they were made
    __shared__ unsigned short s_code[EMIT ? MKB_SLOTS : 1]; // slot -> code (1 + its number; records only)
    __shared__ unsigned int s_hv[NW][MKB_HVCAP][NCH * 4];   // per wave: its quarter of the bitmaps the row's heavy observations selected
    __shared__ int s_hvoff[NW][MKB_HVCAP];
    __shared__ double s_wmax[NW];
    __shared__ long long s_off;
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);  // wave-uniform, and the compiler must know it (scalar offsets below)
    // Addresses: everything a lane touches per round is  uniform base + 16 * lane (+ a constant), so the tables' bitmaps
    // and the output row go through buffer descriptors (base and round offset in scalar registers, one VGPR of lane
    // offset for the whole kernel) -- per-round 64-bit addresses in vector registers were what spilled.
    const auto hv_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(tb.heavy_bits), 0,
                                                           tb.n_heavy > 0 ? tb.n_heavy * 16 * tb.ldw : 0, 0x00020000);
    const int row_voff = lane * 16;                         // the bytes of columns 2 * lane, 2 * lane + 1 within a round
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    auto store_row = [&](const double *row_base, const unsigned int (&slots)[NCH], double sum0) {
        // columns past H fall outside the descriptor's range and are dropped by the hardware
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(row_base), 0, H * 8, 0x00020000);
#pragma unroll
        for (int kk = 0; kk < NCH; ++kk) {
            const unsigned int s0 = slots[kk] & 0xffffu, s1 = slots[kk] >> 16;
            const double v0 = s0 == 0xffffu ? sum0 : __longlong_as_double((long long)s_key[s0]);
            const double v1 = s1 == 0xffffu ? sum0 : __longlong_as_double((long long)s_key[s1]);
            const int soff = (wv * NCH + kk) * 1024;         // wave-uniform
            if (vec_ok) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, d2{v0, v1}), rsrc, row_voff, soff, 2);
            } else {                                         // odd leading dimension / unaligned matrix: 8-byte stores
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v0), rsrc, row_voff, soff, 2);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v1), rsrc, row_voff + 8, soff, 2);
            }
            if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four rounds of lookups in flight, not all of them
        }
    };
    auto no_record = [&](int64_t row) {                     // thread 0: the row has no record (dense or fallback)
        if constexpr (EMIT) {
            out.ndist[row] = 0;
            out.rec_off[row] = 0;
            out.rowmax[row] = 0.0;
            atomicAdd(&out.stats[1], 1ull);
        }
    };

    for (int h = lane; h < SPAN; h += 64) s_mask[wv][h] = 0ull;      // once: every row leaves the arrays zeroed
#pragma unroll
    for (int q = 0; q < SPT; ++q) s_key[t + q * MKB_THREADS] = 0ull;
    if (t == 0) {
        s_flag = 0;
        s_count = 0;
    }
    if (t < NW) s_lcnt[t] = 0;

    // ---- the row pipeline: stage A (row index, CSR range: uniform), stage B (site, observation: lane j = observation j),
    // stage C (the site's table entries).  Row i's stage C and row i + 1's stage A / B run while row i - 1 is worked on.
    struct stage_c {
        double lh, lm;
        int beg, len, hvoff, oh;                             // oh = observation byte | (hit of the majority) << 8
    };
    auto load_a = [&](int64_t i, int64_t &r, int64_t &beg, int &n) {
        r = 0;
        beg = 0;
        n = -1;                                              // past the end
        if (i < R) {
            r = order != nullptr ? order[i] : i;
            beg = row_ptr[r];
            const int64_t n64 = row_ptr[r + 1] - beg;
            n = n64 > MKB_MAXN ? MKB_MAXN + 1 : (int)n64;    // > MAXN: the row goes to the fallback list
        }
    };
    auto load_b = [&](int64_t beg, int n, int &s0, int &o) {
        s0 = 0;
        o = 0;
        if (n >= 0 && n <= MKB_MAXN && lane < n) {
            s0 = site[beg + lane];
            o = obs[beg + lane];
        }
    };
    auto load_c = [&](int n, int s0, int o, stage_c &c) {
        c.lh = c.lm = 0.0;
        c.beg = c.len = 0;
        c.hvoff = -1;
        c.oh = 0;
        if (n >= 0 && n <= MKB_MAXN && lane < n) {
            const int mj = tb.maj[s0];
            c.lh = tb.lhit[s0];
            c.lm = tb.lmiss[s0];
            c.beg = tb.mk_ptr[s0];
            c.len = tb.mk_ptr[s0 + 1] - c.beg;
            c.oh = o | ((o == mj) ? 0x100 : 0);
            const int hv = tb.heavy_id[s0];
            if (hv >= 0) {
                int a = -1;                                  // which of the site's bitmaps this observation selects
                if (o == mj) a = 0;                          // the majority hits: EVERY deviating haplogroup flips
                else {                                       // the majority misses: those that expect exactly `o` flip
                    const uint8_t *alt = tb.heavy_alt + hv * 4;
                    if (alt[1] == o) a = 1;
                    else if (alt[2] == o) a = 2;
                    else if (alt[3] == o) a = 3;
                }
                if (a >= 0) c.hvoff = (hv * 4 + a) * tb.ldw;
            }
        }
    };

    int64_t r_cur, beg_cur, r_nxt, beg_nxt;
    int n_cur, n_nxt, s0_nxt, o_nxt;
    stage_c cc;
    {
        int s0, o;
        load_a(blockIdx.x, r_cur, beg_cur, n_cur);
        load_b(beg_cur, n_cur, s0, o);
        load_c(n_cur, s0, o, cc);
        load_a((int64_t)blockIdx.x + gridDim.x, r_nxt, beg_nxt, n_nxt);
        load_b(beg_nxt, n_nxt, s0_nxt, o_nxt);
    }
    __syncthreads();

    for (int64_t i = blockIdx.x; i < R; i += gridDim.x) {
        const int64_t r = r_cur;
        const int n = n_cur;
        // the pipeline moves on: stages A and B of the row after the next here, stage C of the next row once this row's
        // masks are in the table (its eight registers are then only held through the sums and the row store)
        stage_c cn;
        cn.lh = cn.lm = 0.0;
        cn.beg = cn.len = cn.oh = 0;
        cn.hvoff = -1;
        int64_t r_n2, beg_n2;
        int n_n2, s0_n2, o_n2;
        load_a(i + 2 * (int64_t)gridDim.x, r_n2, beg_n2, n_n2);
        load_b(beg_n2, n_n2, s0_n2, o_n2);
        auto advance = [&]() {
            r_cur = r_nxt; beg_cur = beg_nxt; n_cur = n_nxt; cc = cn;
            r_nxt = r_n2; beg_nxt = beg_n2; n_nxt = n_n2; s0_nxt = s0_n2; o_nxt = o_n2;
        };

        // ---- 1. the row's term lists, the light entries' prefix sums (every wave for itself: no barrier) -----------
        int incl = cc.len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        const int total = __shfl(incl, 63, 64);
        if (n > MKB_MAXN || total > MKB_LIST) {              // uniform: the whole workgroup skips the row
            if (t == 0) {
                fallback[atomicAdd(n_fallback, 1ull)] = r;
                no_record(r);
            }
            load_c(n_nxt, s0_nxt, o_nxt, cn);                // (normally issued after barrier B2)
            advance();
            continue;
        }
        const bool mhit = (cc.oh & 0x100) != 0;
        if (wv == 0 && lane < n) {
            s_tref[lane] = mhit ? cc.lh : cc.lm;
            s_talt[lane] = mhit ? cc.lm : cc.lh;
        }
        // ---- 2. the light flips, gathered ONCE into the LDS list (entry e -> its observation by a search over the lanes'
        // prefix sums; haplogroup | observation << 16; 0xffffffff = the entry does not flip)
        for (int e0 = wv * 64; e0 < total; e0 += MKB_THREADS) {     // wave-uniform trip count: the shuffles need every lane
            const int e = e0 + lane;
            int lo = 0, hi = 64;                            // smallest lane with incl > e (six exact halvings of 64)
#pragma unroll
            for (int step = 0; step < 6; ++step) {
                const int mid = (lo + hi) >> 1;
                const int v = __shfl(incl, mid - 1, 64);    // incl of lane mid - 1 (mid >= 1)
                if (v <= e) lo = mid;
                else hi = mid;
            }
            const int j = lo;                               // incl[j-1] <= e < incl[j]
            const int excl = __shfl(incl, j, 64) - __shfl(cc.len, j, 64);
            const int first = __shfl(cc.beg, j, 64);
            const int oh = __shfl(cc.oh, j, 64);
            unsigned int hap = 0xffffu;
            if (e < total) {
                const int idx = first + (e - excl);
                const int base = tb.mk_base[idx];
                const bool flip = (((oh & 0xff) == base) ? 1 : 0) != ((oh >> 8) & 1);
                if (flip) hap = tb.mk_hap[idx];
            }
            // to the list of the wave that owns the haplogroup (one counter bump per wave and bucket, positions by rank)
            const int owner = hap == 0xffffu ? -1 : (int)(hap / (unsigned int)(NCH * 128));
#pragma unroll
            for (int b = 0; b < NW; ++b) {
                const unsigned long long in_b = __builtin_amdgcn_ballot_w64(owner == b);
                if (in_b != 0ull) {                         // uniform
                    int base_pos = 0;
                    if (lane == 0) base_pos = atomicAdd(&s_lcnt[b], __popcll(in_b));
                    base_pos = __builtin_amdgcn_readfirstlane(base_pos);
                    const int pos = base_pos + __popcll(in_b & ((1ull << lane) - 1ull));
                    if (owner == b && pos < MKB_LCAP) s_list[b][pos] = hap | ((unsigned int)j << 16);
                }
            }
        }
        // the heavy sites of the row: every wave parks ITS quarter of the selected bitmaps in LDS -- all loads of the row
        // issued back to back, one L2 round trip beside the list's (fetching them sub-pass by sub-pass cost six in a row:
        // 5 of the kernel's 14.5 ms)
        const unsigned long long hv_lanes = __builtin_amdgcn_ballot_w64(cc.hvoff >= 0);
        const int n_hv = __popcll(hv_lanes);
        const int n_st = n_hv < MKB_HVCAP ? n_hv : MKB_HVCAP;            // uniform
#ifndef MKB_DBG_NO_HEAVY
        if (lane < SUBP) s_hvq[wv][lane] = 0u;
        if (n_st > 0) {
            const int ord = __popcll(hv_lanes & ((1ull << lane) - 1ull));
            if (cc.hvoff >= 0 && ord < MKB_HVCAP) s_hvoff[wv][ord] = cc.hvoff;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            constexpr int WPO = NCH * 4, NU = (MKB_HVCAP * WPO + 63) / 64;     // words per observation, loads per lane
            unsigned int wreg[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int idx = lane + 64 * u, q = idx / WPO, wd = idx - q * WPO;
                wreg[u] = 0u;
                if (q < n_st) wreg[u] = __builtin_amdgcn_raw_buffer_load_b32(hv_rsrc, (s_hvoff[wv][q] + wd) * 4, wv * WPO * 4, 0);
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int idx = lane + 64 * u, q = idx / WPO, wd = idx - q * WPO;
                if (q < n_st) {
                    s_hv[wv][q][wd] = wreg[u];
                    // a heavy site's deviating haplogroups are a few clades = runs of columns: most (observation, sub-pass)
                    // pairs have no bit at all and are skipped below
                    if (wreg[u] != 0u) atomicOr(&s_hvq[wv][(wd >> 2) / KPS], 1u << q);
                }
            }
        }
#endif
        __syncthreads();                                     // B1: list (and term lists) complete

        // ---- 3. per wave: sub-passes over its quarter of the haplogroups -------------------------------------
        const int my_total = s_lcnt[wv] < MKB_LCAP ? s_lcnt[wv] : MKB_LCAP;
        const bool spilled = s_lcnt[0] > MKB_LCAP || s_lcnt[1] > MKB_LCAP || s_lcnt[2] > MKB_LCAP || s_lcnt[3] > MKB_LCAP;   // uniform
        static_assert(NW == 4, "four waves");
        unsigned int slots[NCH];                             // per round: slot of column 2*lane (low half) and 2*lane+1, 0xffff = majority
#pragma unroll
        for (int s = 0; s < SUBP; ++s) {
            const int col_lo = (wv * NCH + s * KPS) * 128;   // first haplogroup of this wave's sub-pass
            // (the last sub-pass may hold fewer than KPS rounds: what lies beyond belongs to the next wave)
            const int KS = (NCH - s * KPS) < KPS ? (NCH - s * KPS) : KPS;     // (a constant once the loop is unrolled)
            unsigned int hlo[KPS][2], hhi[KPS][2];           // the heavy sites' flips of this lane's two haplogroups per round
#pragma unroll
            for (int k = 0; k < KPS; ++k) hlo[k][0] = hlo[k][1] = hhi[k][0] = hhi[k][1] = 0u;
            if (col_lo < H) {
#ifndef MKB_DBG_NO_HEAVY
                // staged observations with a bit in this sub-pass (usually none), then the unstaged ones word by word
                unsigned int qs = __builtin_amdgcn_readfirstlane(s_hvq[wv][s]);
                unsigned long long todo = hv_lanes;          // uniform
                for (int q = 0; todo != 0ull && (qs != 0u || n_hv > MKB_HVCAP); ++q) {
                    const int j = (int)__builtin_ctzll(todo);
                    todo &= todo - 1ull;
                    if (q < MKB_HVCAP) {
                        if (!((qs >> q) & 1u)) continue;
                        qs &= ~(1u << q);
                    }
#pragma unroll
                    for (int k = 0; k < KPS; ++k) {
                        if (k >= KS) continue;
                        const int wd = (s * KPS + k) * 4 + (lane >> 4);
                        unsigned int w;
                        if (q < MKB_HVCAP) w = s_hv[wv][q][wd];
                        else                                 // beyond the staged ones (4 % of rows): word by word
                            w = __builtin_amdgcn_raw_buffer_load_b32(hv_rsrc, (lane >> 4) * 4,
                                                                     (__builtin_amdgcn_readlane(cc.hvoff, j) + (wv * NCH + s * KPS + k) * 4) * 4, 0);
                        const unsigned int two = (w >> ((2 * lane) & 31)) & 3u;
                        if (j < 32) {                        // uniform
                            hlo[k][0] |= (two & 1u) << j;
                            hlo[k][1] |= (two >> 1) << j;
                        } else {
                            hhi[k][0] |= (two & 1u) << (j - 32);
                            hhi[k][1] |= (two >> 1) << (j - 32);
                        }
                    }
                }
#endif
#ifdef MKB_DBG_NO_SCATTER
                for (int e = lane; e < 0; e += 64) {
#else
                for (int e = lane; e < my_total; e += 64) {
#endif
                    const unsigned int ent = s_list[wv][e];
                    const unsigned int local = (ent & 0xffffu) - (unsigned int)col_lo;
                    if (local < (unsigned int)(KS * 128)) atomicOr(&s_mask[wv][local], 1ull << (ent >> 16));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();                 // the wave's own LDS operations execute in order
#pragma unroll
            for (int k = 0; k < KPS; ++k) {
                if (k >= KS) continue;
                const int kk = s * KPS + k;
                const int h0 = (wv * NCH + kk) * 128 + 2 * lane;
                unsigned long long m0 = ((unsigned long long)hhi[k][0] << 32) | hlo[k][0];
                unsigned long long m1 = ((unsigned long long)hhi[k][1] << 32) | hlo[k][1];
                if (col_lo < H) {
                    const unsigned long long l0 = s_mask[wv][k * 128 + 2 * lane], l1 = s_mask[wv][k * 128 + 2 * lane + 1];
                    if ((l0 | l1) != 0ull) {
                        s_mask[wv][k * 128 + 2 * lane] = 0ull;
                        s_mask[wv][k * 128 + 2 * lane + 1] = 0ull;
                    }
                    m0 |= l0;
                    m1 |= l1;
                }
                if (h0 >= H) m0 = 0ull;
                if (h0 + 1 >= H) m1 = 0ull;
#ifdef MKB_DBG_NO_INSERT
                m0 = m1 = 0ull;
#endif
                unsigned int pair = 0xffffffffu;
                if (__builtin_amdgcn_ballot_w64((m0 | m1) != 0ull) != 0ull) {      // clades are runs of columns: many rounds are all-majority
                    // both haplogroups' first probes are issued together (one LDS round trip instead of two in a row);
                    // a probe that hits another mask's slot goes on alone (rare at a load factor below 0.1)
                    auto hash = [](unsigned long long mask) -> unsigned int {
                        const unsigned int fold = (unsigned int)mask ^ ((unsigned int)(mask >> 32) * 0x9E3779B1u);
                        static_assert(MKB_SLOTS == 512, "9 hash bits");
                        return (fold * 0x85EBCA6Bu) >> (32 - 9);
                    };
                    auto claim = [&](unsigned int hs) {      // this lane made the entry: it also gives it its number
                        const int d = atomicAdd(&s_count, 1);
                        if (d < MKB_MAXD) {
                            s_clist[d] = (unsigned short)hs;
                            if constexpr (EMIT) s_code[hs] = (unsigned short)(d + 1);
                        }
                    };
                    // A clade is a run of columns with ONE mask: 60 lanes of a round would hit the same table slot, and the
                    // LDS takes same-address atomics one lane at a time (insertion was 6 of the kernel's 14 ms).  So a lane
                    // whose two masks equal its left neighbour's copies that lane's answer; only the first lane of a run of
                    // equal pairs (its "leader": the nearest lane at or below it whose pair differs from its neighbour's)
                    // goes to the table.
                    const unsigned long long p0 = __shfl_up(m0, 1, 64), p1 = __shfl_up(m1, 1, 64);
                    const bool leads = lane == 0 || p0 != m0 || p1 != m1;
                    const unsigned long long lead_lanes = __builtin_amdgcn_ballot_w64(leads);
                    const int leader = 63 - __builtin_clzll(lead_lanes & ((2ull << lane) - 1ull));     // lane 0 always leads
                    const bool need0 = leads && m0 != 0ull, need1 = leads && m1 != 0ull && m1 != m0;
                    unsigned int hs0 = hash(m0), hs1 = hash(m1);
                    unsigned long long old0 = m0, old1 = m1;
                    if (need0) old0 = atomicCAS(&s_key[hs0], 0ull, m0);
                    if (need1) old1 = atomicCAS(&s_key[hs1], 0ull, m1);
                    if (need0 && old0 == 0ull) claim(hs0);
                    if (need1 && old1 == 0ull) claim(hs1);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const unsigned long long mask = e ? m1 : m0;
                        unsigned long long old = e ? old1 : old0;
                        unsigned int hs = e ? hs1 : hs0;
                        if ((e ? need1 : need0) && old != 0ull && old != mask) {
                            for (int probes = 0;; ++probes) {
                                hs = (hs + 1) & (MKB_SLOTS - 1);
                                old = atomicCAS(&s_key[hs], 0ull, mask);
                                if (old == 0ull) claim(hs);
                                if (old == 0ull || old == mask) break;
                                if (probes >= MKB_SLOTS) {  // full: cannot happen below MKB_MAXD entries, checked below
                                    s_flag = 1;
                                    break;
                                }
                            }
                        }
                        if (e) hs1 = hs;
                        else hs0 = hs;
                    }
                    const unsigned int f0 = need0 ? hs0 : 0xffffu;
                    const unsigned int f1 = need1 ? hs1 : (m1 != 0ull ? f0 : 0xffffu);
                    pair = __shfl(f0 | (f1 << 16), leader, 64);          // (a leader reads its own)
                }
                slots[kk] = pair;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();                                     // B2: every mask of the row is in the table, numbered
        if (t < NW) s_lcnt[t] = 0;                           // (read by everyone before B2, bumped again only after B4)
        load_c(n_nxt, s0_nxt, o_nxt, cn);

        // ---- 4. one in-order sum per distinct mask (prob_for_vars' own loop), by the first D threads -------------------
        const int D = s_count;
        const bool codable = EMIT && (D + 1 <= ENC_MAX_CODES);
        // uniform: too many distinct values for the table -- or, when no dense row is written, for a record
        const bool bad = (s_flag != 0) || spilled || D > max_distinct || D > MKB_MAXD || (EMIT && M == nullptr && !codable);
        double mine[NQ];
        double wmax = -INFINITY;
        if (!bad) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int d = t + q * MKB_THREADS;
                double a = 0.0;
                if (d < D) {
                    const int slot = s_clist[d];
                    const unsigned long long mask = s_key[slot];
#ifdef MKB_DBG_NO_SUMS
                    for (int j = 0; j < 1; ++j) a += ((mask >> j) & 1ull) ? s_talt[j] : s_tref[j];
#else
#pragma unroll 4
                    for (int j = 0; j < n; ++j) a += ((mask >> j) & 1ull) ? s_talt[j] : s_tref[j];
#endif
                    s_key[slot] = (unsigned long long)__double_as_longlong(a);     // nobody else looks at this slot before B3
                    wmax = fmax(wmax, a);
                }
                mine[q] = a;
            }
            if (t == MKB_THREADS - 1) {
                double a = 0.0;
#pragma unroll 4
                for (int j = 0; j < n; ++j) a += s_tref[j];
                s_sum0 = a;
                wmax = fmax(wmax, a);
            }
            if constexpr (EMIT) {
                if (codable) {                               // uniform
                    wmax = wave_max(wmax);
                    if (lane == 0) s_wmax[wv] = wmax;
                    if (t == 0) {                            // the record: codes ++ P table ++ table of the sums
                        const long long bytes = (long long)out.ldc + 16ll * (D + 1);
                        long long off = (long long)atomicAdd(&out.stats[0], (unsigned long long)bytes);
                        if (off + bytes > out.rec_cap) off = -1;
                        s_off = off;
                    }
                }
            }
        }
        __syncthreads();                                     // B3: sums ready
        // ---- 5. the row --------------------------------------------------------------------------------------
        if (!bad) {
            const double sum0 = s_sum0;
            if constexpr (EMIT) {
                const long long off = codable ? s_off : -1;
                if (off >= 0) {
                    const double shift = fmax(fmax(s_wmax[0], s_wmax[1]), fmax(s_wmax[2], s_wmax[3]));   // sums of logs: finite
                    double *ptab = reinterpret_cast<double *>(out.rec + off + out.ldc);
                    double *mtab = ptab + (D + 1);
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int d = t + q * MKB_THREADS;
                        if (d < D) {
                            ptab[d + 1] = exp(mine[q] - shift);
                            mtab[d + 1] = mine[q];
                        }
                    }
                    if (t == MKB_THREADS - 1) {
                        ptab[0] = exp(sum0 - shift);
                        mtab[0] = sum0;
                    }
                    unsigned short *cw = reinterpret_cast<unsigned short *>(out.rec + off);
#pragma unroll
                    for (int kk = 0; kk < NCH; ++kk) {
                        const int h = (wv * NCH + kk) * 128 + 2 * lane;
                        if (h < out.ldc) {
                            const unsigned int s0 = slots[kk] & 0xffffu, s1 = slots[kk] >> 16;
                            const unsigned int c0 = (s0 != 0xffffu) ? s_code[s0] : 0u;
                            const unsigned int c1 = (s1 != 0xffffu) ? s_code[s1] : 0u;
                            cw[h >> 1] = (unsigned short)(c0 | (c1 << 8));
                        }
                    }
                    if (t == 0) {
                        out.rec_off[r] = off;
                        out.ndist[r] = D + 1;
                        out.rowmax[r] = shift;
                    }
                } else if (t == 0) {
                    no_record(r);                           // more than 256 values: the dense row below is its form
                }
            }
#ifndef MKB_DBG_NO_STORE
            if (M != nullptr) store_row(M + r * ldm, slots, sum0);
#endif
        }
        __syncthreads();                                     // B4: everyone has its values: the table can be zeroed
        if (bad) {
#pragma unroll
            for (int q = 0; q < SPT; ++q) s_key[SPT * t + q] = 0ull;
        } else {
            for (int d = t; d < D; d += MKB_THREADS) s_key[s_clist[d]] = 0ull;
        }
        if (t == 0) {
            if (bad) {
                fallback[atomicAdd(n_fallback, 1ull)] = r;
                no_record(r);
            }
            s_flag = 0;
            s_count = 0;
        }
        advance();
        // (zeroes and counters are ordered before the next row's inserts by that row's barrier B1)
    }
}

#endif  // MIXEMT_BUILD_MARKERS_KERNELS_HPP
