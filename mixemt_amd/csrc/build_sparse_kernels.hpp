// build_sparse_kernels.hpp -- part of libmixemt_hip.so (gfx950); included by mixemt_hip.hip only.
// build_em_matrix (preprocess.py:177-198) from the haplogroups' MARKERS instead of cell by cell.
#ifndef MIXEMT_BUILD_SPARSE_KERNELS_HPP
#define MIXEMT_BUILD_SPARSE_KERNELS_HPP

// ------------------------------------------------------------------------------------------
// K1d build_sparse: the same sums, the same order, the same bits as K1 / K1c -- formed once per
// DISTINCT cell value of a row instead of once per cell.
//
// M[r][h] = sum over the row's sites j, in signature order, of  (obs_j == expected(h, s_j)) ? lhit : lmiss
// (prob_for_vars, preprocess.py:86-96).  At a site almost every haplogroup expects the same base
// (Build 17: 113 027 of 22 million (site, haplogroup) pairs differ from their site's majority base), so
//   * the row's "majority" term list tref[j] is what a haplogroup WITHOUT a deviating marker in the
//     read's window adds up -- 85 % of the row's cells;
//   * a haplogroup's cell is decided by the set of sites where its term flips against that list: a 64-bit
//     mask, OR-ed together from the per-site marker lists (CSR over sites: ~1000 entries per row instead of
//     5408 x 37 table lookups);
//   * the row holds a few dozen DISTINCT masks (median 25): they are deduplicated in an LDS hash table and
//     each one's sum is formed exactly as the reference forms it -- start at 0.0, add the n terms in order --
//     so every cell carries the bits the cell-by-cell kernels produce;
//   * the row is written from that table: 43 KB of stores per row instead of 2e11 additions per 10^6 rows.
// Rows with more than 128 observations, or more than 352 distinct non-zero masks, are appended to
// `fallback` for the cell-by-cell kernel (3 % of synth-v1 rows).  Between rows both LDS arrays are zero: a mask
// is cleared by the thread that reads it, a table slot through the compacted list of occupied slots.
// Round 6: rows of 65 .. 128 observations (merged 2 x 150 mates: two thirds of paired-end fragments; 250-bp reads) have
// an instance of their own, W = 2 mask words per haplogroup, launched after the 64-bit one over the same rows -- each
// instance takes the rows of its length class and skips the others (the long instance by looking at 64 rows at a time).
// Its masks are 128 bits: twice the column ranges per row for the same LDS, and a table whose CAS key is a 64-bit TAG of
// the mask with the mask itself beside it -- claimed by tag, the mask compared after the range's barrier; two masks with
// one tag (2^-64 per pair) send the row to the fallback list instead of sharing a slot.
// What bounds it is the LDS unit, shared by the six rows a CU works on: ~750 LDS wave-instructions per row in round
// 2's form, a third of them atomics or 8-byte gathers, with the data FIFO full 40 % of the time
// (profiles/r03/build_sparse_pmc_lds.txt) -- hiding latency (more rows per CU, prefetching the next row's lists, issue
// priority) and cheaper arithmetic (the hash) all measured as noise, while every cut in LDS traffic showed: the marker
// entries looked up once per row instead of once per column range (-8 %), a thread's two neighbouring masks read and
// cleared as one 16-byte access and the second one reusing the first one's slot when equal (-5 %), the two prefix
// sums by DPP instead of ds_bpermute (-5 %), both term lists in one 16-byte read per observation (-4 %), a thread
// taking CONSECUTIVE entries so that one search serves four (-3 %): 13.7 -> 10.5 ms, bit-exact throughout
// (profiles/r03/build_kernel_experiments.txt, G).
// ------------------------------------------------------------------------------------------
#define SPB_THREADS 256
#define SPB_MAXN 64                   // observations per mask word
#ifndef SPB_SLOTS
#define SPB_SLOTS 512                 // hash slots for the row's distinct non-zero masks (8 bytes of LDS each)
#endif
#define SPB_MAXD (SPB_SLOTS * 11 / 16) // ... of which at most this many may fill up (load factor < 0.7); beyond: fallback
                                      // (rows of up to 64 observations: median 25 distinct values, 0.2 % above 256, 0.17 % above 352)
#ifndef SPB_GATHER
#define SPB_GATHER 4                  // marker entries a thread looks up and keeps (x 256 threads = the 1024 entries that 79 % of the rows stay below; 8: no gain)
#endif
#ifndef SPB_GATHER_LONG
#define SPB_GATHER_LONG 16            // ... of the long rows' instance (their marker lists are twice as long and more: 4096 kept)
#endif
#ifndef SPB_SLOTS_LONG
#define SPB_SLOTS_LONG 1024           // ... and its table: merged mates hold twice the distinct values (median 52; 3.7 % of the
#endif
#define SPB_MAXD_LONG (SPB_SLOTS_LONG * 11 / 16)   //     rows above 256, which the 512-slot table's 352 would hand to the fallback kernel)
#define SPB_LONG_CHUNK 64             // rows the long instance examines at a time


// NCH column pairs per thread (ceil(H / 2 / 256)); the haplogroups are taken in PASSES column ranges so that the
// mask array is 1 / PASSES of a row (LDS per workgroup decides how many rows a CU has in flight, and a row is
// mostly latency: dependent table loads, LDS atomics, eight barriers)
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

// (left alone the record variant takes 164 VGPRs for its exponentials -- 3 rows per CU, 21.9 against 15 ms in round 2;
// compiled for 4 / 5 / 6 waves per SIMD the build -> records call takes 18.3 / 18.3 / 16.9 ms at 10^6 rows)
#ifndef SPB_WAVES
#define SPB_WAVES 6                   // waves per SIMD the dense variant is compiled for (80 VGPRs, 100 bytes of scratch): with three
#endif                                // column ranges and a 512-slot table 6 rows fit a CU (round 2: 4) -- 13.7 against 14.7 ms
#ifndef SPB_WAVES_EMIT
#define SPB_WAVES_EMIT 6
#endif
template <int NCH, int PASSES, bool EMIT, int W>
__global__ __launch_bounds__(SPB_THREADS, (W > 1 ? 4 : (EMIT ? SPB_WAVES_EMIT : SPB_WAVES))) void build_sparse_kernel(
    const uint8_t *__restrict__ maj, const double *__restrict__ lhit, const double *__restrict__ lmiss,
    const int32_t *__restrict__ mk_ptr, const uint16_t *__restrict__ mk_hap, const uint8_t *__restrict__ mk_base,
    const int64_t *__restrict__ row_ptr, const uint16_t *__restrict__ site, const uint8_t *__restrict__ obs,
    const int64_t *__restrict__ order, int64_t R, int H, double *__restrict__ M, int64_t ldm, int vec_ok,
    int64_t *__restrict__ fallback, unsigned long long *__restrict__ n_fallback, int max_distinct, spb_records out,
    int long_follows, int max_entries) {
    // W = 1: the rows of up to 64 observations (longer ones: the fallback list, or -- long_follows -- left to the W = 2
    // launch behind this one); W = 2: the rows of 65 .. 128 observations (longer: fallback; shorter: skipped)
    static_assert(W == 1 || W == 2, "64 or 128 observations per row");
    constexpr int MAXN = SPB_MAXN * W, GATHER = W > 1 ? SPB_GATHER_LONG : SPB_GATHER;
    constexpr int SLOTS = W > 1 ? SPB_SLOTS_LONG : SPB_SLOTS, MAXD = W > 1 ? SPB_MAXD_LONG : SPB_MAXD;
    constexpr int NW = SPB_THREADS / 64, SPT = SLOTS / SPB_THREADS;          // slots scanned per thread
    constexpr int KPP = (NCH + PASSES - 1) / PASSES;       // column-pair chunks per pass
    constexpr int SPAN = KPP * 2 * SPB_THREADS;            // haplogroups per pass
    __shared__ __attribute__((aligned(16))) unsigned long long s_dev[SPAN * W];   // flip mask of the pass's haplogroups (zero between passes)
    __shared__ unsigned long long s_key[SLOTS];        // distinct masks (W = 2: their tags), then their sums (zero between rows)
    __shared__ __attribute__((aligned(16))) unsigned long long s_wide[W > 1 ? 2 * SLOTS : 2];   // W = 2: the masks themselves
    __shared__ unsigned short s_list[MAXD + SPB_THREADS];   // occupied slots, compacted
    __shared__ d2 s_t2[MAXN];                              // {majority term, flipped term} of observation j: one 16-byte read
    __shared__ int s_beg[MAXN], s_cum[MAXN + 1];
    __shared__ unsigned char s_obs[MAXN], s_hit[MAXN];
    __shared__ int64_t s_rows[W > 1 ? SPB_LONG_CHUNK : 1];
    __shared__ int s_nrows;
    __shared__ int s_wcnt[NW];
    __shared__ double s_sum0;
    __shared__ int s_flag;
    __shared__ unsigned short s_code[EMIT ? SLOTS : 1];    // slot -> code (1 + compact index; 0 = the majority value)
    __shared__ double s_wmax[NW];
    __shared__ long long s_off;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    auto no_record = [&](int64_t row) {                     // thread 0: the row has no record (dense or fallback)
        if constexpr (EMIT) {
            out.ndist[row] = 0;
            out.rec_off[row] = 0;
            out.rowmax[row] = 0.0;
            atomicAdd(&out.stats[1], 1ull);
        }
    };

    for (int h = t; h < SPAN * W; h += SPB_THREADS) s_dev[h] = 0ull;     // once: every row leaves both arrays zeroed
#pragma unroll
    for (int q = 0; q < SPT; ++q) s_key[t + q * SPB_THREADS] = 0ull;
    if (t == 0) s_flag = 0;

    auto do_row = [&](const int64_t r) {
        // the thread's number, in the long rows' instance opaque once per row: what is derived from it and from the kernel's
        // arguments alone (three predicates per column chunk: `h < H`, `h + 1 < H`, `h < ldc`) is otherwise computed before
        // the row loop and KEPT -- scalar registers parked in vector lanes and read back by v_readlane pairs in front of
        // every store (137 -> 13 spilled SGPRs in the short rows' record instance, which nevertheless ran 4 % SLOWER with it:
        // 16.4 -> 17.1 ms; the long rows' 38.4 -> 37.6 ms on fragments: profiles/r06/opaque_thread_index_ab.txt)
        int tr = t;
        if constexpr (W > 1) asm volatile("" : "+v"(tr));
        const int64_t beg = row_ptr[r];
        const int64_t n64 = row_ptr[r + 1] - beg;
        if (n64 > MAXN) {                                   // uniform: the whole workgroup skips the row
            if (W == 1 && long_follows != 0) return;        // (the long rows' launch takes it, or hands it to the fallback list)
            if (tr == 0) {
                fallback[atomicAdd(n_fallback, 1ull)] = r;
                no_record(r);
            }
            return;
        }
        const int n = (int)n64;
        // ---- 1. the row's term lists -----------------------------------------------------------------
        if (tr < 64) {                                        // wave 0, lane j = observation j (W = 2: and j + 64)
            int carry = 0;
#pragma unroll
            for (int j0 = 0; j0 < MAXN; j0 += 64) {
                const int j = j0 + tr;
                int len = 0;
                if (j < n) {
                    const int s0 = site[beg + j];
                    const unsigned char o = obs[beg + j];
                    const bool hit = (o == maj[s0]);
                    const double lh = lhit[s0], lm = lmiss[s0];
                    s_t2[j] = hit ? d2{lh, lm} : d2{lm, lh};
                    s_obs[j] = o;
                    s_hit[j] = hit ? 1 : 0;
                    const int b = mk_ptr[s0];
                    s_beg[j] = b;
                    len = mk_ptr[s0 + 1] - b;
                }
                const int incl = carry + wave_inclusive_scan_i32(len);   // prefix sums of the marker-list lengths
                s_cum[j + 1] = incl;
                if constexpr (W > 1) carry = __builtin_amdgcn_readlane(incl, 63);
            }
            if (tr == 0) s_cum[0] = 0;
        }
        __syncthreads();
        const int total = s_cum[n];
        if (W > 1 && max_entries > 0 && total > max_entries) {
            // a long row over the control region: thousands of marker entries (rows of 65+ sites among 150-bp reads: 6000-12000)
            // -- the cell-by-cell kernel, whose cost does not depend on them, is the faster one there
            __syncthreads();                                 // (everyone has read the row's total before wave 0 moves on)
            if (tr == 0) {
                fallback[atomicAdd(n_fallback, 1ull)] = r;
                no_record(r);
            }
            return;
        }
        // table slot of the thread's two haplogroups per chunk, PACKED (low / high 16 bits; 0xffff = the majority value):
        // eleven registers instead of twenty-two at H = 5408 -- the kernel is compiled for 80 VGPRs (six rows per CU)
        // and used to spill 34 of them
        unsigned int slot2[NCH];
        // ---- 2a. the row's first GATHER * 256 marker entries are looked up ONCE (site by a search in the prefix
        // sums, haplogroup and base from the global lists, flip against the majority term) and kept packed in
        // registers {haplogroup: 13 bits, observation index: 6 (7) bits}, ~0 = no flip; every column range then only
        // filters and ORs them.  79 % of the rows have no more entries than that; the rest of a longer row's entries
        // are walked per range as before.
        unsigned int cached[GATHER];
        int e_per = 0;                                       // kept entries per thread (uniform): ceil(kept / 256) <= GATHER
        int lo_keep = 0;                                     // the site of this thread's last kept entry: where its walk over
        {                                                    // a long row's further entries starts (they lie behind it)
            int jj[GATHER];
            unsigned int hap[GATHER], base[GATHER];
            // thread t takes entries GATHER * t ..: one search for the first, the others a few steps further on
            const int tot_c = total < GATHER * SPB_THREADS ? total : GATHER * SPB_THREADS;
            e_per = (tot_c + SPB_THREADS - 1) / SPB_THREADS;
            // (round 6) e_per entries per thread, just enough for the row's kept ones: the unrolled loops over the kept entries
            // -- here and once per column range below -- leave at a uniform branch instead of running all GATHER rounds
            // masked (a median row keeps 600 of its 1024 / 1500 of its 4096 slots)
            int lo = 0;
            {
                const int e = e_per * tr;
                int hi = n;
                if (e < tot_c) {
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (s_cum[mid] <= e) lo = mid;
                        else hi = mid;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < GATHER; ++u) {
                jj[u] = -1;
                hap[u] = base[u] = 0u;
            }
#pragma unroll
            for (int u = 0; u < GATHER; ++u) {
                if (u >= e_per) break;                      // uniform
                const int e = e_per * tr + u;
                if (e < tot_c) {
                    while (s_cum[lo + 1] <= e) ++lo;        // e < total = s_cum[n]: stops at lo < n
                }
                jj[u] = (e < tot_c) ? lo : -1;
                const int idx = (e < tot_c) ? s_beg[lo] + (e - s_cum[lo]) : 0;
                hap[u] = mk_hap[idx];
                base[u] = mk_base[idx];
            }
#pragma unroll
            for (int u = 0; u < GATHER; ++u) {
                cached[u] = 0xffffffffu;
                if (u >= e_per) continue;
                if (jj[u] >= 0) {
                    const bool hit = (s_obs[jj[u]] == base[u]);
                    if ((hit ? 1 : 0) != s_hit[jj[u]]) cached[u] = hap[u] | ((unsigned int)jj[u] << 13);
                }
            }
            lo_keep = lo;
        }
#pragma unroll
        for (int pass = 0; pass < PASSES; ++pass) {
            const int h_lo = pass * SPAN;
            if (h_lo >= H) {
#pragma unroll
                for (int k = pass * KPP; k < NCH && k < (pass + 1) * KPP; ++k) slot2[k] = 0xffffffffu;
                continue;
            }
            // ---- 2. OR the flips into the masks of this pass's haplogroups: the kept entries first, then what a long
            // row has beyond them.  The row's marker lists are ONE flat range (entry e -> its site by a search in the
            // prefix sums): a wave per site with its lanes on the list measured slower (22.4 against 16.4 ms at 10^6
            // rows), the per-site table loads then queue up behind each other
#pragma unroll
            for (int u = 0; u < GATHER; ++u) {
                if (u >= e_per) break;                      // uniform
                const unsigned int local = (cached[u] & 0x1fffu) - (unsigned int)h_lo;
                if (cached[u] != 0xffffffffu && local < (unsigned int)SPAN) {
                    const unsigned int jb = cached[u] >> 13;
                    atomicOr(&s_dev[local * W + (W > 1 ? (jb >> 6) : 0u)], 1ull << (jb & 63u));
                }
            }
            // (a thread's further entries t + 256 i come in ascending order, all behind its kept ones: their sites are found
            // by walking on from lo_keep -- a binary search per entry and column range was most of a long row's time:
            // rows over the control region hold 6000-12000 entries, profiles/r06/build_long_rows.txt)
            int lo = lo_keep;
            for (int e0 = tr + SPB_THREADS * GATHER; e0 < total; e0 += SPB_THREADS * GATHER) {
                int jj[GATHER];
                unsigned int hap[GATHER], base[GATHER];
#pragma unroll
                for (int u = 0; u < GATHER; ++u) {
                    const int e = e0 + u * SPB_THREADS;
                    if (e < total) {
                        while (s_cum[lo + 1] <= e) ++lo;    // e < total = s_cum[n]: stops at lo < n
                    }
                    jj[u] = (e < total) ? lo : -1;
                    const int idx = (e < total) ? s_beg[lo] + (e - s_cum[lo]) : 0;
                    hap[u] = mk_hap[idx];
                    base[u] = mk_base[idx];
                }
#pragma unroll
                for (int u = 0; u < GATHER; ++u) {
                    const unsigned int local = hap[u] - (unsigned int)h_lo;
                    if (jj[u] >= 0 && local < (unsigned int)SPAN) {
                        const bool hit = (s_obs[jj[u]] == base[u]);
                        if ((hit ? 1 : 0) != s_hit[jj[u]]) atomicOr(&s_dev[local * W + (W > 1 ? (jj[u] >> 6) : 0)], 1ull << (jj[u] & 63));
                    }
                }
            }
            __syncthreads();
            // ---- 3. distinct non-zero masks -> table slots; the masks are zeroed again on the way -------
            if constexpr (W == 1) {
#pragma unroll
            for (int k = pass * KPP; k < NCH && k < (pass + 1) * KPP; ++k) {
                // the thread's two haplogroups are neighbours: both masks in one 16-byte read, zeroed by one write;
                // the second one takes the first one's slot when they are equal (haplogroups of one clade)
                typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
                const int h0 = 2 * (tr + k * SPB_THREADS);
                ull2 both = {0ull, 0ull};
                if (h0 < H) {
                    both = *reinterpret_cast<const ull2 *>(&s_dev[h0 - h_lo]);
                    if (h0 + 1 >= H) both.y = 0ull;
                    if ((both.x | both.y) != 0ull) *reinterpret_cast<ull2 *>(&s_dev[h0 - h_lo]) = ull2{0ull, 0ull};
                }
                int sl_pair[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int h = 2 * (tr + k * SPB_THREADS) + e;
                    int sl = -1;
                    if (h < H) {
                        const unsigned long long mask = e == 0 ? both.x : both.y;
                        if (e == 1 && mask != 0ull && mask == both.x) {
                            sl = sl_pair[0];
                        } else if (mask != 0ull) {
                            unsigned int hs = (unsigned int)((mask ^ (mask >> 29)) * 0x9E3779B97F4A7C15ull >> 40) & (SLOTS - 1);
                            for (int probes = 0;; ++probes) {
                                const unsigned long long old = atomicCAS(&s_key[hs], 0ull, mask);
                                if (old == 0ull || old == mask) break;
                                hs = (hs + 1) & (SLOTS - 1);
                                if (probes >= SLOTS) {  // full: cannot happen below MAXD entries, checked next
                                    s_flag = 1;
                                    break;
                                }
                            }
                            sl = (int)hs;
                        }
                    }
                    sl_pair[e] = sl;
                }
                slot2[k] = ((unsigned int)sl_pair[0] & 0xffffu) | ((unsigned int)sl_pair[1] << 16);
            }
            __syncthreads();                                 // masks read and zeroed: the next pass may scatter
            } else {
            // 128-bit masks: the table is claimed by a 64-bit TAG of the mask (compare-and-swap works on one word), the
            // claimer leaves the mask beside it, and whoever found its tag already there compares the mask after the
            // barrier: two masks with one tag send the row to the fallback list (s_flag)
            typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
            ull2 keep[KPP][2];
            bool mine_claim[KPP][2];
#pragma unroll
            for (int kq = 0; kq < KPP; ++kq) {
                const int k = pass * KPP + kq;
                int sl_pair[2] = {-1, -1};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    keep[kq][e] = ull2{0ull, 0ull};
                    mine_claim[kq][e] = true;
                    const int h = 2 * (tr + k * SPB_THREADS) + e;
                    if (k < NCH && h < H) {
                        const ull2 m = *reinterpret_cast<const ull2 *>(&s_dev[(h - h_lo) * 2]);
                        if ((m.x | m.y) != 0ull) {
                            *reinterpret_cast<ull2 *>(&s_dev[(h - h_lo) * 2]) = ull2{0ull, 0ull};
                            keep[kq][e] = m;
                            if (e == 1 && m.x == keep[kq][0].x && m.y == keep[kq][0].y) {
                                sl_pair[1] = sl_pair[0];
                            } else {
                                unsigned long long tag = (m.x ^ (m.x >> 29)) * 0x9E3779B97F4A7C15ull ^ (m.y ^ (m.y >> 31)) * 0xC2B2AE3D27D4EB4Full;
                                tag ^= tag >> 32;
                                if (tag == 0ull) tag = 1ull;
                                unsigned int hs = (unsigned int)(tag >> 40) & (SLOTS - 1);
                                for (int probes = 0;; ++probes) {
                                    // (reading the slot before the compare-and-swap -- equal masks of a clade queue up on one
                                    // LDS word -- measured 3 % SLOWER here, unlike in the quad encoder: profiles/r06/experiments.md 2)
                                    const unsigned long long old = atomicCAS(&s_key[hs], 0ull, tag);
                                    if (old == 0ull) {
                                        *reinterpret_cast<ull2 *>(&s_wide[2 * hs]) = m;
                                        break;
                                    }
                                    if (old == tag) {
                                        mine_claim[kq][e] = false;
                                        break;
                                    }
                                    hs = (hs + 1) & (SLOTS - 1);
                                    if (probes >= SLOTS) {
                                        s_flag = 1;
                                        break;
                                    }
                                }
                                sl_pair[e] = (int)hs;
                            }
                        }
                    }
                }
                if (k < NCH) slot2[k] = ((unsigned int)sl_pair[0] & 0xffffu) | ((unsigned int)sl_pair[1] << 16);
            }
            __syncthreads();                                 // masks read and zeroed, every claimed slot holds its mask
#pragma unroll
            for (int kq = 0; kq < KPP; ++kq) {
                const int k = pass * KPP + kq;
                if (k < NCH) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const unsigned int sl = e == 0 ? (slot2[k] & 0xffffu) : (slot2[k] >> 16);
                        if (!mine_claim[kq][e] && sl != 0xffffu) {
                            const ull2 have = *reinterpret_cast<const ull2 *>(&s_wide[2 * sl]);
                            if (have.x != keep[kq][e].x || have.y != keep[kq][e].y) s_flag = 1;
                        }
                    }
                }
            }
            }
        }
        // ---- 4. compact the occupied slots, one in-order sum per distinct mask (prob_for_vars' own loop) ----
        unsigned long long kk[SPT];
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            kk[q] = s_key[SPT * tr + q];
            cnt += (kk[q] != 0ull) ? 1 : 0;
        }
        const int incl = wave_inclusive_scan_i32(cnt);
        if (lane == 63) s_wcnt[wv] = incl;
        __syncthreads();
        int base_d = 0, D = 0;
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            if (q < wv) base_d += s_wcnt[q];
            D += s_wcnt[q];
        }
        // a record with byte codes up to 256 distinct values, with 16-bit codes beyond ("wide" records: round 6 -- merged
        // mates hold more than 256 values in 3.7 % of the rows, which used to leave for the dense-slab detour)
        const bool wide_rec = EMIT && (D + 1 > ENC_MAX_CODES);
        const bool codable = EMIT && (D + 1 <= ENC_MAX_WIDE);
        // uniform: too many distinct values for the table -- or, when no dense row is written, for a record
        const bool bad = (s_flag != 0) || D > max_distinct || (EMIT && M == nullptr && !codable);
        if (!bad) {
            int d = base_d + incl - cnt;
#pragma unroll
            for (int q = 0; q < SPT; ++q) {
                if (kk[q] != 0ull) {
                    if constexpr (EMIT) s_code[SPT * tr + q] = (unsigned short)(d + 1);
                    s_list[d++] = (unsigned short)(SPT * tr + q);
                }
            }
        }
        __syncthreads();
        if (bad) {
            // more distinct values than the table is sized for: clear it, hand the row to the cell-by-cell kernel
#pragma unroll
            for (int q = 0; q < SPT; ++q) s_key[SPT * tr + q] = 0ull;
            if (tr == 0) {
                fallback[atomicAdd(n_fallback, 1ull)] = r;
                s_flag = 0;
                no_record(r);
            }
            __syncthreads();
            return;
        }
        double mine[(MAXD + SPB_THREADS - 1) / SPB_THREADS];
        double wmax = -INFINITY;
#pragma unroll
        for (int q = 0; q < (MAXD + SPB_THREADS - 1) / SPB_THREADS; ++q) {
            const int d = tr + q * SPB_THREADS;
            double a = 0.0;
            if (d < D) {
                if constexpr (W == 1) {
                    const unsigned long long mask = s_key[s_list[d]];
#pragma unroll 4
                    for (int j = 0; j < n; ++j) {
                        const d2 tt = s_t2[j];
                        a += ((mask >> j) & 1ull) ? tt.y : tt.x;
                    }
                } else {
                    const unsigned long long lo = s_wide[2 * s_list[d]], hi = s_wide[2 * s_list[d] + 1];
                    const int n_lo = n < 64 ? n : 64;
#pragma unroll 4
                    for (int j = 0; j < n_lo; ++j) {
                        const d2 tt = s_t2[j];
                        a += ((lo >> j) & 1ull) ? tt.y : tt.x;
                    }
#pragma unroll 4
                    for (int j = 64; j < n; ++j) {
                        const d2 tt = s_t2[j];
                        a += ((hi >> (j - 64)) & 1ull) ? tt.y : tt.x;
                    }
                }
            }
            mine[q] = a;
        }
        if (tr == SPB_THREADS - 1) {
            double a = 0.0;
#pragma unroll 4
            for (int j = 0; j < n; ++j) a += s_t2[j].x;
            s_sum0 = a;
            wmax = a;
        }
        if constexpr (EMIT) {
            if (codable) {                                  // uniform
#pragma unroll
                for (int q = 0; q < (MAXD + SPB_THREADS - 1) / SPB_THREADS; ++q)
                    if (tr + q * SPB_THREADS < D) wmax = fmax(wmax, mine[q]);
                wmax = wave_max(wmax);
                if (lane == 0) s_wmax[wv] = wmax;
                if (tr == 0) {                               // the record: codes ++ P table ++ table of the sums
                    const long long bytes = (long long)out.ldc * (wide_rec ? 2 : 1) + 16ll * (D + 1);
#ifdef RECORDS_FIXED_SLOTS                                  // (timing experiment: no shared bump pointer)
                    long long off = (long long)r * ((long long)out.ldc + 16ll * (MAXD + 1));
#else
                    long long off = (long long)atomicAdd(&out.stats[0], (unsigned long long)bytes);
#endif
                    if (off + bytes > out.rec_cap) off = -1;
                    s_off = off;
                }
            }
        }
        __syncthreads();                                     // every mask has been read
#pragma unroll
        for (int q = 0; q < (MAXD + SPB_THREADS - 1) / SPB_THREADS; ++q) {
            const int d = tr + q * SPB_THREADS;
            if (d < D) s_key[s_list[d]] = (unsigned long long)__double_as_longlong(mine[q]);
        }
        __syncthreads();
        // ---- 5. the row -----------------------------------------------------------------------------
        const double sum0 = s_sum0;
        if constexpr (EMIT) {
            const long long off = codable ? s_off : -1;
            if (off >= 0) {
                const double shift = fmax(fmax(s_wmax[0], s_wmax[1]), fmax(s_wmax[2], s_wmax[3]));   // sums of logs: finite
                double *ptab = reinterpret_cast<double *>(out.rec + off + out.ldc * (wide_rec ? 2 : 1));
                double *mtab = ptab + (D + 1);
#pragma unroll
                for (int q = 0; q < (MAXD + SPB_THREADS - 1) / SPB_THREADS; ++q) {
                    const int d = tr + q * SPB_THREADS;
                    if (d < D) {
                        ptab[d + 1] = exp(mine[q] - shift);
                        mtab[d + 1] = mine[q];
                    }
                }
                if (tr == SPB_THREADS - 1) {
                    ptab[0] = exp(sum0 - shift);
                    mtab[0] = sum0;
                }
                unsigned short *cw = reinterpret_cast<unsigned short *>(out.rec + off);
                unsigned int *cw32 = reinterpret_cast<unsigned int *>(out.rec + off);
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const int h = 2 * (tr + k * SPB_THREADS);
                    if (h < out.ldc) {
                        const unsigned int sa = slot2[k] & 0xffffu, sb = slot2[k] >> 16;
                        const unsigned int c0 = (h < H && sa != 0xffffu) ? s_code[sa] : 0u;
                        const unsigned int c1 = (h + 1 < H && sb != 0xffffu) ? s_code[sb] : 0u;
                        if (wide_rec) cw32[tr + k * SPB_THREADS] = c0 | (c1 << 16);      // (uniform)
                        else cw[tr + k * SPB_THREADS] = (unsigned short)(c0 | (c1 << 8));
                    }
                }
                if (tr == 0) {
                    out.rec_off[r] = off;
                    out.ndist[r] = D + 1;
                    out.rowmax[r] = shift;
                }
            } else if (tr == 0) {
                no_record(r);                               // the record buffer is full: the dense row below is its form
            }
        }
        // the row goes out through ONE descriptor over it (scalar registers) with the thread's offset t * 16 and the
        // chunk as an immediate: as eleven 64-bit addresses per thread the compiler hoisted them out of the row loop and
        // spilled them (22 of the kernel's 80 VGPRs; ScratchSize 100 -> 0).  A store past the row's H doubles is dropped.
        if (M != nullptr) {                                  // uniform
            typedef unsigned int su4 __attribute__((ext_vector_type(4)));
            typedef unsigned int su2 __attribute__((ext_vector_type(2)));
            const auto mrs = __builtin_amdgcn_make_buffer_rsrc(M + r * ldm, 0, H * 8, 0x00020000);
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const unsigned int sa = slot2[k] & 0xffffu, sb = slot2[k] >> 16;
                const double v0 = sa == 0xffffu ? sum0 : __longlong_as_double((long long)s_key[sa]);
                const double v1 = sb == 0xffffu ? sum0 : __longlong_as_double((long long)s_key[sb]);
                if (vec_ok && ((H & 1) == 0)) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(su4, d2{v0, v1}), mrs, tr * 16, k * SPB_THREADS * 16, 2 /* nt */);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(su2, v0), mrs, tr * 16, k * SPB_THREADS * 16, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(su2, v1), mrs, tr * 16, k * SPB_THREADS * 16 + 8, 0);
                }
            }
        }
        __syncthreads();                                     // everyone has its values: the table can be zeroed
        for (int d = tr; d < D; d += SPB_THREADS) s_key[s_list[d]] = 0ull;
        __syncthreads();
    };

    if constexpr (W == 1) {
        for (int64_t i = blockIdx.x; i < R; i += gridDim.x) do_row(order != nullptr ? order[i] : i);
    } else {
        // the long rows are few among short reads and most among merged mates: 64 rows are looked at at a time (wave 0,
        // a lane each) and the ones of this instance's length class taken in ascending order
        for (int64_t i0 = (int64_t)blockIdx.x * SPB_LONG_CHUNK; i0 < R; i0 += (int64_t)gridDim.x * SPB_LONG_CHUNK) {
            if (t < 64) {
                const int64_t i = i0 + t;
                int64_t r = -1;
                bool want = false;
                if (i < R) {
                    r = order != nullptr ? order[i] : i;
                    want = (row_ptr[r + 1] - row_ptr[r]) > SPB_MAXN;
                }
                const unsigned long long votes = __ballot(want);
                if (want) s_rows[__popcll(votes & ((1ull << t) - 1ull))] = r;
                if (t == 0) s_nrows = __popcll(votes);
            }
            __syncthreads();
            const int n_here = s_nrows;                      // uniform
            for (int e = 0; e < n_here; ++e) do_row(s_rows[e]);
            __syncthreads();                                 // (the list is read to its end before the next chunk fills it)
        }
    }
}

#endif  // MIXEMT_BUILD_SPARSE_KERNELS_HPP
