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
// Rows with more than 64 observations, or more than 704 distinct non-zero masks, are appended to
// `fallback` for the cell-by-cell kernel (3 % of synth-v1 rows).  Between rows both LDS arrays are zero: a mask
// is cleared by the thread that reads it, a table slot through the compacted list of occupied slots.
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
#define SPB_MAXN 64
#ifndef SPB_SLOTS
#define SPB_SLOTS 512                 // hash slots for the row's distinct non-zero masks (8 bytes of LDS each)
#endif
#define SPB_MAXD (SPB_SLOTS * 11 / 16) // ... of which at most this many may fill up (load factor < 0.7); beyond: fallback
                                      // (rows of up to 64 observations: median 25 distinct values, 0.2 % above 256, 0.17 % above 352)
#ifndef SPB_GATHER
#define SPB_GATHER 4                  // marker entries a thread looks up and keeps (x 256 threads = the 1024 entries that 79 % of the rows stay below; 8: no gain)
#endif

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
template <int NCH, int PASSES, bool EMIT>
__global__ __launch_bounds__(SPB_THREADS, (EMIT ? SPB_WAVES_EMIT : SPB_WAVES)) void build_sparse_kernel(
    const uint8_t *__restrict__ maj, const double *__restrict__ lhit, const double *__restrict__ lmiss,
    const int32_t *__restrict__ mk_ptr, const uint16_t *__restrict__ mk_hap, const uint8_t *__restrict__ mk_base,
    const int64_t *__restrict__ row_ptr, const uint16_t *__restrict__ site, const uint8_t *__restrict__ obs,
    const int64_t *__restrict__ order, int64_t R, int H, double *__restrict__ M, int64_t ldm, int vec_ok,
    int64_t *__restrict__ fallback, unsigned long long *__restrict__ n_fallback, int max_distinct, spb_records out) {
    constexpr int NW = SPB_THREADS / 64, SPT = SPB_SLOTS / SPB_THREADS;      // slots scanned per thread
    constexpr int KPP = (NCH + PASSES - 1) / PASSES;       // column-pair chunks per pass
    constexpr int SPAN = KPP * 2 * SPB_THREADS;            // haplogroups per pass
    __shared__ __attribute__((aligned(16))) unsigned long long s_dev[SPAN];   // flip mask of the pass's haplogroups (zero between passes)
    __shared__ unsigned long long s_key[SPB_SLOTS];        // distinct masks, then their sums (zero between rows)
    __shared__ unsigned short s_list[SPB_MAXD + SPB_THREADS];   // occupied slots, compacted
    __shared__ d2 s_t2[SPB_MAXN];                          // {majority term, flipped term} of observation j: one 16-byte read
    __shared__ int s_beg[SPB_MAXN], s_cum[SPB_MAXN + 1];
    __shared__ unsigned char s_obs[SPB_MAXN], s_hit[SPB_MAXN];
    __shared__ int s_wcnt[NW];
    __shared__ double s_sum0;
    __shared__ int s_flag;
    __shared__ unsigned short s_code[EMIT ? SPB_SLOTS : 1];    // slot -> code (1 + compact index; 0 = the majority value)
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

    for (int h = t; h < SPAN; h += SPB_THREADS) s_dev[h] = 0ull;     // once: every row leaves both arrays zeroed
#pragma unroll
    for (int q = 0; q < SPT; ++q) s_key[t + q * SPB_THREADS] = 0ull;
    if (t == 0) s_flag = 0;

    for (int64_t i = blockIdx.x; i < R; i += gridDim.x) {
        const int64_t r = order != nullptr ? order[i] : i;
        const int64_t beg = row_ptr[r];
        const int64_t n64 = row_ptr[r + 1] - beg;
        if (n64 > SPB_MAXN) {                               // uniform: the whole workgroup skips the row
            if (t == 0) {
                fallback[atomicAdd(n_fallback, 1ull)] = r;
                no_record(r);
            }
            continue;
        }
        const int n = (int)n64;
        // ---- 1. the row's term lists -----------------------------------------------------------------
        if (t < 64) {                                        // wave 0, lane j = observation j
            int len = 0;
            if (t < n) {
                const int s0 = site[beg + t];
                const unsigned char o = obs[beg + t];
                const bool hit = (o == maj[s0]);
                const double lh = lhit[s0], lm = lmiss[s0];
                s_t2[t] = hit ? d2{lh, lm} : d2{lm, lh};
                s_obs[t] = o;
                s_hit[t] = hit ? 1 : 0;
                const int b = mk_ptr[s0];
                s_beg[t] = b;
                len = mk_ptr[s0 + 1] - b;
            }
            const int incl = wave_inclusive_scan_i32(len);   // prefix sums of the marker-list lengths
            s_cum[t + 1] = incl;
            if (t == 0) s_cum[0] = 0;
        }
        __syncthreads();
        const int total = s_cum[n];
        // table slot of the thread's two haplogroups per chunk, PACKED (low / high 16 bits; 0xffff = the majority value):
        // eleven registers instead of twenty-two at H = 5408 -- the kernel is compiled for 80 VGPRs (six rows per CU)
        // and used to spill 34 of them
        unsigned int slot2[NCH];
        // ---- 2a. the row's first SPB_GATHER * 256 marker entries are looked up ONCE (site by a search in the prefix
        // sums, haplogroup and base from the global lists, flip against the majority term) and kept packed in
        // registers {haplogroup: 13 bits, observation index: 6 bits}, ~0 = no flip; every column range then only
        // filters and ORs them.  79 % of the rows have no more entries than that; the rest of a longer row's entries
        // are walked per range as before.
        unsigned int cached[SPB_GATHER];
        {
            int jj[SPB_GATHER];
            unsigned int hap[SPB_GATHER], base[SPB_GATHER];
            // thread t takes entries SPB_GATHER * t ..: one search for the first, the others a few steps further on
            const int tot_c = total < SPB_GATHER * SPB_THREADS ? total : SPB_GATHER * SPB_THREADS;
            int lo = 0;
            {
                const int e = SPB_GATHER * t;
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
            for (int u = 0; u < SPB_GATHER; ++u) {
                const int e = SPB_GATHER * t + u;
                if (e < tot_c) {
                    while (s_cum[lo + 1] <= e) ++lo;        // e < total = s_cum[n]: stops at lo < n
                }
                jj[u] = (e < tot_c) ? lo : -1;
                const int idx = (e < tot_c) ? s_beg[lo] + (e - s_cum[lo]) : 0;
                hap[u] = mk_hap[idx];
                base[u] = mk_base[idx];
            }
#pragma unroll
            for (int u = 0; u < SPB_GATHER; ++u) {
                cached[u] = 0xffffffffu;
                if (jj[u] >= 0) {
                    const bool hit = (s_obs[jj[u]] == base[u]);
                    if ((hit ? 1 : 0) != s_hit[jj[u]]) cached[u] = hap[u] | ((unsigned int)jj[u] << 13);
                }
            }
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
            for (int u = 0; u < SPB_GATHER; ++u) {
                const unsigned int local = (cached[u] & 0x1fffu) - (unsigned int)h_lo;
                if (cached[u] != 0xffffffffu && local < (unsigned int)SPAN) atomicOr(&s_dev[local], 1ull << (cached[u] >> 13));
            }
            for (int e0 = t + SPB_THREADS * SPB_GATHER; e0 < total; e0 += SPB_THREADS * SPB_GATHER) {
                int jj[SPB_GATHER];
                unsigned int hap[SPB_GATHER], base[SPB_GATHER];
#pragma unroll
                for (int u = 0; u < SPB_GATHER; ++u) {
                    const int e = e0 + u * SPB_THREADS;
                    int lo = 0, hi = n;                     // s_cum[lo] <= e < s_cum[hi]
                    if (e < total) {
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (s_cum[mid] <= e) lo = mid;
                            else hi = mid;
                        }
                    }
                    jj[u] = (e < total) ? lo : -1;
                    const int idx = (e < total) ? s_beg[lo] + (e - s_cum[lo]) : 0;
                    hap[u] = mk_hap[idx];
                    base[u] = mk_base[idx];
                }
#pragma unroll
                for (int u = 0; u < SPB_GATHER; ++u) {
                    const unsigned int local = hap[u] - (unsigned int)h_lo;
                    if (jj[u] >= 0 && local < (unsigned int)SPAN) {
                        const bool hit = (s_obs[jj[u]] == base[u]);
                        if ((hit ? 1 : 0) != s_hit[jj[u]]) atomicOr(&s_dev[local], 1ull << jj[u]);
                    }
                }
            }
            __syncthreads();
            // ---- 3. distinct non-zero masks -> table slots; the masks are zeroed again on the way -------
#pragma unroll
            for (int k = pass * KPP; k < NCH && k < (pass + 1) * KPP; ++k) {
                // the thread's two haplogroups are neighbours: both masks in one 16-byte read, zeroed by one write;
                // the second one takes the first one's slot when they are equal (haplogroups of one clade)
                typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
                const int h0 = 2 * (t + k * SPB_THREADS);
                ull2 both = {0ull, 0ull};
                if (h0 < H) {
                    both = *reinterpret_cast<const ull2 *>(&s_dev[h0 - h_lo]);
                    if (h0 + 1 >= H) both.y = 0ull;
                    if ((both.x | both.y) != 0ull) *reinterpret_cast<ull2 *>(&s_dev[h0 - h_lo]) = ull2{0ull, 0ull};
                }
                int sl_pair[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int h = 2 * (t + k * SPB_THREADS) + e;
                    int sl = -1;
                    if (h < H) {
                        const unsigned long long mask = e == 0 ? both.x : both.y;
                        if (e == 1 && mask != 0ull && mask == both.x) {
                            sl = sl_pair[0];
                        } else if (mask != 0ull) {
                            unsigned int hs = (unsigned int)((mask ^ (mask >> 29)) * 0x9E3779B97F4A7C15ull >> 40) & (SPB_SLOTS - 1);
                            for (int probes = 0;; ++probes) {
                                const unsigned long long old = atomicCAS(&s_key[hs], 0ull, mask);
                                if (old == 0ull || old == mask) break;
                                hs = (hs + 1) & (SPB_SLOTS - 1);
                                if (probes >= SPB_SLOTS) {  // full: cannot happen below SPB_MAXD entries, checked next
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
        }
        // ---- 4. compact the occupied slots, one in-order sum per distinct mask (prob_for_vars' own loop) ----
        unsigned long long kk[SPT];
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            kk[q] = s_key[SPT * t + q];
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
        const bool codable = EMIT && (D + 1 <= ENC_MAX_CODES);
        // uniform: too many distinct values for the table -- or, when no dense row is written, for a record
        const bool bad = (s_flag != 0) || D > max_distinct || (EMIT && M == nullptr && !codable);
        if (!bad) {
            int d = base_d + incl - cnt;
#pragma unroll
            for (int q = 0; q < SPT; ++q) {
                if (kk[q] != 0ull) {
                    if constexpr (EMIT) s_code[SPT * t + q] = (unsigned short)(d + 1);
                    s_list[d++] = (unsigned short)(SPT * t + q);
                }
            }
        }
        __syncthreads();
        if (bad) {
            // more distinct values than the table is sized for: clear it, hand the row to the cell-by-cell kernel
#pragma unroll
            for (int q = 0; q < SPT; ++q) s_key[SPT * t + q] = 0ull;
            if (t == 0) {
                fallback[atomicAdd(n_fallback, 1ull)] = r;
                s_flag = 0;
                no_record(r);
            }
            __syncthreads();
            continue;
        }
        double mine[(SPB_MAXD + SPB_THREADS - 1) / SPB_THREADS];
        double wmax = -INFINITY;
#pragma unroll
        for (int q = 0; q < (SPB_MAXD + SPB_THREADS - 1) / SPB_THREADS; ++q) {
            const int d = t + q * SPB_THREADS;
            double a = 0.0;
            if (d < D) {
                const unsigned long long mask = s_key[s_list[d]];
#pragma unroll 4
                for (int j = 0; j < n; ++j) {
                    const d2 tt = s_t2[j];
                    a += ((mask >> j) & 1ull) ? tt.y : tt.x;
                }
            }
            mine[q] = a;
        }
        if (t == SPB_THREADS - 1) {
            double a = 0.0;
#pragma unroll 4
            for (int j = 0; j < n; ++j) a += s_t2[j].x;
            s_sum0 = a;
            wmax = a;
        }
        if constexpr (EMIT) {
            if (codable) {                                  // uniform
#pragma unroll
                for (int q = 0; q < (SPB_MAXD + SPB_THREADS - 1) / SPB_THREADS; ++q)
                    if (t + q * SPB_THREADS < D) wmax = fmax(wmax, mine[q]);
                wmax = wave_max(wmax);
                if (lane == 0) s_wmax[wv] = wmax;
                if (t == 0) {                               // the record: codes ++ P table ++ table of the sums
                    const long long bytes = (long long)out.ldc + 16ll * (D + 1);
#ifdef RECORDS_FIXED_SLOTS                                  // (timing experiment: no shared bump pointer)
                    long long off = (long long)r * ((long long)out.ldc + 16ll * (SPB_MAXD + 1));
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
        for (int q = 0; q < (SPB_MAXD + SPB_THREADS - 1) / SPB_THREADS; ++q) {
            const int d = t + q * SPB_THREADS;
            if (d < D) s_key[s_list[d]] = (unsigned long long)__double_as_longlong(mine[q]);
        }
        __syncthreads();
        // ---- 5. the row -----------------------------------------------------------------------------
        const double sum0 = s_sum0;
        if constexpr (EMIT) {
            const long long off = codable ? s_off : -1;
            if (off >= 0) {
                const double shift = fmax(fmax(s_wmax[0], s_wmax[1]), fmax(s_wmax[2], s_wmax[3]));   // sums of logs: finite
                double *ptab = reinterpret_cast<double *>(out.rec + off + out.ldc);
                double *mtab = ptab + (D + 1);
#pragma unroll
                for (int q = 0; q < (SPB_MAXD + SPB_THREADS - 1) / SPB_THREADS; ++q) {
                    const int d = t + q * SPB_THREADS;
                    if (d < D) {
                        ptab[d + 1] = exp(mine[q] - shift);
                        mtab[d + 1] = mine[q];
                    }
                }
                if (t == SPB_THREADS - 1) {
                    ptab[0] = exp(sum0 - shift);
                    mtab[0] = sum0;
                }
                unsigned short *cw = reinterpret_cast<unsigned short *>(out.rec + off);
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const int h = 2 * (t + k * SPB_THREADS);
                    if (h < out.ldc) {
                        const unsigned int sa = slot2[k] & 0xffffu, sb = slot2[k] >> 16;
                        const unsigned int c0 = (h < H && sa != 0xffffu) ? s_code[sa] : 0u;
                        const unsigned int c1 = (h + 1 < H && sb != 0xffffu) ? s_code[sb] : 0u;
                        cw[t + k * SPB_THREADS] = (unsigned short)(c0 | (c1 << 8));
                    }
                }
                if (t == 0) {
                    out.rec_off[r] = off;
                    out.ndist[r] = D + 1;
                    out.rowmax[r] = shift;
                }
            } else if (t == 0) {
                no_record(r);                               // more than 256 values: the dense row below is its form
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
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(su4, d2{v0, v1}), mrs, t * 16, k * SPB_THREADS * 16, 2 /* nt */);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(su2, v0), mrs, t * 16, k * SPB_THREADS * 16, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(su2, v1), mrs, t * 16, k * SPB_THREADS * 16 + 8, 0);
                }
            }
        }
        __syncthreads();                                     // everyone has its values: the table can be zeroed
        for (int d = t; d < D; d += SPB_THREADS) s_key[s_list[d]] = 0ull;
        __syncthreads();
    }
}

#endif  // MIXEMT_BUILD_SPARSE_KERNELS_HPP
