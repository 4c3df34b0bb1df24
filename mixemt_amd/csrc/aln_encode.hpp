// aln_encode.hpp -- part of libmixemt_hip.so; HOST code only (no device work), included by mixemt_hip.hip.
//
// The alignment front end of the EM input, batched:
//   preprocess.process_reads   /root/reference/mixemt/preprocess.py:99-139   alignments -> {fragment: {site: base}}
//   preprocess.read_signature  :142-148                                      'pos:base,pos:base,...' by ascending pos
//   preprocess.reduce_reads    :163-174                                      {signature: [fragment ids]}
//   preprocess.build_em_input  :218-220, :225                                rows = sorted(signatures), weights, id lists
// The reference walks every aligned base of every read in the interpreter (58 s per 10^6 alignments against
// 0.8 s for everything the device does afterwards).  Here the alignments arrive as COLUMNS (mxm_aln_columns) and
// one call produces what those four functions produce together: the CSR observations of the distinct signatures in
// the reference's row order (Python's sorted() over the signature STRINGS), their weights and their fragment lists.
//
//   1. count    per alignment: walk the CIGAR, count the variant sites under M / = / X whose base passes min_bq
//               (the site list is sparse in the reference: next_site[pos] jumps from site to site)
//   2. scatter  the (site, upper-cased base) observations into their FRAGMENT's range (mates share a fragment)
//   3. resolve  per fragment: order by site; a site seen with two different bases, or as 'N', is dropped
//               (preprocess.py:126-138: 'N' absorbs everything that follows and is removed at the end)
//   4. de-dup   fragments with equal observation lists share a row: 64-bit hash + exact compare, fragments taken in
//               the order the reference's dict would hold them (first accepted observation, alignment order)
//   5. order    the distinct signatures by their TEXT: the text is formed ('%d:%s' joined by ',') and compared
//               bytewise -- exactly Python's str comparison for ASCII -- with the first 16 bytes as a sort key
// Steps 1-3 and the text of step 5 run on `n_threads` host threads.
#ifndef MIXEMT_ALN_ENCODE_HPP
#define MIXEMT_ALN_ENCODE_HPP

#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <exception>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#include "mixemt_hip.h"

struct mxm_aln_enc {
    // rows = distinct non-empty signatures in sorted-text order
    std::vector<int64_t> row_ptr;        // [n_rows + 1]
    std::vector<int64_t> row_frag;       // [n_rows] a fragment that carries row r's observations (site / obs are copied
                                         // out of o_site / o_base on fetch: no second copy is held)
    std::vector<int64_t> weights;        // [n_rows] fragments per signature (preprocess.py:220)
    std::vector<int64_t> group_ptr;      // [n_rows + 1]
    std::vector<int64_t> group_frag;     // fragment ids, per row in the reference's list order (:173)
    std::vector<int64_t> dropped;        // fragments whose every site was conflicted away (signature '')
    std::vector<int64_t> text_off;       // [n_rows + 1] into the text as fetched (rows in order)
    std::vector<int64_t> row_text;       // [n_rows] where row r's signature starts in `text`
    std::unique_ptr<char[]> text;        // the signatures in the order they were first seen, one '\n' after each (not
                                         // zero-filled first: 100 MB at 10^6 fragments, written once by the threads)
    // fragments in the reference's dict order (process_reads' result, for the callers that want it: formed on fetch)
    std::vector<int64_t> frag_id;        // [n_frag_seen]
    std::vector<int64_t> f_ptr, f_len;   // [n_frag]: fragment f keeps f_len[f] observations at o_site / o_base [f_ptr[f] ...)
    std::vector<uint16_t> o_site;
    std::vector<uint8_t> o_base;
    int64_t frag_nnz = 0;
};

namespace aln_detail {

struct walker {
    const mxm_aln_columns *c;
    const int32_t *next_site;            // [ref_len + 1] first site index at or after a position
    const int64_t *site_pos;             // [n_sites]
    int64_t ref_len;
    int32_t n_sites, min_mq, min_bq;

    // calls emit(site index, upper-cased base) for every accepted observation of alignment i; returns the count, or
    // -1 for an alignment the reference would not get through either (CIGAR runs past the sequence, unknown operation)
    template <typename F>
    inline int64_t walk(int64_t i, F &&emit) const {
        if (c->mapq[i] < min_mq) return 0;
        const int64_t s0 = c->seq_ptr[i], slen = c->seq_ptr[i + 1] - s0;
        const bool has_q = c->qual != nullptr && (c->has_qual == nullptr || c->has_qual[i] != 0);
        int64_t r = c->ref_start[i], q = 0, n = 0;
        for (int64_t k = c->cig_ptr[i]; k < c->cig_ptr[i + 1]; ++k) {
            const uint32_t op = c->cigar[k] & 15u;
            const int64_t len = (int64_t)(c->cigar[k] >> 4);
            switch (op) {
                case 0: case 7: case 8: {                    // M, =, X: aligned pairs (get_aligned_pairs(matches_only=True))
                    if (q + len > slen) return -1;
                    int64_t lo = r < 0 ? 0 : r;
                    const int64_t hi = r + len;
                    if (lo < ref_len && hi > lo) {
                        for (int32_t s = next_site[lo]; s < n_sites && site_pos[s] < hi; ++s) {
                            const int64_t qp = s0 + q + (site_pos[s] - r);
                            if (has_q && (int32_t)c->qual[qp] < min_bq) continue;
                            uint8_t b = c->seq[qp];
                            if (b >= 'a' && b <= 'z') b = (uint8_t)(b - 32);        // str.upper() on an ASCII base
                            emit(s, b);
                            ++n;
                        }
                    }
                    q += len;
                    r += len;
                    break;
                }
                case 1: case 4: q += len; break;             // I, S: query only
                case 2: case 3: r += len; break;             // D, N: reference only
                case 5: case 6: break;                       // H, P: neither
                default: return -1;
            }
        }
        return n;
    }
};

// Worker threads must not let an exception escape: a std::bad_alloc thrown inside a std::thread body -- the
// per-thread lists and vectors below grow there -- would call std::terminate and take the Python process down, where
// mxm_aln_encode / mxm_bam_read are written to return -5.  Every body runs under a catch-all; the threads are always
// joined; the first exception is rethrown on the calling thread, inside the entry point's own try block.
template <typename F>
static void run_threads(int nt, F &&body) {
    if (nt <= 1) {
        body(0);
        return;
    }
    std::vector<std::exception_ptr> err((size_t)nt);
    std::vector<std::thread> pool;
    std::exception_ptr spawn_err;
    try {
        pool.reserve((size_t)nt);
        for (int t = 0; t < nt; ++t)
            pool.emplace_back([&body, &err, t]() {
                try {
                    body(t);
                } catch (...) {
                    err[(size_t)t] = std::current_exception();
                }
            });
    } catch (...) {
        spawn_err = std::current_exception();               // (a thread could not be started: join the ones that were)
    }
    for (auto &th : pool) th.join();
    if (spawn_err) std::rethrow_exception(spawn_err);
    for (auto &e : err)
        if (e) std::rethrow_exception(e);
}

template <typename F>
static void parallel_for(int64_t n, int n_threads, F &&body, int64_t serial_below = 4096) {
    if (n_threads <= 1 || n < serial_below) {
        body(0, n, 0);
        return;
    }
    const int64_t per = (n + n_threads - 1) / n_threads;
    const int used = (int)((n + per - 1) / per);
    run_threads(used, [&](int t) {
        const int64_t lo = t * per, hi = std::min(n, lo + per);
        if (lo < hi) body(lo, hi, t);
    });
}

static inline uint64_t mix64(uint64_t h) {
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return h;
}

// decimal digits of a non-negative position, as '%d' prints them
static inline char *put_int(char *p, int64_t v) {
    char tmp[24];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v > 0);
    while (n > 0) *p++ = tmp[--n];
    return p;
}

}  // namespace aln_detail

static int aln_encode_impl(const mxm_aln_columns *c, const int32_t *site_of_pos, int64_t ref_len, const int64_t *site_pos,
                           int32_t n_sites, int32_t min_mq, int32_t min_bq, int32_t n_threads, mxm_aln_enc *out) {
    using namespace aln_detail;
    const int64_t n_aln = c->n_aln, n_frag = c->n_frag;
    if (n_threads <= 0) {
        n_threads = (int)std::thread::hardware_concurrency();
        if (n_threads < 1) n_threads = 1;
        if (n_threads > 16) n_threads = 16;
    }
    for (int64_t i = 0; i < n_aln; ++i)
        if (c->frag[i] < 0 || c->frag[i] >= n_frag) return fail(-1, "mxm_aln_encode: fragment index of alignment %s%lld outside [0, n_frag)", "", i);
    for (int32_t s = 0; s < n_sites; ++s)
        if (site_pos[s] < 0 || site_pos[s] >= ref_len || site_of_pos[site_pos[s]] != s || (s > 0 && site_pos[s] <= site_pos[s - 1]))
            return fail(-1, "mxm_aln_encode: site_pos / site_of_pos do not describe the same ascending site list%s (site %lld)", "", s);
    std::vector<int32_t> next_site((size_t)ref_len + 1);
    {
        int32_t s = n_sites;
        next_site[ref_len] = n_sites;
        for (int64_t p = ref_len - 1; p >= 0; --p) {
            if (site_of_pos[p] >= 0) s = site_of_pos[p];
            next_site[p] = s;
        }
    }
    walker w{c, next_site.data(), site_pos, ref_len, n_sites, min_mq, min_bq};
    // MXM_ALN_TIMING=1: stage times on stderr (tools/time_frontend.py)
    const bool timing = getenv("MXM_ALN_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto stamp = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[mxm_aln_encode] %-10s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };

    // ---- 1. count ------------------------------------------------------------------------------------------
    std::vector<int64_t> cnt((size_t)n_aln);
    std::vector<int64_t> bad((size_t)n_threads, -1);
    parallel_for(n_aln, n_threads, [&](int64_t lo, int64_t hi, int t) {
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t n = w.walk(i, [](int32_t, uint8_t) {});
            if (n < 0) {
                if (bad[t] < 0) bad[t] = i;
                cnt[i] = 0;
            } else
                cnt[i] = n;
        }
    });
    for (int t = 0; t < n_threads; ++t)
        if (bad[t] >= 0)
            return fail(-4, "mxm_aln_encode: alignment %s%lld has a CIGAR that runs past its sequence or an unknown operation", "", bad[t]);
    stamp("count");
    // fragment ranges; a fragment's place in the reference's dict = the first alignment that gave it an observation
    std::vector<int64_t> f_cnt((size_t)n_frag, 0), f_first((size_t)n_frag, -1), a_off((size_t)n_aln);
    for (int64_t i = 0; i < n_aln; ++i) {
        const int64_t f = c->frag[i];
        if (cnt[i] > 0 && f_first[f] < 0) f_first[f] = i;
        f_cnt[f] += cnt[i];
    }
    std::vector<int64_t> f_ptr((size_t)n_frag + 1);
    f_ptr[0] = 0;
    for (int64_t f = 0; f < n_frag; ++f) f_ptr[f + 1] = f_ptr[f] + f_cnt[f];
    const int64_t total = f_ptr[n_frag];
    {
        std::vector<int64_t> cursor(f_ptr.begin(), f_ptr.end() - 1);
        for (int64_t i = 0; i < n_aln; ++i) {
            a_off[i] = cursor[c->frag[i]];
            cursor[c->frag[i]] += cnt[i];
        }
    }
    stamp("ranges");
    // ---- 2. scatter ----------------------------------------------------------------------------------------
    std::vector<uint16_t> o_site((size_t)total);
    std::vector<uint8_t> o_base((size_t)total);
    parallel_for(n_aln, n_threads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; ++i) {
            if (cnt[i] == 0) continue;
            int64_t at = a_off[i];
            w.walk(i, [&](int32_t s, uint8_t b) {
                o_site[at] = (uint16_t)s;
                o_base[at] = b;
                ++at;
            });
        }
    });
    stamp("scatter");
    // ---- 3. resolve ----------------------------------------------------------------------------------------
    // in place: fragment f keeps f_len[f] observations at the head of its range, ascending by site
    std::vector<int64_t> f_len((size_t)n_frag, 0);
    std::vector<uint64_t> f_hash((size_t)n_frag, 0);
    parallel_for(n_frag, n_threads, [&](int64_t flo, int64_t fhi, int) {
        for (int64_t f = flo; f < fhi; ++f) {
            const int64_t a = f_ptr[f], n = f_cnt[f];
            if (n == 0) continue;
            uint16_t *st = &o_site[a];
            uint8_t *bs = &o_base[a];
            // insertion sort by site (an alignment's observations ascend already; mates interleave)
            for (int64_t i = 1; i < n; ++i) {
                const uint16_t s = st[i];
                const uint8_t b = bs[i];
                int64_t j = i - 1;
                while (j >= 0 && st[j] > s) {
                    st[j + 1] = st[j];
                    bs[j + 1] = bs[j];
                    --j;
                }
                st[j + 1] = s;
                bs[j + 1] = b;
            }
            int64_t m = 0;
            uint64_t h = 0x9E3779B97F4A7C15ull;
            for (int64_t i = 0; i < n;) {
                int64_t j = i + 1;
                bool same = true;
                while (j < n && st[j] == st[i]) {
                    same = same && bs[j] == bs[i];
                    ++j;
                }
                if (same && bs[i] != 'N') {
                    st[m] = st[i];
                    bs[m] = bs[i];
                    h = mix64(h ^ (((uint64_t)st[m] << 8) | bs[m]));
                    ++m;
                }
                i = j;
            }
            f_len[f] = m;
            f_hash[f] = h;
        }
    });
    stamp("resolve");
    // fragments in dict order
    // (f_first = the fragment's first counted alignment: distinct per fragment, so the order is a scatter, not a sort)
    std::vector<int64_t> order;
    order.reserve((size_t)n_frag);
    {
        std::vector<int64_t> by_first((size_t)n_aln, -1);
        for (int64_t f = 0; f < n_frag; ++f)
            if (f_first[f] >= 0) by_first[(size_t)f_first[f]] = f;
        for (int64_t i = 0; i < n_aln; ++i)
            if (by_first[(size_t)i] >= 0) order.push_back(by_first[(size_t)i]);
    }
    const int64_t n_seen = (int64_t)order.size();
    stamp("dict order");
    // ---- 4. de-dup -----------------------------------------------------------------------------------------
    // by 64-bit hash + exact compare, in 64 partitions of the hash on the threads (one table on one thread was 50 of the
    // encoder's 180 ms at 16 threads): each partition walks ITS fragments in dict order, so a signature's representative
    // is the first fragment that shows it; signature ids are then handed out in dict order of the representatives
    std::vector<int64_t> sig_rep;                            // representative fragment of each signature
    std::vector<int64_t> sig_of((size_t)n_frag, -1);
    std::vector<int64_t> sig_count;
    out->dropped.clear();
    {
        constexpr int NPART = 64;
        const int nt = (n_threads > 1 && n_seen >= 4096) ? n_threads : 1;
        std::vector<int64_t> first_k((size_t)n_seen, -1), cnt_k((size_t)n_seen, 0);
        std::vector<std::vector<int64_t>> lists((size_t)nt * NPART);
        auto run = [&](const std::function<void(int)> &body) { run_threads(nt, body); };
        run([&](int t) {
            const int64_t per = (n_seen + nt - 1) / nt, lo = t * per, hi = std::min(n_seen, lo + per);
            for (int64_t k = lo; k < hi; ++k) {
                const int64_t f = order[(size_t)k];
                if (f_len[f] > 0) lists[(size_t)t * NPART + (size_t)(f_hash[f] >> 58)].push_back(k);
            }
        });
        run([&](int t) {
            std::vector<int64_t> slot;
            for (int part = t; part < NPART; part += nt) {
                size_t count = 0;
                for (int u = 0; u < nt; ++u) count += lists[(size_t)u * NPART + part].size();
                size_t cap = 16;
                while (cap < count * 2) cap <<= 1;
                slot.assign(cap, -1);                        // -> k of the signature's first fragment
                for (int u = 0; u < nt; ++u) {               // thread order = ascending k = dict order
                    for (const int64_t k : lists[(size_t)u * NPART + part]) {
                        const int64_t f = order[(size_t)k], m = f_len[f];
                        size_t h = (size_t)(f_hash[f] * 0x9e3779b97f4a7c15ull >> 20) & (cap - 1);
                        for (;;) {
                            const int64_t kr = slot[h];
                            if (kr < 0) {
                                slot[h] = k;
                                first_k[(size_t)k] = k;
                                cnt_k[(size_t)k] = 1;
                                break;
                            }
                            const int64_t g = order[(size_t)kr];
                            if (f_hash[g] == f_hash[f] && f_len[g] == m &&
                                memcmp(&o_site[f_ptr[g]], &o_site[f_ptr[f]], (size_t)m * 2) == 0 &&
                                memcmp(&o_base[f_ptr[g]], &o_base[f_ptr[f]], (size_t)m) == 0) {
                                first_k[(size_t)k] = kr;
                                ++cnt_k[(size_t)kr];
                                break;
                            }
                            h = (h + 1) & (cap - 1);
                        }
                    }
                }
            }
        });
        std::vector<int64_t> id_of_k((size_t)n_seen, -1);
        for (int64_t k = 0; k < n_seen; ++k) {               // ids (and the dropped fragments) in dict order
            const int64_t f = order[(size_t)k];
            if (f_len[f] == 0) {
                out->dropped.push_back(f);
            } else if (first_k[(size_t)k] == k) {
                id_of_k[(size_t)k] = (int64_t)sig_rep.size();
                sig_rep.push_back(f);
                sig_count.push_back(cnt_k[(size_t)k]);
            }
        }
        run([&](int t) {
            const int64_t per = (n_seen + nt - 1) / nt, lo = t * per, hi = std::min(n_seen, lo + per);
            for (int64_t k = lo; k < hi; ++k) {
                const int64_t f = order[(size_t)k];
                if (f_len[f] > 0) sig_of[f] = id_of_k[(size_t)first_k[(size_t)k]];
            }
        });
    }
    const int64_t n_sig = (int64_t)sig_rep.size();
    stamp("de-dup");
    // ---- 5. order by text ----------------------------------------------------------------------------------
    std::vector<int64_t> t_off((size_t)n_sig + 1);
    t_off[0] = 0;
    {
        // length of '%d' per site, then of every signature (+ 1 for the separator that follows it)
        std::vector<uint8_t> digits((size_t)n_sites);
        for (int32_t s = 0; s < n_sites; ++s) {
            int d = 1;
            for (int64_t v = site_pos[s]; v >= 10; v /= 10) ++d;
            digits[s] = (uint8_t)d;
        }
        parallel_for(n_sig, n_threads, [&](int64_t lo, int64_t hi, int) {
            for (int64_t sg = lo; sg < hi; ++sg) {
                const int64_t f = sig_rep[sg], a = f_ptr[f], m = f_len[f];
                int64_t len = m * 2 + (m - 1);               // ':' + base per item, ',' between items
                for (int64_t i = 0; i < m; ++i) len += digits[o_site[a + i]];
                t_off[sg + 1] = len + 1;                     // (its own length: summed up below)
            }
        });
        for (int64_t sg = 0; sg < n_sig; ++sg) t_off[sg + 1] += t_off[sg];
    }
    std::unique_ptr<char[]> text(new char[(size_t)t_off[n_sig] + 1]);
    parallel_for(n_sig, n_threads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t sg = lo; sg < hi; ++sg) {
            const int64_t f = sig_rep[sg], a = f_ptr[f], m = f_len[f];
            char *p = &text[t_off[sg]];
            for (int64_t i = 0; i < m; ++i) {
                if (i) *p++ = ',';
                p = put_int(p, site_pos[o_site[a + i]]);
                *p++ = ':';
                *p++ = (char)o_base[a + i];
            }
            *p++ = '\n';
        }
    });
    stamp("text");
    struct key {
        uint64_t k0, k1;
        int64_t sg;
    };
    std::vector<key> keys((size_t)n_sig);
    auto be64 = [&](int64_t off, int64_t len) -> uint64_t {
        uint64_t v = 0;
        for (int i = 0; i < 8; ++i) v = (v << 8) | (uint64_t)(uint8_t)(i < len ? text[off + i] : 0);
        return v;
    };
    for (int64_t sg = 0; sg < n_sig; ++sg) {
        const int64_t off = t_off[sg], len = t_off[sg + 1] - off - 1;
        keys[sg].k0 = be64(off, len);
        keys[sg].k1 = be64(off + 8, len > 8 ? len - 8 : 0);
        keys[sg].sg = sg;
    }
    auto less = [&](const key &x, const key &y) -> bool {
        if (x.k0 != y.k0) return x.k0 < y.k0;
        if (x.k1 != y.k1) return x.k1 < y.k1;
        // (a zero byte stands for "past the end" in the keys; a signature holds none, so equal keys mean equal
        // first 16 bytes AND both at least as long as the shorter's prefix: the rest decides)
        const int64_t xo = t_off[x.sg], xl = t_off[x.sg + 1] - xo - 1, yo = t_off[y.sg], yl = t_off[y.sg + 1] - yo - 1;
        const int64_t n = xl < yl ? xl : yl;
        const int cmp = n > 0 ? memcmp(&text[xo], &text[yo], (size_t)n) : 0;
        if (cmp != 0) return cmp < 0;
        return xl < yl;
    };
    if (n_threads > 1 && n_sig >= 4096) {
        // sample sort: splitters from a sorted sample, every thread counts and scatters its share of the keys into the
        // splitters' buckets, then the buckets are sorted side by side -- no serial merge at the end (the pairwise merges
        // of sorted runs this replaces spent their last pass, all n_sig keys, on one thread)
        const int parts = n_threads;
        const int64_t step = std::max<int64_t>(1, n_sig / ((int64_t)parts * 64));
        std::vector<key> sample;
        for (int64_t i = step / 2; i < n_sig; i += step) sample.push_back(keys[(size_t)i]);
        std::sort(sample.begin(), sample.end(), less);
        std::vector<key> split;
        for (int b = 1; b < parts; ++b) split.push_back(sample[(size_t)((int64_t)sample.size() * b / parts)]);
        auto bucket_of = [&](const key &x) -> int {
            return (int)(std::upper_bound(split.begin(), split.end(), x, less) - split.begin());
        };
        std::vector<int64_t> bcnt((size_t)parts * parts, 0);    // [thread][bucket]
        std::vector<int32_t> bkt((size_t)n_sig);
        parallel_for(parts, parts, [&](int64_t lo, int64_t hi, int) {
            for (int64_t t = lo; t < hi; ++t)
                for (int64_t i = n_sig * t / parts; i < n_sig * (t + 1) / parts; ++i) {
                    bkt[(size_t)i] = bucket_of(keys[(size_t)i]);
                    ++bcnt[(size_t)t * parts + bkt[(size_t)i]];
                }
        }, 2);
        std::vector<int64_t> start((size_t)parts * parts), bstart((size_t)parts + 1, 0);
        int64_t run = 0;
        for (int b = 0; b < parts; ++b) {
            bstart[b] = run;
            for (int t = 0; t < parts; ++t) {
                start[(size_t)t * parts + b] = run;
                run += bcnt[(size_t)t * parts + b];
            }
        }
        bstart[parts] = run;
        std::vector<key> tmp((size_t)n_sig);
        parallel_for(parts, parts, [&](int64_t lo, int64_t hi, int) {
            for (int64_t t = lo; t < hi; ++t) {
                std::vector<int64_t> at(start.begin() + t * parts, start.begin() + (t + 1) * parts);
                for (int64_t i = n_sig * t / parts; i < n_sig * (t + 1) / parts; ++i) tmp[(size_t)at[bkt[(size_t)i]]++] = keys[(size_t)i];
            }
        }, 2);
        parallel_for(parts, parts, [&](int64_t lo, int64_t hi, int) {
            for (int64_t b = lo; b < hi; ++b) std::sort(tmp.begin() + bstart[b], tmp.begin() + bstart[b + 1], less);
        }, 2);
        keys.swap(tmp);
    } else {
        std::sort(keys.begin(), keys.end(), less);
    }
    stamp("sort");
    // ---- outputs -------------------------------------------------------------------------------------------
    std::vector<int64_t> rank_of((size_t)n_sig);             // signature id -> row
    out->row_ptr.assign((size_t)n_sig + 1, 0);
    out->weights.resize((size_t)n_sig);
    out->group_ptr.assign((size_t)n_sig + 1, 0);
    out->text_off.assign((size_t)n_sig + 1, 0);
    out->row_frag.resize((size_t)n_sig);
    out->row_text.resize((size_t)n_sig);
    for (int64_t r = 0; r < n_sig; ++r) {
        const int64_t sg = keys[r].sg, f = sig_rep[sg];
        rank_of[sg] = r;
        out->row_frag[r] = f;
        out->row_text[r] = t_off[sg];
        out->row_ptr[r + 1] = out->row_ptr[r] + f_len[f];
        out->weights[r] = sig_count[sg];
        out->group_ptr[r + 1] = out->group_ptr[r] + sig_count[sg];
        out->text_off[r + 1] = out->text_off[r] + (t_off[sg + 1] - t_off[sg]);
    }
    out->group_frag.resize((size_t)out->group_ptr[n_sig]);
    {
        std::vector<int64_t> cursor(out->group_ptr.begin(), out->group_ptr.end() - 1);
        for (int64_t k = 0; k < n_seen; ++k) {               // dict order -> each list in the reference's order
            const int64_t f = order[k];
            if (sig_of[f] >= 0) out->group_frag[cursor[rank_of[sig_of[f]]]++] = f;
        }
    }
    out->frag_nnz = 0;
    for (int64_t k = 0; k < n_seen; ++k) out->frag_nnz += f_len[order[k]];
    out->frag_id.swap(order);
    out->f_ptr.swap(f_ptr);
    out->f_len.swap(f_len);
    out->o_site.swap(o_site);
    out->o_base.swap(o_base);
    out->text = std::move(text);
    stamp("outputs");
    return 0;
}

extern "C" int mxm_aln_encode(const mxm_aln_columns *cols, const int32_t *site_of_pos, int64_t ref_len,
                              const int64_t *site_pos, int32_t n_sites, int32_t min_mq, int32_t min_bq, int32_t n_threads,
                              mxm_aln_enc **out) {
    if (out == nullptr) return fail(-1, "mxm_aln_encode: out is NULL%s", "");
    *out = nullptr;
    if (cols == nullptr || cols->n_aln < 0 || cols->n_frag < 0 || site_of_pos == nullptr || site_pos == nullptr || ref_len <= 0 ||
        n_sites < 0 || n_sites > 65536)
        return fail(-1, "mxm_aln_encode: bad arguments%s", "");
    if (cols->n_aln > 0 && (cols->ref_start == nullptr || cols->mapq == nullptr || cols->frag == nullptr || cols->cig_ptr == nullptr ||
                            cols->seq_ptr == nullptr || (cols->cig_ptr[cols->n_aln] > 0 && cols->cigar == nullptr) ||
                            (cols->seq_ptr[cols->n_aln] > 0 && cols->seq == nullptr)))
        return fail(-1, "mxm_aln_encode: alignment columns missing%s", "");
    mxm_aln_enc *res = nullptr;
    try {
        res = new mxm_aln_enc();
        const int rc = aln_encode_impl(cols, site_of_pos, ref_len, site_pos, n_sites, min_mq, min_bq, n_threads, res);
        if (rc != 0) {
            delete res;
            return rc;
        }
    } catch (const std::exception &e) {
        delete res;
        return fail(-5, "mxm_aln_encode: %s", e.what());
    }
    *out = res;
    return 0;
}

extern "C" int mxm_aln_sizes_of(const mxm_aln_enc *e, mxm_aln_sizes *s) {
    if (e == nullptr || s == nullptr) return fail(-1, "mxm_aln_sizes_of: NULL argument%s", "");
    s->n_rows = (int64_t)e->weights.size();
    s->nnz = e->row_ptr.empty() ? 0 : e->row_ptr.back();
    s->n_grouped = (int64_t)e->group_frag.size();
    s->n_dropped = (int64_t)e->dropped.size();
    s->text_bytes = e->text_off.empty() ? 0 : e->text_off.back();
    s->n_frag_seen = (int64_t)e->frag_id.size();
    s->frag_nnz = e->frag_nnz;
    return 0;
}

template <typename V, typename P>
static inline void aln_copy_out(const std::vector<V> &v, P *dst) {
    if (dst != nullptr && !v.empty()) memcpy(dst, v.data(), v.size() * sizeof(V));
}

extern "C" int mxm_aln_fetch(const mxm_aln_enc *e, int64_t *row_ptr, uint16_t *site, uint8_t *obs, int64_t *weights,
                             int64_t *group_ptr, int64_t *group_frag, int64_t *dropped, char *text, int64_t *text_off) {
    if (e == nullptr) return fail(-1, "mxm_aln_fetch: NULL handle%s", "");
    aln_copy_out(e->row_ptr, row_ptr);
    const int64_t n_rows = (int64_t)e->weights.size();
    if (site != nullptr || obs != nullptr || text != nullptr) {
        int nt = (int)std::thread::hardware_concurrency();
        nt = nt < 1 ? 1 : (nt > 8 ? 8 : nt);
        aln_detail::parallel_for(n_rows, nt, [&](int64_t lo, int64_t hi, int) {
            for (int64_t r = lo; r < hi; ++r) {
                const int64_t f = e->row_frag[r], m = e->row_ptr[r + 1] - e->row_ptr[r];
                if (m > 0 && site != nullptr) memcpy(site + e->row_ptr[r], &e->o_site[e->f_ptr[f]], (size_t)m * 2);
                if (m > 0 && obs != nullptr) memcpy(obs + e->row_ptr[r], &e->o_base[e->f_ptr[f]], (size_t)m);
                if (text != nullptr) memcpy(text + e->text_off[r], &e->text[e->row_text[r]], (size_t)(e->text_off[r + 1] - e->text_off[r]));
            }
        });
    }
    aln_copy_out(e->weights, weights);
    aln_copy_out(e->group_ptr, group_ptr);
    aln_copy_out(e->group_frag, group_frag);
    aln_copy_out(e->dropped, dropped);
    aln_copy_out(e->text_off, text_off);
    return 0;
}

extern "C" int mxm_aln_fetch_fragments(const mxm_aln_enc *e, int64_t *frag_id, int64_t *frag_ptr, uint16_t *site, uint8_t *obs) {
    if (e == nullptr) return fail(-1, "mxm_aln_fetch_fragments: NULL handle%s", "");
    aln_copy_out(e->frag_id, frag_id);
    int64_t at = 0;
    for (size_t k = 0; k < e->frag_id.size(); ++k) {
        const int64_t f = e->frag_id[k], m = e->f_len[f];
        if (frag_ptr != nullptr) frag_ptr[k] = at;
        if (m > 0 && site != nullptr) memcpy(site + at, &e->o_site[e->f_ptr[f]], (size_t)m * 2);
        if (m > 0 && obs != nullptr) memcpy(obs + at, &e->o_base[e->f_ptr[f]], (size_t)m);
        at += m;
    }
    if (frag_ptr != nullptr) frag_ptr[e->frag_id.size()] = at;
    return 0;
}

extern "C" void mxm_aln_free(mxm_aln_enc *e) { delete e; }

#endif  // MIXEMT_ALN_ENCODE_HPP
