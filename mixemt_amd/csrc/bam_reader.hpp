// bam_reader.hpp -- part of libmixemt_hip.so; HOST code only (zlib + threads), included by mixemt_hip.hip.
//
// A BAM file straight into the COLUMNS mxm_aln_encode takes (aln_encode.hpp) -- the front end of the EM input without one
// Python object per alignment.  The reference reads its alignments through pysam (bin/mixemt:139-147 opens the file,
// preprocess.py:209 iterates `bamfile.fetch()`, :118-132 reads mapping_quality / query_name / query_sequence /
// query_qualities / get_aligned_pairs of every AlignedSegment); with the batched encoder in place that object-by-object
// hand-over is the slowest stage left (4.3 s per 10^6 alignments against 0.2-0.4 s for the encoder).  This reader does
// what `fetch()` + those five attributes amount to:
//   * BGZF: the file is a series of gzip members of at most 64 KiB, each carrying its own size in a 'BC' extra field
//     (SAM/BAM specification, section 4.1); the members are found by one walk over the headers and inflated on
//     `n_threads` threads, each into its own slice of one output buffer;
//   * BAM records (section 4.2): refID, pos, mapq, n_cigar_op, l_seq, read_name, cigar (already `len << 4 | op`),
//     4-bit packed bases ("=ACMGRSVTWYHKDBN"), qualities (0xFF in the first byte = absent: pysam's None);
//   * like `fetch()` without a region: every record that is placed on a reference (refID >= 0), in file order; records
//     without a reference (unplaced, at the end of a sorted file) are skipped.  No flag is looked at -- the reference
//     filters by mapping quality only (preprocess.py:119);
//   * fragments: records of equal read_name share a fragment index, numbered by first appearance (mates merge,
//     preprocess.py:124-132).
// Not supported (returns -4, the caller falls back to pysam where there is one): CIGARs of more than 65535 operations
// (kept in a CG tag), CRAM, uncompressed SAM.
#ifndef MIXEMT_BAM_READER_HPP
#define MIXEMT_BAM_READER_HPP

#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>
#include <functional>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <new>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "mixemt_hip.h"

namespace bam_detail {
// an array that is NOT zero-filled when sized (std::vector would touch every page of the 100 MB columns on one thread;
// here the first touch is the filling threads' own)
template <class T>
struct raw {
    T *p = nullptr;
    size_t n = 0;
    raw() = default;
    raw(const raw &) = delete;
    raw &operator=(const raw &) = delete;
    ~raw() { free(p); }
    void resize(size_t count) {
        free(p);
        p = nullptr;
        n = 0;
        if (count) {
            p = static_cast<T *>(malloc(count * sizeof(T)));
            if (p == nullptr) throw std::bad_alloc();
        }
        n = count;
    }
    T *data() { return p; }
    const T *data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
};

// the file, mapped read-only (the inflating threads read the page cache directly)
struct mapped_file {
    const uint8_t *p = nullptr;
    size_t n = 0;
    ~mapped_file() {
        if (p != nullptr && n) munmap(const_cast<uint8_t *>(p), n);
    }
    size_t size() const { return n; }
    const uint8_t *data() const { return p; }
    const uint8_t &operator[](size_t i) const { return p[i]; }
};
}  // namespace bam_detail

struct mxm_bam {
    std::vector<int64_t> cig_ptr, seq_ptr, name_off;
    bam_detail::raw<int64_t> ref_start, frag;
    bam_detail::raw<int32_t> mapq, ref_id;
    bam_detail::raw<uint16_t> flag;
    bam_detail::raw<uint32_t> cigar;
    bam_detail::raw<uint8_t> seq, qual, has_qual;
    std::vector<char> names;             // fragment names back to back (no separators; name_off[n_frag + 1])
    std::vector<std::string> ref_names;
    std::vector<int64_t> ref_lens;
    int64_t n_records_total = 0, n_skipped_unplaced = 0;
    bool any_qual = false;
};

namespace bam_detail {

static inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

struct bgzf_block {
    size_t in_off, in_len;               // deflate payload
    size_t out_off, out_len;             // its place in the inflated stream
    uint32_t crc;                        // CRC-32 of the inflated bytes (the gzip trailer)
};

}  // namespace bam_detail

static int bam_read_impl(const char *path, int n_threads, mxm_bam *out) {
    using namespace bam_detail;
    // MXM_ALN_TIMING=1: stage times on stderr (tools/time_bam_reader.py)
    const bool timing = getenv("MXM_ALN_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto stamp = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[mxm_bam_read] %-10s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    mapped_file raw;
    {
        const int fd = open(path, O_RDONLY);
        if (fd < 0) return fail(-6, "mxm_bam_read: cannot open %s", path);
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
            close(fd);
            return fail(-6, "mxm_bam_read: %s is not a regular file", path);
        }
        if (st.st_size > 0) {
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) {
                close(fd);
                return fail(-6, "mxm_bam_read: cannot map %s", path);
            }
            raw.p = static_cast<const uint8_t *>(m);
            raw.n = (size_t)st.st_size;
            madvise(m, raw.n, MADV_WILLNEED);
        }
        close(fd);
    }
    stamp("file");
    // ---- BGZF members ----------------------------------------------------------------------------------------
    std::vector<bgzf_block> blocks;
    size_t at = 0, total_out = 0;
    while (at < raw.size()) {
        if (raw.size() - at < 18 + 8 || raw[at] != 31 || raw[at + 1] != 139 || raw[at + 2] != 8 || (raw[at + 3] & 4) == 0)
            return fail(-4, "mxm_bam_read: %s is not a BGZF file (no gzip member with an extra field at byte %lld)", path, (long long)at);
        const size_t xlen = le16(&raw[at + 10]);
        if (raw.size() - at < 12 + xlen) return fail(-4, "mxm_bam_read: truncated BGZF header in %s (byte %lld)", path, (long long)at);
        long bsize = -1;
        for (size_t x = at + 12; x + 4 <= at + 12 + xlen;) {
            const size_t slen = le16(&raw[x + 2]);
            if (raw[x] == 'B' && raw[x + 1] == 'C' && slen == 2 && x + 6 <= at + 12 + xlen) bsize = (long)le16(&raw[x + 4]) + 1;
            x += 4 + slen;
        }
        if (bsize < (long)(12 + xlen + 8) || at + (size_t)bsize > raw.size())
            return fail(-4, "mxm_bam_read: bad BGZF block size in %s (byte %lld)", path, (long long)at);
        bgzf_block b;
        b.in_off = at + 12 + xlen;
        b.in_len = (size_t)bsize - xlen - 12 - 8;
        b.out_len = le32(&raw[at + (size_t)bsize - 4]);
        b.crc = le32(&raw[at + (size_t)bsize - 8]);
        b.out_off = total_out;
        if (b.out_len > 65536) return fail(-4, "mxm_bam_read: BGZF block of more than 64 KiB in %s (byte %lld)", path, (long long)at);
        total_out += b.out_len;
        blocks.push_back(b);
        at += (size_t)bsize;
    }
    bam_detail::raw<uint8_t> data;
    data.resize(total_out);
    if (n_threads <= 0) {
        n_threads = (int)std::thread::hardware_concurrency();
        if (n_threads < 1) n_threads = 1;
        if (n_threads > 16) n_threads = 16;
    }
    {
        std::vector<long long> bad((size_t)n_threads, -1);
        auto work = [&](int t) {
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -15) != Z_OK) { bad[t] = 0; return; }
            for (size_t i = (size_t)t; i < blocks.size(); i += (size_t)n_threads) {
                const bgzf_block &b = blocks[i];
                if (b.out_len == 0) continue;
                inflateReset(&zs);
                zs.next_in = const_cast<Bytef *>(raw.data() + b.in_off);
                zs.avail_in = (uInt)b.in_len;
                zs.next_out = data.data() + b.out_off;
                zs.avail_out = (uInt)b.out_len;
                const int rc = inflate(&zs, Z_FINISH);
                if (rc != Z_STREAM_END || zs.avail_out != 0 || zs.avail_in != 0 ||
                    (uint32_t)crc32(crc32(0L, Z_NULL, 0), data.data() + b.out_off, (uInt)b.out_len) != b.crc) {
                    bad[t] = (long long)i;
                    break;
                }
            }
            inflateEnd(&zs);
        };
        if (n_threads == 1 || blocks.size() < 8) {
            n_threads = 1;
            work(0);
        } else {
            aln_detail::run_threads(n_threads, work);                    // (aln_encode.hpp: no exception leaves a worker thread)
        }
        for (int t = 0; t < n_threads; ++t)
            if (bad[t] >= 0) return fail(-4, "mxm_bam_read: BGZF block %s%lld does not inflate to its recorded size and CRC", "", bad[t]);
    }
    stamp("inflate");
    // ---- BAM header ------------------------------------------------------------------------------------------
    const uint8_t *p = data.data(), *end = data.data() + data.size();
    if (end - p < 12 || memcmp(p, "BAM\1", 4) != 0) return fail(-4, "mxm_bam_read: %s has no BAM magic", path);
    const uint32_t l_text = le32(p + 4);
    p += 8;
    if ((size_t)(end - p) < (size_t)l_text + 4) return fail(-4, "mxm_bam_read: truncated header text in %s", path);
    p += l_text;
    const uint32_t n_ref = le32(p);
    p += 4;
    for (uint32_t r = 0; r < n_ref; ++r) {
        if (end - p < 4) return fail(-4, "mxm_bam_read: truncated reference list in %s", path);
        const uint32_t l_name = le32(p);
        p += 4;
        if ((size_t)(end - p) < (size_t)l_name + 4 || l_name == 0) return fail(-4, "mxm_bam_read: truncated reference list in %s", path);
        out->ref_names.emplace_back(reinterpret_cast<const char *>(p), l_name - 1);
        p += l_name;
        out->ref_lens.push_back((int64_t)le32(p));
        p += 4;
    }
    // ---- record boundaries (one sequential walk), then the columns ------------------------------------------
    std::vector<const uint8_t *> recs;
    int64_t n_cig = 0, n_bases = 0;
    while (p < end) {
        if (end - p < 4) return fail(-4, "mxm_bam_read: truncated alignment record in %s", path);
        const uint32_t block_size = le32(p);
        if (block_size < 32 || (size_t)(end - p - 4) < block_size) return fail(-4, "mxm_bam_read: truncated alignment record in %s", path);
        const uint8_t *r = p + 4;
        ++out->n_records_total;
        const int32_t ref_id = (int32_t)le32(r);
        const uint32_t l_read_name = r[8], n_cigar_op = le16(r + 12), l_seq = le32(r + 16);
        if (32ull + l_read_name + 4ull * n_cigar_op + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq > (uint64_t)block_size || l_read_name == 0)
            return fail(-4, "mxm_bam_read: alignment record %s%lld is inconsistent", "", (long long)out->n_records_total - 1);
        if (ref_id >= 0) {
            recs.push_back(r);
            n_cig += n_cigar_op;
            n_bases += l_seq;
        } else {
            ++out->n_skipped_unplaced;
        }
        p += 4 + (size_t)block_size;
    }
    stamp("walk");
    const int64_t n = (int64_t)recs.size();
    out->ref_start.resize((size_t)n);
    out->mapq.resize((size_t)n);
    out->ref_id.resize((size_t)n);
    out->flag.resize((size_t)n);
    out->frag.resize((size_t)n);
    out->has_qual.resize((size_t)n);
    out->cig_ptr.assign((size_t)n + 1, 0);
    out->seq_ptr.assign((size_t)n + 1, 0);
    for (int64_t i = 0; i < n; ++i) {
        out->cig_ptr[i + 1] = out->cig_ptr[i] + le16(recs[i] + 12);
        out->seq_ptr[i + 1] = out->seq_ptr[i] + le32(recs[i] + 16);
    }
    out->cigar.resize((size_t)n_cig);
    out->seq.resize((size_t)n_bases);
    out->qual.resize((size_t)n_bases);
    static const char decode[17] = "=ACMGRSVTWYHKDBN";
    std::vector<long long> too_long((size_t)n_threads, -1);
    std::vector<uint64_t> name_hash((size_t)n);
    std::vector<uint8_t> name_len((size_t)n);
    auto fill = [&](int t) {
        const int64_t per = (n + n_threads - 1) / n_threads, lo = t * per, hi = std::min(n, lo + per);
        for (int64_t i = lo; i < hi; ++i) {
            const uint8_t *r = recs[i];
            const uint32_t l_read_name = r[8], n_cigar_op = le16(r + 12), l_seq = le32(r + 16);
            {
                const uint8_t *nm = r + 32;
                const size_t len = strnlen(reinterpret_cast<const char *>(nm), l_read_name);
                uint64_t h = 1469598103934665603ull;                      // FNV-1a
                for (size_t k = 0; k < len; ++k) h = (h ^ nm[k]) * 1099511628211ull;
                name_hash[i] = h ^ (h >> 29);
                name_len[i] = (uint8_t)len;
            }
            out->ref_id[i] = (int32_t)le32(r);
            out->ref_start[i] = (int64_t)(int32_t)le32(r + 4);
            out->mapq[i] = r[9];
            out->flag[i] = le16(r + 14);
            const uint8_t *c = r + 32 + l_read_name;
            uint32_t *cg = out->cigar.data() + out->cig_ptr[i];
            for (uint32_t k = 0; k < n_cigar_op; ++k) cg[k] = le32(c + 4 * k);
            // a CIGAR of more than 65535 operations is stored as kSmN with the real one in a CG tag (section 4.2.2)
            if (n_cigar_op == 2 && (cg[0] & 15u) == 4 && (cg[0] >> 4) == l_seq && (cg[1] & 15u) == 3 && too_long[t] < 0) too_long[t] = i;
            const uint8_t *s = c + 4 * n_cigar_op;
            uint8_t *sq = out->seq.data() + out->seq_ptr[i];
            for (uint32_t k = 0; k < l_seq; ++k) sq[k] = (uint8_t)decode[(s[k >> 1] >> ((~k & 1u) << 2)) & 15u];
            const uint8_t *q = s + (l_seq + 1) / 2;
            const bool has = l_seq > 0 && q[0] != 0xFF;
            out->has_qual[i] = has ? 1 : 0;
            if (l_seq) memcpy(out->qual.data() + out->seq_ptr[i], q, l_seq);
        }
    };
    if (n_threads == 1 || n < 4096) {
        const int keep = n_threads;
        n_threads = 1;
        fill(0);
        n_threads = keep;
    } else {
        aln_detail::run_threads(n_threads, fill);
    }
    for (size_t t = 0; t < too_long.size(); ++t)
        if (too_long[t] >= 0) return fail(-4, "mxm_bam_read: alignment %s%lld keeps its CIGAR in a CG tag (more than 65535 operations)", "", too_long[t]);
    for (int64_t i = 0; i < n; ++i) out->any_qual = out->any_qual || out->has_qual[i] != 0;
    stamp("columns");
    // ---- fragments: equal read names share an index, numbered by first appearance ---------------------------
    // Partitioned by the top bits of the names' hashes (taken by the threads above): every partition de-duplicates its own
    // records in a table of its own (open addressing over the records' name bytes), in ascending record order, so that
    // each name's representative is its FIRST record; the numbering by first appearance and the name bytes follow from
    // one prefix pass.  (As one table on one thread this was 106 ms of a 300 ms read at 1.4 * 10^6 records.)
    {
        constexpr int NPART = 64;
        const int nt = (n_threads > 1 && n >= 4096) ? n_threads : 1;
        std::vector<int64_t> first((size_t)n);
        // (1) per thread and partition: its records, ascending
        std::vector<std::vector<int64_t>> lists((size_t)nt * NPART);
        auto split = [&](int t) {
            const int64_t per = (n + nt - 1) / nt, lo = t * per, hi = std::min(n, lo + per);
            for (int64_t i = lo; i < hi; ++i) lists[(size_t)t * NPART + (size_t)(name_hash[i] >> 58)].push_back(i);
        };
        // (2) per partition: first[i] = the first record with i's name
        auto dedup = [&](int t) {
            std::vector<int64_t> slot;
            for (int part = t; part < NPART; part += nt) {
                size_t count = 0;
                for (int u = 0; u < nt; ++u) count += lists[(size_t)u * NPART + part].size();
                size_t cap = 16;
                while (cap < count * 2) cap <<= 1;
                slot.assign(cap, -1);
                for (int u = 0; u < nt; ++u) {               // thread order = ascending record order
                    for (const int64_t i : lists[(size_t)u * NPART + part]) {
                        const uint8_t *name = recs[i] + 32;
                        const size_t len = name_len[i];
                        size_t at_slot = (size_t)(name_hash[i] * 0x9e3779b97f4a7c15ull >> 20) & (cap - 1);
                        for (;;) {
                            const int64_t j = slot[at_slot];
                            if (j < 0) {
                                slot[at_slot] = i;
                                first[i] = i;
                                break;
                            }
                            if (name_hash[j] == name_hash[i] && name_len[j] == len && memcmp(recs[j] + 32, name, len) == 0) {
                                first[i] = j;
                                break;
                            }
                            at_slot = (at_slot + 1) & (cap - 1);
                        }
                    }
                }
            }
        };
        auto run = [&](const std::function<void(int)> &body) { aln_detail::run_threads(nt, body); };
        run(split);
        run(dedup);
        // (3) number the first records in file order; the names' offsets
        out->name_off.assign(1, 0);
        int64_t n_frag = 0, bytes = 0;
        for (int64_t i = 0; i < n; ++i) {
            if (first[i] == i) {
                out->frag[i] = n_frag++;
                bytes += name_len[i];
                out->name_off.push_back(bytes);
            }
        }
        out->names.resize((size_t)bytes);
        // (4) every record's fragment; the name bytes
        auto fill_frag = [&](int t) {
            const int64_t per = (n + nt - 1) / nt, lo = t * per, hi = std::min(n, lo + per);
            for (int64_t i = lo; i < hi; ++i) {
                if (first[i] == i) memcpy(out->names.data() + out->name_off[(size_t)out->frag[i]], recs[i] + 32, name_len[i]);
                else out->frag[i] = out->frag[first[i]];     // (its first record lies earlier in the file: numbered in (3))
            }
        };
        run(fill_frag);
    }
    stamp("names");
    return 0;
}

extern "C" int mxm_bam_read(const char *path, int32_t n_threads, mxm_bam **out) {
    if (out == nullptr || path == nullptr) return fail(-1, "mxm_bam_read: NULL argument%s", "");
    *out = nullptr;
    mxm_bam *res = nullptr;
    try {
        res = new mxm_bam();
        const int rc = bam_read_impl(path, n_threads, res);
        if (rc != 0) {
            delete res;
            return rc;
        }
    } catch (const std::exception &e) {
        delete res;
        return fail(-5, "mxm_bam_read: %s", e.what());
    }
    *out = res;
    return 0;
}

extern "C" int mxm_bam_sizes_of(const mxm_bam *b, mxm_bam_sizes *s) {
    if (b == nullptr || s == nullptr) return fail(-1, "mxm_bam_sizes_of: NULL argument%s", "");
    s->n_aln = (int64_t)b->ref_start.size();
    s->n_frag = (int64_t)b->name_off.size() - 1;
    s->n_cigar = (int64_t)b->cigar.size();
    s->n_bases = (int64_t)b->seq.size();
    s->names_bytes = (int64_t)b->names.size();
    s->n_ref = (int64_t)b->ref_names.size();
    s->n_records_total = b->n_records_total;
    s->n_skipped_unplaced = b->n_skipped_unplaced;
    return 0;
}

extern "C" int mxm_bam_columns(const mxm_bam *b, mxm_aln_columns *cols) {
    if (b == nullptr || cols == nullptr) return fail(-1, "mxm_bam_columns: NULL argument%s", "");
    cols->n_aln = (int64_t)b->ref_start.size();
    cols->n_frag = (int64_t)b->name_off.size() - 1;
    cols->ref_start = b->ref_start.data();
    cols->mapq = b->mapq.data();
    cols->frag = b->frag.data();
    cols->cig_ptr = b->cig_ptr.data();
    cols->cigar = b->cigar.data();
    cols->seq_ptr = b->seq_ptr.data();
    cols->seq = b->seq.data();
    cols->qual = b->any_qual ? b->qual.data() : nullptr;
    cols->has_qual = b->has_qual.data();
    return 0;
}

extern "C" int mxm_bam_fetch_names(const mxm_bam *b, char *names, int64_t *name_off, int32_t *ref_id, uint16_t *flag) {
    if (b == nullptr) return fail(-1, "mxm_bam_fetch_names: NULL handle%s", "");
    if (names != nullptr && !b->names.empty()) memcpy(names, b->names.data(), b->names.size());
    if (name_off != nullptr) memcpy(name_off, b->name_off.data(), b->name_off.size() * sizeof(int64_t));
    if (ref_id != nullptr && !b->ref_id.empty()) memcpy(ref_id, b->ref_id.data(), b->ref_id.size() * sizeof(int32_t));
    if (flag != nullptr && !b->flag.empty()) memcpy(flag, b->flag.data(), b->flag.size() * sizeof(uint16_t));
    return 0;
}

extern "C" void mxm_bam_free(mxm_bam *b) { delete b; }

#endif  // MIXEMT_BAM_READER_HPP
