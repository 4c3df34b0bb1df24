// Host-only harness for csrc/bam_reader.hpp (tests/test_bam_reader.py builds it with g++ -fsanitize=address,undefined):
// reads every file named on the command line and prints the return code and the sizes; a sanitizer report is a failure.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, const char *a = "", long long b = 0, long long c = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, b, c);
    return code;
}

#include "aln_encode.hpp"
#include "bam_reader.hpp"

int main(int argc, char **argv) {
    for (int i = 1; i < argc; ++i) {
        for (int threads = 1; threads <= 3; threads += 2) {
            mxm_bam *bam = nullptr;
            const int rc = mxm_bam_read(argv[i], threads, &bam);
            mxm_bam_sizes sz = {};
            long long check = 0;
            if (rc == 0) {
                mxm_bam_sizes_of(bam, &sz);
                mxm_aln_columns c;
                mxm_bam_columns(bam, &c);
                for (int64_t k = 0; k < c.n_aln; ++k) check += c.ref_start[k] + c.mapq[k] + c.frag[k] + c.has_qual[k];
                for (int64_t k = 0; k < sz.n_cigar; ++k) check += c.cigar[k];
                for (int64_t k = 0; k < sz.n_bases; ++k) check += c.seq[k] + (c.qual ? c.qual[k] : 0);
                // ... and through the batched encoder (csrc/aln_encode.hpp) on the same threads: every 5th position a site
                const int64_t ref_len = 17000;
                std::vector<int32_t> site_of_pos((size_t)ref_len, -1);
                std::vector<int64_t> site_pos;
                for (int64_t pos = 3; pos < ref_len; pos += 5) {
                    site_of_pos[(size_t)pos] = (int32_t)site_pos.size();
                    site_pos.push_back(pos);
                }
                mxm_aln_enc *enc = nullptr;
                const int erc = mxm_aln_encode(&c, site_of_pos.data(), ref_len, site_pos.data(), (int32_t)site_pos.size(), 20, 20,
                                               threads, &enc);
                if (erc == 0) {
                    mxm_aln_sizes es = {};
                    mxm_aln_sizes_of(enc, &es);
                    check += es.n_rows * 31 + es.nnz;
                    mxm_aln_free(enc);
                } else {
                    check -= erc;
                }
            }
            printf("%s threads=%d rc=%d n_aln=%lld n_frag=%lld check=%lld %s\n", argv[i], threads, rc, (long long)sz.n_aln,
                   (long long)sz.n_frag, check, rc ? g_err : "");
            mxm_bam_free(bam);
        }
    }
    return 0;
}
