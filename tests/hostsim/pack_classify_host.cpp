// Test-only host build of the two read classifiers of spliser_amd/csrc/spl_pack.h: classify_ops (what the host packer runs, the
// statement tests/test_pack_host.py holds to a Python restatement of the layout rules) and classify_lean (what the layout kernel
// runs on the device: the same decision in fewer instructions).  Not part of the product.
#include <cstdint>
#include <cstring>

#include "../../spliser_amd/csrc/spl_pack.h"

// -> number of reads on which the two disagree in any field the layout uses; first_bad = index of the first one (or -1)
extern "C" int64_t classify_compare(int64_t n_reads, const int32_t *pos, const uint16_t *flag, const uint32_t *cig_off, const uint32_t *cigar,
                                    int64_t *first_bad)
{
    int64_t bad = 0;
    *first_bad = -1;
    for (int64_t i = 0; i < n_reads; ++i) {
        splrec::Rec a, b;
        memset(&a, 0, sizeof a);
        memset(&b, 0, sizeof b);
        const uint32_t *ops = cigar + cig_off[i];
        const uint32_t n = cig_off[i + 1] - cig_off[i];
        splrec::classify_ops(pos[i], flag[i], splrec::PtrOps{ops}, n, 12345u + (uint32_t)i, a);
        splrec::classify_lean(pos[i], flag[i], splrec::PtrOps{ops}, n, 12345u + (uint32_t)i, b);
        const uint32_t words = a.run == SPL_RC_SIMPLE ? 2u : (a.run == SPL_RC_MNM ? 4u : 6u);
        bool same = a.run == b.run && a.n_wide == b.n_wide && a.weight == b.weight;
        for (uint32_t q = 0; q < words && same; ++q) same = a.w[q] == b.w[q];
        // ... and the straight-line path of the layout kernel: the read's first five ops with WHATEVER follows them in the array
        // standing in for those it does not have (the caller pads the array); it must say "mine" exactly for the reads of at most
        // five ops that all consume the reference, and give classify_ops's record for those
        {
            splrec::Rec f;
            memset(&f, 0, sizeof f);
            const bool mine = splrec::classify_fast5(pos[i], flag[i], ops[0], ops[1], ops[2], ops[3], ops[4], n, 12345u + (uint32_t)i, f);
            bool all = n <= 5u;
            for (uint32_t q = 0; q < n && all; ++q) all = splrec::kind_of(ops[q]) != 0u;
            same = same && mine == all;
            if (mine && all) {
                same = same && a.run == f.run && a.n_wide == f.n_wide && a.weight == f.weight;
                for (uint32_t q = 0; q < words && same; ++q) same = a.w[q] == f.w[q];
            }
        }
        // ... and the path for reads with a clip in front and / or behind (classify_clipped): "mine" exactly for CIGARs of at most seven
        // ops that are [one non-consuming op] + at most five ops that all consume + [one non-consuming op], classify_ops's record for those
        {
            splrec::Rec f;
            memset(&f, 0, sizeof f);
            const bool mine = splrec::classify_clipped(pos[i], flag[i], splrec::PtrOps{ops}, n, 12345u + (uint32_t)i, f);
            uint32_t lo = 0, hi = n;
            if (hi > lo && splrec::kind_of(ops[lo]) == 0u) ++lo;
            if (hi > lo && splrec::kind_of(ops[hi - 1u]) == 0u) --hi;
            bool all = n <= 7u && hi - lo <= 5u;
            for (uint32_t q = lo; q < hi && all; ++q) all = splrec::kind_of(ops[q]) != 0u;
            same = same && mine == all;
            if (mine && all) {
                same = same && a.run == f.run && a.n_wide == f.n_wide && a.weight == f.weight;
                for (uint32_t q = 0; q < words && same; ++q) same = a.w[q] == f.w[q];
            }
        }
        // ... and the first thing the fused kernel asks (classify_plain): "mine" for the reads that are M, M N M or M N M N M with op M
        // itself and get a record of those runs; classify_ops's record then
        {
            splrec::Rec f;
            memset(&f, 0, sizeof f);
            const bool mine = splrec::classify_plain(pos[i], flag[i], ops[0], ops[1], ops[2], ops[3], ops[4], n, f);
            bool shape = (n == 1u || n == 3u || n == 5u) && a.run != SPL_RC_OTHER;
            for (uint32_t q = 0; q < n && shape; ++q) shape = (ops[q] & 15u) == ((q & 1u) ? 3u : 0u);
            same = same && mine == shape;
            if (mine && shape) {
                same = same && a.run == f.run && a.n_wide == f.n_wide && a.weight == f.weight;
                for (uint32_t q = 0; q < words && same; ++q) same = a.w[q] == f.w[q];
            }
        }
        if (!same) {
            if (*first_bad < 0) *first_bad = i;
            ++bad;
        }
    }
    return bad;
}
