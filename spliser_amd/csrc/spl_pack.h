// spl_pack.h -- the read layout the counting kernels consume, and the host packer that produces it.
//
// What checkBam reads from a SAM line is flag, POS and CIGAR (SpliSER_v0_1_8.py:434-437); the ABI hands those over BAM-native
// (spl_reads: pos, flag, cig_off, cigar).  The kernels want something else: everything a typical read needs in ONE memory trip,
// reads of one kind next to each other so that a wave runs one code path, and no wider than the kind needs.  That layout is
// produced on the HOST, once per read set, while the data is on its way to the GPU anyway (by the BAM decoder's threads for a
// file, by spl_reads_upload's threads for caller arrays) -- the device never sees the BAM-native arrays and never spends a pass
// on re-arranging them.
//
// A read set is cut into CHUNKS of SPL_CHUNK consecutive reads (file order; one workgroup of the range kernel per chunk).
// Inside a chunk the reads are stably partitioned into four RUNS by class, each run an array of fixed-size records:
//
//   SIMPLE  8 B   {pos, flag | len << 16}                                  one aligned op (M, =, X), mapped, len < 65536:
//                                                                           every unspliced short read
//   MNM    16 B   {pos, flag | a << 16, d, b}                              aligned a, N d, aligned b; a < 65536: once-spliced
//   M2     24 B   {pos, flag | a << 16, d1, b | c << 16, d2, 0}            aligned, N, aligned, N, aligned; a, b, c < 65536
//   OTHER  24 B   {pos, fn, op0, op1, op2 | index, n_ops}                  everything else.  fn = flag | min(n_ops, SPL_NOPS_SAT)
//                                                                           << 16 | sub-class << 29; NARROW: at most three ops,
//                                                                           all inline (absent = 0xf); WIDE: op0, op1 and the
//                                                                           index of op 0 in the segment's array of wide ops
//
// Ops that do not consume the reference (S, H, I, P, undefined codes) change nothing for any path of checkBam (:457-464: no
// progress, no test): a CIGAR of at most SPL_PACK_SCAN_OPS ops is classified and packed without them ("5S95M100N50M" is a
// once-spliced read).  Longer CIGARs are WIDE as they are.  Every class except OTHER also requires: not flagged unmapped (0x4),
// POS >= 0 and the read's end within SPL_COORD_MAX -- reads that fail are OTHER and get the general path's checks.
//
// Run r of a chunk starts at byte spl_run_offset(n, r) of the chunk's record area (16-byte aligned).
#ifndef SPL_PACK_H
#define SPL_PACK_H

#include <stdint.h>

#ifndef SPL_CHUNK
#define SPL_CHUNK 2048                   // consecutive reads per chunk = per workgroup of the range kernel
#define SPL_CHUNK_SHIFT 11
#endif
// Read sets of SPL_BIG_SET_READS reads and more are cut into chunks of twice the size: a workgroup's fixed costs (window base,
// LDS clearing, hand-over of its lists) are paid half as often, 3.5 % on a 100 M-read launch -- and 15 % the other way on a
// 20 M-read one, whose grid then has too few workgroups for its last round.
#define SPL_CHUNK_BIG (2 * SPL_CHUNK)
#define SPL_CHUNK_BIG_SHIFT (SPL_CHUNK_SHIFT + 1)
#define SPL_BIG_SET_READS 64000000
#define SPL_PACK_SCAN_OPS 8              // CIGARs up to this many ops are packed without their non-consuming ops
#define SPL_NOPS_SAT 0x1fffu
#define SPL_RC_SHIFT 29
// runs of a chunk, in order
#define SPL_RC_SIMPLE 0u
#define SPL_RC_MNM 1u
#define SPL_RC_M2 2u
#define SPL_RC_OTHER 3u
#define SPL_RC_RUNS 4
// sub-classes of OTHER (top bits of fn)
#define SPL_RC_NARROW 3u
#define SPL_RC_WIDE 4u
#define SPL_REC_SIMPLE 8
#define SPL_REC_MNM 16
#define SPL_REC_M2 24
#define SPL_REC_OTHER 24
#ifndef SPL_W_SIMPLE                     // what a read of each class costs the range kernel, roughly (chunk order: longest first)
#define SPL_W_SIMPLE 2u
#define SPL_W_MNM 5u
#define SPL_W_M2 9u
#define SPL_W_NARROW 6u
#define SPL_W_WIDE 14u
#endif
#define SPL_INLINE_OPS 3                 // CIGAR ops of an OTHER record resolved in the straight-line part
// Coordinates (read end, site position) must stay <= SPL_COORD_MAX so that t+1, cur and x - dbase never wrap int32.
#define SPL_COORD_MAX 2147483581
// op code -> 2-bit kind: M(0)=1 D(2)=3 N(3)=2 =(7)=1 X(8)=1, others 0 (does not consume the reference)
#define SPL_KIND_TABLE ((1u << 0) | (3u << 4) | (2u << 6) | (1u << 14) | (1u << 16))

#if defined(__HIPCC__)
#define SPL_PACK_HD __host__ __device__ inline
#else
#define SPL_PACK_HD inline
#endif

// Byte offset of run r (0..3; 4 = size of the record area) inside a chunk with n[0..3] reads per run.
SPL_PACK_HD uint32_t spl_run_offset(const uint16_t *n, int r)
{
    uint32_t off = 0;
    if (r >= 1) off = ((uint32_t)n[0] * SPL_REC_SIMPLE + 15u) & ~15u;
    if (r >= 2) off += (uint32_t)n[1] * SPL_REC_MNM;
    if (r >= 3) off += (uint32_t)n[2] * SPL_REC_M2;
    if (r >= 4) off = (off + (uint32_t)n[3] * SPL_REC_OTHER + 15u) & ~15u;
    return off;
}

// What the kernels get per chunk (one 32-byte scalar load): where its records are, where its segment's wide ops are, the
// shift that moves its segment into the shard's coordinate space (spliser_amd/shard.py) and the reads per run.
struct spl_chunk_meta {
    uint64_t rec;        // device address of the chunk's record area
    uint64_t wide;       // device address of the wide-op array of the chunk's segment
    int32_t shift;       // added to every pos of the chunk
    int32_t first_pos;   // POS of the chunk's first read in file order (unshifted): base of the workgroup's LDS window
    uint16_t n[SPL_RC_RUNS];
};

// One read -> its record: shared by the host packer (spl_pack.cpp) and the device packer (spl_devpack.hip), so that a read set
// packed on either side is the same bytes.
namespace splrec {
SPL_PACK_HD uint32_t kind_of(uint32_t op) { return (SPL_KIND_TABLE >> (2u * (op & 15u))) & 3u; }

struct Rec {
    uint32_t run;   // SPL_RC_SIMPLE .. SPL_RC_OTHER
    uint32_t w[6];  // the record's words (as many as the run's record size says)
    uint32_t n_wide; // ops that go to the wide array (WIDE reads: all of them)
    uint32_t weight;
};

// What checkBam's walk depends on (:436-559), in the form the range kernel wants it.  One read; ops(k) is its k-th CIGAR op
// (the host packer reads them through a pointer, the layout kernel out of its workgroup's LDS: one classifier, the same bytes).
template <class Ops>
SPL_PACK_HD void classify_ops(int32_t pos, uint32_t flag, const Ops &ops, uint32_t n_all, uint32_t wide_index, Rec &r)
{
    uint32_t n = n_all;
    uint32_t c5[5] = {0xfu, 0xfu, 0xfu, 0xfu, 0xfu}; // the first five reference-consuming ops
    uint32_t m = 0xffffffffu;                        // their number, if the CIGAR was short enough to look
    uint32_t w0 = 0xfu, w1 = 0xfu, w2 = 0xfu;
    if (n_all <= (uint32_t)SPL_PACK_SCAN_OPS) {
        m = 0;
        for (uint32_t k = 0; k < n_all; ++k) {
            const uint32_t op = ops(k);
            if (kind_of(op) == 0u) continue;
            for (uint32_t q = 0; q < 5u; ++q) c5[q] = m == q ? op : c5[q]; // (no indexing by m: the five stay in registers on the device)
            ++m;
        }
        if (m <= 3u) { n = m; w0 = c5[0]; w1 = c5[1]; w2 = c5[2]; }
    }
    const bool wide = n > 3u;
    if (wide) { w0 = ops(0u); w1 = ops(1u); w2 = wide_index; n = n_all; }
    const bool placed = !(flag & 4u) && pos >= 0;
    const int64_t room = (int64_t)SPL_COORD_MAX - (int64_t)pos;
    r.n_wide = 0;
    if (placed && !wide) {
        if (n == 1u && kind_of(w0) == 1u && (w0 >> 4) < 65536u && (int64_t)(w0 >> 4) <= room) {
            r.run = SPL_RC_SIMPLE;
            r.w[0] = (uint32_t)pos;
            r.w[1] = flag | ((w0 >> 4) << 16);
            r.weight = SPL_W_SIMPLE;
            return;
        }
        if (n == 3u && kind_of(w0) == 1u && kind_of(w1) == 2u && kind_of(w2) == 1u && (w0 >> 4) < 65536u &&
            (int64_t)(w0 >> 4) + (int64_t)(w1 >> 4) + (int64_t)(w2 >> 4) <= room) {
            r.run = SPL_RC_MNM;
            r.w[0] = (uint32_t)pos;
            r.w[1] = flag | ((w0 >> 4) << 16);
            r.w[2] = w1 >> 4;
            r.w[3] = w2 >> 4;
            r.weight = SPL_W_MNM;
            return;
        }
    }
    if (placed && m == 5u) { // twice-spliced: five lengths
        const uint32_t la = c5[0] >> 4, d1 = c5[1] >> 4, lb = c5[2] >> 4, d2 = c5[3] >> 4, lc = c5[4] >> 4;
        if (kind_of(c5[0]) == 1u && kind_of(c5[1]) == 2u && kind_of(c5[2]) == 1u && kind_of(c5[3]) == 2u && kind_of(c5[4]) == 1u &&
            la < 65536u && lb < 65536u && lc < 65536u && (int64_t)la + d1 + lb + d2 + lc <= room) {
            r.run = SPL_RC_M2;
            r.w[0] = (uint32_t)pos;
            r.w[1] = flag | (la << 16);
            r.w[2] = d1;
            r.w[3] = lb | (lc << 16);
            r.w[4] = d2;
            r.w[5] = 0u;
            r.weight = SPL_W_M2;
            return;
        }
    }
    r.run = SPL_RC_OTHER;
    r.w[0] = (uint32_t)pos;
    r.w[1] = flag | ((n < SPL_NOPS_SAT ? n : SPL_NOPS_SAT) << 16) | ((wide ? SPL_RC_WIDE : SPL_RC_NARROW) << SPL_RC_SHIFT);
    r.w[2] = w0; r.w[3] = w1; r.w[4] = w2;
    r.w[5] = n;
    r.n_wide = wide ? n_all : 0u;
    r.weight = wide ? SPL_W_WIDE : SPL_W_NARROW;
}

// The same decision with fewer instructions, for the layout kernel (spl_devpack.hip), where classifying 200 M reads was what the
// launch took: a CIGAR of up to SPL_PACK_SCAN_OPS ops is looked at as a bit mask of its reference-consuming ops -- when every op
// consumes (no clips, no insertions: most reads) the first five are the ops as they stand, otherwise the j-th consuming op is
// the op at the mask's j-th set bit (count-trailing-zeros, no per-op select chains).  Must give what classify_ops gives, field
// for field: tests/test_pack_host.py holds the two against each other on the host.
SPL_PACK_HD uint32_t ctz32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_ctz(x);
#else
    uint32_t n = 0;
    while (n < 32u && !((x >> n) & 1u)) ++n;
    return n;
#endif
}
// The record of a read whose reference-consuming ops are known: m of them (0xffffffff: a CIGAR too long to look at), the first
// five c0 .. c4 (what stands in those beyond m is looked at by nobody), and the CIGAR's first two ops as they stand, op0 and op1
// (a WIDE record's words; anything when the CIGAR has fewer than four ops).  Straight-line: selects, no branches -- on the device
// every branch here is a mask juggle and a merge of seven registers; coordinates in 32 bits (five lengths of 28 bits cannot wrap).
SPL_PACK_HD void finish_record(int32_t pos, uint32_t flag, uint32_t m, uint32_t n_all, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t c4,
                               uint32_t op0, uint32_t op1, uint32_t wide_index, Rec &r)
{
    const bool wide = m > 3u;
    const uint32_t n = wide ? n_all : m;
    // placed, and in the coordinate space at all (a POS beyond it leaves no room for any length, not even 0: OTHER)
    const bool placed = !(flag & 4u) && pos >= 0 && pos <= (int32_t)SPL_COORD_MAX;
    const uint32_t room = (uint32_t)SPL_COORD_MAX - (uint32_t)pos;
    const uint32_t l0 = c0 >> 4, l1 = c1 >> 4, l2 = c2 >> 4, l3 = c3 >> 4, l4 = c4 >> 4;
    const bool k0 = placed && kind_of(c0) == 1u && l0 < 65536u, k2 = kind_of(c2) == 1u && kind_of(c1) == 2u;
    const bool simple = k0 && m == 1u && l0 <= room;
    const bool mnm = k0 && m == 3u && k2 && l0 + l1 + l2 <= room;
    const bool m2 = k0 && m == 5u && k2 && kind_of(c3) == 2u && kind_of(c4) == 1u && l2 < 65536u && l4 < 65536u && l0 + l1 + l2 + l3 + l4 <= room;
    const bool other = !(simple || mnm || m2);
    const uint32_t x0 = wide ? op0 : (m > 0u ? c0 : 0xfu), x1 = wide ? op1 : (m > 1u ? c1 : 0xfu), x2 = wide ? wide_index : (m > 2u ? c2 : 0xfu);
    r.run = simple ? SPL_RC_SIMPLE : (mnm ? SPL_RC_MNM : (m2 ? SPL_RC_M2 : SPL_RC_OTHER));
    r.w[0] = (uint32_t)pos;
    r.w[1] = flag | (other ? ((n < SPL_NOPS_SAT ? n : SPL_NOPS_SAT) << 16) | ((wide ? SPL_RC_WIDE : SPL_RC_NARROW) << SPL_RC_SHIFT) : l0 << 16);
    r.w[2] = other ? x0 : l1;
    r.w[3] = other ? x1 : (m2 ? l2 | (l4 << 16) : l2);
    r.w[4] = other ? x2 : l3;
    r.w[5] = other ? n : 0u;
    r.n_wide = wide && other ? n_all : 0u;
    r.weight = simple ? SPL_W_SIMPLE : (mnm ? SPL_W_MNM : (m2 ? SPL_W_M2 : (wide ? SPL_W_WIDE : SPL_W_NARROW)));
}

// The layout kernel's straight-line path: a read whose CIGAR is at most five ops that all consume the reference, between at most ONE
// op in front and ONE behind that do not (soft / hard clips as a local aligner writes them: "5S95M100N50M", "70M80N75M5S" -- to
// checkBam's walk such ops are nothing, SpliSER_v0_1_8.py:457-464).  p0 .. p4 = the ops from the first one behind the leading clip on
// (what stands in those beyond n_eff is anything and is masked), n_eff = the ops between the clips; op0, op1, n_all = the CIGAR as
// it stands (a WIDE record keeps its first two ops and its length as they are).  -> true when the n_eff ops ARE at most five and
// all consume; r is the read's record then (field for field classify_ops's), garbage otherwise.
// The five kinds side by side, two bits each: a class is one comparison of that signature.
SPL_PACK_HD bool classify_clip5(int32_t pos, uint32_t flag, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, uint32_t p4, uint32_t n_eff,
                                uint32_t op0, uint32_t op1, uint32_t n_all, uint32_t wide_index, Rec &r)
{
    const uint32_t m = n_eff < 5u ? n_eff : 5u;
    const uint32_t used = (1u << (2u * m)) - 1u;
    const uint32_t sig = (kind_of(p0) | (kind_of(p1) << 2) | (kind_of(p2) << 4) | (kind_of(p3) << 6) | (kind_of(p4) << 8)) & used;
    const bool fast = n_eff <= 5u && ((sig | (sig >> 1)) & 0x155u) == (0x155u & used); // every op between the clips consumes
    const bool placed = !(flag & 4u) && pos >= 0 && pos <= (int32_t)SPL_COORD_MAX;
    const uint32_t room = (uint32_t)SPL_COORD_MAX - (uint32_t)pos;
    const uint32_t l0 = p0 >> 4, l1 = p1 >> 4, l2 = p2 >> 4, l3 = p3 >> 4, l4 = p4 >> 4;
    const bool lim0 = placed && l0 < 65536u;
    // (all ops consume, so a signature says how many there are: 01 = M; 01 10 01 = M N M; 01 10 01 10 01 = M N M N M)
    const bool simple = lim0 && sig == 0x1u && l0 <= room;
    const bool mnm = lim0 && sig == 0x19u && l0 + l1 + l2 <= room;
    const bool m2 = lim0 && sig == 0x199u && l2 < 65536u && l4 < 65536u && l0 + l1 + l2 + l3 + l4 <= room;
    const bool other = !(simple || mnm || m2);
    const bool wide = m > 3u;
    const uint32_t n_rec = wide ? (n_all < SPL_NOPS_SAT ? n_all : SPL_NOPS_SAT) : m;
    const uint32_t x0 = wide ? op0 : (m > 0u ? p0 : 0xfu), x1 = wide ? op1 : (m > 1u ? p1 : 0xfu), x2 = wide ? wide_index : (m > 2u ? p2 : 0xfu);
    r.run = simple ? SPL_RC_SIMPLE : (mnm ? SPL_RC_MNM : (m2 ? SPL_RC_M2 : SPL_RC_OTHER));
    r.w[0] = (uint32_t)pos;
    r.w[1] = flag | (other ? (n_rec << 16) | ((wide ? SPL_RC_WIDE : SPL_RC_NARROW) << SPL_RC_SHIFT) : l0 << 16);
    r.w[2] = other ? x0 : l1;
    r.w[3] = other ? x1 : (m2 ? l2 | (l4 << 16) : l2);
    r.w[4] = other ? x2 : l3;
    r.w[5] = other ? (wide ? n_all : m) : 0u;
    r.n_wide = wide && other ? n_all : 0u;
    r.weight = simple ? SPL_W_SIMPLE : (mnm ? SPL_W_MNM : (m2 ? SPL_W_M2 : (wide ? SPL_W_WIDE : SPL_W_NARROW)));
    return fast;
}
// The three shapes nine reads in ten have, decided from the op CODES alone: "M", "M N M", "M N M N M" with op M itself (an aligner
// writes = and X only when asked to; those, like everything else, are the other classifiers' business).  A read that is one of the
// three AND within the limits of its record (placed, lengths below 65536 where the record holds them in 16 bits, its end inside the
// coordinate space) -> true and its record in r, field for field classify_ops's; false -> r is garbage.  Half the instructions of
// classify_fast5, which decides every class from the kinds' signature: in the fused kernel this is what every read takes first.
SPL_PACK_HD bool classify_plain(int32_t pos, uint32_t flag, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3, uint32_t o4, uint32_t n_all, Rec &r)
{
    const bool placed = !(flag & 4u) && pos >= 0 && pos <= (int32_t)SPL_COORD_MAX;
    const uint32_t room = (uint32_t)SPL_COORD_MAX - (uint32_t)pos;
    const uint32_t l0 = o0 >> 4, l1 = o1 >> 4, l2 = o2 >> 4, l3 = o3 >> 4, l4 = o4 >> 4;
    const bool m0 = placed && (o0 & 15u) == 0u && l0 < 65536u;
    const bool mnm_codes = ((o2 & 15u) | ((o1 & 15u) ^ 3u)) == 0u;   // op 1 is N, op 2 is M
    const bool m2_codes = ((o4 & 15u) | ((o3 & 15u) ^ 3u)) == 0u;    // op 3 is N, op 4 is M
    const bool simple = m0 && n_all == 1u && l0 <= room;
    const bool mnm = m0 && n_all == 3u && mnm_codes && l0 + l1 + l2 <= room;
    const bool m2 = m0 && n_all == 5u && mnm_codes && m2_codes && l2 < 65536u && l4 < 65536u && l0 + l1 + l2 + l3 + l4 <= room;
    r.run = simple ? SPL_RC_SIMPLE : (mnm ? SPL_RC_MNM : SPL_RC_M2);
    r.w[0] = (uint32_t)pos;
    r.w[1] = flag | l0 << 16;
    r.w[2] = l1;
    r.w[3] = m2 ? l2 | (l4 << 16) : l2;
    r.w[4] = l3;
    r.w[5] = 0u;
    r.n_wide = 0u;
    r.weight = simple ? SPL_W_SIMPLE : (mnm ? SPL_W_MNM : SPL_W_M2);
    return simple || mnm || m2;
}

// ... without clips: the CIGAR's first five ops as they stand
SPL_PACK_HD bool classify_fast5(int32_t pos, uint32_t flag, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3, uint32_t o4, uint32_t n_all,
                                uint32_t wide_index, Rec &r)
{
    return classify_clip5(pos, flag, o0, o1, o2, o3, o4, n_all, o0, o1, n_all, wide_index, r);
}
// ... and with them: the clips found and stepped over.  `ops` must be readable for every k < SPL_PACK_SCAN_OPS whatever the read's
// number of ops (padded).  -> true for CIGARs of at most seven ops that are [clip] + at most five consuming ops + [clip].
template <class Ops>
SPL_PACK_HD bool classify_clipped(int32_t pos, uint32_t flag, const Ops &ops, uint32_t n_all, uint32_t wide_index, Rec &r)
{
    const uint32_t f0 = ops(0u), last = ops((n_all - 1u) & 7u);
    const uint32_t lead = (n_all > 0u && kind_of(f0) == 0u) ? 1u : 0u;
    const uint32_t trail = (n_all > lead && kind_of(last) == 0u) ? 1u : 0u;
    const uint32_t n_eff = n_all - lead - trail;
    const bool fast = classify_clip5(pos, flag, ops(lead), ops(lead + 1u), ops(lead + 2u), ops(lead + 3u), ops(lead + 4u), n_eff, f0, ops(1u), n_all, wide_index, r);
    return fast && n_all <= 7u;
}

// bit (code) set: the op consumes the reference (M D N = X)
#define SPL_CONSUMES_BITS ((1u << 0) | (1u << 2) | (1u << 3) | (1u << 7) | (1u << 8))
// Which of a CIGAR's first five ops consume the reference, as a bit mask cut to its n_all ops (n_all <= 5).
SPL_PACK_HD uint32_t consuming_mask5(uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3, uint32_t o4, uint32_t n_all)
{
    return (((SPL_CONSUMES_BITS >> (o0 & 15u)) & 1u) | (((SPL_CONSUMES_BITS >> (o1 & 15u)) & 1u) << 1) | (((SPL_CONSUMES_BITS >> (o2 & 15u)) & 1u) << 2) |
            (((SPL_CONSUMES_BITS >> (o3 & 15u)) & 1u) << 3) | (((SPL_CONSUMES_BITS >> (o4 & 15u)) & 1u) << 4)) & ((1u << n_all) - 1u);
}

// An accessor with `padded = true` promises that ops(k) may be called for every k < SPL_PACK_SCAN_OPS whatever the read's number
// of ops (what comes back beyond them is ignored): the first five ops are then five unconditional reads.
template <class Ops>
SPL_PACK_HD void classify_lean(int32_t pos, uint32_t flag, const Ops &ops, uint32_t n_all, uint32_t wide_index, Rec &r)
{
    uint32_t m = 0xffffffffu, c0 = 0xfu, c1 = 0xfu, c2 = 0xfu, c3 = 0xfu, c4 = 0xfu;
    uint32_t o0 = 0xfu, o1 = 0xfu, o2 = 0xfu, o3 = 0xfu, o4 = 0xfu;
    if (Ops::padded) {
        o0 = ops(0u); o1 = ops(1u); o2 = ops(2u); o3 = ops(3u); o4 = ops(4u);
    } else {
        if (n_all > 0u) o0 = ops(0u);
        if (n_all > 1u) o1 = ops(1u);
        if (n_all > 2u) o2 = ops(2u);
        if (n_all > 3u) o3 = ops(3u);
        if (n_all > 4u) o4 = ops(4u);
    }
    if (n_all <= (uint32_t)SPL_PACK_SCAN_OPS) {
        uint32_t hi = 0; // the consuming bits of ops 5..7
        if (n_all > 5u) {
            hi = ((SPL_CONSUMES_BITS >> (ops(5u) & 15u)) & 1u) << 5;
            if (n_all > 6u) hi |= ((SPL_CONSUMES_BITS >> (ops(6u) & 15u)) & 1u) << 6;
            if (n_all > 7u) hi |= ((SPL_CONSUMES_BITS >> (ops(7u) & 15u)) & 1u) << 7;
        }
        const uint32_t full = (1u << n_all) - 1u;
        const uint32_t mask = consuming_mask5(o0, o1, o2, o3, o4, n_all < 5u ? n_all : 5u) | hi;
        if (mask == full) { // every op consumes the reference: nothing to drop, the first five are the ops as they stand
            m = n_all;
            c0 = o0; c1 = o1; c2 = o2; c3 = o3; c4 = o4;
        } else { // the j-th consuming op is the op at the mask's j-th set bit
            uint32_t mm = mask;
            m = 0;
            if (mm) { c0 = ops(ctz32(mm)); mm &= mm - 1u; ++m; }
            if (mm) { c1 = ops(ctz32(mm)); mm &= mm - 1u; ++m; }
            if (mm) { c2 = ops(ctz32(mm)); mm &= mm - 1u; ++m; }
            if (mm) { c3 = ops(ctz32(mm)); mm &= mm - 1u; ++m; }
            if (mm) { c4 = ops(ctz32(mm)); mm &= mm - 1u; ++m; }
            while (mm) { mm &= mm - 1u; ++m; }
        }
    }
    finish_record(pos, flag, m, n_all, c0, c1, c2, c3, c4, o0, o1, wide_index, r);
}

struct PtrOps {
    static constexpr bool padded = false;
    const uint32_t *p;
    SPL_PACK_HD uint32_t operator()(uint32_t k) const { return p[k]; }
};
SPL_PACK_HD void classify(int32_t pos, uint32_t flag, const uint32_t *ops, uint32_t n_all, uint32_t wide_index, Rec &r)
{
    classify_ops(pos, flag, PtrOps{ops}, n_all, wide_index, r);
}

} // namespace splrec

#include <stddef.h>

#include <vector>

namespace splpack {

// One stretch of BAM-native reads: ops of read k are cigar[cig_off[k] .. cig_off[k + 1]).
struct Part {
    const int32_t *pos;
    const uint16_t *flag;
    const uint32_t *cig_off;
    const uint32_t *cigar;
    int64_t n;
};

// Parts laid end to end = the reads of one segment (one reference of a BAM file, or a caller's arrays), file order.
struct Source {
    std::vector<Part> parts;
    std::vector<int64_t> first; // first[k] = reads before part k; first[parts.size()] = n_reads
    int64_t n_reads = 0, n_ops = 0;
    void add(const Part &p)
    {
        if (first.empty()) first.push_back(0);
        parts.push_back(p);
        n_reads += p.n;
        n_ops += p.n ? (int64_t)(p.cig_off[p.n] - p.cig_off[0]) : 0;
        first.push_back(n_reads);
    }
};

struct ChunkDesc {
    uint64_t rec_off;  // byte offset of the chunk's record area in the segment's record blob
    uint64_t wide_off; // index of the chunk's first wide op in the segment's wide-op array
    int32_t first_pos;
    uint32_t cost;     // what the chunk will cost the range kernel, roughly (SPL_W_*)
    uint16_t n[SPL_RC_RUNS];
};

struct Plan {
    uint32_t chunk = SPL_CHUNK; // reads per chunk (SPL_CHUNK or SPL_CHUNK_BIG): set before plan()
    std::vector<ChunkDesc> chunks;
    uint64_t rec_bytes = 0, n_wide = 0;
};

// Pass 1: classify every read, size every chunk.  Pass 2: write the records of chunks [c0, c1) to rec_dst (= where chunk c0's
// record area goes) and their wide ops to wide_dst (= where chunk c0's wide ops go), with the wide-op indexes in the records
// relative to the segment's array.  emit() is single-threaded over its range; callers run ranges in parallel.
void plan(const Source &src, Plan &out, int n_threads);
void emit(const Source &src, const Plan &plan, size_t c0, size_t c1, uint8_t *rec_dst, uint32_t *wide_dst);

// Run fn(k) for k in [0, n) on up to n_threads threads (the calling thread included).
void parallel_for(size_t n, int n_threads, void (*fn)(size_t, void *), void *arg);

} // namespace splpack

#endif // SPL_PACK_H
