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

struct PtrOps {
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
