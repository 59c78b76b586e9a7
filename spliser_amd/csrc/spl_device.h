// spl_device.h -- kernel parameter blocks and launch geometry shared by spl_kernels.hip and spl_capi.cpp.
#ifndef SPL_DEVICE_H
#define SPL_DEVICE_H

#include <stdint.h>
#include <hip/hip_vector_types.h>

#include "spl_pack.h"
#include "spl_devpack.h"

// Launch geometry of the classification kernels (see DESIGN.md "Kernels").
#ifndef SPL_BLOCK
#define SPL_BLOCK 256                    // threads per workgroup = 4 waves
#endif
#define SPL_COUNTER_STRIDE 64            // words between the 8 queue counters: a 256-byte line each (counters sharing a
                                         // line serialise the atomics of all XCDs: measured 11 ns per atomic, chip-wide)
#define SPL_WAVES (SPL_BLOCK / 64)
// The range kernel deals the wave-iterations of a chunk (64 * K reads of ONE run each, K = 4 / 2 / 1 / 1 by run; a run's last one
// may be partial) round-robin to its four waves.  Simple reads are never listed; of the others a wave can get a quarter of the
// chunk plus what the alignment of the runs' partial iterations adds: at most 577 of a 2048-read chunk -- s_q has 640 entries per
// wave -- and 1089 of a 4096-read one, which 640 entries hold in every chunk but one of nothing but flagged spliced reads: a list
// that is full hands its entries straight to the literal queue (push_direct), no read is dropped.  (Round 4 gave chunks of 4096
// reads 1152 entries a wave: 24.5 KB of LDS per workgroup in the stranded instantiation, six workgroups per CU instead of eight.)
#ifndef SPL_K_SIMPLE
#define SPL_K_SIMPLE 4                   // reads per lane and wave-iteration: simple reads (8-byte records)
#define SPL_K_MNM 2                      // ... once-spliced reads (16-byte records)
#endif
#ifndef SPL_LIST_K
#define SPL_LIST_K 2                     // entries per lane and round of the range kernel's list pass (3: spills 2 registers, 4: 17)
#endif
#ifndef SPL_WAVE_ITERS
#define SPL_WAVE_ITERS 10
#endif
#define SPL_WAVE_READS (64 * SPL_WAVE_ITERS)
#define SPL_TILE_FUSED_SHIFT 10           // reads per tile of a fused pass: the records of 1024 reads are 24.8 KB of LDS at most
#define SPL_TILE_FUSED (1 << SPL_TILE_FUSED_SHIFT)
#define SPL_BLOCK_FUSED 256              // threads of its workgroups: four reads of a tile a thread (512 threads, two reads each, eight
                                         // waves to count a tile: 1.25 ms a launch against 0.94 -- what a wave does per tile whatever its
                                         // share of it is what the pass spends its instructions on)
#define SPL_WAVE_READS_FUSED 512         // a wave's list entries there (4 workgroups of 39 KB on a CU)
#define SPL_WIN 1020                     // distinct site positions a workgroup privatises in LDS (pair kernel; range kernel unstranded)
#define SPL_WIN_STRANDED 956             // ... range kernel, stranded: 4 windows + the lists, 8 workgroups in 160 KB
#define SPL_WIN_STRANDED_FUSED 508       // ... the fused pass, stranded: 4 windows of 2 KB beside a tile's records, 4 workgroups in 160 KB
#define SPL_SERIAL_MAX 8                 // pair kernel: sites a lane classifies alone before the wave takes over
#ifndef SPL_AGG_ROUNDS
#define SPL_AGG_ROUNDS 2                 // distinct addresses agg_add merges across the wave before it falls back to plain atomics
#endif
#define SPL_LITERAL_WAVES 8192           // one-wave workgroups of the literal kernel (grid-stride over the queue)
#ifndef SPL_SCAN_BLOCK
#define SPL_SCAN_BLOCK 256               // distinct positions per workgroup of the scan kernels (a multiple of 256; 1024 left the chip half empty)
#endif


#define SPL_DEV_ERR_RANGE 1
#define SPL_DEV_ERR_TABLE 2

// junction table entry, word 3: flags above ...
#define SPL_JF_COUNT_MASK 0x00ffffffu // ... the number of rivals
#define SPL_JF_COMPLEX 0x40000000u  // needs the literal walk (a rival is a junction end itself, duplicate partner edges, ...)
#define SPL_JF_MULTIROW 0x80000000u // a rival shares its position with another row: unstranded runs cannot address it by dpos

// per-row flag byte (built at upload)
#define SPL_SF_PLUS 1u    // Site.strand == '+'
#define SPL_SF_MINUS 2u   // Site.strand == '-'

// Position index entry: one per 32 bp bucket of the shard's coordinate space.
struct spl_dbk {
    uint32_t first;  // distinct-position index ("dpos") of the first site at or after the bucket start
    uint32_t occ;    // which of the bucket's 32 positions are sites
    uint32_t rival;  // ... are ends of junctions that have rivals (not necessarily sites)
};

struct spl_count_params {
    // reads: the packed layout of spl_pack.h, one descriptor per chunk
    int64_t n_reads;
    uint32_t n_chunks;
    const spl_chunk_meta *chunk_meta;
    // sites
    int32_t n_sites;
    const int32_t *site_pos;
    const uint8_t *site_strand;  // ASCII, pair kernel
    const uint8_t *site_flags;   // SPL_SF_*, range kernel
    const uint4 *site_meta;      // {part_off, n_part, comp_off, n_comp}
    const int32_t *part_pos;
    const int32_t *part_site;
    const int32_t *comp_pos;
    // position -> first-row index
    const uint32_t *bucket;      // n_buckets + 1 entries
    uint32_t n_buckets;
    int32_t bucket_base;
    int32_t bucket_shift;
    // position -> distinct-position index (range kernel): 64 bp buckets {first dpos, -, occupancy mask lo, hi}
    const spl_dbk *dbucket;    // 32 bp buckets
    uint32_t n_dbuckets;
    int32_t dbase;
    int32_t n_dpos;
    const int32_t *dpos_first_row;
    const uint4 *jhash;          // junction table (see spl_hot_params)
    uint32_t jhash_mask;
    const uint4 *jrivals;
    // options
    int32_t stranded;            // 0 none, 1 fr, 2 rf
    int32_t combine_mode;
    // outputs
    uint32_t *beta1;             // point counters (pair kernel, rival-site corrections); the scan adds the ranges
    uint32_t *beta2s_reads;
    uint32_t *dbl;
    int32_t *diff;               // range kernel: difference arrays over distinct positions, diff_stride apart
    int32_t diff_stride;         // n_dpos + 1 rounded up
    int32_t *err;
};

// Argument block of the range kernel: only what the straight-line path touches (keeps it out of SGPR spills).
struct spl_hot_params {
    uint32_t n_chunks;           // SLOTS of the chunk order = workgroups of the launch (8 equal XCD shares)
    uint32_t chunk_shift;        // log2 of the reads per chunk of this read set (SPL_CHUNK_SHIFT or SPL_CHUNK_BIG_SHIFT)
    const spl_chunk_meta *chunk_meta; // [chunks] (spl_pack.h)
    const uint32_t *chunk_order; // [n_chunks] slot of an XCD share -> chunk, longest chunk first within every share; 0xffffffff = none
    const int32_t *part_pos;     // partner positions (CSR values): the twice-spliced junction-table pass scans a rival's list
    const spl_dbk *dbucket;    // 32 bp buckets
    uint32_t n_dbuckets;
    int32_t dbase;
    int32_t n_dpos;
    int32_t stranded;
    int32_t *diff;
    int32_t diff_stride;
    // junction table: BED junctions with a flagged end -> their rival sites (built at upload)
    const uint4 *jhash;          // two quads per slot: {l, r, first rival record, n_rivals | SPL_JF_*} (l == 0x80000000 = empty), first rival
    uint32_t jhash_mask;
    const uint4 *jrivals;        // {t_pos, t_dpos | strand code << 30, double-count edge 0, edge 1 (0xffffffff = none)}
    uint32_t *dbl;
    int32_t combine_mode;
    uint32_t *queue;             // reads handed to spl_count_literal_kernel (packed indexes): 8 regions (workgroup & 7) of queue_cap
    uint32_t *queue_n;           // entries used per region, counter k at word k * SPL_COUNTER_STRIDE
    uint32_t queue_cap;
    int32_t *err;
    // the FUSED instantiation (reads counted straight from the BAM-native arrays, see spl_count_ranges_kernel): the chunks are
    // cells of the grid over the arrays' indexes; chunk_meta is not looked at
    const spl_layout_chunk *cells; // [chunks] (spl_devpack.h; made by spl_layout_map_kernel)
    spl_devreads src;
    int64_t src_n_rec, src_n_ops;
};

// the literal kernel's view of the queue (entries: chunk << SPL_CHUNK_SHIFT | slot of the read in its chunk, run order)
struct spl_queue_params {
    const uint32_t *queue;       // 8 regions of queue_cap entries
    const uint32_t *queue_n;     // counter k at word k * SPL_COUNTER_STRIDE
    uint32_t queue_cap;
    uint32_t chunk_shift;        // (how to take an entry apart)
    uint32_t *queue_total;       // the number of entries, for the host (diagnostic), written by this launch
    // the block sums of the difference arrays are taken by this launch too (see spl_count_literal_kernel)
    const int32_t *diff;
    int32_t *block_sums;         // [scan_arrays][scan_blocks]
    int32_t diff_stride, n_dpos, scan_blocks, scan_arrays;
    // ... and the other copy of the counter region is cleared for the next counting pass
    uint4 *clear_region;
    size_t clear_n16;
    // entries of a FUSED pass: chunk << chunk_shift | index of the read in the chunk's cell of the arrays
    const spl_layout_chunk *cells; // null: the entries are slots of packed chunks
    spl_devreads src;
};

struct spl_sse_params {
    int64_t n_sites;
    const int32_t *site_pos;
    const uint32_t *part_off;
    const int32_t *part_pos;
    const int32_t *part_site;
    const int64_t *alpha;
    const int64_t *edge_cnt;
    const uint32_t *beta1;
    const uint32_t *beta2s_reads;
    const uint32_t *dbl;
    int32_t cryptic;
    int64_t *beta2_simple;
    int64_t *beta2_cryptic;
    double *beta2_weighted;
    double *sse;
};

struct spl_scan_params {
    int32_t n_dpos;
    const int32_t *dpos_first_row; // [n_dpos + 1]
    int32_t n_arrays;            // 2 unstranded {beta1, ME}; 4 stranded {beta1+, beta1-, ME+, ME-}
    int32_t diff_stride;
    int32_t n_blocks;
    const int32_t *diff;
    int32_t *block_sums;         // [n_arrays][n_blocks]
    const uint8_t *site_flags;
    uint32_t *beta1;
    uint32_t *beta2s_reads;
    int32_t with_sse;            // the table has the inputs of findBeta2Counts: compute beta2 / SSE of every row right here
    spl_sse_params sse;          // (its beta1 / beta2s_reads / cryptic members are not used by the fused path)
    double *sse_with_cryptic;    // SSE as --beta2Cryptic defines it; sse.sse holds the plain one
};


#ifdef __cplusplus
extern "C" {
#endif
// variant: 0 = range kernel (any table: the junction table is built from each row's own lists), 1 = pair kernel (the
// literal cross-check), 2 = range kernel WITH wave-level aggregation of LDS atomics (SPL_OPT_WAVE_AGGREGATION)
int spl_dev_launch_count(const spl_count_params *p, const spl_hot_params *h, int variant, void *stream, int *grid_out, int *lds_out, void *ev_start, void *ev_stop); // (h->chunk_shift picks the instantiation)
int spl_dev_launch_junctions(const spl_chunk_meta *chunk_meta, uint32_t n_chunks,
                             int stranded, uint32_t min_anchor, uint32_t min_intron, uint32_t max_intron, unsigned long long *keys,
                             uint32_t *vals, uint32_t n_slots, unsigned long long *out_keys,
                             uint32_t *out_vals, uint32_t *n_out, int32_t *err, void *stream);
int spl_dev_launch_build_dbuckets(const int32_t *site_pos, const int32_t *dpos_first_row, int32_t n_dpos, const int32_t *flag_pos,
                                  int32_t n_flag, int32_t dbase, uint32_t n_dbuckets, spl_dbk *out, void *stream);
int spl_dev_launch_clear(void *region, size_t bytes, void *stream);
int spl_dev_launch_literal(const spl_count_params *p, const spl_queue_params *q, void *stream);
int spl_dev_launch_scan(const spl_scan_params *p, void *stream);
int spl_dev_launch_sse(const spl_sse_params *p, void *stream);
#ifdef __cplusplus
}
#endif

#endif // SPL_DEVICE_H
