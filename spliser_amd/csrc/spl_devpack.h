// spl_devpack.h -- the layout kernel's launchers (spl_devpack.hip) and the handle by which BAM-native reads stay on the device
// for it (what spl_bam_decode_device extracts, what spl_soa_upload brings up).
#ifndef SPL_DEVPACK_H
#define SPL_DEVPACK_H
#include <stdint.h>

#include "spl_pack.h"

struct spl_devreads {          // BAM-native reads in device memory, file order
    const int32_t *pos;        // 1-based POS
    const uint16_t *flag;
    const uint32_t *cig_off;   // n + 1 offsets into cigar
    const uint32_t *cigar;
};

// One segment (a reference of a BAM file, one caller array) of the reads a layout launch works on.  Its chunks are the cells of
// the grid of SPL_CHUNK (or SPL_CHUNK_BIG) reads over the ARRAYS' indexes that it touches -- the first and the last one partly --
// so that a thread's four reads are one aligned 16-byte load of POS and of the CIGAR offsets wherever the segment begins.
struct spl_layout_seg {
    int64_t first;             // index of the segment's first read in the arrays
    int64_t n_reads;
    uint32_t chunk0;           // index of its first chunk in the read set's flat chunk list (spl_chunk_meta, cost)
    uint32_t dev0;             // ... among the chunks of this launch (record slots, chunk_seg)
    uint32_t n_chunks;
    int32_t shift;             // added to every POS of the segment by the counting kernels (spliser_amd/shard.py)
};

// Every chunk of a launch has a record slot of the worst-case size (every read 24 bytes): no workgroup has to know what the
// others need, nothing is counted before it is written, and 288 GB are there to be used.  The slot's tail is slack for the
// range kernel's last loads of a run (spl_kernels.hip: fetch).
#define SPL_LAYOUT_SLOT(chunk_reads) ((size_t)(chunk_reads) * SPL_REC_OTHER + 256u)

// What a workgroup of the layout kernel needs to know about its chunk, in ONE scalar load (made from the segments and the
// arrays' CIGAR offsets by spl_layout_map_kernel: a workgroup that first had to look its segment up, then the offsets of its
// first and last read, had three memory trips behind it before its first op was asked for).
struct spl_layout_chunk {
    int64_t lo;                // index of the chunk's first read in the arrays (its cell of the grid: lo & ~(chunk - 1))
    uint32_t n;                // reads
    uint32_t o_lo, o_hi;       // its ops: cigar[o_lo .. o_hi)
    uint32_t seg_op0;          // the first op of its segment (WIDE reads' indexes count from there)
    int32_t shift;
    uint32_t flat;             // its index in the read set's flat chunk list
};

struct spl_layout_params {
    spl_devreads src;
    int64_t n_rec;             // reads in the arrays (cig_off has n_rec + 1 entries): no load goes beyond them
    int64_t n_ops;             // ops in the arrays
    const spl_layout_chunk *chunks; // [chunks of this launch]
    uint8_t *rec_base;         // record slots, SPL_LAYOUT_SLOT apart
    spl_chunk_meta *meta;      // [flat chunk list]
};

static inline uint32_t spl_layout_seg_chunks(int64_t first, int64_t n_reads, uint32_t chunk)
{
    return n_reads > 0 ? (uint32_t)((first + n_reads - 1) / chunk - first / chunk + 1) : 0u;
}
// Slots per XCD share of the range kernel's grid for n chunks (blocks of 8 consecutive chunks dealt round-robin to the eight).
static inline uint32_t spl_order_per(uint32_t n_chunks) { return ((n_chunks + 7u) / 8u + 7u) / 8u * 8u; }

#ifdef __cplusplus
extern "C" {
#endif
// chunks[k] = the launch's k-th chunk (which segment's, where its reads and ops lie), cost[its flat index] = what it will cost the
// range kernel, roughly, from its numbers of reads and ops (device arrays; chunk = reads per chunk)
int spl_dev_launch_layout_map(const spl_devreads *src, const spl_layout_seg *segs, uint32_t n_segs, uint32_t n_chunks, uint32_t chunk, spl_layout_chunk *chunks, uint32_t *cost,
                              void *stream);
// the layout itself: one workgroup per chunk of the launch (chunk = reads per chunk, 2048 or 4096)
int spl_dev_launch_layout(const spl_layout_params *p, uint32_t n_dev_chunks, uint32_t chunk, void *stream, void *ev_start, void *ev_stop);
// cost[n_chunks] -> order[8 * spl_order_per(n_chunks)]: the range kernel's slots, XCD share by XCD share, longest chunk first;
// 0xffffffff = an empty slot
int spl_dev_launch_chunk_order(const uint32_t *cost, uint32_t n_chunks, uint32_t chunk, uint32_t *order, void *stream);
#ifdef __cplusplus
}
#endif
#endif
