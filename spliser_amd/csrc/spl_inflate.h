// spl_inflate.h -- DEFLATE (RFC 1951) on the device: the launchers of spl_inflate.hip and what they exchange with the host.
//
// A BGZF block is a gzip member of its own holding at most 64 KiB: a BAM file is hundreds of thousands of independent DEFLATE
// streams.  Two kernels inflate them (spl_inflate_wave.h has the method): the Huffman decoding, a WAVE per block -- 64 lanes
// share one pair of look-up tables in LDS and find their places in the block's bits by decoding from guessed starts until the
// guesses agree -- writes what the block's symbols say as a stream of tokens; the block's bytes are made from that stream by a
// LANE per block.  Every loop of either is bounded by the block's own sizes: a corrupt block ends with an error code in its status
// word, never with a hang or a read beyond the image's padding.  (Round 2's decoder -- a lane per block for everything, canonical
// Huffman decoding by code length with tables per lane -- was kept beside them for comparison until round 5: git history.)
//
// The inflate and CRC kernels are what `process` replaces the host's libdeflate + CRC32 threads with (bam_reader.cpp,
// decode_worker; `--hostDecode` keeps those); replaces SpliSER_v0_1_8.py:422 (samtools view) all the same.
#ifndef SPL_INFLATE_H
#define SPL_INFLATE_H
#include <stddef.h>
#include <stdint.h>

struct spl_zblock {
    uint64_t in;      // offset of the block's DEFLATE data in the file image
    uint64_t out;     // offset of the block's payload in the inflated stream
    uint32_t in_len;  // bytes of DEFLATE data
    uint32_t out_len; // ISIZE
    uint32_t crc;     // CRC32 of the payload (from the block's trailer)
    uint32_t pad;
};

// The file image must be readable for this many bytes past the end of any block's DEFLATE data (inside the file that is the
// next block; behind the last one the caller pads), and the inflated stream writable for 16 bytes past its end: the wave kernel
// moves 16 bytes at a time.
#define SPL_Z_IMAGE_PAD 64u
// Between the decoding and the copying kernel every block has a stream of tokens (spl_inflate_wave.h: literal runs of up to
// 128 bytes behind a length byte, matches in three bytes): at most five bytes for four of output (a literal of its own and a
// match of three), so this many bytes per block, 16-byte aligned:
#define SPL_Z_TOKEN_STRIDE 82048u

// status codes written per block (0 = fine)
#define SPL_Z_OK 0u
#define SPL_Z_BAD_BLOCK_TYPE 1u
#define SPL_Z_BAD_STORED 2u
#define SPL_Z_BAD_LENGTHS 3u
#define SPL_Z_BAD_CODE 4u
#define SPL_Z_BAD_DISTANCE 5u
#define SPL_Z_OVERRUN 6u
#define SPL_Z_SHORT 7u
#define SPL_Z_BAD_CRC 8u
#define SPL_Z_TOKENS 9u        // the block's tokens do not fit the room it was given (a caller that gave less than SPL_Z_TOKEN_STRIDE: again, with all of it)

// ---- BAM records out of the inflated stream, one BGZF block per lane ------------------------------------------------
// What a lane reports about the records that START in its block (spl_bam_scan_kernel).  `start` = the first record boundary at
// or after the block's first byte -- GUESSED from the bytes (a header that satisfies the BAM specification and whose
// successors chain), except in the block the BAM header ends in; `reached` = the first boundary at or after the block's end,
// by walking the records from `start`.  The host accepts the lot only if reached[b] == start[b + 1] for every b: by induction
// from the end of the BAM header every boundary is then a true one (the same argument as the host decoder's, bam_reader.cpp).
// (a record is 36 bytes at least: no more than this many begin in a block of 64 KiB)
#define SPL_BS_REC_CAP 1824u
struct spl_bscan {
    uint64_t start, reached;
    uint32_t n_all;      // records starting in the block
    uint32_t n_placed;   // ... with a reference and a position: the ones that are extracted
    uint32_t n_ops;      // CIGAR ops of those
    uint32_t flags;      // SPL_BS_*
    int32_t tid_first, tid_last; // of the placed records (tid_first = -1: none)
    uint32_t n_foreign;  // records starting in the block that belong to references outside [tid_lo, tid_hi): another device's
    uint32_t n_foreign_hi; // ... of those, the ones of references at or behind tid_hi (the others lie in front of tid_lo)
};
#define SPL_BS_CORRUPT 1u     // a record that contradicts itself (block_size < 32, fields beyond block_size, stream ends inside it)
#define SPL_BS_NEEDS_HOST 2u  // a CIGAR parked in a CG tag (more than 65535 ops): the host decoder's business
#define SPL_BS_UNSORTED 4u    // reference ids go down inside the block (over all its records, placed or not, this share's or not)
#define SPL_BS_NO_START 8u    // no plausible record start found within reach of the block
#define SPL_BS_INCOMPLETE 16u  // a record of the block runs past the end of what is inflated at the moment (a window of the stream)

#ifdef __cplusplus
extern "C" {
#endif
// stream_len = bytes of the inflated stream; header_end = where the first record starts; blocks[b].out / out_len say where block b lies.
// The stream may be there in part only: `stream` is indexed with offsets into the WHOLE stream all the same (the caller passes
// the window's address minus the window's offset), stream_len = where the window ends, more = the stream goes on behind it.
// Only records of references tid_lo <= tid < tid_hi count (records without a reference count as reference n_ref): a device that
// decodes a stretch of the file takes the references that begin there.
// recs (or null): n_blocks x SPL_BS_REC_CAP 16-bit places; block b's entry j = where its j-th placed record begins, counted from
// the block's first byte -- what lets the extraction take a block's records 64 at a time instead of walking them.
int spl_dev_launch_bam_scan(const uint8_t *stream, uint64_t stream_len, uint64_t header_end, int32_t n_ref, int32_t tid_lo, int32_t tid_hi, const spl_zblock *blocks,
                            uint32_t n_blocks, spl_bscan *scan, int more, uint16_t *recs, void *stream_handle);
// rec_off[b] / op_off[b]: index of the block's first placed record / first op in the output arrays.  cig_off gets n + 1 entries
// (the caller sets entry 0); ref_max_end[tid] = largest last base of a read of the reference (atomicMax; zero it first).
// recs: what the scan of THESE blocks (same first block, same order) left, or null: then a lane walks its block's records.
int spl_dev_launch_bam_extract(const uint8_t *stream, uint64_t stream_len, int32_t n_ref, int32_t tid_lo, int32_t tid_hi, const spl_zblock *blocks, uint32_t n_blocks, const spl_bscan *scan,
                               const uint64_t *rec_off, const uint64_t *op_off, int32_t *pos, uint16_t *flag, uint32_t *cig_off, uint32_t *cigar,
                               int32_t *tid, unsigned long long *ref_max_end, const uint16_t *recs, void *stream_handle);
// where the reference id changes along the placed records: (index of the first record of a run, its tid) pairs, unordered
int spl_dev_launch_bam_bounds(const int32_t *tid, const uint32_t *cig_off, uint64_t n, uint64_t *bounds, uint32_t *n_bounds, uint32_t cap, void *stream_handle);
// image: the whole file in device memory, readable SPL_Z_IMAGE_PAD bytes past its end; out: writable 16 bytes past the last block.
// work: spl_dev_inflate_work_bytes(n_blocks) bytes of device memory (the blocks' token streams between the two kernels); null is
// an error (hipErrorInvalidValue) for every launcher that takes it.  stream: the stream to launch on.
size_t spl_dev_inflate_work_bytes(uint32_t n_blocks);
// ... with `stride` bytes of token room a block instead of SPL_Z_TOKEN_STRIDE (a multiple of 16; a block that needs more gets
// SPL_Z_TOKENS): what real files' blocks need is a third of the worst case, and a first call waits for every gigabyte it is given
size_t spl_dev_inflate_work_bytes2(uint32_t n_blocks, uint32_t stride);
int spl_dev_launch_inflate_decode2(const uint8_t *image, const spl_zblock *blocks, uint32_t n_blocks, uint32_t *status, void *work, uint32_t stride, void *stream);
// ... told which of the two decoding kernels to take (SPL_Z_LAUNCH_DENSE: the one with five waves a SIMD and 3 KB of token room a
// tile, for files whose blocks deflate to 12 KB or less on average; decode2 decides by the stride alone).  SPL_Z_DENSE=0 / 1 in the
// environment overrides either.
#define SPL_Z_LAUNCH_DENSE 1u
int spl_dev_launch_inflate_decode3(const uint8_t *image, const spl_zblock *blocks, uint32_t n_blocks, uint32_t *status, void *work, uint32_t stride, uint32_t flags, void *stream);
int spl_dev_launch_inflate_copy2(const spl_zblock *blocks, uint32_t n_blocks, uint8_t *out, uint32_t *status, void *work, uint32_t stride, void *stream);
int spl_dev_launch_inflate(const uint8_t *image, const spl_zblock *blocks, uint32_t n_blocks, uint8_t *out, uint32_t *status, void *work, void *stream);
// ... in two halves, for callers that put them on different streams: the Huffman decoding (token streams into `work`), then the
// copies (the inflated bytes into `out`; a block whose tokens do not give out_len bytes gets SPL_Z_SHORT)
int spl_dev_launch_inflate_decode(const uint8_t *image, const spl_zblock *blocks, uint32_t n_blocks, uint32_t *status, void *work, void *stream);
int spl_dev_launch_inflate_copy(const spl_zblock *blocks, uint32_t n_blocks, uint8_t *out, uint32_t *status, void *work, void *stream);
// CRC32 of every block's bytes against the value in its trailer (blocks that failed before keep their status)
int spl_dev_launch_crc32(const uint8_t *out, const spl_zblock *blocks, uint32_t n_blocks, uint32_t *status, void *stream);
#ifdef __cplusplus
}
#endif
#endif
