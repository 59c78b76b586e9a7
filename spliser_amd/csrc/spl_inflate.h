// spl_inflate.h -- DEFLATE (RFC 1951) on the device, one BGZF block per lane.
//
// A BGZF block is a gzip member of its own holding at most 64 KiB: a BAM file is hundreds of thousands of independent
// DEFLATE streams, which is all the parallelism a GPU needs -- no cooperation inside a stream.  Each lane decodes its block
// from the copy of the file in device memory straight to its place in the inflated stream.  Canonical Huffman decoding by
// code length (count-per-length tables: the fifteen counts of a table live in registers, the symbols in the lane's scratch),
// bits from a 64-bit buffer refilled by aligned 32-bit loads.  Every loop is bounded by the block's own sizes: a corrupt block
// ends with an error code in its status word, never with a hang.
//
// The inflate and CRC kernels are what `process --gpuDecode` replaces the host's libdeflate + CRC32 threads with
// (bam_reader.cpp, decode_worker); replaces SpliSER_v0_1_8.py:422 (samtools view) all the same.
#ifndef SPL_INFLATE_H
#define SPL_INFLATE_H
#include <stdint.h>

struct spl_zblock {
    uint64_t in;      // offset of the block's DEFLATE data in the file image
    uint64_t out;     // offset of the block's payload in the inflated stream
    uint32_t in_len;  // bytes of DEFLATE data
    uint32_t out_len; // ISIZE
    uint32_t crc;     // CRC32 of the payload (from the block's trailer)
    uint32_t pad;
};

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

#ifdef __cplusplus
extern "C" {
#endif
// image: the whole file in device memory, padded with 8 readable bytes.  stream: the stream to launch on.
int spl_dev_launch_inflate(const uint8_t *image, const spl_zblock *blocks, uint32_t n_blocks, uint8_t *out, uint32_t *status, void *stream);
int spl_dev_launch_crc32(const uint8_t *out, const spl_zblock *blocks, uint32_t n_blocks, uint32_t *status, void *stream);
#ifdef __cplusplus
}
#endif
#endif
