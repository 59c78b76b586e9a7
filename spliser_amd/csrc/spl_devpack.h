// spl_devpack.h -- the device packer's launchers (spl_devpack.hip) and the handle by which a decoded BAM keeps its reads on the
// device for it.
#ifndef SPL_DEVPACK_H
#define SPL_DEVPACK_H
#include <stdint.h>

struct spl_devreads {          // BAM-native reads in device memory, file order (what spl_bam_decode_device extracts)
    const int32_t *pos;        // 1-based POS
    const uint16_t *flag;
    const uint32_t *cig_off;   // n + 1 offsets into cigar
    const uint32_t *cigar;
};

#ifdef __cplusplus
extern "C" {
#endif
// chunk = reads per chunk (2048 or 4096); descs: device array of splpack::ChunkDesc, one per chunk of reads [first, first + n_reads)
int spl_dev_launch_pack_count(const spl_devreads *src, int64_t first, int64_t n_reads, uint32_t chunk, void *descs, void *stream);
// sizes -> places, in place; totals: two 64-bit words (record bytes, wide ops)
int spl_dev_launch_pack_offsets(void *descs, uint32_t n_chunks, void *totals, void *stream);
int spl_dev_launch_pack_emit(const spl_devreads *src, int64_t first, int64_t n_reads, uint32_t chunk, const void *descs, void *rec_base, void *wide_base, void *stream);
#ifdef __cplusplus
}
#endif
#endif
