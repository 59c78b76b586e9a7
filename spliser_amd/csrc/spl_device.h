// spl_device.h -- kernel parameter blocks and launch geometry shared by spl_kernels.hip and spl_capi.cpp.
#ifndef SPL_DEVICE_H
#define SPL_DEVICE_H

#include <stdint.h>
#include <hip/hip_vector_types.h>

// Launch geometry of spl_count_kernel (see DESIGN.md "Kernels").
#define SPL_BLOCK 256                    // threads per workgroup = 4 waves
#define SPL_RPT 8                        // reads per thread
#define SPL_CHUNK (SPL_BLOCK * SPL_RPT)  // consecutive reads per workgroup
#define SPL_WIN 2048                     // sites whose counters a workgroup privatises in LDS (2 x 8 KiB)
#define SPL_SERIAL_MAX 8                 // sites a lane classifies alone before the wave takes the read over

// Coordinates (read end, site position) must stay <= SPL_COORD_MAX so that t+1 and cur never wrap int32.
#define SPL_COORD_MAX 2147483645

#define SPL_DEV_ERR_RANGE 1

struct spl_count_params {
    // reads
    int64_t n_reads;
    uint32_t n_chunks;
    const int32_t *r_pos;
    const uint16_t *r_flag;
    const uint32_t *cig_off;
    const uint32_t *cigar;
    // sites
    int32_t n_sites;
    const int32_t *site_pos;
    const uint8_t *site_strand;
    const uint4 *site_meta;   // {part_off, n_part, comp_off, n_comp}
    const int32_t *part_pos;
    const int32_t *comp_pos;
    // position -> first-row index
    const uint32_t *bucket;   // n_buckets + 1 entries
    uint32_t n_buckets;
    int32_t bucket_base;
    int32_t bucket_shift;
    // options
    int32_t stranded;         // 0 none, 1 fr, 2 rf
    int32_t combine_mode;
    // outputs
    uint32_t *beta1;
    uint32_t *beta2s_reads;
    uint32_t *dbl;
    int32_t *err;
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

#ifdef __cplusplus
extern "C" {
#endif
int spl_dev_launch_count(const spl_count_params *p, void *stream, int *grid_out);
int spl_dev_launch_sse(const spl_sse_params *p, void *stream);
#ifdef __cplusplus
}
#endif

#endif // SPL_DEVICE_H
