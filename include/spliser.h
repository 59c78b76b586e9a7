/*
 * spliser.h -- C ABI of libspliser_hip.so, the MI355X (gfx950) implementation of the hot path of
 * SpliSER v0.1.8 `process`: per-splice-site beta1 / beta2Simple / double-count read classification
 * and the beta2 + SSE vector pass.
 *
 * The reference (pure Python) has no FFI; the seam this library replaces is the batch
 *
 *     processSites(inBAM, qChrom, isStranded, strandedType, isbeta2Cryptic)   SpliSER_v0_1_8.py:681-692
 *       = for every site: checkBam(...)                                       SpliSER_v0_1_8.py:408-559
 *         for every site: findBeta2Counts(...); calculateSSE(...)             SpliSER_v0_1_8.py:581-639
 *
 * and, upstream of it, the per-site `samtools view` child process (SpliSER_v0_1_8.py:422), which is
 * replaced by one whole-file BAM decode into structure-of-arrays buffers (spl_bam_*).
 *
 * Conventions
 *   - every function returns 0 on success and a negative spl_status on failure; the message of the
 *     last failure on the calling thread is available from spl_last_error();
 *   - the caller owns every buffer it passes in; the library never keeps a caller pointer past the
 *     return of the call that received it;
 *   - a spl_ctx is bound to one GPU and one HIP stream; calls on different contexts may run
 *     concurrently from different threads, calls on the same context must be serialised;
 *   - there is NO CPU fallback: spl_create() fails when no gfx950 device is visible.
 *
 * Coordinates: all positions are the reference's integers -- 1-based SAM POS for reads; for sites the
 * values computed at SpliSER_v0_1_8.py:275-276 (left site = last exonic base, right site = last
 * intronic base, both 1-based), so a CIGAR N op of length d ending at cur gives lSite = cur-d-1,
 * rSite = cur-1 (SpliSER_v0_1_8.py:482-483) and equality tests are direct.  One call covers one
 * *shard*: a single int32 coordinate space (one chromosome, or several chromosomes the host has laid
 * side by side with offsets -- see spliser_amd/shard.py); every coordinate must stay at or below 2^31-67.
 */
#ifndef SPLISER_H
#define SPLISER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPL_ABI_VERSION 1

typedef enum spl_status {
    SPL_OK = 0,
    SPL_ERR_ARG = -1,       /* bad argument (null pointer, negative size, unsorted table, ...)   */
    SPL_ERR_NO_DEVICE = -2, /* no usable gfx950 device / HIP runtime failure at start-up          */
    SPL_ERR_HIP = -3,       /* a HIP call failed; message carries hipGetErrorString               */
    SPL_ERR_IO = -4,        /* file missing / unreadable / truncated                              */
    SPL_ERR_FORMAT = -5,    /* not BGZF/BAM, corrupt block, unsupported CIGAR op                   */
    SPL_ERR_RANGE = -6,     /* coordinate overflow: a read or site exceeds the int32 shard space   */
    SPL_ERR_NOMEM = -7
} spl_status;

typedef struct spl_ctx spl_ctx;       /* one GPU + one stream                                       */
typedef struct spl_dsites spl_dsites; /* a site table resident in HBM (+ its output counters)       */
typedef struct spl_dreads spl_dreads; /* a read set resident in HBM                                  */
typedef struct spl_bam spl_bam;       /* an opened + decoded BAM file                                */

/* Site table of one shard: the state the reference keeps in Site objects
 * (Gene_Site_Iter_Graph_v0_1_8.py:98-120) after findAlphaCounts + findCompetitorPos
 * (SpliSER_v0_1_8.py:227-372), flattened to SoA + CSR.  Row order = Site.__lt__ order
 * (Gene_Site_Iter_Graph_v0_1_8.py:123-136): ascending pos, '+' before '-' at equal pos. */
typedef struct spl_sites {
    int64_t n_sites;
    const int32_t *pos;        /* [n_sites]   Site.pos, non-decreasing                               */
    const uint8_t *strand;     /* [n_sites]   Site.strand as ASCII ('+', '-', anything else = none)  */
    const uint32_t *part_off;  /* [n_sites+1] CSR offsets into part_*                                */
    const int32_t *part_pos;   /* [n_part]    keys of Site.PartnerCounts, insertion order            */
    const int32_t *part_site;  /* [n_part]    row of that partner in this table, -1 when absent (the
                                              one-way lists of combine's query tables); NULL: spl_sse
                                              is unavailable.  The range kernel takes either.        */
    const uint32_t *comp_off;  /* [n_sites+1] CSR offsets into comp_pos                              */
    const int32_t *comp_pos;   /* [n_comp]    Site.CompetitorPos (sorted unique)                     */
    const int64_t *alpha;      /* [n_sites]   Site.alphaCounts[sample]; may be NULL for spl_count    */
    const int64_t *edge_cnt;   /* [n_part]    Site.PartnerCounts[part_pos][sample]; may be NULL for
                                              spl_count                                             */
} spl_sites;

/* Reads of one shard, in file order (coordinate-sorted input gives the best locality but is not
 * required for correctness).  This is exactly what checkBam consumes from each SAM line:
 * column 2 (flag), column 4 (POS), column 6 (CIGAR)  -- SpliSER_v0_1_8.py:434-437. */
typedef struct spl_reads {
    int64_t n_reads;
    const int32_t *pos;       /* [n_reads]   1-based leftmost position (SAM POS)                     */
    const uint16_t *flag;     /* [n_reads]   SAM FLAG                                                */
    const uint32_t *cig_off;  /* [n_reads+1] offsets into cigar; cig_off[0] == 0                     */
    const uint32_t *cigar;    /* [cig_off[n_reads]] BAM-native ops: len<<4 | op, op = MIDNSHP=X 0..8 */
} spl_reads;

typedef struct spl_opts {
    int32_t stranded;     /* 0 = unstranded, 1 = "fr", 2 = "rf"   (check_strand, SpliSER_v0_1_8.py:374-406) */
    int32_t combine_mode; /* 0 = `process`; 1 = `combine`/`combineShallow`: a flanking read also counts
                             toward beta2Simple (SpliSER_v0_1_8.py:529-536)                            */
    int32_t flags;        /* SPL_OPT_* bits                                                            */
} spl_opts;

/* Force the literal per-(read, site) kernel: an on-device cross-check (tests).  Nothing in the product selects it -- the
 * range kernel takes every table, combine's query tables with their one-way partner lists included (its junction table is
 * built from each row's own lists); both give identical counters. */
#define SPL_OPT_PAIR_KERNEL 1
/* Variant of the range kernel that merges the LDS atomics of neighbouring lanes before issuing them (same results;
 * measured never faster than the plain atomics the default uses -- kept for parity tests and experiments). */
#define SPL_OPT_WAVE_AGGREGATION 2

/* ---- library / context ------------------------------------------------------------------------ */
int spl_abi_version(void);
const char *spl_last_error(void);
int spl_device_count(int *n_out);
/* The library keeps device memory it was given back (read sets, site tables, the buffers of a device decode) for its next call
 * in the same process -- fresh device memory costs 30 ms per GB on this stack -- up to half of what was free on the device when it
 * first asked (SPL_DEV_CACHE_GB overrides).  spl_trim returns all of it to the driver: device_id, or -1 for every device. */
int spl_trim(int device_id);
int spl_create(int device_id, spl_ctx **out);
/* Same, but work is enqueued on a caller-provided hipStream_t (e.g. torch's current stream). */
int spl_create_on_stream(int device_id, void *hip_stream, spl_ctx **out);
void spl_destroy(spl_ctx *ctx);
int spl_sync(spl_ctx *ctx);
/* Device-side barrier: what is launched on the context after this call starts after everything launched before it has
 * finished (the tails of counting passes, which run on a second stream, included).  The host does not wait. */
int spl_pass_barrier(spl_ctx *ctx);
/* HIP-event stopwatch on the context's stream (what bench.py times the kernels with). */
int spl_timer_begin(spl_ctx *ctx);
int spl_timer_end(spl_ctx *ctx, float *elapsed_ms_out);

/* Per-launch stopwatch around the classification kernel ALONE (HIP events recorded on the context's
 * stream immediately before and after each spl_count_kernel launch).  begin() arms up to max_records
 * launches; collect() synchronises and returns their durations in launch order. */
int spl_kernel_timing_begin(spl_ctx *ctx, int max_records);
int spl_kernel_timing_collect(spl_ctx *ctx, float *ms_out, int capacity, int *n_out);
/* The library's own stopwatch over ALL its kernels (process-wide): spl_prof_enable(1) clears what was recorded and brackets every
 * kernel launch from then on -- the device decode's, the device packer's, the counting passes' -- with two events on its stream;
 * spl_prof_report waits for the devices and writes JSON text, [{"kernel", "calls", "ms", "bytes"}, ...] most time first, `bytes`
 * being what the kernel was given to work on (the stretch of the inflated stream, the reads: algorithmic, not fetched).  Returns
 * the length the text needs.  What bench.py's end-to-end legs quote as their `kernels` table. */
int spl_prof_enable(int on);
int spl_prof_report(char *buf, int cap);

/* ---- one-shot entry points on host buffers (what the `process` driver calls per shard) ---------
 * spl_count  == the checkBam loop of processSites (SpliSER_v0_1_8.py:686-688) for all sites at once.
 *   beta1[s]         += reads classified "beta1" for site s           (SpliSER_v0_1_8.py:558-559)
 *   beta2s_reads[s]  += reads that add to beta2SimpleCounts in checkBam (:532, :541, :552)
 *   dbl[e]           += PartnerBeta2DoubleCounts increments for partner edge e (:527, :551)
 * Outputs are overwritten (not accumulated). */
int spl_count(spl_ctx *ctx, const spl_sites *sites, const spl_reads *reads, const spl_opts *opts,
              uint32_t *beta1, uint32_t *beta2s_reads, uint32_t *dbl);

/* spl_sse == findBeta2Counts + calculateSSE for every site (SpliSER_v0_1_8.py:690-692, 581-639).
 * Needs sites->alpha, sites->edge_cnt and sites->part_site.  All arithmetic is IEEE binary64 with
 * no contraction, in the reference's operation order, so results are bit-identical to CPython's. */
int spl_sse(spl_ctx *ctx, const spl_sites *sites, const uint32_t *beta1, const uint32_t *beta2s_reads,
            const uint32_t *dbl, int beta2_cryptic, int64_t *beta2_simple, int64_t *beta2_cryptic_count,
            double *beta2_weighted, double *sse);

/* ---- device-resident pipeline (bench.py, multi-shard overlap) ---------------------------------- */
int spl_sites_upload(spl_ctx *ctx, const spl_sites *sites, spl_dsites **out);
void spl_sites_free(spl_ctx *ctx, spl_dsites *ds);
/* A read set on the device.  The counting kernels do not read the BAM-native arrays: a read set is cut into chunks of 2048
 * (or 4096) reads, each chunk partitioned by the kind of read (unspliced / once-spliced / twice-spliced / anything else) with
 * records as wide as the kind needs (8 / 16 / 24 bytes; spliser_amd/csrc/spl_pack.h).  A caller's HOST arrays are packed that
 * way by host threads on their way up, piece by piece through a ring of page-locked staging buffers, the DMA of one piece
 * running while the next is packed (the records are two thirds of the arrays' bytes, and a hand-over is bound by PCIe); reads
 * that are in device memory already (spl_bam_decode_device, spl_soa_upload) are laid out by the layout kernel
 * (spliser_amd/csrc/spl_devpack.hip).  The caller's arrays are not needed after the call returns. */
int spl_reads_upload(spl_ctx *ctx, const spl_reads *reads, spl_dreads **out);
/* Same, from n_seg host segments laid end to end: segment k is moved by pos_shift[k] into the shard's coordinate space
 * (spliser_amd/shard.py packs several chromosomes into one launch that way).  Reads keep segment order. */
int spl_reads_upload_segments(spl_ctx *ctx, int n_seg, const spl_reads *segs, const int32_t *pos_shift, spl_dreads **out);
/* The same in steps, for segments that become available one after the other (the references of a BAM file while it is still
 * being decoded): begin, add ... add, finish; counting passes need a finished read set.  spl_reads_add_bam takes the reads of
 * reference `tid` straight from the decoder's buffers -- host memory, or the device's own after spl_bam_decode_device -- and
 * waits until that reference is complete (spl_bam_wait_ref). */
int spl_reads_begin(spl_ctx *ctx, spl_dreads **out);
/* ... with the number of reads that are going to be added, if the caller knows it (0 = no idea): sets of 64 M reads and more are
 * cut into chunks of 4096 instead of 2048 reads, which suits launches of that size (3.5 % on 100 M reads) and no others. */
int spl_reads_begin_sized(spl_ctx *ctx, int64_t expected_reads, spl_dreads **out);
int spl_reads_add(spl_ctx *ctx, spl_dreads *dr, const spl_reads *reads, int32_t pos_shift);
/* ... known_max_end: the last base (1-based) any of the reads covers, if the caller knows it (< 0: computed when needed) */
int spl_reads_add2(spl_ctx *ctx, spl_dreads *dr, const spl_reads *reads, int32_t pos_shift, int64_t known_max_end);
int spl_reads_add_bam(spl_ctx *ctx, spl_dreads *dr, spl_bam *bam, int tid, int32_t pos_shift);
int spl_reads_finish(spl_ctx *ctx, spl_dreads *dr);
void spl_reads_free(spl_ctx *ctx, spl_dreads *dr);
/* BAM-native reads RESIDENT IN HBM, as the arrays they are (what checkBam reads from a SAM line, SpliSER_v0_1_8.py:434-437, and
 * nothing else): n_seg host segments laid end to end in device arrays pos / flag / cig_off / cigar -- what a decode on the
 * device leaves (spl_bam_decode_device) and what SURVEY.md 8(d)'s "kernel-only from device-resident SoA" starts from.  A read
 * set is made from them ON THE DEVICE by spl_reads_add_soa + spl_reads_finish.  A set whose segments all lie in ONE such handle
 * stays arrays ("fused"): spl_count_launch reads them itself and makes its records in LDS (spl_kernels.hip, the FUSED range
 * kernel) -- no records in memory, no layout launch; what needs records (the pair kernel, the merging variant,
 * spl_junctions), a set of several handles or with host-packed segments, or SPL_FUSED=0, gets them from the layout kernel
 * (spl_devpack.hip: one launch, every read fetched and classified once).  spl_reads_relayout does again what spl_reads_finish
 * launched -- the chunks' descriptors and order, and the layout kernel where the set has records -- so that a bench.py step is
 * arrays -> counters, every step.  The handle may be freed while read sets made from it live (they share the arrays). */
typedef struct spl_dsoa spl_dsoa;
int spl_soa_upload(spl_ctx *ctx, int n_seg, const spl_reads *segs, spl_dsoa **out);
/* ... with max_end[k] = the last base (1-based) any read of segment k covers, where the caller knows it (null, or < 0: computed) */
int spl_soa_upload2(spl_ctx *ctx, int n_seg, const spl_reads *segs, const int64_t *max_end, spl_dsoa **out);
void spl_soa_free(spl_ctx *ctx, spl_dsoa *soa);
int spl_reads_add_soa(spl_ctx *ctx, spl_dreads *dr, spl_dsoa *soa, int seg, int32_t pos_shift);
int spl_reads_relayout(spl_ctx *ctx, spl_dreads *dr);
/* What the layout moves for a finished read set: the BAM-native arrays read (10 bytes a read + 4 an op) and the records written
 * (0 while the set is fused). */
int spl_reads_layout_bytes(spl_ctx *ctx, const spl_dreads *dr, int64_t *soa_bytes_out, int64_t *record_bytes_out);
/* ... and the durations of the layout kernel's launches since spl_kernel_timing_begin (call before spl_kernel_timing_collect) */
int spl_layout_timing_collect(spl_ctx *ctx, float *ms_out, int capacity, int *n_out);
/* The host packer alone (no GPU involved; diagnostic and test hook): sizes of what an upload of `reads` would send, and --
 * into buffers of those sizes, when given -- the bytes: chunk descriptors (32 bytes each: record offset u64, wide-op offset
 * u64, first POS i32, cost u32, reads per run u16[4]), the record blob, the wide ops. */
int spl_pack_host(const spl_reads *reads, int n_threads, int64_t *n_chunks_out, int64_t *rec_bytes_out, int64_t *n_wide_out,
                  void *chunk_desc, void *rec, uint32_t *wide);
/* Enqueue one counting pass of dr over ds (asynchronous): the range kernel on the context's stream, then the literal kernel
 * and the scan (which also computes beta2 / SSE when the table has the inputs) on a second stream the context owns, so that
 * the range kernel of the NEXT launch -- next shard, sample or step -- starts as soon as this one's is done.  The counters
 * start from zero (a clean copy of the counter region; the table keeps three).  spl_sync and the download calls wait for
 * everything; a download returns the results of the LAST pass launched on that table.  Environment, read when the context
 * is created: SPL_TAIL_STREAM=0 -- one stream; SPL_TAIL_HOST_WAIT=1 -- the call blocks while more than two passes are in
 * flight instead of putting a wait into the queue (a few percent faster when the host keeps up, idle GPU when it does not). */
int spl_count_launch(spl_ctx *ctx, spl_dsites *ds, const spl_dreads *dr, const spl_opts *opts);
/* Enqueue the beta2/SSE kernel on the counters currently held by ds (asynchronous). */
int spl_sse_launch(spl_ctx *ctx, spl_dsites *ds, int beta2_cryptic);
/* Synchronise and copy results back; any output pointer may be NULL. */
int spl_counters_download(spl_ctx *ctx, const spl_dsites *ds, uint32_t *beta1, uint32_t *beta2s_reads,
                          uint32_t *dbl);
int spl_sse_download(spl_ctx *ctx, const spl_dsites *ds, int64_t *beta2_simple,
                     int64_t *beta2_cryptic_count, double *beta2_weighted, double *sse);
/* Bytes the classification kernel must move at minimum for (ds, dr): each input once, each output
 * once -- the numerator of bench.py's roofline.achieved. */
int spl_count_algorithmic_bytes(const spl_dsites *ds, const spl_dreads *dr, int64_t *bytes_out);
/* Reads the last range-kernel launch on dr handed to the literal kernel (unmapped-but-placed records and reads
 * whose junction ends have rival sites); synchronises.  Diagnostic. */
int spl_literal_queue_size(spl_ctx *ctx, const spl_dreads *dr, int64_t *n_out);
/* Launch geometry of the last spl_count_launch on this context (for DESIGN.md / profiles). */
int spl_last_launch_info(const spl_ctx *ctx, int32_t *grid_out, int32_t *block_out, int32_t *lds_bytes_out);

/* ---- BAM ingest: replaces `samtools view` (SpliSER_v0_1_8.py:422) ------------------------------
 * The whole BGZF file is inflated and its alignment records are split per reference sequence on n_threads host threads
 * (0 = all cores up to 32).  Like `samtools view` without -F/-q every record that has a reference id is kept (secondary,
 * supplementary, duplicate, QC-fail, unmapped-but-placed ...).
 * spl_bam_open_stream returns once the block directory and the header are read; the decode goes on on threads of its own.
 * spl_bam_wait_ref waits until reference `tid` is complete -- in a file sorted by reference that is when a record of a later
 * reference has been seen, long before the end of the file -- so that its reads can go to the GPU (spl_reads_add_bam) while
 * the rest is still being inflated.  spl_bam_wait_all waits for the end of the file and says whether the file WAS sorted by
 * reference (*sorted_out = 0: records of an earlier reference came after a later one; a consumer that took references early
 * must then take them again).  spl_bam_open = open_stream + wait_all. */
int spl_bam_open(const char *path, int n_threads, spl_bam **out);
int spl_bam_open_stream(const char *path, int n_threads, spl_bam **out);
/* Decode on the DEVICE instead: spl_bam_open_deferred reads the header and starts nothing; spl_bam_decode_device then sends the
 * file to the GPU as it is, inflates its BGZF blocks there (Huffman decoding a wave per block, the copies it leaves a lane per
 * block, CRC32 checked; a window of the stream at a time, the next window's decoding beside this window's checks and the upload
 * beside both), finds and extracts the alignment records there, and keeps what checkBam reads (POS, FLAG, CIGAR: a fifteenth of
 * the inflated bytes) in device memory: spl_reads_add_bam on a context of the same device lays a reference's reads out for the
 * counting kernels with kernels, nothing crosses PCIe again; spl_bam_reads (or a context on another device) makes the host
 * copies, once.  Every reference is complete when the call returns.  *on_device_out = 0: the file is one the device path does not take (not sorted by
 * reference, CIGARs parked in CG tags, anything malformed) and the host threads have been started on it instead -- results and
 * error reporting are the host decoder's either way.  Waiting on a deferred file nobody decoded starts the host decode. */
int spl_bam_open_deferred(const char *path, int n_threads, spl_bam **out);
int spl_bam_decode_device(spl_ctx *ctx, spl_bam *bam, int *on_device_out);
/* For a caller that will call spl_bam_decode_device from another thread in a moment while others may already wait for
 * references: the file is marked as taken by the device decoder now, so that those waits wait instead of starting the host
 * decode.  The promise must be kept (spl_bam_decode_device) or taken back (spl_bam_start), or the waits never end. */
int spl_bam_reserve_device(spl_bam *bam);
/* The same decode in SHARES, one per device (replaces SpliSER_v0_1_8.py:422 for a stretch of the file): a
 * BAM file's BGZF blocks are independent, so every device can take its own stretch -- over its own PCIe link, into its own memory,
 * no exchange.  spl_bam_share_plan cuts the file into up to n_shares stretches of EQUAL size in file bytes, at any BGZF block: a
 * share owns the records that begin in its blocks (the plan finds the record boundary at every cut by inflating a few blocks on
 * the host; a share's decoder must arrive exactly at the next share's first record, or the plan is dropped), so a reference may
 * lie in several shares.  That is what the per-site loop allows (processSites, :681-692: nothing in it needs a whole chromosome)
 * and what checkBam's counters allow (:519-559 only ever add one per read): each device counts its stretch of a reference
 * against the reference's whole site table, the host adds the partial beta1 / beta2s_reads / dbl arrays, and one spl_sse call
 * runs on the sums -- still no collective.  *n_out = the shares made (fewer than asked for only when the file has fewer blocks).
 * spl_bam_share_range: the references share k CAN hold records of (tid_lo <= tid < tid_hi; a superset by at most one at the upper
 * end; the last share also holds the records without a reference); spl_bam_share_info: the bytes of the file its own blocks take
 * (what the balance of the plan is judged by), the stretch [u_lo, u_hi) of the inflated stream its records begin in, and the
 * blocks it inflates behind its own for the end of its last record.  The file is then reserved (spl_bam_reserve_device) and
 * EVERY share decoded by a spl_bam_decode_device_share call, each on the context of the device that is to count it; the file is
 * complete when the last of them returns.  Should any share not be decodable on its device, all of them are dropped and the host
 * threads decode the file (*on_device_out = 0 from that share's call; spl_bam_decoded_on_device says how it ended).  Afterwards
 * spl_bam_share_ref says what share k holds of a reference and spl_reads_add_bam_share adds exactly that to a read set on the
 * share's device.  spl_bam_share_count_host (diagnostic, no GPU): the records of share k per reference by the host's inflate and
 * a plain walk from u_lo that must arrive at u_hi -- per_tid[n_ref + 1], the last entry the records without a reference. */
int spl_bam_share_plan(spl_bam *bam, int n_shares, int *n_out);
int spl_bam_share_range(spl_bam *bam, int k, int *tid_lo_out, int *tid_hi_out);
int spl_bam_share_info(spl_bam *bam, int k, int64_t *file_bytes_out, int64_t *u_lo_out, int64_t *u_hi_out, int64_t *tail_blocks_out);
int spl_bam_share_count_host(spl_bam *bam, int k, int64_t *per_tid);
int spl_bam_decode_device_share(spl_ctx *ctx, spl_bam *bam, int k, int *on_device_out);
int spl_bam_share_ref(spl_bam *bam, int k, int tid, int64_t *n_reads_out, int64_t *max_end_out);
int spl_reads_add_bam_share(spl_ctx *ctx, spl_dreads *dr, spl_bam *bam, int share, int tid, int32_t pos_shift);
int spl_bam_decoded_on_device(spl_bam *bam, int *on_device_out);
/* For whoever runs spl_bam_decode_device / _share on threads of his own and waits for the outcome elsewhere: returns when the
 * file is no longer the device decoders' to decide about -- *on_device_out = 1: its reads are on the device(s), every reference
 * complete; 0: the host threads have it (and may still be decoding: spl_bam_wait_ref / _all).  It does not wait for the decoders'
 * calls to RETURN: a call gives its buffers, streams and events back after it has made the references complete (10 ms for a
 * large file), and the counting need not stand behind that (`process` Step 3, SpliSER_v0_1_8.py:681-692 per chromosome). */
int spl_bam_wait_device(spl_bam *bam, int *on_device_out);
/* A deferred file's other option, said out loud: decode on the host's threads, starting now.  Also ends a reservation that
 * nobody has taken up (its maker failed before it could call spl_bam_decode_device). */
int spl_bam_start(spl_bam *bam);
/* Inflated bytes per file byte over the first record blocks of a deferred file (0 = cannot tell).  Diagnostic: BGZF inflate is
 * what a decode costs; since the wave-per-block decoder the GPU is the faster side for a real library's file (3...4) and for one
 * that inflates at memset speed (synthetic data: 50) alike, and `process` no longer asks. */
int spl_bam_compression_ratio(spl_bam *bam, double *ratio_out);
/* Records, their CIGAR ops, and the inflated bytes both were counted in, sampled on the host at three places of the file (the
 * whole block directory is walked first): out3 = {records, ops, bytes}.  What the device decoder sizes its extracted arrays by
 * before anything runs on the device (spl_capi.cpp: early_room); here for tests and diagnostics.  SPL_ERR_FORMAT: cannot tell. */
int spl_bam_sample(spl_bam *bam, int64_t *out3);
int spl_bam_wait_ref(spl_bam *bam, int tid, int64_t *n_reads_out, int64_t *max_end_out);
int spl_bam_wait_all(spl_bam *bam, int *sorted_out);
/* spl_bam_close waits for a decode in progress to END.  A caller who only wants to leave (something else failed) says so first:
 * after spl_bam_cancel the decoders stop at their next batch (host) or window (device), a decode that has not begun never
 * does, and waiting calls return with an error. */
void spl_bam_cancel(spl_bam *bam);
/* Why the device decoder (spl_bam_decode_device / _share) left this file to the host threads: "" if it did not, otherwise a few
 * words (not sorted by reference, a CIGAR parked in a CG tag, a block that did not inflate, not enough device memory, ...).  The
 * pointer is good while the file is open.  (The reference has no counterpart: samtools reads whatever it is given, :422.) */
const char *spl_bam_decline_reason(spl_bam *bam);
void spl_bam_close(spl_bam *bam);
int spl_bam_n_ref(const spl_bam *bam);
const char *spl_bam_ref_name(const spl_bam *bam, int tid);
int64_t spl_bam_ref_length(const spl_bam *bam, int tid);
int64_t spl_bam_n_records(const spl_bam *bam); /* all records, including those without a reference */
/* Borrowed view (valid until spl_bam_close) of the reads placed on reference `tid` as BAM-native arrays, assembled on the
 * first call (waits for the whole file); *max_end_out = largest 1-based end coordinate any of them reaches (for shard
 * packing).  The GPU path does not need these arrays (spl_reads_add_bam). */
int spl_bam_reads(const spl_bam *bam, int tid, spl_reads *out, int64_t *max_end_out);

/* Test / synthetic-workload utility (no reference counterpart): write per-reference read sets as a coordinate-ordered
 * BAM with dummy names, SEQ and QUAL of the query length, BGZF blocks deflated on n_threads (0 = all cores). */
int spl_bam_write(const char *path, int n_ref, const char *const *ref_names, const int64_t *ref_lengths,
                  const spl_reads *per_ref, int level, int n_threads);
/* Same with a choice of what SEQ / QUAL hold: seq_mode 0 = constant bytes (spl_bam_write: a file that deflates to a few bytes
 * per record), 1 = pseudo-random bases and binned qualities in runs (deflates about 4x, like a real library). */
int spl_bam_write2(const char *path, int n_ref, const char *const *ref_names, const int64_t *ref_lengths,
                   const spl_reads *per_ref, int level, int n_threads, int seq_mode);

/* ---- junction table of a read set (what the pipeline otherwise takes from `regtools junctions extract`) --------------
 * Every N op of every mapped read is a junction (left, right) in SpliSER's site convention (left = last base before the
 * intron, right = last intronic base, SpliSER_v0_1_8.py:482-483 -- the numbers findAlphaCounts derives from a BED12 line,
 * :275-276).  spl_junctions builds, on the device, the table of distinct (left, right[, read strand]) with the number of
 * reads carrying each and the longest anchors on either side (reference bases of the read between the junction and the
 * previous / next N op or read end: the block sizes of a BED12 line), and returns their number; spl_junctions_get copies
 * the table of the last call, sorted by (left, right, strand), into caller arrays (any may be NULL).  stranded: 0 ->
 * strand '?', 1 = fr / 2 = rf -> '+' / '-' of the read by check_strand's rule (:374-406).  Policy knobs in the sense of
 * regtools' -a / -m / -M: a read supports a junction only if both of ITS anchors are >= min_anchor and the intron length
 * is in [min_intron, max_intron] (max_intron 0 = no upper limit); 0, 0, 0 counts every N op. */
int spl_junctions(spl_ctx *ctx, const spl_dreads *dr, int stranded, int32_t min_anchor, int32_t min_intron, int32_t max_intron,
                  int64_t *n_out);
int spl_junctions_get(const spl_ctx *ctx, int32_t *left, int32_t *right, uint8_t *strand, uint32_t *count,
                      uint32_t *anchor_left, uint32_t *anchor_right);

/* ---- host helper of Step 1 ---------------------------------------------------------------------------
 * binary_gene_search (SpliSER_v0_1_8.py:118-173) for a batch of query positions against one chromosome's gene list
 * (in list order), probe for probe like the reference.  Strand bytes are '+', '-' or 0 for anything else;
 * out[q] = index of the gene found or -1.  No GPU involved. */
int spl_gene_search(const int64_t *left, const int64_t *right, const uint8_t *gene_strand, int64_t n_genes,
                    const int64_t *q_pos, const uint8_t *q_strand, int64_t n_queries, int is_stranded, int32_t *out);

/* ---- host helpers of Steps 0-1: the text inputs as columns ----------------------------------------------------------------
 * spl_bed_open: every line of exactly 12 tab-separated columns of a BED12 junction file as findAlphaCounts reads it
 * (SpliSER_v0_1_8.py:257-277): chromosome (index into the file's chromosome names, first-appearance order), left = chromStart +
 * blockSizes[0], right = chromEnd - blockSizes[1], alpha = score, strand byte (0 = empty column).  spl_gff_open: every `gene`
 * line of a GFF / GTF file as createGenes keeps it (:81-87, HTSeq conventions): chromosome, left = column 4 - 1, right = column 5,
 * strand byte, name = value of the first attribute.  Both fail with SPL_ERR_FORMAT on anything they are not sure to read the way
 * the reference's Python would (the caller then reads line by line).  No GPU involved. */
typedef struct spl_textfile spl_textfile;
int spl_bed_open(const char *path, spl_textfile **out);
int spl_gff_open(const char *path, spl_textfile **out);
void spl_text_close(spl_textfile *t);
int64_t spl_text_rows(const spl_textfile *t);
int32_t spl_text_n_chrom(const spl_textfile *t);
const char *spl_text_chrom_name(const spl_textfile *t, int32_t k);
const int32_t *spl_text_chrom(const spl_textfile *t);                 /* [rows] */
const int64_t *spl_text_i64(const spl_textfile *t, int which);        /* [rows] which: 0 left, 1 right, 2 alpha (BED) */
const uint8_t *spl_text_strand(const spl_textfile *t);                /* [rows] */
const char *spl_text_names(const spl_textfile *t, const uint32_t **off_out); /* GFF: gene names, one blob + rows + 1 offsets */

/* ---- output (outputBedFile, SpliSER_v0_1_8.py:641-664) -------------------------------------------------------
 * Appends the rows of one chromosome to a .SpliSER.tsv file (the caller writes the header line): 12 tab-separated
 * columns, SSE as "%.3f", the two cryptic columns as an integer and "%.5f" or "NA NA" when cryptic == 0, Partners as
 * Python's str(dict) "{pos: count, ...}" in partner order, Competitors as str(list).  Strand and gene texts come as one
 * blob each with n_sites + 1 offsets.  No GPU involved.  spl_tsv_append_many: the same for several chromosomes at once, in the
 * order given (their rows are formatted side by side). */
typedef struct spl_tsv_rows {
    const char *chrom;
    int64_t n_sites;
    const int64_t *pos;
    const char *strand_blob; const uint32_t *strand_off;
    const char *gene_blob; const uint32_t *gene_off;
    const double *sse;
    const int64_t *alpha;
    const uint32_t *beta1;
    const int64_t *beta2_simple, *beta2_cryptic; /* beta2_cryptic, beta2_weighted: read when cryptic != 0 */
    const double *beta2_weighted;
    const uint32_t *part_off; const int64_t *part_pos, *edge_cnt;
    const uint32_t *comp_off; const int64_t *comp_pos;
} spl_tsv_rows;
int spl_tsv_append_many(const char *path, int32_t n_chrom, const spl_tsv_rows *rows, int cryptic);
int spl_tsv_append(const char *path, const char *chrom, int64_t n_sites, const int64_t *pos, const char *strand_blob,
                   const uint32_t *strand_off, const char *gene_blob, const uint32_t *gene_off, const double *sse,
                   const int64_t *alpha, const uint32_t *beta1, const int64_t *beta2_simple, int cryptic,
                   const int64_t *beta2_cryptic, const double *beta2_weighted, const uint32_t *part_off,
                   const int64_t *part_pos, const int64_t *edge_cnt, const uint32_t *comp_off, const int64_t *comp_pos);

/* The writer's "%.3f" / "%.5f": x >= 0 with `digits` (0..6) decimals into out64 (NUL-terminated), correctly rounded from the
 * double's exact value, ties to even -- what Python's "{0:.3f}".format gives (SpliSER_v0_1_8.py:655-660).  Exported so that the
 * tests can hold it against Python over many values. */
int spl_fmt_fixed(double x, int digits, char *out64);

/* ---- `combine` / `combineShallow`: the host walk on columns (csrc/spl_combine.cpp) --------------------------------------------
 * Replaces the lock-step loop of SpliSER_v0_1_8.py:820-915 (combine) and :1007-1166 (combineShallow) around the calls of
 * checkBam (:903): spl_combine_open parses the per-sample .SpliSER.tsv files into columns (SPL_ERR_FORMAT on anything it is
 * not sure to read the way Python's str.split / int / float / literal_eval would: the caller then walks the files in Python);
 * spl_combine_region_runs gives what the region order (:761-790) is deduced from; spl_combine_merge is the walk -- merged
 * sites in output order, and for every sample the gap-fill queries as tables (rows by position per region; strand, partners
 * and competitors as the walk had them when it reached that sample); the queries are answered through spl_count in
 * combine mode and handed back with spl_combine_answers; spl_combine_write is outputCombinedLines (:722-740).  No GPU. */
typedef struct spl_combine spl_combine;
typedef struct spl_query_table {
    const char *chrom;            /* region name */
    int64_t n;                    /* rows */
    const int64_t *pos;           /* [n] site positions, ascending */
    const int64_t *site;          /* [n] index of the merged site a row answers */
    const uint8_t *strand;        /* [n] first byte of the site's strand text at that moment, 0 if it had none */
    const uint32_t *part_off;     /* [n + 1] */
    const int64_t *part_pos;      /* partner positions (one-way lists) */
    const uint32_t *comp_off;     /* [n + 1] */
    const int64_t *comp_pos;      /* competitor positions */
} spl_query_table;
int spl_combine_open(const char *const *tsv_paths, int32_t n_samples, spl_combine **out);
void spl_combine_close(spl_combine *c);
int64_t spl_combine_rows(const spl_combine *c, int32_t idx);
int32_t spl_combine_n_texts(const spl_combine *c);
const char *spl_combine_text(const spl_combine *c, int32_t id);
int64_t spl_combine_region_runs(const spl_combine *c, int32_t idx, int32_t *ids, int64_t cap);
int spl_combine_keep_gene(spl_combine *c, const char *gene);
int spl_combine_merge(spl_combine *c, const char *const *chroms, int32_t n_chroms, int is_stranded, const char *q_gene, int shallow,
                      int64_t min_samples, int64_t min_reads, double min_sse);
int64_t spl_combine_n_sites(const spl_combine *c);
int64_t spl_combine_n_gap_sites(const spl_combine *c);
int64_t spl_combine_skipped(const spl_combine *c, const int64_t **pairs);
int32_t spl_combine_n_tables(const spl_combine *c, int32_t idx);
int spl_combine_table(const spl_combine *c, int32_t idx, int32_t k, spl_query_table *out);
int spl_combine_answers(spl_combine *c, int32_t idx, int64_t n, const int64_t *site, const uint32_t *beta1, const uint32_t *beta2_simple);
int spl_combine_write(const spl_combine *c, const char *path, const char *const *titles, int cryptic);

/* str(x) of a Python float (the beta2_weighted column of .combined.tsv, :735) into out64, NUL-terminated.  Test hook. */
int spl_fmt_repr(double x, char *out64);

#ifdef __cplusplus
}
#endif
#endif /* SPLISER_H */
