// spl_bam.h -- internal: what spl_capi.cpp needs from the BAM decoder (bam_reader.cpp).
#ifndef SPL_BAM_H
#define SPL_BAM_H

#include "../../include/spliser.h"
#include "spl_pack.h"

// The reads of reference `tid` as a packer source: the decoder's own parts, in file order, nothing copied.  Waits until the
// reference is complete (spl_bam_wait_ref).  The views stay valid until spl_bam_release_ref(tid) or spl_bam_close.
int spl_bam_source(spl_bam *bam, int tid, splpack::Source *out, int64_t *max_end_out);

// ---- for the device decoder (spl_capi.cpp: spl_bam_decode_device) -------------------------------------------------------
// A file opened with spl_bam_open_deferred has its header read and nothing else started: the caller then either hands the decode
// to the host threads (spl_bam_start_host) or does it elsewhere and gives the result back (spl_bam_adopt).
struct spl_bam_block_info { uint64_t data_off; uint64_t uoff; uint32_t data_len, isize, crc; };
int spl_bam_walk_all(spl_bam *bam);                       // the whole block directory, now (SPL_OK or the file's error)
void spl_bam_walk_some(spl_bam *bam, size_t bytes);
bool spl_bam_walk_complete(const spl_bam *bam);
size_t spl_bam_block_count(const spl_bam *bam);
void spl_bam_block_get(const spl_bam *bam, size_t i, spl_bam_block_info *out);
const uint8_t *spl_bam_image(const spl_bam *bam, size_t *fsize_out);
int spl_bam_fd(const spl_bam *bam);                        // the open file (pread: bytes without touching the mapping's page tables)
uint64_t spl_bam_header_end(const spl_bam *bam);           // where the first record starts in the inflated stream
int spl_bam_thread_count(const spl_bam *bam);
// records, CIGAR ops and the inflated bytes they were counted in, sampled on the host at three places of blocks [b_lo, b_hi)
bool spl_bam_sample_density(spl_bam *bam, size_t b_lo, size_t b_hi, uint64_t *n_rec_out, uint64_t *n_ops_out, uint64_t *n_bytes_out);
// The placed records of the whole file in file order as four malloc'ed arrays (the file takes them over and frees them with
// free()); reference t has records [ref_first[t], ref_first[t] + ref_n[t]), cig_off holds n_total + 1 offsets into cigar.
int spl_bam_adopt(spl_bam *bam, int32_t *pos, uint16_t *flag, uint32_t *cig_off, uint32_t *cigar, const int64_t *ref_first, const int64_t *ref_n,
                  const int64_t *ref_max_end, int64_t n_records_total);
// what the device decoder keeps in device memory for the device packer: an opaque handle, freed with the file
void spl_bam_set_device_reads(spl_bam *bam, void *handle, void (*free_fn)(void *));
void *spl_bam_device_reads(spl_bam *bam, int tid);          // the handle that holds ALL of reference `tid` (null: none does -- no device decode, or the reference lies in several shares)
void *spl_bam_share_reads(spl_bam *bam, int share);         // what share `share` left in device memory (null: nothing)

// ---- a decode in SHARES: each device takes a stretch of the file, cut at ANY BGZF block ---------------------------------
// A share owns the records that BEGIN in blocks [block_lo, block_own) -- in the inflated stream: at offsets [u_lo, u_hi), both
// record boundaries the plan found by inflating a few blocks at every cut on the host -- and inflates blocks [block_lo,
// block_hi): the tail [block_own, block_hi) holds the end of its last record and nothing else of its own.  A reference's records
// may lie in several shares (counters are additive per read, SpliSER_v0_1_8.py:519-559: each device counts its stretch against the
// reference's whole site table); tid_lo <= tid < tid_hi are the references a share CAN hold records of (records without a
// reference count as n_ref; the last share's tid_hi = n_ref + 1).
struct spl_bam_share { uint64_t block_lo, block_hi; int32_t tid_lo, tid_hi; uint64_t block_own, u_lo, u_hi; };
// Cuts the file into up to n_shares stretches of equal size in file bytes.  The whole block directory is walked first.  Once per
// file; later calls return the first plan.
int spl_bam_share_get(spl_bam *bam, int k, spl_bam_share *out);
// A share's decoder is done: `handle` holds its records (share-local first record per reference in ref_first), or failed != 0.
// When the last share has reported the file is complete -- or, if one failed, everything is dropped and the host threads decode.
int spl_bam_share_done(spl_bam *bam, int k, void *handle, void (*free_fn)(void *), const int64_t *ref_first, const int64_t *ref_n,
                       const int64_t *ref_max_end, int64_t n_records, int failed);
int spl_bam_shares_on_device(spl_bam *bam);
bool spl_bam_cancelled(const spl_bam *bam);                   // spl_bam_cancel was called: stop at the next window                  // 1: all shares reported and none failed
// spl_bam_adopt with null arrays = the reads stay on the device; `fetch(handle, ...)` brings malloc'ed host copies when a host-side
// reader asks for them (spl_bam_source, spl_bam_reads)
void spl_bam_set_fetch(spl_bam *bam, int (*fetch)(void *, int32_t **, uint16_t **, uint32_t **, uint32_t **));
int spl_bam_start_host(spl_bam *bam);                      // decode on the host's threads unless somebody decodes already
bool spl_bam_claim_for_device(spl_bam *bam);               // the device decoder takes the file (false: it is taken)
void spl_bam_note_decline(spl_bam *bam, const char *why); // why the device decoder leaves the file to the host threads
int spl_bam_device_gives_up(spl_bam *bam);                 // ... and hands it to the host threads after all
void spl_bam_linger(spl_bam *bam, double seconds);       // a decoder with only its clearing up left: until spl_bam_cancel, `seconds` at most

#endif
