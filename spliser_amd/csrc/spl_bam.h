// spl_bam.h -- internal: what spl_capi.cpp needs from the BAM decoder (bam_reader.cpp).
#ifndef SPL_BAM_H
#define SPL_BAM_H

#include "../../include/spliser.h"
#include "spl_pack.h"

// The reads of reference `tid` as a packer source: the decoder's own parts, in file order, nothing copied.  Waits until the
// reference is complete (spl_bam_wait_ref).  The views stay valid until spl_bam_release_ref(tid) or spl_bam_close.
int spl_bam_source(spl_bam *bam, int tid, splpack::Source *out, int64_t *max_end_out);

#endif
