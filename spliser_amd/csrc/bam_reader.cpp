// bam_reader.cpp -- BGZF/BAM ingest for the MI355X build (host side, no htslib / samtools in the image).
//
// Replaces the per-site `samtools view BAM chr:t-(t+1)` child process of the reference
// (SpliSER_v0_1_8.py:422): the file is decoded ONCE, BGZF blocks are inflated on a pool of host threads,
// and only the three fields checkBam reads from each SAM line (flag, POS, CIGAR -- :434-437) are kept,
// per reference sequence, as structure-of-arrays ready for spl_reads_upload().
//
// Like `samtools view` without -F/-q, no record is filtered by flag or MAPQ.  CIGARs of more than
// 65535 ops stored in a CG:B,I tag behind an `<l_seq>S<rlen>N` placeholder are restored the way htslib's
// bam_tag2cigar does, because that is what samtools would print.
//
// Format: SAM/BAM specification sections 4.1 (BGZF) and 4.2 (BAM).
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <math.h>
#include <zlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spliser.h"
#include "spl_bam.h"
#include "spl_error.h"

namespace {

// The decoder's threads stay on the NUMA node the calling thread is on: inflated data is written by some threads and walked
// by others, and on a two-socket host every such hand-over across the sockets goes over the inter-socket link (measured on the
// MI355X box, 2 x 64 cores: 100 M records in 1.31 s with the threads left to roam, 0.98 s on one node).  SPL_BAM_NO_PIN=1
// turns it off.  The caller's own affinity is never changed.
struct NodeCpus {
    cpu_set_t set;
    bool valid = false;
    NodeCpus()
    {
        if (getenv("SPL_BAM_NO_PIN")) return;
        const int cpu = sched_getcpu();
        if (cpu < 0) return;
        cpu_set_t allowed;
        if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
        for (int node = 0; node < 64 && !valid; ++node) {
            char path[96];
            snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
            FILE *f = fopen(path, "r");
            if (!f) break;
            char text[4096];
            const bool got = fgets(text, sizeof text, f) != nullptr;
            fclose(f);
            if (!got) continue;
            cpu_set_t s;
            CPU_ZERO(&s);
            bool mine = false;
            for (char *p = text; *p;) { // "0-63,128-191"
                char *end = nullptr;
                const long a = strtol(p, &end, 10);
                if (end == p) break;
                long b = a;
                if (*end == '-') { p = end + 1; b = strtol(p, &end, 10); }
                for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { if (CPU_ISSET((int)c, &allowed)) CPU_SET((int)c, &s); if (c == cpu) mine = true; }
                p = (*end == ',') ? end + 1 : end;
                if (*end != ',') break;
            }
            if (mine && CPU_COUNT(&s) >= 8) { set = s; valid = true; }
        }
    }
    void pin_this_thread() const { if (valid) (void)sched_setaffinity(0, sizeof(set), &set); }
};


// ---- optional libdeflate (present in the image as libdeflate.so.0 without a header) -------------------
struct Deflate {
    void *(*alloc)(void) = nullptr;
    int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*free_)(void *) = nullptr;
    uint32_t (*crc32)(uint32_t, const void *, size_t) = nullptr;
    void *(*calloc_)(int) = nullptr;                                           // compressor (the BAM writer)
    size_t (*compress)(void *, const void *, size_t, void *, size_t) = nullptr;
    void (*cfree)(void *) = nullptr;
    bool ok = false, cok = false;
    Deflate()
    {
        if (getenv("SPL_BAM_NO_LIBDEFLATE")) return;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void *(*)(void))dlsym(h, "libdeflate_alloc_decompressor");
        decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
        free_ = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        crc32 = (uint32_t(*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32");
        ok = alloc && decompress && free_ && crc32;
        calloc_ = (void *(*)(int))dlsym(h, "libdeflate_alloc_compressor");
        compress = (size_t(*)(void *, const void *, size_t, void *, size_t))dlsym(h, "libdeflate_deflate_compress");
        cfree = (void (*)(void *))dlsym(h, "libdeflate_free_compressor");
        cok = ok && calloc_ && compress && cfree;
    }
};
const Deflate &deflate_lib()
{
    static Deflate d;
    return d;
}

inline uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline int32_t le32s(const uint8_t *p) { return (int32_t)le32(p); }

struct Block {
    size_t coff;    // offset of the block in the file
    uint64_t uoff;  // offset of its payload in the inflated stream
    uint32_t csize; // whole block
    uint32_t xlen;
    uint32_t isize; // uncompressed payload
    uint32_t crc;   // of the payload, as the block's trailer says (read with ISIZE: whoever wants it later would miss the cache again)
};

// The file's BGZF blocks, found by walking the block headers from the start (each header says where the next block begins).
// The walk is one thread's pointer chase over the whole file -- a tenth of a second per gigabyte -- so only the blocks of the
// BAM header are walked by the opening call; the rest is walked beside the decode, which takes the blocks as they are
// announced.  Blocks live in chunks that never move: readers index below n_ready without a lock.
struct BlockDir {
    static constexpr size_t CHUNK = (size_t)1 << 15;
    std::vector<Block *> chunks;        // sized for the worst case when the file is opened, filled by the walk
    std::atomic<size_t> n_ready{0};     // blocks [0, n_ready) are final
    std::atomic<int> state{0};          // 0 = being walked, 1 = complete, -1 = failed (`error`, `err_code`)
    std::string error;
    int err_code = 0;
    size_t off = 0;                     // the walk's position in the file
    uint64_t uoff = 0;                  // ... and in the inflated stream
    const Block &at(size_t i) const { return chunks[i / CHUNK][i % CHUNK]; }
    ~BlockDir() { for (Block *c : chunks) free(c); }
};

// Walk up to max_blocks further block headers (0 = to the end of the file).  Returns false when the walk has failed.
bool walk_blocks(BlockDir &dir, const uint8_t *file, size_t fsize, const char *path, size_t max_blocks)
{
    auto failed = [&](int code, const char *what, size_t where) {
        char text[512];
        snprintf(text, sizeof text, "%s: %s at offset %zu", path, what, where);
        dir.error = text;
        dir.err_code = code;
        dir.state.store(-1, std::memory_order_release);
        return false;
    };
    size_t n = dir.n_ready.load(std::memory_order_relaxed), done = 0;
    size_t off = dir.off;
    while (off < fsize && (max_blocks == 0 || done < max_blocks)) {
        if (fsize - off < 18 || file[off] != 0x1f || file[off + 1] != 0x8b || file[off + 2] != 8 || !(file[off + 3] & 4))
            return failed(SPL_ERR_FORMAT, "not BGZF (BAM files are BGZF-compressed)", off);
        const uint32_t xlen = le16(file + off + 10);
        if (fsize - off < 12 + (size_t)xlen) return failed(SPL_ERR_FORMAT, "truncated BGZF header", off);
        uint32_t bsize = 0;
        bool have = false;
        const size_t x_end = off + 12 + xlen;
        for (size_t x = off + 12; x + 4 <= x_end;) {
            const uint32_t slen = le16(file + x + 2);
            if (x + 4 + (size_t)slen > x_end) break; // a subfield that runs past the extra area: corrupt, and not ours to read
            if (file[x] == 'B' && file[x + 1] == 'C' && slen == 2) { bsize = (uint32_t)le16(file + x + 4) + 1; have = true; }
            x += 4 + slen;
        }
        if (!have || bsize < 12 + xlen + 8 || fsize - off < bsize) return failed(SPL_ERR_FORMAT, "corrupt or truncated BGZF block", off);
        Block b;
        b.coff = off; b.uoff = dir.uoff; b.csize = bsize; b.xlen = xlen; b.isize = le32(file + off + bsize - 4); b.crc = le32(file + off + bsize - 8);
        if (b.isize > 65536) return failed(SPL_ERR_FORMAT, "BGZF ISIZE > 64 KiB", off);
        if (n / BlockDir::CHUNK >= dir.chunks.size()) return failed(SPL_ERR_FORMAT, "more BGZF blocks than the file has room for", off);
        Block *&chunk = dir.chunks[n / BlockDir::CHUNK];
        if (!chunk) chunk = (Block *)malloc(sizeof(Block) * BlockDir::CHUNK);
        if (!chunk) return failed(SPL_ERR_NOMEM, "out of host memory for the block directory", off);
        chunk[n % BlockDir::CHUNK] = b;
        ++n;
        ++done;
        off += bsize;
        dir.uoff += b.isize;
        if ((n & 255u) == 0) dir.n_ready.store(n, std::memory_order_release);
    }
    dir.off = off;
    dir.n_ready.store(n, std::memory_order_release);
    if (off >= fsize) {
        // htslib only warns about a missing EOF marker; a truncated file is far more likely than a writer that omits it, and
        // silently counting fewer reads is the reference's worst failure mode: refuse.  (The opening call has looked at the
        // file's last 28 bytes already; this is the same statement about the walked blocks.)
        if (n == 0 || dir.at(n - 1).isize != 0) return failed(SPL_ERR_IO, "BGZF EOF marker missing -- file is truncated", off);
        dir.state.store(1, std::memory_order_release);
    }
    return true;
}

// ---- the rest of the directory by several threads at once ---------------------------------------------------------------
// walk_blocks is one thread's pointer chase: 0.25 us a block, 0.3 s for the 865 000 blocks of a 200 M-read file -- longer than
// that file takes to reach the device.  Whoever wants the WHOLE directory (spl_bam_walk_all: the device decoder) gets the
// remainder of the file cut into stretches, one thread each: the first starts where the walk stands, the others at the first
// offset of their stretch where a block header stands and two more follow it -- a guess, made good by the stitching: a
// stretch's walk must end exactly where the next one's began, and by induction from the first every block is then one the
// one-thread walk would have found.  Anything else (a stretch without a block start, a header that does not hold, ends that do
// not meet) and nothing is kept: the one-thread walk does the file, and finds the words for what is wrong with it.

bool block_at(const uint8_t *file, size_t fsize, size_t off, Block &b)
{
    if (fsize - off < 18 || file[off] != 0x1f || file[off + 1] != 0x8b || file[off + 2] != 8 || !(file[off + 3] & 4)) return false;
    const uint32_t xlen = le16(file + off + 10);
    if (fsize - off < 12 + (size_t)xlen) return false;
    uint32_t bsize = 0;
    bool have = false;
    const size_t x_end = off + 12 + xlen;
    for (size_t x = off + 12; x + 4 <= x_end;) {
        const uint32_t slen = le16(file + x + 2);
        if (x + 4 + (size_t)slen > x_end) break;
        if (file[x] == 'B' && file[x + 1] == 'C' && slen == 2) { bsize = (uint32_t)le16(file + x + 4) + 1; have = true; }
        x += 4 + slen;
    }
    if (!have || bsize < 12 + xlen + 8 || fsize - off < bsize) return false;
    b.coff = off; b.uoff = 0; b.csize = bsize; b.xlen = xlen; b.isize = le32(file + off + bsize - 4); b.crc = le32(file + off + bsize - 8);
    return b.isize <= 65536;
}

struct Stretch {
    std::vector<Block> blocks;
    size_t start = 0, end = 0;
    bool ok = false;
};

void walk_stretch(const uint8_t *file, size_t fsize, size_t from, size_t until, bool exact, Stretch &out)
{
    size_t off = from;
    if (!exact) {
        bool found = false;
        while (off < until) {
            const void *hit = memchr(file + off, 0x1f, until - off);
            if (!hit) break;
            off = (size_t)((const uint8_t *)hit - file);
            Block b;
            size_t q = off;
            int chain = 0;
            while (chain < 3 && q < fsize && block_at(file, fsize, q, b)) { q += b.csize; ++chain; }
            if (chain == 3 || (chain > 0 && q == fsize)) { found = true; break; }
            ++off;
        }
        if (!found) return;
    }
    out.start = off;
    out.blocks.reserve((until - off) / 4096 + 16);
    while (off < until) {
        Block b;
        if (!block_at(file, fsize, off, b)) return;
        out.blocks.push_back(b);
        off += b.csize;
    }
    out.end = off;
    out.ok = true;
}

// The directory from where the walk stands up to the first block that begins at or behind `until` (fsize: the remainder of the
// file) in one go; false = nothing done (the caller walks on by itself).
bool walk_rest_in_parallel(BlockDir &dir, const uint8_t *file, size_t fsize, int n_threads, size_t until = (size_t)-1)
{
    const size_t from = dir.off;
    until = std::min(until, fsize);
    size_t least = (size_t)64 << 20; // (below this the one-thread walk is done before the threads have started)
    if (const char *e = getenv("SPL_WALK_PARALLEL_MIN")) least = (size_t)std::max(4096ll, atoll(e));
    if (dir.state.load() != 0 || from >= until || until - from < least) return false;
    const size_t T = std::min<size_t>((size_t)std::max(2, std::min(n_threads, 32)), (until - from) / (least / 4));
    if (T < 2) return false;
    std::vector<Stretch> st(T);
    const size_t span = (until - from + T - 1) / T;
    std::vector<std::thread> pool;
    auto work = [&](size_t t) { walk_stretch(file, fsize, from + t * span, std::min(until, from + (t + 1) * span), t == 0, st[t]); };
    for (size_t t = 1; t < T; ++t) pool.emplace_back(work, t);
    work(0);
    for (std::thread &th : pool) th.join();
    size_t n_new = 0;
    for (size_t t = 0; t < T; ++t) {
        if (!st[t].ok || (t && st[t].start != st[t - 1].end)) return false;
        n_new += st[t].blocks.size();
    }
    const bool whole = until == fsize;
    if (whole && st[T - 1].end != fsize) return false;
    size_t n = dir.n_ready.load(std::memory_order_relaxed);
    if ((n + n_new + BlockDir::CHUNK - 1) / BlockDir::CHUNK > dir.chunks.size()) return false;
    uint64_t uoff = dir.uoff;
    for (size_t t = 0; t < T; ++t) {
        for (Block b : st[t].blocks) {
            Block *&chunk = dir.chunks[n / BlockDir::CHUNK];
            if (!chunk) chunk = (Block *)malloc(sizeof(Block) * BlockDir::CHUNK);
            if (!chunk) return false; // (what has been written lies beyond n_ready: the one-thread walk writes it again)
            b.uoff = uoff;
            uoff += b.isize;
            chunk[n % BlockDir::CHUNK] = b;
            ++n;
        }
    }
    if (whole && (n == 0 || dir.at(n - 1).isize != 0)) return false; // (no EOF marker: the one-thread walk says so)
    dir.off = st[T - 1].end;
    dir.uoff = uoff;
    dir.n_ready.store(n, std::memory_order_release);
    if (whole) dir.state.store(1, std::memory_order_release);
    return true;
}

// The reads one batch found for one reference, BAM-native: exact-size arrays carved out of the extracting thread's arena.
struct RefReads {
    int32_t *pos = nullptr;
    uint16_t *flag = nullptr;
    uint32_t *cig_off = nullptr; // n + 1 entries, cig_off[0] = 0: ops of read k are cigar[cig_off[k] .. cig_off[k + 1])
    uint32_t *cigar = nullptr;
    size_t n = 0, n_ops = 0;
    int64_t max_end = 0;
};

// Bump allocation out of large slabs (2 MiB-aligned, huge pages asked for).  What the decoder extracts lives until the file
// is closed, a decode produces gigabytes of it in pieces of a few hundred kilobytes, and malloc serves pieces of that size
// with an mmap and an munmap each: on 64 threads the address-space lock and the TLB shoot-downs of those calls were the
// decode's ceiling.  A slab is never given back before spl_bam_close.
struct Arena {
    std::vector<void *> slabs;
    uint8_t *cur = nullptr, *end = nullptr;
    bool failed = false;
    void *take(size_t bytes)
    {
        bytes = (bytes + 63) & ~(size_t)63;
        if ((size_t)(end - cur) < bytes) {
            const size_t huge = 2u << 20, want = std::max<size_t>(bytes, (size_t)16 << 20);
            const size_t size = (want + huge - 1) / huge * huge;
            void *slab = nullptr;
            if (posix_memalign(&slab, huge, size) != 0) { failed = true; return nullptr; }
            (void)madvise(slab, size, MADV_HUGEPAGE);
            slabs.push_back(slab);
            cur = (uint8_t *)slab;
            end = cur + size;
        }
        void *out = cur;
        cur += bytes;
        return out;
    }
};

// Inflate one BGZF block into dst (exactly b.isize bytes) and verify its CRC32.
bool inflate_block(const uint8_t *file, const Block &b, uint8_t *dst, void *ld)
{
    const uint8_t *cdata = file + b.coff + 12 + b.xlen;
    const size_t clen = (size_t)b.csize - 12 - b.xlen - 8;
    const uint8_t *tail = file + b.coff + b.csize - 8;
    const uint32_t want_crc = le32(tail);
    const Deflate &L = deflate_lib();
    if (b.isize == 0) return true;
    if (ld) {
        size_t got = 0;
        if (L.decompress(ld, cdata, clen, dst, b.isize, &got) != 0 || got != b.isize) return false;
        return L.crc32(0, dst, b.isize) == want_crc;
    }
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef *>(cdata);
    zs.avail_in = (uInt)clen;
    zs.next_out = dst;
    zs.avail_out = b.isize;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = (rc == Z_STREAM_END) && zs.total_out == b.isize;
    inflateEnd(&zs);
    if (!ok) return false;
    return (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, b.isize) == want_crc;
}

// Find a CG:B,I|i tag in the aux area; returns pointer to its uint32 array and its length, or nullptr.
const uint8_t *find_cg_tag(const uint8_t *aux, const uint8_t *end, uint32_t *n_out)
{
    const uint8_t *p = aux;
    while (p + 3 <= end) {
        const uint8_t t0 = p[0], t1 = p[1], ty = p[2];
        p += 3;
        size_t sz = 0;
        switch (ty) {
        case 'A': case 'c': case 'C': sz = 1; break;
        case 's': case 'S': sz = 2; break;
        case 'i': case 'I': case 'f': sz = 4; break;
        case 'd': sz = 8; break;
        case 'Z': case 'H': {
            const uint8_t *q = p;
            while (q < end && *q) ++q;
            if (q >= end) return nullptr;
            sz = (size_t)(q - p) + 1;
            break;
        }
        case 'B': {
            if (p + 5 > end) return nullptr;
            const uint8_t sub = p[0];
            const uint32_t n = le32(p + 1);
            size_t es;
            switch (sub) {
            case 'c': case 'C': es = 1; break;
            case 's': case 'S': es = 2; break;
            case 'i': case 'I': case 'f': es = 4; break;
            default: return nullptr;
            }
            if (t0 == 'C' && t1 == 'G' && (sub == 'I' || sub == 'i')) {
                if (p + 5 + (size_t)n * 4 > end) return nullptr;
                *n_out = n;
                return p + 5;
            }
            sz = 5 + (size_t)n * es;
            break;
        }
        default:
            return nullptr; // unknown type: cannot walk further
        }
        if (p + sz > end) return nullptr;
        p += sz;
    }
    return nullptr;
}

} // namespace

// Large buffers ask for transparent huge pages (the usual system setting is "madvise"): a decode touches about a gigabyte
// of fresh memory, and 4 KiB at a time that is a quarter of a million page faults contending for one address-space lock.
static void *big_alloc(size_t bytes)
{
    const size_t huge = 2u << 20;
    if (bytes < 4 * huge) return malloc(bytes ? bytes : 1);
    void *p = nullptr;
    if (posix_memalign(&p, huge, (bytes + huge - 1) / huge * huge) != 0) return nullptr;
    (void)madvise(p, (bytes + huge - 1) / huge * huge, MADV_HUGEPAGE);
    return p;
}

// What spl_bam_reads hands out: one exact-size allocation per array and reference, filled once after the last segment.
struct RefFinal {
    int64_t n = 0, n_cigar = 0, max_end = 0;
    int32_t *pos = nullptr;
    uint16_t *flag = nullptr;
    uint32_t *cig_off = nullptr; // n + 1
    uint32_t *cigar = nullptr;
    RefFinal() = default;
    RefFinal(const RefFinal &) = delete;
    RefFinal &operator=(const RefFinal &) = delete;
    ~RefFinal() { free(pos); free(flag); free(cig_off); free(cigar); }
};

struct PendingPart {
    int32_t tid = 0;
    RefReads reads;
    int share = -1;     // reads left on a device: the share whose arrays hold them (-1: a whole-file decode's) ...
    int64_t first = 0;  // ... and the part's first record in those arrays
};

// A BAM file being decoded (or decoded).  The decode runs on a thread of its own (which drives the inflate and parse pools);
// consumers wait per reference: in a coordinate-sorted file reference t is complete as soon as a record of a later reference
// has been seen, so its reads can be packed and sent to the GPU while the rest of the file is still being inflated.
struct spl_bam {
    std::vector<std::string> ref_names;
    std::vector<int64_t> ref_lengths;
    int n_refs = 0;
    // ---- filled by the decode thread, read under `mu` ----
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::vector<PendingPart *>> parts; // per reference, file order (owned; their arrays live in `slabs`)
    std::vector<void *> slabs;
    std::vector<int64_t> ref_max_end, ref_reads;
    int complete_upto = 0;      // every reference < this is complete -- if the file is sorted by reference
    int max_tid_seen = -1;
    bool out_of_order = false;  // a record of an earlier reference came after one of a later reference
    bool done = false;
    int err_code = 0;
    std::string error;
    int64_t n_records = 0;
    // ---- BAM-native arrays per reference, assembled on demand (spl_bam_reads) ----
    std::vector<RefFinal> refs_storage; // (never resized after the header: RefFinal is not copyable)
    std::vector<char> assembled;
    // ---- the decode thread and what it works on ----
    std::thread worker;
    void *map = nullptr;
    size_t fsize = 0;
    int fd = -1;               // kept open for readers that want the bytes without the mapping (spl_bam_decode_device)
    BlockDir dir;
    int n_threads = 1;
    // reads adopted from the device decoder without their host copies: fetched (dev_fetch) when somebody wants to read them
    bool lazy = false;
    bool fetching = false;     // somebody is copying a share's reads to the host right now (fetch_lazy)
    int (*dev_fetch)(void *, int32_t **, uint16_t **, uint32_t **, uint32_t **) = nullptr;
    // what the device decoder(s) left in device memory for the device packer (spl_capi.cpp), and how to give it back: one handle
    // for a whole-file decode, one per share otherwise
    struct DevShare { void *handle = nullptr; void (*free_fn)(void *) = nullptr; int share = -1; bool fetched = false; };
    std::vector<DevShare> dev_shares;
    // a decode in shares (spl_bam_share_plan): the plan, and what the shares' decoders have reported so far
    std::vector<spl_bam_share> shares;
    struct ShareResult { bool reported = false, failed = false; void *handle = nullptr; void (*free_fn)(void *) = nullptr; std::vector<int64_t> first, n, max_end; int64_t n_records = 0; };
    std::vector<ShareResult> share_results;
    bool shares_on_device = false;
    std::atomic<bool> cancel{false};       // spl_bam_cancel: whoever decodes stops at the next batch / window; nobody starts
    bool reserved = false;     // claim == 1 on behalf of a spl_bam_decode_device call that is still to come
    bool decoders_apart = false; // the device decoders run on threads of their own (spl_bam_reserve_device was called): somebody else waits for their outcome
    int claim = 0;             // 0 = nobody decodes yet (deferred open), 1 = the device decoder is at it, 2 = host worker started / arrays adopted
    uint64_t header_bytes = 0; // magic, text and reference dictionary: the first record starts here in the inflated stream
    std::string path;
    std::string decline_reason; // why the device decoder handed the file to the host threads, if it did
    ~spl_bam();
};

spl_bam::~spl_bam()
{
    if (worker.joinable()) worker.join();
    for (DevShare &d : dev_shares) if (d.handle && d.free_fn) d.free_fn(d.handle);
    for (ShareResult &r : share_results) if (r.handle && r.free_fn) r.free_fn(r.handle);
    for (auto &list : parts) for (PendingPart *p : list) delete p;
    for (void *slab : slabs) free(slab);
    if (map) munmap(map, fsize);
    if (fd >= 0) close(fd);
}

namespace {


// What a thread extracts: (reference id, reads) parts in file order.  A coordinate-sorted BAM changes reference rarely, so a
// stretch of records usually is one part.
struct Part { int32_t tid; RefReads reads; };

// The CIGAR of the record whose fixed fields start at r (block_size bs, `need` = fixed + name + cigar + seq + qual bytes):
// the record's own ops, or the real ones parked in a CG tag (htslib bam_tag2cigar).
inline const uint8_t *record_cigar(const uint8_t *r, uint32_t bs, size_t need, uint32_t l_name, uint32_t l_seq, uint32_t &n_cig)
{
    const uint8_t *cig = r + 32 + l_name;
    if (n_cig > 0 && (le32(cig) & 15u) == 4u && (le32(cig) >> 4) == l_seq) {
        uint32_t n_real = 0;
        const uint8_t *cg = find_cg_tag(r + need, r + bs, &n_real);
        if (cg && n_real >= n_cig && n_real < (1u << 29)) { cig = cg; n_cig = n_real; }
    }
    return cig;
}

// Extract the alignment records in [p, end); p MUST be a record boundary.  Stops at the last complete record and returns
// the position reached (a record boundary).  Two walks: the first checks every record and sizes the parts, the second fills
// arrays of exactly that size -- the bytes are in the caller's cache both times.
const uint8_t *extract_records(const uint8_t *p, const uint8_t *end, int n_ref, Arena &arena, std::vector<Part> &parts, int64_t &n_records,
                               std::string &err, bool &fatal)
{
    struct Run { int32_t tid; size_t n, ops; const uint8_t *begin; };
    Run few[4];
    std::vector<Run> many; // (more than four changes of reference in one stretch: an unsorted file)
    size_t n_runs = 0;
    auto run_at = [&](size_t k) -> Run & { return k < 4 ? few[k] : many[k - 4]; };
    const uint8_t *q = p;
    while (end - q >= 4) {
        const uint32_t bs = le32(q);
        if (bs < 32) { err = "corrupt record (block_size < 32)"; fatal = true; break; }
        if ((size_t)(end - q) < 4 + (size_t)bs) break;
        // the walk is a pointer chase (the next record starts where this one ends); records of one library are about the
        // same size, so the headers a few records ahead are probably where this record's size says
        __builtin_prefetch(q + 3 * (4 + (size_t)bs));
        __builtin_prefetch(q + 4 * (4 + (size_t)bs));
        __builtin_prefetch(q + 4 * (4 + (size_t)bs) + 64);
        const uint8_t *r = q + 4;
        const int32_t tid = le32s(r), pos0 = le32s(r + 4);
        const uint32_t l_name = r[8], l_seq = le32(r + 16);
        uint32_t n_cig = le16(r + 12);
        const size_t need = 32 + (size_t)l_name + 4ull * n_cig + ((size_t)l_seq + 1) / 2 + l_seq;
        if (need > bs) { err = "corrupt record (fields exceed block_size)"; fatal = true; break; }
        n_records++;
        if (tid >= 0 && tid < n_ref && pos0 >= 0) {
            (void)record_cigar(r, bs, need, l_name, l_seq, n_cig);
            if (n_runs == 0 || run_at(n_runs - 1).tid != tid) {
                const Run fresh = {tid, 0, 0, q};
                if (n_runs < 4) few[n_runs] = fresh; else many.push_back(fresh);
                ++n_runs;
            }
            Run &run = run_at(n_runs - 1);
            run.n++;
            run.ops += n_cig;
        }
        q += 4 + (size_t)bs;
    }
    const uint8_t *reached = q;
    for (size_t k = 0; k < n_runs; ++k) {
        const Run &run = run_at(k);
        if (run.ops > 0xfffffff0ull) { err = "more than 2^32 CIGAR operations in one stretch of records"; fatal = true; return reached; }
        RefReads rr;
        rr.n = run.n;
        rr.n_ops = run.ops;
        rr.pos = (int32_t *)arena.take(sizeof(int32_t) * run.n);
        rr.flag = (uint16_t *)arena.take(sizeof(uint16_t) * run.n);
        rr.cig_off = (uint32_t *)arena.take(sizeof(uint32_t) * (run.n + 1));
        rr.cigar = (uint32_t *)arena.take(sizeof(uint32_t) * std::max<size_t>(run.ops, 1));
        if (!rr.pos || !rr.flag || !rr.cig_off || !rr.cigar) { err = "out of host memory"; fatal = true; return reached; }
        rr.cig_off[0] = 0;
        size_t i = 0, at = 0;
        for (q = run.begin; i < run.n; q += 4 + (size_t)le32(q)) {
            const uint32_t bs = le32(q);
            __builtin_prefetch(q + 3 * (4 + (size_t)bs));
            const uint8_t *r = q + 4;
            const int32_t tid = le32s(r), pos0 = le32s(r + 4);
            if (tid != run.tid || pos0 < 0) continue; // (an unplaced record inside the run)
            const uint32_t l_name = r[8], l_seq = le32(r + 16);
            uint32_t n_cig = le16(r + 12);
            const size_t need = 32 + (size_t)l_name + 4ull * n_cig + ((size_t)l_seq + 1) / 2 + l_seq;
            const uint8_t *cig = record_cigar(r, bs, need, l_name, l_seq, n_cig);
            int64_t ref_len = 0;
            uint32_t *dst = rr.cigar + at;
            for (uint32_t c = 0; c < n_cig; ++c) {
                const uint32_t op = le32(cig + 4ull * c);
                dst[c] = op;
                const uint32_t code = op & 15u;
                if (code == 0 || code == 2 || code == 3 || code == 7 || code == 8) ref_len += op >> 4;
            }
            at += n_cig;
            rr.pos[i] = pos0 + 1;
            rr.flag[i] = le16(r + 14);
            rr.cig_off[++i] = (uint32_t)at;
            const int64_t e = (int64_t)pos0 + 1 + (ref_len > 0 ? ref_len : 1) - 1;
            if (e > rr.max_end) rr.max_end = e;
        }
        parts.push_back(Part{run.tid, rr});
    }
    return reached;
}

// Does a record plausibly start at c?  Every fixed field is checked against the BAM specification; the caller also
// requires the next records to chain.  A false positive only costs time: the chunk results are accepted only when
// every chunk's walk ends exactly on the next chunk's start (see parse_segment_parallel).
bool plausible_record(const uint8_t *c, const uint8_t *end, int n_ref, size_t *len_out)
{
    if (end - c < 36) return false;
    const uint32_t bs = le32(c);
    if (bs < 33 || bs > (1u << 29)) return false;
    const uint8_t *r = c + 4;
    const int32_t tid = le32s(r), pos0 = le32s(r + 4);
    const uint32_t l_name = r[8];
    const uint32_t n_cig = le16(r + 12);
    const int32_t l_seq = le32s(r + 16), next_tid = le32s(r + 20), next_pos = le32s(r + 24);
    if (tid < -1 || tid >= n_ref || next_tid < -1 || next_tid >= n_ref || pos0 < -1 || next_pos < -1 || l_seq < 0 || l_name < 1) return false;
    const size_t need = 32 + (size_t)l_name + 4ull * n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
    if (need > bs) return false;
    if ((size_t)(end - c) >= 4 + 32 + (size_t)l_name) {
        const uint8_t *name = r + 32;
        if (name[l_name - 1] != 0) return false;
        for (uint32_t i = 0; i + 1 < l_name; ++i) if (name[i] < 33 || name[i] > 126) return false;
    }
    *len_out = 4 + (size_t)bs;
    return true;
}

const uint8_t *find_record_start(const uint8_t *from, const uint8_t *end, int n_ref)
{
    for (const uint8_t *c = from; c + 36 <= end; ++c) {
        size_t len = 0;
        if (!plausible_record(c, end, n_ref, &len)) continue;
        const uint8_t *q = c + len;
        bool ok = true;
        for (int k = 0; k < 3 && ok && q + 36 <= end; ++k) { // the next three records must chain
            size_t l2 = 0;
            ok = plausible_record(q, end, n_ref, &l2);
            q += l2;
        }
        if (ok) return c;
    }
    return end;
}

// The parse threads' parts, in file order, are handed to their references (nothing is copied); what is complete by now is
// announced to the waiting consumers.
void merge_parts(spl_bam *bam, std::vector<Part> &parts)
{
    std::lock_guard<std::mutex> lock(bam->mu);
    const int before = bam->complete_upto;
    for (Part &pt : parts) {
        if (pt.reads.n == 0) continue;
        if (pt.tid < bam->max_tid_seen) bam->out_of_order = true;
        if (pt.tid > bam->max_tid_seen) bam->max_tid_seen = pt.tid;
        std::vector<PendingPart *> &list = bam->parts[(size_t)pt.tid];
        bam->ref_reads[(size_t)pt.tid] += (int64_t)pt.reads.n;
        if (pt.reads.max_end > bam->ref_max_end[(size_t)pt.tid]) bam->ref_max_end[(size_t)pt.tid] = pt.reads.max_end;
        PendingPart *pp = new PendingPart();
        pp->tid = pt.tid;
        pp->reads = pt.reads;
        list.push_back(pp);
    }
    if (bam->max_tid_seen > bam->complete_upto) bam->complete_upto = bam->max_tid_seen;
    if (bam->complete_upto != before) bam->cv.notify_all();
}

// BAM-native arrays of one reference (spl_bam_reads): one exact-size allocation per array, copied from the parts.
bool assemble_ref(spl_bam *bam, int tid, std::string &err)
{
    RefFinal &dst = bam->refs_storage[(size_t)tid];
    const std::vector<PendingPart *> &parts = bam->parts[(size_t)tid];
    int64_t n = 0, g = 0;
    for (const PendingPart *pt : parts) { n += (int64_t)pt->reads.n; g += (int64_t)pt->reads.n_ops; }
    if (g > 0xfffffff0LL) { err = "more than 2^32 CIGAR operations on one reference"; return false; }
    dst.n = n;
    dst.n_cigar = g;
    dst.max_end = bam->ref_max_end[(size_t)tid];
    dst.pos = (int32_t *)big_alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(n, 1));
    dst.flag = (uint16_t *)big_alloc(sizeof(uint16_t) * (size_t)std::max<int64_t>(n, 1));
    dst.cig_off = (uint32_t *)big_alloc(sizeof(uint32_t) * (size_t)(n + 1));
    dst.cigar = (uint32_t *)big_alloc(sizeof(uint32_t) * (size_t)std::max<int64_t>(g, 1));
    if (!dst.pos || !dst.flag || !dst.cig_off || !dst.cigar) { err = "out of host memory"; return false; }
    dst.cig_off[0] = 0;
    std::vector<int64_t> read_at(parts.size() + 1, 0), op_at(parts.size() + 1, 0);
    for (size_t i = 0; i < parts.size(); ++i) {
        read_at[i + 1] = read_at[i] + (int64_t)parts[i]->reads.n;
        op_at[i + 1] = op_at[i] + (int64_t)parts[i]->reads.n_ops;
    }
    std::atomic<size_t> next(0);
    auto work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= parts.size()) break;
            const RefReads &src = parts[i]->reads;
            const size_t k = src.n;
            if (!k) continue;
            memcpy(dst.pos + read_at[i], src.pos, sizeof(int32_t) * k);
            memcpy(dst.flag + read_at[i], src.flag, sizeof(uint16_t) * k);
            const uint32_t base = (uint32_t)op_at[i];
            uint32_t *off = dst.cig_off + read_at[i]; // entry j + 1 = end of read j
            const uint32_t c0 = src.cig_off[0]; // (0 for the host decoder's parts; a file-wide offset for adopted arrays)
            for (size_t j = 1; j <= k; ++j) off[j] = base + (src.cig_off[j] - c0);
            if (src.n_ops) memcpy(dst.cigar + op_at[i], src.cigar + c0, sizeof(uint32_t) * src.n_ops);
        }
    };
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::min(bam->n_threads, 16), parts.size()));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
    return true;
}

// The decode proper, on the file's own thread.  The blocks are dealt out in BATCHES of a few consecutive blocks (two
// megabytes inflated); a worker thread inflates a batch into a buffer of its own and extracts the records right there, while the
// bytes are still in its cache -- the inflated stream is never written out to memory and read back, and every thread does both
// halves of the work.  Where the first record of a batch starts is not known when the batch is taken (records straddle blocks and
// batches freely): the worker GUESSES the first boundary (a plausible record header whose successors chain, find_record_start)
// and keeps a copy of the bytes in front of it (`head`) and of the incomplete record at the end (`tail`).  This thread commits
// the batches in file order and accepts a batch only if the bytes carried over from its predecessors plus its head are whole
// records, i.e. if a sequential walk from the last known boundary arrives exactly at the guessed start -- by induction from the
// end of the header every accepted start is a true boundary, whatever the guess was based on.  A batch that fails the test is
// inflated again and walked sequentially from the known boundary (never seen on a well-formed file; the tests force it).
struct BatchOut {
    std::vector<Part> parts;
    std::vector<uint8_t> head, tail;
    size_t len = 0, start = 0, i0 = 0, i1 = 0; // bytes inflated, guessed first boundary; the batch's blocks
    uint64_t u0 = 0;                            // offset of the batch in the inflated stream
    int64_t nrec = 0;
    bool inflate_bad = false, parse_bad = false;
    bool known = false;                         // the first boundary is the end of the BAM header, not a guess
    bool skip = false;                          // nothing but BAM header in it
    std::atomic<int> state{0}; // 0 = the worker's, 1 = ready to commit
};

void decode_worker(spl_bam *bam)
{
    const uint8_t *file = (const uint8_t *)bam->map;
    BlockDir &dir = bam->dir;
    const int n_ref = bam->n_refs;
    const NodeCpus node; // the NUMA node this thread runs on (the opening thread's, inherited): all worker threads stay there
    auto env_num = [](const char *name, long dflt) { const char *e = getenv(name); const long v = e ? atol(e) : 0; return v > 0 ? v : dflt; };
    const size_t BATCH = (size_t)env_num("SPL_BAM_BATCH_BLOCKS", 32);
    const bool force_slow = getenv("SPL_BAM_FORCE_RESYNC") != nullptr; // (tests: every batch takes the sequential path)
    const bool timing = getenv("SPL_BAM_TIMING") != nullptr;
    const uint64_t H = bam->header_bytes; // the first record starts here
    std::string fail;
    // (a batch of real data is a few hundred kilobytes of file: no more workers than the file can have batches)
    const int n_workers = (int)std::max<size_t>(1, std::min<size_t>((size_t)bam->n_threads, bam->fsize / 65536 + 1));
    const size_t W = 4 * (size_t)n_workers; // batches in flight: finished but not yet committed, or being worked on
    std::vector<BatchOut> ring(W);
    // No lock anywhere on the batches' way: a slot is handed over by its state word, room in the ring is the frontier counter.
    // (Condition variables were tried first: with 64 workers every commit woke the lot of them, 6 of 9 seconds went into that.)
    std::atomic<size_t> frontier(0); // the next batch to commit; slot b % W is free for batch b once frontier > b - W
    std::atomic<bool> stop(false);
    std::atomic<size_t> next(0);
    auto nap = [](int &spins) { // wait a little: busy first (the other side is usually microseconds away), then off the core
        if (++spins < 2000) { __builtin_ia32_pause(); return; }
        std::this_thread::sleep_for(std::chrono::microseconds(spins < 4000 ? 20 : 200));
    };
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();

    double sum_inflate = 0, sum_extract = 0; // (SPL_BAM_TIMING: thread-seconds)
    auto keep_slabs = [&](Arena &a) { // (the file owns what its parts point into)
        std::lock_guard<std::mutex> lock(bam->mu);
        bam->slabs.insert(bam->slabs.end(), a.slabs.begin(), a.slabs.end());
        a.slabs.clear();
    };
    // Blocks [i0, i1) of batch b, once the directory has them; false = there is no batch b (the file has ended, or its walk failed).
    auto batch_blocks = [&](size_t b, size_t &i0, size_t &i1) {
        i0 = b * BATCH;
        for (int spins = 0;;) {
            const int st = dir.state.load(std::memory_order_acquire);
            const size_t n = dir.n_ready.load(std::memory_order_acquire);
            if (n >= i0 + BATCH) { i1 = i0 + BATCH; return true; }
            if (st != 0) { i1 = n; return n > i0; }
            if (stop.load(std::memory_order_acquire)) return false;
            nap(spins);
        }
    };
    auto inflate_batch = [&](size_t i0, size_t i1, uint8_t *buf, void *ld) {
        size_t at = 0;
        bool ok = true;
        for (size_t i = i0; i < i1; ++i) {
            const Block &blk = dir.at(i);
            ok = inflate_block(file, blk, buf + at, ld) && ok;
            at += blk.isize;
        }
        return ok;
    };
    auto work = [&]() {
        void *ld = deflate_lib().ok ? deflate_lib().alloc() : nullptr;
        std::vector<uint8_t> store(BATCH * 65536 + 64);
        uint8_t *buf = store.data();
        Arena arena;
        double my_inflate = 0, my_extract = 0;
        for (;;) {
            const size_t b = next.fetch_add(1);
            size_t i0, i1;
            if (!batch_blocks(b, i0, i1)) break;
            for (int spins = 0; !stop.load(std::memory_order_acquire) && b >= frontier.load(std::memory_order_acquire) + W;) nap(spins);
            if (stop.load(std::memory_order_acquire)) break;
            BatchOut &o = ring[b % W];
            o.i0 = i0;
            o.i1 = i1;
            o.u0 = dir.at(i0).uoff;
            const uint64_t u1 = dir.at(i1 - 1).uoff + dir.at(i1 - 1).isize;
            o.len = (size_t)(u1 - o.u0);
            o.start = o.len;
            o.known = false;
            o.skip = u1 <= H && u1 > o.u0;   // BAM header bytes only
            o.inflate_bad = o.parse_bad = false;
            if (o.skip) { o.state.store(1, std::memory_order_release); continue; }
            const double w0 = timing ? now() : 0.0;
            o.inflate_bad = !inflate_batch(i0, i1, buf, ld);
            const double w1 = timing ? now() : 0.0;
            size_t reached = o.len;
            if (!o.inflate_bad) {
                const uint8_t *end = buf + o.len;
                o.known = o.u0 <= H; // the batch the header ends in: the first record starts right there
                const uint8_t *p = o.known ? buf + (size_t)(H - o.u0) : find_record_start(buf, end, n_ref);
                o.start = (size_t)(p - buf);
                std::string err;
                bool fatal = false;
                reached = (size_t)(extract_records(p, end, n_ref, arena, o.parts, o.nrec, err, fatal) - buf);
                o.parse_bad = fatal;
                if (!o.known) o.head.assign((const uint8_t *)buf, (const uint8_t *)buf + o.start);
                o.tail.assign((const uint8_t *)buf + reached, end);
            }
            o.state.store(1, std::memory_order_release);
            if (timing) { const double w2 = now(); my_inflate += w1 - w0; my_extract += w2 - w1; }
        }
        if (timing) { std::lock_guard<std::mutex> lock(bam->mu); sum_inflate += my_inflate; sum_extract += my_extract; }
        keep_slabs(arena);
        if (ld) deflate_lib().free_(ld);
    };
    // the rest of the block directory, beside the decode
    std::thread walker([&]() { node.pin_this_thread(); walk_blocks(dir, file, bam->fsize, bam->path.c_str(), 0); });
    std::vector<std::thread> pool;
    for (int t = 0; t < n_workers; ++t) pool.emplace_back([&]() { node.pin_this_thread(); work(); });

    // ---- commit, in file order
    std::vector<uint8_t> carry, joined, again; // carry: the bytes between the last committed record and the frontier batch
    void *ld_mine = nullptr;
    Arena arena_mine;
    double t_wait = 0, t_bridge = 0, t_merge = 0, t_release = 0;
    size_t n_resync = 0;
    auto walk = [&](const uint8_t *p0, const uint8_t *p1, bool &fatal) { // sequential, authoritative: commits what it parses
        std::vector<Part> seq;
        int64_t n = 0;
        const uint8_t *r = extract_records(p0, p1, n_ref, arena_mine, seq, n, fail, fatal);
        merge_parts(bam, seq);
        bam->n_records += n;
        return r;
    };
    size_t n_batches = 0;
    for (size_t b = 0; fail.empty(); ++b) {
        if (bam->cancel.load(std::memory_order_acquire)) { fail = "the file was closed while it was being decoded"; break; }
        BatchOut &o = ring[b % W];
        if (o.state.load(std::memory_order_acquire) != 1) {
            const double w0 = now();
            bool ended = false;
            for (int spins = 0; o.state.load(std::memory_order_acquire) != 1 && !ended;) {
                // (no batch b: the directory is final and ends before it -- a worker that takes b finds the same and leaves)
                ended = dir.state.load(std::memory_order_acquire) != 0 && dir.n_ready.load(std::memory_order_acquire) <= b * BATCH;
                if (!ended) nap(spins);
            }
            t_wait += now() - w0;
            if (ended) break;
        }
        n_batches = b + 1;
        if (o.skip) {
            o.state.store(0, std::memory_order_relaxed);
            frontier.store(b + 1, std::memory_order_release);
            continue;
        }
        const double c0 = now();
        if (o.inflate_bad) { fail = "inflate or CRC32 failure in a BGZF block (corrupt file)"; break; }
        bool accept = !o.parse_bad && !force_slow;
        bool fatal = false, handled = false;
        if (accept && (!carry.empty() || !o.head.empty())) {
            joined.assign(carry.begin(), carry.end());
            joined.insert(joined.end(), o.head.begin(), o.head.end());
            // (peek first: nothing may be committed from a walk that turns out not to arrive at the guessed start)
            const uint8_t *q = joined.data(), *qe = joined.data() + joined.size();
            while (qe - q >= 4) {
                const uint32_t bs = le32(q);
                if (bs < 32 || (size_t)(qe - q) < 4 + (size_t)bs) break;
                q += 4 + (size_t)bs;
            }
            if (q == qe) {
                const uint8_t *r = walk(joined.data(), qe, fatal);
                if (fatal) break;
                accept = (r == qe);
                carry.clear();
            } else if (o.start == o.len) { // no record starts in this batch: one record spans it (or the file ends inside one)
                const uint8_t *r = walk(joined.data(), q, fatal);
                if (fatal) break;
                carry.assign(r, qe);
                handled = true;
            } else {
                accept = false;
            }
        }
        const double c1 = now();
        t_bridge += c1 - c0;
        if (handled) {
        } else if (accept) {
            merge_parts(bam, o.parts);
            bam->n_records += o.nrec;
            carry.swap(o.tail);
        } else { // the guess did not hold: this batch again, sequentially, from the known boundary
            ++n_resync;
            if (!ld_mine && deflate_lib().ok) ld_mine = deflate_lib().alloc();
            again.resize(carry.size() + o.len + 64);
            if (!carry.empty()) memcpy(again.data(), carry.data(), carry.size());
            if (!inflate_batch(o.i0, o.i1, again.data() + carry.size(), ld_mine)) { fail = "inflate or CRC32 failure in a BGZF block (corrupt file)"; break; }
            const uint8_t *p0 = again.data(), *p1 = again.data() + carry.size() + o.len;
            if (o.known) p0 += (size_t)(H - o.u0);
            const uint8_t *r = walk(p0, p1, fatal);
            if (fatal) break;
            carry.assign(r, p1);
        }
        o.parts.clear();
        o.head.clear();
        o.tail.clear();
        o.nrec = 0;
        const double c2 = now();
        t_merge += c2 - c1;
        o.state.store(0, std::memory_order_relaxed);
        frontier.store(b + 1, std::memory_order_release);
        t_release += now() - c2;
    }
    stop.store(true, std::memory_order_release);
    for (auto &t : pool) t.join();
    walker.join();
    if (dir.state.load() < 0) { // the walk's own words (and error class) for what is wrong with the file
        std::lock_guard<std::mutex> lock(bam->mu);
        bam->error = dir.error;
        bam->err_code = dir.err_code;
        fail.clear();
    } else if (fail.empty() && H > dir.uoff) {
        fail = "no BAM header found";
    }
    keep_slabs(arena_mine);
    if (ld_mine) deflate_lib().free_(ld_mine);
    if (timing)
        fprintf(stderr, "[spl_bam_open] %zu blocks in %zu batches, %d threads: %.3f s, the committing thread waited %.3f s of that (straddling records %.3f s, hand-over %.3f s, "
                "ring %.3f s), %zu batches re-walked; workers: inflate + CRC %.2f thread-s, records %.2f thread-s\n", dir.n_ready.load(), n_batches, n_workers,
                now() - t_begin, t_wait, t_bridge, t_merge, t_release, n_resync, sum_inflate, sum_extract);
    if (fail.empty() && !carry.empty() && dir.state.load() > 0) fail = "file ends inside a record (truncated)";
    {
        std::lock_guard<std::mutex> lock(bam->mu);
        if (!fail.empty() && !bam->err_code) { bam->error = bam->path + ": " + fail; bam->err_code = SPL_ERR_FORMAT; }
        bam->complete_upto = bam->n_refs;
        bam->done = true;
    }
    bam->cv.notify_all();
}

// The BAM header (magic, text, reference dictionary) from the first blocks of the file, inflated one by one until it is whole.
int read_header(spl_bam *bam, std::string &fail, int &code)
{
    code = SPL_ERR_FORMAT;
    const uint8_t *file = (const uint8_t *)bam->map;
    std::vector<uint8_t> head;
    void *ld = deflate_lib().ok ? deflate_lib().alloc() : nullptr;
    int rc = 1; // 1 = need more, 0 = done, -1 = failed
    for (size_t i = 0; rc == 1; ++i) {
        if (i >= bam->dir.n_ready.load() && bam->dir.state.load() == 0) walk_blocks(bam->dir, file, bam->fsize, bam->path.c_str(), 1);
        if (bam->dir.state.load() < 0) { fail = bam->dir.error; code = bam->dir.err_code; rc = -1; break; }
        if (i >= bam->dir.n_ready.load()) break;
        const Block &b = bam->dir.at(i);
        const size_t at = head.size();
        head.resize(at + b.isize);
        if (!inflate_block(file, b, head.data() + at, ld)) { fail = "inflate or CRC32 failure in a BGZF block (corrupt file)"; rc = -1; break; }
        const uint8_t *p = head.data(), *end = head.data() + head.size();
        if (end - p < 12) continue;
        if (memcmp(p, "BAM\1", 4) != 0) { fail = "not a BAM file (bad magic)"; rc = -1; break; }
        const uint32_t l_text = le32(p + 4);
        if ((size_t)(end - p) < 12 + (size_t)l_text) continue;
        const uint8_t *q = p + 8 + l_text;
        const int32_t n_ref = le32s(q);
        q += 4;
        if (n_ref < 0) { fail = "negative n_ref"; rc = -1; break; }
        std::vector<std::string> names;
        std::vector<int64_t> lens;
        bool whole = true;
        for (int32_t k = 0; k < n_ref; ++k) {
            if (end - q < 4) { whole = false; break; }
            const uint32_t l_name = le32(q);
            if ((size_t)(end - q) < 4 + (size_t)l_name + 4) { whole = false; break; }
            names.emplace_back((const char *)q + 4, l_name ? l_name - 1 : 0);
            lens.push_back(le32(q + 4 + l_name));
            q += 4 + l_name + 4;
        }
        if (!whole) continue;
        bam->ref_names.swap(names);
        bam->ref_lengths.swap(lens);
        bam->refs_storage = std::vector<RefFinal>((size_t)n_ref);
        bam->assembled.assign((size_t)n_ref, 0);
        bam->parts.assign((size_t)n_ref, std::vector<PendingPart *>());
        bam->ref_max_end.assign((size_t)n_ref, 0);
        bam->ref_reads.assign((size_t)n_ref, 0);
        bam->n_refs = n_ref;
        bam->header_bytes = (uint64_t)(q - p);
        rc = 0;
    }
    if (ld) deflate_lib().free_(ld);
    if (rc == 1) fail = "no BAM header found";
    return rc == 0 ? 0 : -1;
}

} // namespace

// Opens the file, reads the block directory and the header, and starts the decode on a thread of its own.
static int open_file(const char *path, int n_threads, bool start_now, spl_bam **out)
{
    if (!path || !out) return spl_set_error(SPL_ERR_ARG, "spl_bam_open: null argument");
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return spl_set_error(SPL_ERR_IO, "cannot open %s", path);
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) { close(fd); return spl_set_error(SPL_ERR_IO, "cannot stat %s (or empty file)", path); }
    const size_t fsize = (size_t)st.st_size;
    void *map = mmap(nullptr, fsize, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) { close(fd); return spl_set_error(SPL_ERR_IO, "mmap failed for %s", path); }
    struct FdGuard { int fd; ~FdGuard() { if (fd >= 0) close(fd); } } fd_guard{fd}; // (handed to the spl_bam below, or closed on the way out)
    madvise(map, fsize, MADV_SEQUENTIAL);
    const uint8_t *file = (const uint8_t *)map;

    // The end of the file first: a BGZF file ends with an empty block (htslib's 28-byte EOF marker).
    static const uint8_t eof_marker[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bool bgzf = fsize >= 18 && file[0] == 0x1f && file[1] == 0x8b && file[2] == 8 && (file[3] & 4);
    if (bgzf && (fsize < 28 || memcmp(file + fsize - 28, eof_marker, 28) != 0)) {
        // Not the standard marker: a truncated file, or (legal, unusual) an empty last block written with other header fields.
        // The caller must not be handed results of a truncated file chromosome by chromosome before anybody notices: walk
        // all blocks here and now -- the one case that pays for the walk up front.
        BlockDir probe;
        probe.chunks.assign(fsize / 28 / BlockDir::CHUNK + 2, nullptr);
        walk_blocks(probe, file, fsize, path, 0);
        if (probe.state.load() < 0) {
            const int code = probe.err_code;
            const std::string text = probe.error;
            munmap(map, fsize);
            return spl_set_error(code, "%s", text.c_str());
        }
    }
    spl_bam *bam = new (std::nothrow) spl_bam();
    if (!bam) { munmap(map, fsize); return spl_set_error(SPL_ERR_NOMEM, "out of host memory"); }
    bam->map = map;
    bam->fsize = fsize;
    bam->path = path;
    bam->fd = fd_guard.fd;
    fd_guard.fd = -1;
    bam->dir.chunks.assign(fsize / 28 / BlockDir::CHUNK + 2, nullptr); // (a block is at least 28 bytes)
    if (n_threads <= 0) { // default: all hardware threads up to SPL_BAM_THREADS (32 unless the environment says otherwise)
        const char *e = getenv("SPL_BAM_THREADS");
        const int cap = e && atoi(e) > 0 ? atoi(e) : 32;
        n_threads = (int)std::thread::hardware_concurrency();
        if (n_threads > cap) n_threads = cap;
    }
    bam->n_threads = n_threads > 0 ? n_threads : 1;
    std::string fail;
    int code = SPL_ERR_FORMAT;
    if (read_header(bam, fail, code) != 0) {
        delete bam;
        if (fail.compare(0, strlen(path), path) == 0) return spl_set_error(code, "%s", fail.c_str()); // (the walk's message names the file already)
        return spl_set_error(code, "%s: %s", path, fail.c_str());
    }
    if (start_now) { bam->claim = 2; bam->worker = std::thread(decode_worker, bam); }
    *out = bam;
    return SPL_OK;
}

extern "C" int spl_bam_open_stream(const char *path, int n_threads, spl_bam **out) { return open_file(path, n_threads, true, out); }
extern "C" int spl_bam_open_deferred(const char *path, int n_threads, spl_bam **out) { return open_file(path, n_threads, false, out); }

// (call with bam->mu held) nobody will decode a file that is being closed: whoever waits is told so
static bool cancelled_locked(spl_bam *bam)
{
    if (!bam->cancel.load(std::memory_order_acquire)) return false;
    if (!bam->done) {
        bam->claim = 2;
        bam->err_code = SPL_ERR_IO;
        bam->error = bam->path + ": closed before it was decoded";
        bam->done = true;
        bam->cv.notify_all();
    }
    return true;
}

// Decode on the host's threads unless somebody is decoding already (waits on a deferred file come through here).
int spl_bam_start_host(spl_bam *bam)
{
    if (!bam) return spl_set_error(SPL_ERR_ARG, "spl_bam_start_host: null argument");
    std::lock_guard<std::mutex> lock(bam->mu);
    if (bam->claim == 0 && !cancelled_locked(bam)) { bam->claim = 2; bam->worker = std::thread(decode_worker, bam); }
    bam->cv.notify_all(); // (spl_bam_wait_device)
    return SPL_OK;
}

// The device decoder takes the file (false: somebody else has it) / gives it to the host threads after all.
bool spl_bam_claim_for_device(spl_bam *bam)
{
    std::lock_guard<std::mutex> lock(bam->mu);
    if (bam->claim == 1 && bam->reserved) { bam->reserved = false; return true; } // (reserved earlier for exactly this call)
    if (bam->claim != 0) return false;
    bam->claim = 1;
    return true;
}

extern "C" int spl_bam_reserve_device(spl_bam *bam)
{
    if (!bam) return spl_set_error(SPL_ERR_ARG, "spl_bam_reserve_device: null argument");
    std::lock_guard<std::mutex> lock(bam->mu);
    if (bam->claim != 0) return spl_set_error(SPL_ERR_ARG, "spl_bam_reserve_device: the file is being decoded already");
    bam->claim = 1;
    bam->reserved = true;
    bam->decoders_apart = true;
    return SPL_OK;
}
// Why the device decoder did not take the file (spl_capi.cpp's to_host): kept for whoever tells the user (process.py logs it once)
void spl_bam_note_decline(spl_bam *bam, const char *why)
{
    std::lock_guard<std::mutex> lock(bam->mu);
    if (bam->decline_reason.empty()) bam->decline_reason = why ? why : "";
}
extern "C" const char *spl_bam_decline_reason(spl_bam *bam)
{
    if (!bam) return "";
    std::lock_guard<std::mutex> lock(bam->mu);
    return bam->decline_reason.c_str(); // (set at most once, never changed: the pointer stays good while the file is open)
}
int spl_bam_device_gives_up(spl_bam *bam)
{
    std::lock_guard<std::mutex> lock(bam->mu);
    if (bam->claim == 1 && !cancelled_locked(bam)) { bam->claim = 2; bam->worker = std::thread(decode_worker, bam); }
    bam->cv.notify_all(); // (spl_bam_wait_device)
    return SPL_OK;
}

// Whoever started device decoders on the file (spl_bam_decode_device / _share on threads of the caller) and wants to know how
// that went: waits until the file is no longer theirs to decide about -- its reads are on the device(s) and every reference is
// complete (1), or the host threads have the file (0: not sorted by reference, a CIGAR in a CG tag, no device memory ...; they
// may still be at it, spl_bam_wait_ref / _all wait for them).  It does NOT wait for the decoders' calls to return: what they
// give back on the way out (buffers, streams, events: 10 ms for a large file) is nobody's business but theirs.
extern "C" int spl_bam_wait_device(spl_bam *bam, int *on_device_out)
{
    if (!bam || !on_device_out) return spl_set_error(SPL_ERR_ARG, "spl_bam_wait_device: null argument");
    std::unique_lock<std::mutex> lock(bam->mu);
    bam->cv.wait(lock, [&]() { return bam->claim != 1 || bam->done; });
    *on_device_out = (bam->shares_on_device || (bam->done && !bam->dev_shares.empty())) ? 1 : 0;
    return SPL_OK;
}

// A device decoder that has told the waiters and has only its clearing up left: until the file is being closed (spl_bam_cancel), `seconds` at most.
void spl_bam_linger(spl_bam *bam, double seconds)
{
    std::unique_lock<std::mutex> lock(bam->mu);
    if (!bam->decoders_apart) return; // (the decoder's caller is the one who goes on with the reads: nobody to stand aside for)
    (void)bam->cv.wait_for(lock, std::chrono::duration<double>(seconds), [&]() { return bam->cancel.load(std::memory_order_acquire); });
}

// The file is about to be closed: a decode in progress stops at its next batch (host threads) or window (device), a decode that
// has not begun never does.  Waiting calls return with an error.  (spl_bam_close alone waits for the decode to END: minutes on a
// large file when the caller only wants to leave, because something else failed.)
extern "C" void spl_bam_cancel(spl_bam *bam)
{
    if (!bam) return;
    bam->cancel.store(true, std::memory_order_release);
    std::lock_guard<std::mutex> lock(bam->mu);
    if (bam->claim == 0) (void)cancelled_locked(bam);
    bam->cv.notify_all();
}
bool spl_bam_cancelled(const spl_bam *bam) { return bam->cancel.load(std::memory_order_acquire); }

// Decode on the host's threads (public face of spl_bam_start_host: a deferred file whose caller has made up his mind).  A
// reservation nobody has taken up (spl_bam_reserve_device, and then no spl_bam_decode_device: its caller failed on the way
// there) ends here -- the file must not be left waiting for a decoder that will not come.
extern "C" int spl_bam_start(spl_bam *bam)
{
    if (!bam) return spl_set_error(SPL_ERR_ARG, "spl_bam_start: null argument");
    {
        std::lock_guard<std::mutex> lock(bam->mu);
        if (bam->claim == 1 && bam->reserved) { bam->reserved = false; bam->claim = 0; }
    }
    return spl_bam_start_host(bam);
}

// How well the file's first blocks behind the BAM header are compressed (inflated bytes per file byte, over up to 256 blocks):
// what decides whether the host's inflate or the GPU's is the faster one for this file.  Deferred files only (nobody else may be
// walking the block directory).  0 = could not tell (no record blocks among the first, or the directory is not walkable).
extern "C" int spl_bam_compression_ratio(spl_bam *bam, double *ratio_out)
{
    if (!bam || !ratio_out) return spl_set_error(SPL_ERR_ARG, "spl_bam_compression_ratio: null argument");
    *ratio_out = 0.0;
    {
        std::lock_guard<std::mutex> lock(bam->mu);
        if (bam->claim != 0) return SPL_OK;
    }
    const size_t want = 256;
    if (bam->dir.state.load() == 0 && bam->dir.n_ready.load() < want)
        walk_blocks(bam->dir, (const uint8_t *)bam->map, bam->fsize, bam->path.c_str(), want - bam->dir.n_ready.load());
    if (bam->dir.state.load() < 0) return SPL_OK; // (the decoder will report what is wrong with the file)
    uint64_t in = 0, out = 0;
    const size_t n = std::min(want, bam->dir.n_ready.load());
    for (size_t i = 0; i < n; ++i) {
        const Block &b = bam->dir.at(i);
        if (b.uoff < bam->header_bytes || b.isize == 0) continue;
        in += b.csize;
        out += b.isize;
    }
    if (in) *ratio_out = (double)out / (double)in;
    return SPL_OK;
}

int spl_bam_walk_all(spl_bam *bam)
{
    if (bam->dir.state.load() == 0 && !getenv("SPL_WALK_ONE_THREAD")) (void)walk_rest_in_parallel(bam->dir, (const uint8_t *)bam->map, bam->fsize, bam->n_threads);
    if (bam->dir.state.load() == 0) walk_blocks(bam->dir, (const uint8_t *)bam->map, bam->fsize, bam->path.c_str(), 0);
    if (bam->dir.state.load() < 0) return spl_set_error(bam->dir.err_code, "%s", bam->dir.error.c_str());
    return SPL_OK;
}
// The directory of the file's first `bytes` (by several threads when that is worth it): for a caller who wants to begin with the
// first blocks while spl_bam_walk_all does the rest.  Whatever goes wrong here is spl_bam_walk_all's to find and to say.
void spl_bam_walk_some(spl_bam *bam, size_t bytes)
{
    if (bam->dir.state.load() != 0 || getenv("SPL_WALK_ONE_THREAD")) return;
    (void)walk_rest_in_parallel(bam->dir, (const uint8_t *)bam->map, bam->fsize, bam->n_threads, std::min(bam->fsize, bam->dir.off + bytes));
}
bool spl_bam_walk_complete(const spl_bam *bam) { return bam->dir.state.load() == 1; }
size_t spl_bam_block_count(const spl_bam *bam) { return bam->dir.n_ready.load(); }
void spl_bam_block_get(const spl_bam *bam, size_t i, spl_bam_block_info *out)
{
    const Block &b = bam->dir.at(i);
    out->data_off = b.coff + 12 + b.xlen;
    out->data_len = b.csize - 12 - b.xlen - 8;
    out->uoff = b.uoff;
    out->isize = b.isize;
    out->crc = b.crc;
}
void spl_bam_set_device_reads(spl_bam *bam, void *handle, void (*free_fn)(void *))
{
    std::lock_guard<std::mutex> lock(bam->mu);
    for (spl_bam::DevShare &d : bam->dev_shares) if (d.handle && d.free_fn) d.free_fn(d.handle);
    bam->dev_shares.clear();
    if (handle) { spl_bam::DevShare d; d.handle = handle; d.free_fn = free_fn; d.share = -1; bam->dev_shares.push_back(d); }
}
void *spl_bam_device_reads(spl_bam *bam, int tid)
{
    std::lock_guard<std::mutex> lock(bam->mu);
    if (tid < 0 || tid >= bam->n_refs || bam->dev_shares.empty()) return nullptr;
    if (bam->parts[(size_t)tid].empty()) return bam->dev_shares[0].handle; // (no reads: any handle says so)
    const int share = bam->parts[(size_t)tid][0]->share;
    for (const PendingPart *pp : bam->parts[(size_t)tid]) if (pp->share != share) return nullptr; // (in several shares: no single handle)
    for (const spl_bam::DevShare &d : bam->dev_shares) if (d.share == share) return d.handle;
    return nullptr;
}
void *spl_bam_share_reads(spl_bam *bam, int share)
{
    std::lock_guard<std::mutex> lock(bam->mu);
    for (const spl_bam::DevShare &d : bam->dev_shares) if (d.share == share) return d.handle;
    return nullptr;
}
const uint8_t *spl_bam_image(const spl_bam *bam, size_t *fsize_out) { if (fsize_out) *fsize_out = bam->fsize; return (const uint8_t *)bam->map; }
int spl_bam_fd(const spl_bam *bam) { return bam->fd; }
uint64_t spl_bam_header_end(const spl_bam *bam) { return bam->header_bytes; }
int spl_bam_thread_count(const spl_bam *bam) { return bam->n_threads; }

int spl_bam_adopt(spl_bam *bam, int32_t *pos, uint16_t *flag, uint32_t *cig_off, uint32_t *cigar, const int64_t *ref_first, const int64_t *ref_n,
                  const int64_t *ref_max_end, int64_t n_records_total)
{
    std::lock_guard<std::mutex> lock(bam->mu);
    if (bam->claim != 1) return spl_set_error(SPL_ERR_ARG, "spl_bam_adopt: the file is not claimed by the device decoder");
    bam->claim = 2;
    const bool have = pos != nullptr; // (nullptr: the arrays stay on the device until somebody asks, spl_bam_set_fetch)
    if (have) { bam->slabs.push_back(pos); bam->slabs.push_back(flag); bam->slabs.push_back(cig_off); bam->slabs.push_back(cigar); }
    bam->lazy = !have;
    for (int t = 0; t < bam->n_refs; ++t) {
        if (ref_n[t] <= 0) continue;
        PendingPart *pp = new PendingPart();
        pp->tid = t;
        pp->first = ref_first[t];
        RefReads &r = pp->reads;
        r.n = (size_t)ref_n[t];
        if (have) {
            r.pos = pos + ref_first[t];
            r.flag = flag + ref_first[t];
            r.cig_off = cig_off + ref_first[t]; // (offsets into the file-wide op array: cig_off[0] of a part need not be 0)
            r.cigar = cigar;
            r.n_ops = (size_t)(r.cig_off[r.n] - r.cig_off[0]);
        }
        r.max_end = ref_max_end[t];
        bam->parts[(size_t)t].push_back(pp);
        bam->ref_reads[(size_t)t] = ref_n[t];
        bam->ref_max_end[(size_t)t] = ref_max_end[t];
    }
    bam->n_records = n_records_total;
    bam->max_tid_seen = bam->n_refs - 1;
    bam->complete_upto = bam->n_refs;
    bam->done = true;
    bam->cv.notify_all();
    return SPL_OK;
}

// ---- a decode in shares ------------------------------------------------------------------------------------------------
namespace {
// The first record that begins at or after the beginning of block b: where in the inflated stream (*u_out) and of which
// reference (*tid_out; records without one: n_refs) -- by inflating the block and a few behind it.  false: cannot tell.
// (A block's first bytes are the tail of the record before: the boundary is a GUESS, find_record_start's -- a plausible record
// whose successors chain -- and whoever decodes the stretch in front of it must arrive exactly there, or the plan is dropped.)
bool first_record_at(spl_bam *bam, size_t b, void *ld, std::vector<uint8_t> &buf, uint64_t *u_out, int32_t *tid_out)
{
    const uint8_t *file = (const uint8_t *)bam->map;
    const size_t n_blocks = bam->dir.n_ready.load();
    if (b >= n_blocks) return false;
    const uint64_t u0 = bam->dir.at(b).uoff;
    // enough bytes for a record that begins near the block's end and the three behind it that must chain: 6 blocks, more while
    // nothing has been found (a record larger than a block)
    for (size_t want = 6; want <= 96; want *= 4) {
        size_t len = 0, nb = 0;
        for (size_t k = b; k < n_blocks && nb < want; ++k, ++nb) len += bam->dir.at(k).isize;
        buf.resize(len + 64);
        size_t at = 0;
        for (size_t k = b; k < b + nb; ++k) {
            if (!inflate_block(file, bam->dir.at(k), buf.data() + at, ld)) return false;
            at += bam->dir.at(k).isize;
        }
        const uint8_t *const end = buf.data() + len;
        const uint8_t *c = u0 <= bam->header_bytes ? buf.data() + (size_t)(bam->header_bytes - u0) : find_record_start(buf.data(), end, bam->n_refs);
        if (u0 <= bam->header_bytes && (size_t)(bam->header_bytes - u0) > len) return false;
        if (c + 36 <= end) {
            const int32_t tid = le32s(c + 4);
            *u_out = u0 + (uint64_t)(c - buf.data());
            *tid_out = tid < 0 || tid >= bam->n_refs ? bam->n_refs : tid;
            return true;
        }
        if (b + nb >= n_blocks) { // nothing begins behind here: the end of the stream is the boundary
            *u_out = u0 + len;
            *tid_out = bam->n_refs;
            return true;
        }
    }
    return false;
}
} // namespace

// What the records of blocks [b_lo, b_hi) take, from three places in them (the stretch's beginning, a third and two thirds of
// the way; the block directory must hold them): records, their CIGAR ops, and the inflated bytes both were counted in -- the
// host inflates half a dozen blocks at each place and walks the whole records in them.  For the device decoder's first guess
// at the room its extracted arrays need, before anything runs on the device (spl_capi.cpp); false: cannot tell.
bool spl_bam_sample_density(spl_bam *bam, size_t b_lo, size_t b_hi, uint64_t *n_rec_out, uint64_t *n_ops_out, uint64_t *n_bytes_out)
{
    const size_t n_blocks = bam->dir.n_ready.load();
    b_hi = std::min(b_hi, n_blocks);
    if (b_lo >= b_hi) return false;
    void *ld = deflate_lib().ok ? deflate_lib().alloc() : nullptr;
    std::vector<uint8_t> buf;
    uint64_t n_rec = 0, n_ops = 0, n_bytes = 0;
    size_t last = (size_t)-1;
    for (int k = 0; k < 3; ++k) {
        const size_t b = b_lo + (b_hi - b_lo) * (size_t)k / 3;
        if (b == last) continue;
        last = b;
        uint64_t u = 0;
        int32_t tid = 0;
        if (!first_record_at(bam, b, ld, buf, &u, &tid)) continue;
        const uint8_t *const end = buf.data() + (buf.size() - 64);
        const uint8_t *c = buf.data() + (size_t)(u - bam->dir.at(b).uoff), *const c0 = c;
        uint64_t rec = 0, ops = 0;
        while (c + 36 <= end) {
            const uint32_t block_size = le32(c);
            if (block_size < 32 || block_size > (1u << 28) || c + 4 + block_size > end) break;
            ops += (uint64_t)c[16] | ((uint64_t)c[17] << 8);
            ++rec;
            c += 4 + (size_t)block_size;
        }
        if (!rec) continue;
        n_rec += rec;
        n_ops += ops;
        n_bytes += (uint64_t)(c - c0);
    }
    if (ld) deflate_lib().free_(ld);
    *n_rec_out = n_rec;
    *n_ops_out = n_ops;
    *n_bytes_out = n_bytes;
    return n_rec != 0 && n_bytes != 0;
}

extern "C" int spl_bam_sample(spl_bam *bam, int64_t *out3)
{
    if (!bam || !out3) return spl_set_error(SPL_ERR_ARG, "spl_bam_sample: null argument");
    const int rc = spl_bam_walk_all(bam);
    if (rc) return rc;
    uint64_t n_rec = 0, n_ops = 0, n_bytes = 0;
    if (!spl_bam_sample_density(bam, 0, bam->dir.n_ready.load(), &n_rec, &n_ops, &n_bytes))
        return spl_set_error(SPL_ERR_FORMAT, "%s: no whole record found where the sample looked", bam->path.c_str());
    out3[0] = (int64_t)n_rec; out3[1] = (int64_t)n_ops; out3[2] = (int64_t)n_bytes;
    return SPL_OK;
}

extern "C" int spl_bam_share_plan(spl_bam *bam, int n_shares, int *n_out)
{
    if (!bam || n_shares < 1) return spl_set_error(SPL_ERR_ARG, "spl_bam_share_plan: bad argument");
    {
        std::lock_guard<std::mutex> lock(bam->mu);
        if (!bam->shares.empty()) { if (n_out) *n_out = (int)bam->shares.size(); return SPL_OK; }
    }
    int rc = spl_bam_walk_all(bam);
    if (rc) return rc;
    const size_t n_blocks = bam->dir.n_ready.load();
    const int32_t all = bam->n_refs + 1;
    const uint64_t stream_end = n_blocks ? bam->dir.at(n_blocks - 1).uoff + bam->dir.at(n_blocks - 1).isize : 0;
    std::vector<size_t> cut_block{0};
    std::vector<uint64_t> cut_u{bam->header_bytes};
    std::vector<int32_t> cut_tid{0};
    if (n_shares > 1 && n_blocks > 8) {
        void *ld = deflate_lib().ok ? deflate_lib().alloc() : nullptr;
        std::vector<uint8_t> buf;
        const bool dbg = getenv("SPL_BAM_TIMING") != nullptr;
        // where the records begin in the file: the shares are equal parts of what lies behind the header's blocks
        size_t first_rec_block = 0;
        while (first_rec_block + 1 < n_blocks && bam->dir.at(first_rec_block + 1).uoff <= bam->header_bytes) ++first_rec_block;
        const double byte0 = (double)bam->dir.at(first_rec_block).coff, byte1 = (double)bam->fsize;
        for (int k = 1; k < n_shares; ++k) {
            const size_t target = (size_t)(byte0 + (byte1 - byte0) * k / n_shares);
            size_t lo = 0, hi = n_blocks; // the block that begins nearest to the target offset
            while (lo + 1 < hi) { const size_t mid = lo + (hi - lo) / 2; if (bam->dir.at(mid).coff <= target) lo = mid; else hi = mid; }
            size_t b = lo;
            if (lo + 1 < n_blocks && bam->dir.at(lo + 1).coff - target < target - bam->dir.at(lo).coff) b = lo + 1;
            if (b <= cut_block.back() || b <= first_rec_block || b >= n_blocks) continue; // (more shares asked for than the file has blocks to begin them with)
            uint64_t u = 0; int32_t tid = 0;
            if (!first_record_at(bam, b, ld, buf, &u, &tid)) { if (dbg) fprintf(stderr, "[share plan] block %zu: no record boundary found\n", b); continue; }
            if (dbg) fprintf(stderr, "[share plan] cut %d: block %zu, first record at %llu, reference %d\n", k, b, (unsigned long long)u, tid);
            if (u <= cut_u.back() || u >= stream_end || tid < cut_tid.back()) continue; // (nothing begins between two cuts; or a file that is not sorted: the decoders will say so)
            cut_block.push_back(b); cut_u.push_back(u); cut_tid.push_back(tid);
        }
        if (ld) deflate_lib().free_(ld);
    }
    std::vector<spl_bam_share> plan;
    for (size_t k = 0; k < cut_block.size(); ++k) {
        const bool last = k + 1 == cut_block.size();
        spl_bam_share sh;
        sh.block_lo = cut_block[k];
        sh.block_own = last ? n_blocks : cut_block[k + 1];
        sh.u_lo = cut_u[k];
        sh.u_hi = last ? stream_end : cut_u[k + 1];
        sh.tid_lo = cut_tid[k];
        sh.tid_hi = last ? all : cut_tid[k + 1] + 1;
        size_t hi = (size_t)sh.block_own; // the blocks that hold the end of the share's last record
        while (hi < n_blocks && bam->dir.at(hi).uoff < sh.u_hi) ++hi;
        if (hi < n_blocks) ++hi; // (and one more: whoever looks for the boundary u_hi itself -- a last block no record begins in -- needs a record's worth of bytes behind it)
        sh.block_hi = hi;
        plan.push_back(sh);
    }
    std::lock_guard<std::mutex> lock(bam->mu);
    if (bam->shares.empty()) { bam->shares = plan; bam->share_results.assign(plan.size(), spl_bam::ShareResult()); }
    if (n_out) *n_out = (int)bam->shares.size();
    return SPL_OK;
}

// What share k is: the bytes of the file its own blocks take (what its device uploads and inflates, the tail aside), and the
// stretch of the inflated stream its records begin in.  Any output may be null.
extern "C" int spl_bam_share_info(spl_bam *bam, int k, int64_t *file_bytes_out, int64_t *u_lo_out, int64_t *u_hi_out, int64_t *tail_blocks_out)
{
    if (!bam) return spl_set_error(SPL_ERR_ARG, "spl_bam_share_info: null argument");
    spl_bam_share sh;
    const int rc = spl_bam_share_get(bam, k, &sh);
    if (rc) return rc;
    const size_t n_blocks = bam->dir.n_ready.load();
    const uint64_t c0 = bam->dir.at((size_t)sh.block_lo).coff, c1 = sh.block_own < n_blocks ? bam->dir.at((size_t)sh.block_own).coff : bam->fsize;
    if (file_bytes_out) *file_bytes_out = (int64_t)(c1 - c0);
    if (u_lo_out) *u_lo_out = (int64_t)sh.u_lo;
    if (u_hi_out) *u_hi_out = (int64_t)sh.u_hi;
    if (tail_blocks_out) *tail_blocks_out = (int64_t)(sh.block_hi - sh.block_own);
    return SPL_OK;
}

// The records of share k by the HOST's inflate and a plain walk from u_lo: how many per reference (per_tid[n_ref + 1], the last
// entry the records without a reference).  The walk must arrive exactly at u_hi -- what the device decoder of the share is held
// to as well.  Diagnostic / test hook: the sum over the shares is the file.
extern "C" int spl_bam_share_count_host(spl_bam *bam, int k, int64_t *per_tid)
{
    if (!bam || !per_tid) return spl_set_error(SPL_ERR_ARG, "spl_bam_share_count_host: null argument");
    spl_bam_share sh;
    const int rc = spl_bam_share_get(bam, k, &sh);
    if (rc) return rc;
    for (int t = 0; t <= bam->n_refs; ++t) per_tid[t] = 0;
    const uint8_t *file = (const uint8_t *)bam->map;
    void *ld = deflate_lib().ok ? deflate_lib().alloc() : nullptr;
    std::vector<uint8_t> buf;
    size_t len = 0;
    for (size_t b = (size_t)sh.block_lo; b < (size_t)sh.block_hi; ++b) len += bam->dir.at(b).isize;
    buf.resize(len + 64);
    size_t at = 0;
    bool ok = true;
    for (size_t b = (size_t)sh.block_lo; b < (size_t)sh.block_hi && ok; ++b) {
        ok = inflate_block(file, bam->dir.at(b), buf.data() + at, ld);
        at += bam->dir.at(b).isize;
    }
    if (ld) deflate_lib().free_(ld);
    if (!ok) return spl_set_error(SPL_ERR_FORMAT, "%s: a block of share %d does not inflate", bam->path.c_str(), k);
    const uint64_t u0 = bam->dir.at((size_t)sh.block_lo).uoff;
    uint64_t u = sh.u_lo;
    while (u < sh.u_hi) {
        if (u - u0 + 36 > len) return spl_set_error(SPL_ERR_FORMAT, "%s: share %d: a record runs past the share's blocks", bam->path.c_str(), k);
        const uint8_t *c = buf.data() + (size_t)(u - u0);
        const uint32_t bs = le32(c);
        if (bs < 32) return spl_set_error(SPL_ERR_FORMAT, "%s: share %d: corrupt record", bam->path.c_str(), k);
        const int32_t tid = le32s(c + 4);
        per_tid[tid < 0 || tid >= bam->n_refs ? bam->n_refs : tid]++;
        u += 4ull + bs;
    }
    if (u != sh.u_hi) return spl_set_error(SPL_ERR_FORMAT, "%s: share %d: the walk from %llu arrives at %llu, not at the next share's first record %llu", bam->path.c_str(), k,
                                           (unsigned long long)sh.u_lo, (unsigned long long)u, (unsigned long long)sh.u_hi);
    return SPL_OK;
}

int spl_bam_share_get(spl_bam *bam, int k, spl_bam_share *out)
{
    std::lock_guard<std::mutex> lock(bam->mu);
    if (!bam || k < 0 || (size_t)k >= bam->shares.size() || !out) return spl_set_error(SPL_ERR_ARG, "spl_bam_share_get: no such share");
    *out = bam->shares[(size_t)k];
    return SPL_OK;
}

int spl_bam_share_done(spl_bam *bam, int k, void *handle, void (*free_fn)(void *), const int64_t *ref_first, const int64_t *ref_n,
                       const int64_t *ref_max_end, int64_t n_records, int failed)
{
    std::unique_lock<std::mutex> lock(bam->mu);
    if (k < 0 || (size_t)k >= bam->share_results.size() || bam->share_results[(size_t)k].reported) {
        lock.unlock();
        if (handle && free_fn) free_fn(handle);
        return spl_set_error(SPL_ERR_ARG, "spl_bam_share_done: no such share, or reported twice");
    }
    spl_bam::ShareResult &r = bam->share_results[(size_t)k];
    r.reported = true;
    r.failed = failed != 0;
    r.handle = handle;
    r.free_fn = free_fn;
    r.n_records = n_records;
    if (!r.failed) {
        r.first.assign(ref_first, ref_first + bam->n_refs);
        r.n.assign(ref_n, ref_n + bam->n_refs);
        r.max_end.assign(ref_max_end, ref_max_end + bam->n_refs);
    }
    bool all = true, any_failed = false;
    for (const spl_bam::ShareResult &x : bam->share_results) { all = all && x.reported; any_failed = any_failed || x.failed; }
    if (!all) return SPL_OK;
    if (bam->claim != 1) return SPL_OK; // (somebody gave the file to the host threads meanwhile)
    bam->claim = 2;
    if (!any_failed) { // every share was sorted by reference in itself; so must the shares be among each other
        int last_tid = -1;
        for (const spl_bam::ShareResult &x : bam->share_results) {
            int lo = -1, hi = -1;
            for (int t = 0; t < bam->n_refs; ++t) if (x.n[(size_t)t] > 0) { if (lo < 0) lo = t; hi = t; }
            if (lo >= 0 && lo < last_tid) { any_failed = true; if (bam->decline_reason.empty()) bam->decline_reason = "not sorted by reference"; }
            if (hi >= 0) last_tid = hi;
        }
    }
    if (any_failed) { // everything the devices have is dropped: the host threads decode the file
        std::vector<spl_bam::ShareResult> drop;
        drop.swap(bam->share_results);
        bam->share_results.assign(drop.size(), spl_bam::ShareResult());
        for (spl_bam::ShareResult &x : bam->share_results) x.reported = true;
        if (!cancelled_locked(bam)) bam->worker = std::thread(decode_worker, bam);
        bam->cv.notify_all(); // (spl_bam_wait_device)
        lock.unlock();
        for (spl_bam::ShareResult &x : drop) if (x.handle && x.free_fn) x.free_fn(x.handle);
        return SPL_OK;
    }
    bam->lazy = true;
    int64_t n_all = 0;
    for (size_t s = 0; s < bam->share_results.size(); ++s) {
        spl_bam::ShareResult &x = bam->share_results[s];
        spl_bam::DevShare d;
        d.handle = x.handle; d.free_fn = x.free_fn; d.share = (int)s;
        x.handle = nullptr;
        bam->dev_shares.push_back(d);
        n_all += x.n_records;
        for (int t = 0; t < bam->n_refs; ++t) { // (a reference may have a part in several shares: file order = share order)
            if (x.n[(size_t)t] <= 0) continue;
            PendingPart *pp = new PendingPart();
            pp->tid = t;
            pp->share = (int)s;
            pp->first = x.first[(size_t)t];
            pp->reads.n = (size_t)x.n[(size_t)t];
            pp->reads.max_end = x.max_end[(size_t)t];
            bam->parts[(size_t)t].push_back(pp);
            bam->ref_reads[(size_t)t] += x.n[(size_t)t];
            bam->ref_max_end[(size_t)t] = std::max(bam->ref_max_end[(size_t)t], x.max_end[(size_t)t]);
        }
    }
    bam->n_records = n_all;
    bam->max_tid_seen = bam->n_refs - 1;
    bam->complete_upto = bam->n_refs;
    bam->shares_on_device = true;
    bam->done = true;
    bam->cv.notify_all();
    return SPL_OK;
}

extern "C" int spl_bam_share_range(spl_bam *bam, int k, int *tid_lo_out, int *tid_hi_out)
{
    if (!bam) return spl_set_error(SPL_ERR_ARG, "spl_bam_share_range: null argument");
    spl_bam_share sh;
    const int rc = spl_bam_share_get(bam, k, &sh);
    if (rc) return rc;
    if (tid_lo_out) *tid_lo_out = sh.tid_lo;
    if (tid_hi_out) *tid_hi_out = sh.tid_hi;
    return SPL_OK;
}

// What share k holds of reference `tid`, once the shares' decoders are done (spl_bam_wait_device): the number of its records
// there and the last base any of them covers.
extern "C" int spl_bam_share_ref(spl_bam *bam, int k, int tid, int64_t *n_reads_out, int64_t *max_end_out)
{
    if (!bam) return spl_set_error(SPL_ERR_ARG, "spl_bam_share_ref: null argument");
    std::lock_guard<std::mutex> lock(bam->mu);
    if (k < 0 || (size_t)k >= bam->share_results.size() || tid < 0 || tid >= bam->n_refs) return spl_set_error(SPL_ERR_ARG, "spl_bam_share_ref: no such share or reference");
    const spl_bam::ShareResult &x = bam->share_results[(size_t)k];
    if (!x.reported || x.failed || x.n.size() != (size_t)bam->n_refs) return spl_set_error(SPL_ERR_ARG, "spl_bam_share_ref: share %d has not been decoded on a device", k);
    if (n_reads_out) *n_reads_out = x.n[(size_t)tid];
    if (max_end_out) *max_end_out = x.max_end[(size_t)tid];
    return SPL_OK;
}

extern "C" int spl_bam_decoded_on_device(spl_bam *bam, int *on_device_out)
{
    if (!bam || !on_device_out) return spl_set_error(SPL_ERR_ARG, "spl_bam_decoded_on_device: null argument");
    std::lock_guard<std::mutex> lock(bam->mu);
    *on_device_out = (bam->shares_on_device || (bam->done && !bam->dev_shares.empty())) ? 1 : 0;
    return SPL_OK;
}

int spl_bam_shares_on_device(spl_bam *bam)
{
    std::lock_guard<std::mutex> lock(bam->mu);
    return bam->shares_on_device ? 1 : 0;
}

void spl_bam_set_fetch(spl_bam *bam, int (*fetch)(void *, int32_t **, uint16_t **, uint32_t **, uint32_t **))
{
    std::lock_guard<std::mutex> lock(bam->mu);
    bam->dev_fetch = fetch;
}

// The host copies of reads that were adopted without them: of every share that has not brought its yet.  Called with bam->mu
// held; the copy itself (gigabytes over PCIe) runs WITHOUT it -- whoever else wants a reference that is there already, or wants
// to wait for one, is not kept out -- and a second caller waits for the first one's copy instead of making its own.
static int fetch_lazy(spl_bam *bam, std::unique_lock<std::mutex> &lock)
{
    for (;;) {
        if (!bam->lazy) return SPL_OK;
        if (!bam->dev_fetch || bam->dev_shares.empty()) return spl_set_error(SPL_ERR_ARG, "%s: decoded reads are neither on the host nor fetchable", bam->path.c_str());
        if (bam->fetching) { bam->cv.wait(lock, [&]() { return !bam->fetching; }); continue; }
        size_t k = 0;
        while (k < bam->dev_shares.size() && bam->dev_shares[k].fetched) ++k;
        if (k == bam->dev_shares.size()) { bam->lazy = false; return SPL_OK; }
        bam->fetching = true;
        void *const handle = bam->dev_shares[k].handle;
        const int share = bam->dev_shares[k].share;
        int32_t *pos = nullptr; uint16_t *flag = nullptr; uint32_t *cig_off = nullptr, *cigar = nullptr;
        lock.unlock();
        const int rc = bam->dev_fetch(handle, &pos, &flag, &cig_off, &cigar);
        lock.lock();
        bam->fetching = false;
        if (rc) { bam->cv.notify_all(); return rc; }
        bam->slabs.push_back(pos); bam->slabs.push_back(flag); bam->slabs.push_back(cig_off); bam->slabs.push_back(cigar);
        for (int t = 0; t < bam->n_refs; ++t) {
            for (PendingPart *pp : bam->parts[(size_t)t]) {
                if (pp->share != share) continue;
                RefReads &r = pp->reads;
                const int64_t first = pp->first;
                r.pos = pos + first;
                r.flag = flag + first;
                r.cig_off = cig_off + first;
                r.cigar = cigar;
                r.n_ops = (size_t)(r.cig_off[r.n] - r.cig_off[0]);
            }
        }
        bam->dev_shares[k].fetched = true;
        bam->cv.notify_all();
    }
}

static int decode_status(spl_bam *bam) // (call with bam->mu held)
{
    if (bam->err_code) return spl_set_error(bam->err_code, "%s", bam->error.c_str());
    return SPL_OK;
}

extern "C" int spl_bam_wait_ref(spl_bam *bam, int tid, int64_t *n_reads_out, int64_t *max_end_out)
{
    if (!bam) return spl_set_error(SPL_ERR_ARG, "spl_bam_wait_ref: null argument");
    if (tid < 0 || tid >= bam->n_refs) return spl_set_error(SPL_ERR_ARG, "tid %d out of range", tid);
    (void)spl_bam_start_host(bam); // (a deferred file nobody has decoded yet: on the host then)
    std::unique_lock<std::mutex> lock(bam->mu);
    bam->cv.wait(lock, [&]() { return bam->done || tid < bam->complete_upto; });
    const int rc = decode_status(bam);
    if (rc) return rc;
    if (n_reads_out) *n_reads_out = bam->ref_reads[(size_t)tid];
    if (max_end_out) *max_end_out = bam->ref_max_end[(size_t)tid];
    return SPL_OK;
}

extern "C" int spl_bam_wait_all(spl_bam *bam, int *sorted_out)
{
    if (!bam) return spl_set_error(SPL_ERR_ARG, "spl_bam_wait_all: null argument");
    (void)spl_bam_start_host(bam);
    std::unique_lock<std::mutex> lock(bam->mu);
    bam->cv.wait(lock, [&]() { return bam->done; });
    if (sorted_out) *sorted_out = bam->out_of_order ? 0 : 1;
    return decode_status(bam);
}

extern "C" int spl_bam_open(const char *path, int n_threads, spl_bam **out)
{
    int rc = spl_bam_open_stream(path, n_threads, out);
    if (rc != SPL_OK) return rc;
    rc = spl_bam_wait_all(*out, nullptr);
    if (rc != SPL_OK) { delete *out; *out = nullptr; }
    return rc;
}

extern "C" void spl_bam_close(spl_bam *bam) { delete bam; }
extern "C" int spl_bam_n_ref(const spl_bam *bam) { return bam ? bam->n_refs : 0; }
extern "C" const char *spl_bam_ref_name(const spl_bam *bam, int tid)
{
    if (!bam || tid < 0 || (size_t)tid >= bam->ref_names.size()) return nullptr;
    return bam->ref_names[(size_t)tid].c_str();
}
extern "C" int64_t spl_bam_ref_length(const spl_bam *bam, int tid)
{
    if (!bam || tid < 0 || (size_t)tid >= bam->ref_lengths.size()) return -1;
    return bam->ref_lengths[(size_t)tid];
}
extern "C" int64_t spl_bam_n_records(const spl_bam *bam)
{
    if (!bam) return 0;
    spl_bam *b = const_cast<spl_bam *>(bam);
    (void)spl_bam_start_host(b);
    std::unique_lock<std::mutex> lock(b->mu);
    b->cv.wait(lock, [&]() { return b->done; });
    return b->n_records;
}

extern "C" int spl_bam_reads(const spl_bam *cbam, int tid, spl_reads *out, int64_t *max_end_out)
{
    spl_bam *bam = const_cast<spl_bam *>(cbam);
    if (!bam || !out) return spl_set_error(SPL_ERR_ARG, "spl_bam_reads: null argument");
    if (tid < 0 || tid >= bam->n_refs) return spl_set_error(SPL_ERR_ARG, "tid %d out of range", tid);
    int rc = spl_bam_wait_all(bam, nullptr); // (the whole file: these arrays must hold every record of the reference, sorted file or not)
    if (rc) return rc;
    std::unique_lock<std::mutex> lock(bam->mu);
    rc = fetch_lazy(bam, lock);
    if (rc) return rc;
    if (!bam->assembled[(size_t)tid]) {
        std::string err;
        if (!assemble_ref(bam, tid, err)) return spl_set_error(SPL_ERR_NOMEM, "%s: %s", bam->path.c_str(), err.c_str());
        bam->assembled[(size_t)tid] = 1;
    }
    const RefFinal &rr = bam->refs_storage[(size_t)tid];
    out->n_reads = rr.n;
    out->pos = rr.pos;
    out->flag = rr.flag;
    out->cig_off = rr.cig_off;
    out->cigar = rr.cigar;
    if (max_end_out) *max_end_out = rr.max_end;
    return SPL_OK;
}

int spl_bam_source(spl_bam *bam, int tid, splpack::Source *out, int64_t *max_end_out)
{
    if (!bam || !out) return spl_set_error(SPL_ERR_ARG, "spl_bam_source: null argument");
    int rc = spl_bam_wait_ref(bam, tid, nullptr, max_end_out);
    if (rc) return rc;
    std::unique_lock<std::mutex> lock(bam->mu);
    rc = fetch_lazy(bam, lock);
    if (rc) return rc;
    for (const PendingPart *pt : bam->parts[(size_t)tid]) {
        const RefReads &r = pt->reads;
        if (r.n == 0) continue;
        out->add(splpack::Part{r.pos, r.flag, r.cig_off, r.cigar, (int64_t)r.n});
    }
    return SPL_OK;
}

// ---- BAM writer (synthetic workloads, tests): the inverse of the reader above -------------------------------------
// Records carry a dummy read name, SEQ and QUAL of the query length so that the file has the size and block structure
// of a real BAM.  The reads are cut into slices; a slice is formatted and deflated into BGZF blocks of its own by one
// thread (concatenated BGZF blocks are a BGZF file, wherever the records are cut).  Only what spl_bam_open reads back is
// meaningful.
namespace {

void put32(std::vector<uint8_t> &v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((uint8_t)(x >> (8 * i))); }
void put16(std::vector<uint8_t> &v, uint32_t x) { v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); }

// raw[0..n) -> one BGZF block appended to out.  ld: a libdeflate compressor or null (zlib).
bool deflate_block(const uint8_t *raw, size_t n, int level, void *ld, std::vector<uint8_t> &out)
{
    const size_t at = out.size();
    out.resize(at + 18 + n + n / 8 + 64 + 8);
    uint8_t *dst = out.data() + at;
    size_t clen = 0;
    if (ld) {
        clen = deflate_lib().compress(ld, raw, n, dst + 18, n + n / 8 + 64);
        if (clen == 0) return false;
    } else {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
        zs.next_in = const_cast<Bytef *>(raw); zs.avail_in = (uInt)n;
        zs.next_out = dst + 18; zs.avail_out = (uInt)(n + n / 8 + 64);
        const int rc = deflate(&zs, Z_FINISH);
        clen = zs.total_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END) return false;
    }
    const size_t bsize = 18 + clen + 8;
    if (bsize > 65536) return false;
    static const uint8_t head[12] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0};
    memcpy(dst, head, 12);
    dst[12] = 'B'; dst[13] = 'C'; dst[14] = 2; dst[15] = 0;
    dst[16] = (uint8_t)((bsize - 1) & 0xff); dst[17] = (uint8_t)((bsize - 1) >> 8);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), raw, (uInt)n);
    uint8_t *t = dst + 18 + clen;
    for (int i = 0; i < 4; ++i) { t[i] = (uint8_t)(crc >> (8 * i)); t[4 + i] = (uint8_t)((uint32_t)n >> (8 * i)); }
    out.resize(at + bsize);
    return true;
}

bool deflate_all(const std::vector<uint8_t> &raw, int level, void *ld, std::vector<uint8_t> &out)
{
    const size_t BLOCK = 0xff00;
    for (size_t at = 0; at < raw.size(); at += BLOCK)
        if (!deflate_block(raw.data() + at, std::min(BLOCK, raw.size() - at), level, ld, out)) return false;
    return true;
}

} // namespace

namespace {

inline uint64_t mix64(uint64_t x) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; return x ^ (x >> 31); }

// seq_mode 2: one record the way an aligner's output looks after `samtools sort` (what the reference reads through
// `samtools view`, SpliSER_v0_1_8.py:422): an Illumina read name, MAPQ and bin as htslib computes them, mate fields for paired
// flags, SEQ taken from a reference that is a function of the position -- overlapping reads share their bases, a few mismatches
// aside, which is where a sorted BAM's long, far matches come from --, QUAL from a per-cycle distribution of the four NovaSeq
// bins, and STAR's tags (NH HI AS nM, XS:A on spliced reads, MD:Z on a third).
void realistic_record(std::vector<uint8_t> &cur, int tid, int64_t ref_len, int32_t pos1, uint32_t flag, const uint32_t *cig, uint32_t n_ops, uint64_t &lcg)
{
    auto rnd = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return lcg >> 33; }; // 31 bits
    uint32_t qlen = 0, rlen = 0;
    bool spliced = false;
    for (uint32_t j = 0; j < n_ops; ++j) {
        const uint32_t c = cig[j] & 15u, l = cig[j] >> 4;
        if (c == 0 || c == 1 || c == 4 || c == 7 || c == 8) qlen += l;
        if (c == 0 || c == 2 || c == 3 || c == 7 || c == 8) rlen += l;
        spliced = spliced || c == 3;
    }
    char name[64];
    const uint32_t tile = (rnd() & 1u ? 1101u : 2101u) + (uint32_t)(rnd() % 78u) + 100u * (uint32_t)(rnd() % 6u);
    const int l_name = snprintf(name, sizeof name, "A00741:215:HGV2FDSXY:%u:%u:%u:%u", 1u + (unsigned)(rnd() & 3u), tile, 1000u + (unsigned)(rnd() % 31000u),
                                1000u + (unsigned)(rnd() % 36000u)) + 1;
    const uint32_t nm = (uint32_t)(rnd() % 100u < 74u ? 0u : (rnd() % 100u < 75u ? 1u : 2u + rnd() % 3u));
    const uint32_t mq = rnd() % 100u;
    const uint8_t mapq = mq < 90u ? 255 : (mq < 96u ? 3 : (mq < 99u ? 1 : 0));
    const int32_t pos0 = pos1 - 1, end0 = pos0 + (int32_t)((rlen && !(flag & 4u)) ? rlen : 1u);
    uint32_t bin; // reg2bin (SAM specification 5.3)
    {
        const int32_t b = pos0, e = end0 - 1;
        if (b >> 14 == e >> 14) bin = 4681u + (uint32_t)(b >> 14);
        else if (b >> 17 == e >> 17) bin = 585u + (uint32_t)(b >> 17);
        else if (b >> 20 == e >> 20) bin = 73u + (uint32_t)(b >> 20);
        else if (b >> 23 == e >> 23) bin = 9u + (uint32_t)(b >> 23);
        else if (b >> 26 == e >> 26) bin = 1u + (uint32_t)(b >> 26);
        else bin = 0;
    }
    int32_t mtid = -1, mpos = -1, tlen = 0;
    if (flag & 1u) { // paired: the mate a fragment length away, on the other strand
        const int32_t frag = 180 + (int32_t)(rnd() % 320u);
        mtid = tid;
        if (flag & 16u) { mpos = std::max(0, end0 - frag); tlen = -(end0 - mpos); }
        else { mpos = std::min<int64_t>(std::max<int64_t>(ref_len - 1, 0), (int64_t)pos0 + frag - (int32_t)qlen); if (mpos < pos0) mpos = pos0; tlen = mpos + (int32_t)qlen - pos0; }
    }
    // tags first (their size is part of block_size)
    uint8_t tags[96];
    size_t nt = 0;
    auto tag_u8 = [&](char a, char b, uint32_t v) {
        tags[nt++] = (uint8_t)a; tags[nt++] = (uint8_t)b;
        if (v < 256u) { tags[nt++] = 'C'; tags[nt++] = (uint8_t)v; } else { tags[nt++] = 'S'; tags[nt++] = (uint8_t)v; tags[nt++] = (uint8_t)(v >> 8); }
    };
    const uint32_t nh = mapq == 255 ? 1u : (mapq == 3 ? 2u : (mapq == 1 ? 3u + (uint32_t)(rnd() & 1u) : 5u + (uint32_t)(rnd() % 6u)));
    tag_u8('N', 'H', nh);
    tag_u8('H', 'I', 1u + (uint32_t)(rnd() % nh));
    tag_u8('A', 'S', (flag & 1u ? 2u * qlen : qlen) - 2u - 2u * nm);
    tag_u8('n', 'M', nm);
    if (spliced) { tags[nt++] = 'X'; tags[nt++] = 'S'; tags[nt++] = 'A'; tags[nt++] = (flag & 16u) ? '-' : '+'; }
    uint32_t mm_at[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t m = 0; m < nm && m < 8u; ++m) mm_at[m] = qlen ? (uint32_t)(rnd() % qlen) : 0u;
    if (rnd() % 3u == 0) { // MD:Z (matches between mismatches; which base it was does not matter for what deflate sees)
        char md[48];
        int k = 0;
        if (nm == 0 || qlen == 0) k = snprintf(md, sizeof md, "%u", rlen);
        else {
            const uint32_t a = mm_at[0] % (rlen ? rlen : 1u);
            k = snprintf(md, sizeof md, "%u%c%u", a, "ACGT"[rnd() & 3u], rlen - a - (rlen ? 1u : 0u));
        }
        tags[nt++] = 'M'; tags[nt++] = 'D'; tags[nt++] = 'Z';
        memcpy(tags + nt, md, (size_t)k + 1);
        nt += (size_t)k + 1;
    }
    const uint32_t bs = 32 + (uint32_t)l_name + 4 * n_ops + (qlen + 1) / 2 + qlen + (uint32_t)nt;
    put32(cur, bs);
    put32(cur, (uint32_t)tid);
    put32(cur, (uint32_t)pos0);
    cur.push_back((uint8_t)l_name); cur.push_back(mapq);
    put16(cur, bin); put16(cur, n_ops); put16(cur, flag);
    put32(cur, qlen); put32(cur, (uint32_t)mtid); put32(cur, (uint32_t)mpos); put32(cur, (uint32_t)tlen);
    cur.insert(cur.end(), name, name + l_name);
    for (uint32_t j = 0; j < n_ops; ++j) put32(cur, cig[j]);
    // SEQ: the reference's bases under the aligned blocks, random ones for what the reference does not have (I, S)
    static const uint8_t code[4] = {1, 2, 4, 8};
    uint8_t q[2]; // (two bases make a byte)
    uint32_t qi = 0;
    int64_t rp = pos0;
    auto emit = [&](uint8_t b2) {
        for (uint32_t m = 0; m < nm && m < 8u; ++m) if (mm_at[m] == qi) b2 = (uint8_t)((b2 + 1u + (qi & 1u)) & 3u);
        q[qi & 1u] = code[b2];
        if (qi & 1u) cur.push_back((uint8_t)(q[0] << 4 | q[1]));
        ++qi;
    };
    for (uint32_t j = 0; j < n_ops; ++j) {
        const uint32_t c = cig[j] & 15u, l = cig[j] >> 4;
        if (c == 0 || c == 7 || c == 8) { for (uint32_t x = 0; x < l; ++x, ++rp) emit((uint8_t)(mix64(((uint64_t)(uint32_t)tid << 40) ^ (uint64_t)rp) >> 62)); }
        else if (c == 1 || c == 4) { for (uint32_t x = 0; x < l; ++x) emit((uint8_t)(rnd() & 3u)); }
        else if (c == 2 || c == 3) rp += l;
    }
    if (qi & 1u) cur.push_back((uint8_t)(q[0] << 4));
    // QUAL: NovaSeq's bins (37, 25, 11, 2); the best one less likely towards the read's end, a call like the one before it more often than not
    uint8_t prev = 37;
    for (uint32_t j = 0; j < qlen; ++j) {
        const uint32_t r = (uint32_t)(rnd() % 1000u);
        if (j && r < 550u) { cur.push_back(prev); continue; }
        const uint32_t late = qlen ? 120u * j / qlen * j / qlen : 0u; // 0 .. 120 per mille
        const uint32_t u = (uint32_t)(rnd() % 1000u);
        prev = u < 930u - late ? 37 : (u < 975u - late / 2u ? 25 : (u < 997u ? 11 : 2));
        cur.push_back(prev);
    }
    cur.insert(cur.end(), tags, tags + nt);
}

} // namespace

// seq_mode 0: constant SEQ / QUAL bytes (a file that deflates to a few bytes per record); 1: pseudo-random bases and binned
// qualities in runs, so that records deflate about as well as those of a real library (~4x); 2: records an aligner would write
// (realistic_record above) in blocks cut the way htslib cuts them -- a record that fits a block is never split (bam_write1's
// bgzf_flush_try) -- at the compression level asked for (htslib's default is 6).
extern "C" int spl_bam_write2(const char *path, int n_ref, const char *const *ref_names, const int64_t *ref_lengths,
                              const spl_reads *per_ref, int level, int n_threads, int seq_mode)
{
    if (!path || n_ref < 0 || (n_ref && (!ref_names || !ref_lengths || !per_ref))) return spl_set_error(SPL_ERR_ARG, "spl_bam_write: bad argument");
    FILE *fh = fopen(path, "wb");
    if (!fh) return spl_set_error(SPL_ERR_IO, "cannot create %s", path);
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads <= 0) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    int rc = SPL_OK;
    // header
    {
        std::vector<uint8_t> cur, out;
        std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
        for (int i = 0; i < n_ref; ++i) text += std::string("@SQ\tSN:") + ref_names[i] + "\tLN:" + std::to_string((long long)ref_lengths[i]) + "\n";
        cur.insert(cur.end(), {'B', 'A', 'M', 1});
        put32(cur, (uint32_t)text.size());
        cur.insert(cur.end(), text.begin(), text.end());
        put32(cur, (uint32_t)n_ref);
        for (int i = 0; i < n_ref; ++i) {
            const size_t ln = strlen(ref_names[i]) + 1;
            put32(cur, (uint32_t)ln);
            cur.insert(cur.end(), ref_names[i], ref_names[i] + ln);
            put32(cur, (uint32_t)ref_lengths[i]);
        }
        if (!deflate_all(cur, level, nullptr, out)) rc = spl_set_error(SPL_ERR_IO, "deflate failed while writing %s", path);
        else if (fwrite(out.data(), 1, out.size(), fh) != out.size()) rc = spl_set_error(SPL_ERR_IO, "short write to %s", path);
    }
    struct Slice { int tid; int64_t k0, k1; uint64_t serial; std::vector<uint8_t> out; bool bad = false; };
    std::vector<Slice> slices;
    const int64_t SLICE = 16384;
    uint64_t serial = 0;
    for (int tid = 0; tid < n_ref; ++tid)
        for (int64_t k = 0; k < per_ref[tid].n_reads; k += SLICE) {
            Slice sl;
            sl.tid = tid; sl.k0 = k; sl.k1 = std::min(per_ref[tid].n_reads, k + SLICE); sl.serial = serial;
            serial += (uint64_t)(sl.k1 - sl.k0);
            slices.push_back(std::move(sl));
        }
    const size_t WAVE = (size_t)n_threads * 4;
    for (size_t w0 = 0; w0 < slices.size() && rc == SPL_OK; w0 += WAVE) {
        const size_t w1 = std::min(slices.size(), w0 + WAVE);
        std::atomic<size_t> next(w0);
        auto work = [&]() {
            void *ld = deflate_lib().cok ? deflate_lib().calloc_(level < 1 ? 1 : (level > 12 ? 12 : level)) : nullptr;
            std::vector<uint8_t> cur;
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= w1) break;
                Slice &sl = slices[i];
                const spl_reads &r = per_ref[sl.tid];
                cur.clear();
                uint64_t lcg = 0x9E3779B97F4A7C15ull * (sl.serial + 1);
                if (seq_mode == 2) { // whole records per block, like htslib: a block is closed when the next record does not fit
                    const size_t BLOCK = 0xff00;
                    for (int64_t k = sl.k0; k < sl.k1 && !sl.bad; ++k) {
                        const size_t before = cur.size();
                        realistic_record(cur, sl.tid, ref_lengths[sl.tid], r.pos[k], r.flag[k], r.cigar + r.cig_off[k], r.cig_off[k + 1] - r.cig_off[k], lcg);
                        if (cur.size() > BLOCK && before) { // the block without this record, the record to the front
                            if (!deflate_block(cur.data(), before, level, ld, sl.out)) sl.bad = true;
                            cur.erase(cur.begin(), cur.begin() + (ptrdiff_t)before);
                        }
                        while (cur.size() > BLOCK && !sl.bad) { // (a record larger than a block is cut, as htslib cuts it)
                            if (!deflate_block(cur.data(), BLOCK, level, ld, sl.out)) sl.bad = true;
                            cur.erase(cur.begin(), cur.begin() + (ptrdiff_t)BLOCK);
                        }
                    }
                    if (!cur.empty() && !sl.bad && !deflate_block(cur.data(), cur.size(), level, ld, sl.out)) sl.bad = true;
                    continue;
                }
                for (int64_t k = sl.k0; k < sl.k1; ++k) {
                    const uint32_t o0 = r.cig_off[k], n_ops = r.cig_off[k + 1] - o0;
                    uint32_t qlen = 0;
                    for (uint32_t j = 0; j < n_ops; ++j) { const uint32_t c = r.cigar[o0 + j] & 15u; if (c == 0 || c == 1 || c == 4 || c == 7 || c == 8) qlen += r.cigar[o0 + j] >> 4; }
                    char name[24];
                    const int l_name = snprintf(name, sizeof(name), "r%llu", (unsigned long long)(sl.serial + (uint64_t)(k - sl.k0))) + 1;
                    const uint32_t bs = 32 + (uint32_t)l_name + 4 * n_ops + (qlen + 1) / 2 + qlen;
                    put32(cur, bs);
                    put32(cur, (uint32_t)sl.tid);
                    put32(cur, (uint32_t)(r.pos[k] - 1));
                    cur.push_back((uint8_t)l_name); cur.push_back(60);
                    put16(cur, 4680); put16(cur, n_ops); put16(cur, r.flag[k]);
                    put32(cur, qlen); put32(cur, 0xffffffffu); put32(cur, 0xffffffffu); put32(cur, 0);
                    cur.insert(cur.end(), name, name + l_name);
                    for (uint32_t j = 0; j < n_ops; ++j) put32(cur, r.cigar[o0 + j]);
                    if (seq_mode == 0) {
                        cur.insert(cur.end(), (qlen + 1) / 2, (uint8_t)0x12);
                        cur.insert(cur.end(), qlen, (uint8_t)30);
                    } else {
                        static const uint8_t base[4] = {1, 2, 4, 8};
                        static const uint8_t qbin[4] = {37, 37, 25, 11};
                        for (uint32_t j = 0; j < (qlen + 1) / 2; ++j) {
                            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                            cur.push_back((uint8_t)((base[(lcg >> 60) & 3] << 4) | base[(lcg >> 58) & 3]));
                        }
                        for (uint32_t j = 0; j < qlen;) { // binned qualities in runs
                            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                            const uint32_t run = 1u + (uint32_t)((lcg >> 40) % 24u);
                            const uint8_t q = qbin[(lcg >> 62) & 3];
                            for (uint32_t x = 0; x < run && j < qlen; ++x, ++j) cur.push_back(q);
                        }
                    }
                }
                if (!deflate_all(cur, level, ld, sl.out)) sl.bad = true;
            }
            if (ld) deflate_lib().cfree(ld);
        };
        std::vector<std::thread> pool;
        const int nt = (int)std::min<size_t>((size_t)n_threads, w1 - w0);
        for (int t = 1; t < nt; ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
        for (size_t i = w0; i < w1 && rc == SPL_OK; ++i) {
            if (slices[i].bad) rc = spl_set_error(SPL_ERR_IO, "deflate failed while writing %s", path);
            else if (fwrite(slices[i].out.data(), 1, slices[i].out.size(), fh) != slices[i].out.size()) rc = spl_set_error(SPL_ERR_IO, "short write to %s", path);
            std::vector<uint8_t>().swap(slices[i].out);
        }
    }
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (rc == SPL_OK && fwrite(eof, 1, 28, fh) != 28) rc = spl_set_error(SPL_ERR_IO, "short write to %s", path);
    fclose(fh);
    return rc;
}

extern "C" int spl_bam_write(const char *path, int n_ref, const char *const *ref_names, const int64_t *ref_lengths,
                             const spl_reads *per_ref, int level, int n_threads)
{
    return spl_bam_write2(path, n_ref, ref_names, ref_lengths, per_ref, level, n_threads, 0);
}
