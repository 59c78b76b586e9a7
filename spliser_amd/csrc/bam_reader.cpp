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
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spliser.h"
#include "spl_error.h"

namespace {

// ---- optional libdeflate (present in the image as libdeflate.so.0 without a header) -------------------
struct Deflate {
    void *(*alloc)(void) = nullptr;
    int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*free_)(void *) = nullptr;
    uint32_t (*crc32)(uint32_t, const void *, size_t) = nullptr;
    bool ok = false;
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
    size_t coff;   // offset of the block in the file
    uint32_t csize; // whole block
    uint32_t xlen;
    uint32_t isize; // uncompressed payload
};

struct RefReads {
    std::vector<int32_t> pos;
    std::vector<uint16_t> flag;
    std::vector<uint32_t> cig_off; // n + 1
    std::vector<uint32_t> cigar;
    int64_t max_end = 0;
    RefReads() { cig_off.push_back(0); }
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

struct spl_bam {
    std::vector<std::string> ref_names;
    std::vector<int64_t> ref_lengths;
    std::vector<RefReads> refs;
    int64_t n_records = 0;
};

namespace {

struct Parser {
    spl_bam *bam;
    bool header_done = false;
    std::string err;

    // Consume as many complete items as possible from [p, end); returns bytes consumed.
    size_t feed(const uint8_t *p, const uint8_t *end, bool &fatal)
    {
        const uint8_t *start = p;
        fatal = false;
        if (!header_done) {
            // magic, l_text, text, n_ref, then per ref: l_name, name, l_ref -- needs to be complete in the buffer
            if (end - p < 12) return 0;
            if (memcmp(p, "BAM\1", 4) != 0) { err = "not a BAM file (bad magic)"; fatal = true; return 0; }
            const uint32_t l_text = le32(p + 4);
            if ((size_t)(end - p) < 12 + (size_t)l_text) return 0;
            const uint8_t *q = p + 8 + l_text;
            const int32_t n_ref = le32s(q);
            q += 4;
            if (n_ref < 0) { err = "negative n_ref"; fatal = true; return 0; }
            std::vector<std::string> names;
            std::vector<int64_t> lens;
            for (int32_t i = 0; i < n_ref; ++i) {
                if (end - q < 4) return 0;
                const uint32_t l_name = le32(q);
                if ((size_t)(end - q) < 4 + (size_t)l_name + 4) return 0;
                names.emplace_back((const char *)q + 4, l_name ? l_name - 1 : 0);
                lens.push_back(le32(q + 4 + l_name));
                q += 4 + l_name + 4;
            }
            bam->ref_names.swap(names);
            bam->ref_lengths.swap(lens);
            bam->refs.resize((size_t)n_ref);
            header_done = true;
            p = q;
        }
        while (end - p >= 4) {
            const uint32_t bs = le32(p);
            if (bs < 32) { err = "corrupt record (block_size < 32)"; fatal = true; break; }
            if ((size_t)(end - p) < 4 + (size_t)bs) break;
            const uint8_t *r = p + 4;
            const int32_t tid = le32s(r);
            const int32_t pos0 = le32s(r + 4);
            const uint32_t l_name = r[8];
            uint32_t n_cig = le16(r + 12);
            const uint16_t flag = le16(r + 14);
            const uint32_t l_seq = le32(r + 16);
            const size_t fixed = 32;
            const size_t need = fixed + l_name + 4ull * n_cig + ((size_t)l_seq + 1) / 2 + l_seq;
            if (need > bs) { err = "corrupt record (fields exceed block_size)"; fatal = true; break; }
            bam->n_records++;
            if (tid >= 0 && (size_t)tid < bam->refs.size() && pos0 >= 0) {
                const uint8_t *cig = r + fixed + l_name;
                // real CIGAR parked in a CG tag? (htslib bam_tag2cigar)
                if (n_cig > 0 && (le32(cig) & 15u) == 4u && (le32(cig) >> 4) == l_seq) {
                    uint32_t n_real = 0;
                    const uint8_t *cg = find_cg_tag(r + need, r + bs, &n_real);
                    if (cg && n_real >= n_cig && n_real < (1u << 29)) { cig = cg; n_cig = n_real; }
                }
                RefReads &rr = bam->refs[(size_t)tid];
                int64_t ref_len = 0;
                const size_t base = rr.cigar.size();
                rr.cigar.resize(base + n_cig);
                uint32_t *dst = rr.cigar.data() + base;
                for (uint32_t k = 0; k < n_cig; ++k) {
                    const uint32_t op = le32(cig + 4ull * k);
                    dst[k] = op;
                    const uint32_t code = op & 15u;
                    if (code == 0 || code == 2 || code == 3 || code == 7 || code == 8) ref_len += op >> 4;
                }
                if (rr.cigar.size() > 0xfffffff0ull) { err = "more than 2^32 CIGAR ops on one reference"; fatal = true; break; }
                rr.pos.push_back(pos0 + 1);
                rr.flag.push_back(flag);
                rr.cig_off.push_back((uint32_t)rr.cigar.size());
                const int64_t e = (int64_t)pos0 + 1 + (ref_len > 0 ? ref_len : 1) - 1;
                if (e > rr.max_end) rr.max_end = e;
            }
            p += 4 + (size_t)bs;
        }
        return (size_t)(p - start);
    }
};

} // namespace

extern "C" int spl_bam_open(const char *path, int n_threads, spl_bam **out)
{
    if (!path || !out) return spl_set_error(SPL_ERR_ARG, "spl_bam_open: null argument");
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return spl_set_error(SPL_ERR_IO, "cannot open %s", path);
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) { close(fd); return spl_set_error(SPL_ERR_IO, "cannot stat %s (or empty file)", path); }
    const size_t fsize = (size_t)st.st_size;
    void *map = mmap(nullptr, fsize, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return spl_set_error(SPL_ERR_IO, "mmap failed for %s", path);
    madvise(map, fsize, MADV_SEQUENTIAL);
    const uint8_t *file = (const uint8_t *)map;

    // 1. block directory (headers only)
    std::vector<Block> blocks;
    size_t off = 0;
    int rc = SPL_OK;
    while (off < fsize) {
        if (fsize - off < 18 || file[off] != 0x1f || file[off + 1] != 0x8b || file[off + 2] != 8 || !(file[off + 3] & 4)) {
            rc = spl_set_error(SPL_ERR_FORMAT, "%s: not BGZF at offset %zu (BAM files are BGZF-compressed)", path, off);
            break;
        }
        const uint32_t xlen = le16(file + off + 10);
        if (fsize - off < 12 + (size_t)xlen) { rc = spl_set_error(SPL_ERR_FORMAT, "%s: truncated BGZF header", path); break; }
        uint32_t bsize = 0;
        bool have = false;
        for (size_t x = off + 12; x + 4 <= off + 12 + xlen;) {
            const uint32_t slen = le16(file + x + 2);
            if (file[x] == 'B' && file[x + 1] == 'C' && slen == 2) { bsize = (uint32_t)le16(file + x + 4) + 1; have = true; }
            x += 4 + slen;
        }
        if (!have || bsize < 12 + xlen + 8 || fsize - off < bsize) {
            rc = spl_set_error(SPL_ERR_FORMAT, "%s: corrupt or truncated BGZF block at offset %zu", path, off);
            break;
        }
        Block b;
        b.coff = off; b.csize = bsize; b.xlen = xlen; b.isize = le32(file + off + bsize - 4);
        if (b.isize > 65536) { rc = spl_set_error(SPL_ERR_FORMAT, "%s: BGZF ISIZE > 64 KiB at offset %zu", path, off); break; }
        blocks.push_back(b);
        off += bsize;
    }
    if (rc != SPL_OK) { munmap(map, fsize); return rc; }
    if (blocks.empty() || blocks.back().isize != 0) {
        // htslib only warns about a missing EOF marker; a truncated file is far more likely than a writer
        // that omits it, and silently counting fewer reads is the reference's worst failure mode: refuse.
        munmap(map, fsize);
        return spl_set_error(SPL_ERR_IO, "%s: BGZF EOF marker missing -- file is truncated", path);
    }

    spl_bam *bam = new (std::nothrow) spl_bam();
    if (!bam) { munmap(map, fsize); return spl_set_error(SPL_ERR_NOMEM, "out of host memory"); }
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads <= 0) n_threads = 1;

    // 2. segments of blocks: parallel inflate, then one sequential field-extraction pass.
    const size_t SEG_BLOCKS = 4096; // <= 256 MiB uncompressed
    std::vector<uint8_t> buf;
    size_t carry = 0; // bytes of an incomplete record kept at the front of buf
    Parser parser;
    parser.bam = bam;
    std::string fail;
    for (size_t b0 = 0; b0 < blocks.size() && fail.empty(); b0 += SEG_BLOCKS) {
        const size_t b1 = std::min(blocks.size(), b0 + SEG_BLOCKS);
        std::vector<size_t> uoff(b1 - b0 + 1);
        uoff[0] = carry;
        for (size_t i = b0; i < b1; ++i) uoff[i - b0 + 1] = uoff[i - b0] + blocks[i].isize;
        const size_t total = uoff[b1 - b0];
        if (buf.size() < total) buf.resize(total);
        std::atomic<size_t> next(b0);
        std::atomic<bool> bad(false);
        auto work = [&]() {
            void *ld = deflate_lib().ok ? deflate_lib().alloc() : nullptr;
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= b1) break;
                if (!inflate_block(file, blocks[i], buf.data() + uoff[i - b0], ld)) bad.store(true);
            }
            if (ld) deflate_lib().free_(ld);
        };
        const int nt = (int)std::min<size_t>((size_t)n_threads, b1 - b0);
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
        if (bad.load()) { fail = "inflate or CRC32 failure in a BGZF block (corrupt file)"; break; }
        bool fatal = false;
        const size_t used = parser.feed(buf.data(), buf.data() + total, fatal);
        if (fatal) { fail = parser.err; break; }
        carry = total - used;
        if (carry) memmove(buf.data(), buf.data() + used, carry);
    }
    munmap(map, fsize);
    if (fail.empty() && !parser.header_done) fail = "no BAM header found";
    if (fail.empty() && carry != 0) fail = "file ends inside a record (truncated)";
    if (!fail.empty()) {
        delete bam;
        return spl_set_error(SPL_ERR_FORMAT, "%s: %s", path, fail.c_str());
    }
    *out = bam;
    return SPL_OK;
}

extern "C" void spl_bam_close(spl_bam *bam) { delete bam; }
extern "C" int spl_bam_n_ref(const spl_bam *bam) { return bam ? (int)bam->refs.size() : 0; }
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
extern "C" int64_t spl_bam_n_records(const spl_bam *bam) { return bam ? bam->n_records : 0; }

extern "C" int spl_bam_reads(const spl_bam *bam, int tid, spl_reads *out, int64_t *max_end_out)
{
    if (!bam || !out) return spl_set_error(SPL_ERR_ARG, "spl_bam_reads: null argument");
    if (tid < 0 || (size_t)tid >= bam->refs.size()) return spl_set_error(SPL_ERR_ARG, "tid %d out of range", tid);
    const RefReads &rr = bam->refs[(size_t)tid];
    out->n_reads = (int64_t)rr.pos.size();
    out->pos = rr.pos.data();
    out->flag = rr.flag.data();
    out->cig_off = rr.cig_off.data();
    out->cigar = rr.cigar.data();
    if (max_end_out) *max_end_out = rr.max_end;
    return SPL_OK;
}
