// Sanitizer driver for the host-side BAM code (no GPU needed): write a synthetic BAM with spl_bam_write, read it back
// with spl_bam_open on several thread counts, compare, then feed truncated / corrupted copies and expect clean errors.
// Built by tests/test_native_sanitizers.py with -fsanitize=address,undefined.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/spliser.h"

// (internal entry points of bam_reader.cpp, as spl_bam.h declares them: the device decoder's way to the whole block directory)
int spl_bam_walk_all(spl_bam *bam);
size_t spl_bam_block_count(const spl_bam *bam);

static std::string g_err;
int spl_set_error(int code, const char *fmt, ...) { g_err = fmt; return code; }

struct Ref { std::vector<int32_t> pos; std::vector<uint16_t> flag; std::vector<uint32_t> off{0}, cig; };

int main(int argc, char **argv)
{
    const char *dir = argc > 1 ? argv[1] : "/tmp";
    const int n_reads = argc > 2 ? atoi(argv[2]) : 300000;
    std::mt19937 rng(7);
    const char *names[3] = {"chrA", "chrB", "chrEmpty"};
    int64_t lens[3] = {50000000, 40000000, 1000};
    Ref refs[3];
    for (int t = 0; t < 2; ++t) {
        int32_t p = 100;
        for (int i = 0; i < n_reads / 2; ++i) {
            p += (int32_t)(rng() % 40);
            refs[t].pos.push_back(p);
            refs[t].flag.push_back((uint16_t)(rng() % 4096));
            const int kind = rng() % 10;
            if (kind < 5) refs[t].cig.push_back(150u << 4);
            else if (kind < 8) { refs[t].cig.push_back((uint32_t)(1 + rng() % 100) << 4); refs[t].cig.push_back(((uint32_t)(50 + rng() % 5000) << 4) | 3u); refs[t].cig.push_back((uint32_t)(1 + rng() % 100) << 4); }
            else if (kind < 9) { refs[t].cig.push_back((5u << 4) | 4u); refs[t].cig.push_back(100u << 4); refs[t].cig.push_back((2u << 4) | 1u); refs[t].cig.push_back(43u << 4); }
            refs[t].off.push_back((uint32_t)refs[t].cig.size()); // kind 9: no CIGAR at all ('*')
        }
    }
    spl_reads per[3];
    for (int t = 0; t < 3; ++t) {
        per[t].n_reads = (int64_t)refs[t].pos.size();
        per[t].pos = refs[t].pos.data(); per[t].flag = refs[t].flag.data(); per[t].cig_off = refs[t].off.data(); per[t].cigar = refs[t].cig.data();
    }
    const std::string path = std::string(dir) + "/asan.bam";
    if (spl_bam_write(path.c_str(), 3, names, lens, per, 1, 4) != 0) { fprintf(stderr, "write failed: %s\n", g_err.c_str()); return 1; }
    for (int threads : {1, 3, 8}) {
        spl_bam *b = nullptr;
        if (spl_bam_open(path.c_str(), threads, &b) != 0) { fprintf(stderr, "open failed: %s\n", g_err.c_str()); return 1; }
        if (spl_bam_n_ref(b) != 3 || spl_bam_n_records(b) != (int64_t)(refs[0].pos.size() + refs[1].pos.size())) { fprintf(stderr, "counts differ\n"); return 1; }
        for (int t = 0; t < 3; ++t) {
            spl_reads r; int64_t me = 0;
            spl_bam_reads(b, t, &r, &me);
            if (r.n_reads != per[t].n_reads || (r.n_reads && (memcmp(r.pos, per[t].pos, 4 * r.n_reads) || memcmp(r.flag, per[t].flag, 2 * r.n_reads) ||
                memcmp(r.cig_off, per[t].cig_off, 4 * (r.n_reads + 1)) || memcmp(r.cigar, per[t].cigar, 4 * per[t].cig_off[r.n_reads])))) { fprintf(stderr, "round trip differs on ref %d with %d threads\n", t, threads); return 1; }
        }
        spl_bam_close(b);
    }
    // the whole block directory at once, by several threads on stretches of the file and by one thread: the same blocks
    size_t n_blocks[2] = {0, 0};
    for (int one = 0; one < 2; ++one) {
        if (one) setenv("SPL_WALK_ONE_THREAD", "1", 1); else { unsetenv("SPL_WALK_ONE_THREAD"); setenv("SPL_WALK_PARALLEL_MIN", "65536", 1); }
        spl_bam *b = nullptr;
        if (spl_bam_open_deferred(path.c_str(), 8, &b) != 0) { fprintf(stderr, "deferred open failed: %s\n", g_err.c_str()); return 1; }
        if (spl_bam_walk_all(b) != 0) { fprintf(stderr, "walk failed: %s\n", g_err.c_str()); return 1; }
        n_blocks[one] = spl_bam_block_count(b);
        spl_bam_close(b);
    }
    unsetenv("SPL_WALK_ONE_THREAD");
    if (n_blocks[0] != n_blocks[1] || n_blocks[0] < 4) { fprintf(stderr, "directories differ: %zu blocks by several threads, %zu by one\n", n_blocks[0], n_blocks[1]); return 1; }
    // damaged copies: every one must fail cleanly (no crash, no sanitizer report)
    FILE *fh = fopen(path.c_str(), "rb");
    std::vector<unsigned char> data;
    unsigned char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), fh)) > 0) data.insert(data.end(), buf, buf + n);
    fclose(fh);
    int rejected = 0, accepted = 0;
    for (int trial = 0; trial < 40; ++trial) {
        std::vector<unsigned char> bad = data;
        if (trial < 10) bad.resize(bad.size() * (trial + 1) / 12);
        else for (int k = 0; k < 1 + trial % 5; ++k) bad[rng() % bad.size()] ^= (unsigned char)(1 + rng() % 255);
        const std::string bp = std::string(dir) + "/asan_bad.bam";
        fh = fopen(bp.c_str(), "wb"); fwrite(bad.data(), 1, bad.size(), fh); fclose(fh);
        spl_bam *b = nullptr;
        if (spl_bam_open(bp.c_str(), 4, &b) == 0) { ++accepted; spl_bam_close(b); } else ++rejected;
        b = nullptr;
        if (spl_bam_open_deferred(bp.c_str(), 4, &b) == 0) { (void)spl_bam_walk_all(b); spl_bam_close(b); } // (the stretch-wise walk on the same damage)
    }
    printf("ok: round trip on 3 thread counts; damaged copies rejected %d, accepted %d (a flipped bit inside an unused byte of the\n"
           "gzip header cannot be noticed; every payload bit is covered by CRC32)\n", rejected, accepted);
    return rejected >= 30 ? 0 : 1;
}
