// inflate_wave_host.cpp -- the body of spl_inflate_wave_kernel (spliser_amd/csrc/spl_inflate_wave.h) run on the host, a wave of
// 64 fibers per BGZF block (wave_emul.h): what tests/test_inflate_wave_host.py loads, and a command line over a whole .bam:
//   g++ -O1 -g -o /tmp/iw tests/hostsim/inflate_wave_host.cpp -lz && /tmp/iw file.bam [max_blocks]
#include <zlib.h>

#include <vector>

#define WAVE_EMUL_IMPLEMENTATION
#include "wave_emul.h"
#include "../../spliser_amd/csrc/spl_inflate_wave.h"

extern "C" int emul_inflate_blocks(const uint8_t *image, const spl_zblock *blocks, uint32_t n, uint8_t *out, uint32_t *status)
{
    static splz::Shared sh;
    for (uint32_t b = 0; b < n; ++b) {
        memset(&sh, 0xEE, sizeof sh); // (nothing may depend on what the wave before left)
        static std::vector<uint8_t> tokens(SPL_Z_TOKEN_STRIDE + 256);
        memset(tokens.data(), 0xEE, tokens.size());
        uint32_t n_tok = 0;
        const bool ok = wv::run_wave([&]() {
            uint32_t n = 0;
            const uint32_t st = splz::decode_block(sh, image, blocks[b], tokens.data(), n);
            if (wv::lane() == 0) { status[b] = st; n_tok = st == SPL_Z_OK ? n : 0u; }
        });
        if (!ok) return -1 - (int)b;
        // ... and once more with every tile's tokens by a writing pass of their own (what a tile falls back to): the same status,
        // the same token stream, byte for byte
        {
            static std::vector<uint8_t> tokens2(SPL_Z_TOKEN_STRIDE + 256);
            memset(tokens2.data(), 0xEE, tokens2.size());
            memset(&sh, 0xEE, sizeof sh);
            uint32_t st2 = 99, n2 = 0;
            const bool ok2 = wv::run_wave([&]() {
                uint32_t n = 0;
                const uint32_t st = splz::decode_block(sh, image, blocks[b], tokens2.data(), n, SPL_Z_TOKEN_STRIDE, splz::OPT_WRITING_PASS);
                if (wv::lane() == 0) { st2 = st; n2 = st == SPL_Z_OK ? n : 0u; }
            });
            if (!ok2) return -200000 - (int)b;
            if (st2 != status[b] || n2 != n_tok || (n_tok && memcmp(tokens.data(), tokens2.data(), n_tok) != 0)) return -300000 - (int)b;
        }
        // ... and once more the way the denser kernel does (spl_inflate_decode_dense_kernel: its wave's shared memory ends
        // TOKCAP - TOKCAP_SMALL bytes earlier): the same status, nothing written behind the smaller token room, and -- tiles may
        // be cut elsewhere, so the tokens may be other tokens -- the same bytes out of the copying
        {
            static std::vector<uint8_t> tokens3(SPL_Z_TOKEN_STRIDE + 256);
            memset(tokens3.data(), 0xEE, tokens3.size());
            memset(&sh, 0xEE, sizeof sh);
            uint32_t st3 = 99, n3 = 0;
            const bool ok3 = wv::run_wave([&]() {
                uint32_t n = 0;
                const uint32_t st = splz::decode_block<splz::TOKCAP_SMALL>(sh, image, blocks[b], tokens3.data(), n);
                if (wv::lane() == 0) { st3 = st; n3 = st == SPL_Z_OK ? n : 0u; }
            });
            if (!ok3) return -400000 - (int)b;
            if (st3 != status[b]) return -500000 - (int)b;
            for (uint32_t k = splz::TOKCAP_SMALL / 4u; k < splz::TOKCAP / 4u; ++k)
                if (sh.tok[k] != 0xEEEEEEEEu) return -600000 - (int)b; // (memory the denser kernel's wave does not have)
            if (st3 == SPL_Z_OK) {
                static std::vector<uint8_t> out3(65536 + 64);
                static uint8_t lane_lds3[splz::COPY_LANE_BYTES];
                memset(lane_lds3, 0xEE, sizeof lane_lds3);
                const uint32_t made3 = splz::copy_block(out3.data(), blocks[b].out_len, tokens3.data(), n3, lane_lds3, lane_lds3 + splz::RING_BYTES);
                static uint8_t lane_lds1[splz::COPY_LANE_BYTES];
                static std::vector<uint8_t> out1(65536 + 64);
                memset(lane_lds1, 0xEE, sizeof lane_lds1);
                const uint32_t made1 = splz::copy_block(out1.data(), blocks[b].out_len, tokens.data(), n_tok, lane_lds1, lane_lds1 + splz::RING_BYTES);
                if (made3 != made1 || (made1 == blocks[b].out_len && memcmp(out1.data(), out3.data(), made1) != 0)) return -700000 - (int)b;
            }
        }
        for (size_t k = SPL_Z_TOKEN_STRIDE; k < tokens.size(); ++k) // (a block's room for tokens ends where the next block's begins)
            if (tokens[k] != 0xEE) return -100000 - (int)b;
        if (status[b] == SPL_Z_OK) {
            static uint8_t lane_lds[splz::COPY_LANE_BYTES];
            memset(lane_lds, 0xEE, sizeof lane_lds);
            const uint32_t made = splz::copy_block(out + blocks[b].out, blocks[b].out_len, tokens.data(), n_tok, lane_lds, lane_lds + splz::RING_BYTES);
            if (made != blocks[b].out_len) status[b] = SPL_Z_SHORT;
        }
    }
    return 0;
}

#ifndef EMUL_NO_MAIN
int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s file.bam [max_blocks]\n", argv[0]); return 2; }
    const size_t max_blocks = argc > 2 ? (size_t)atoll(argv[2]) : (size_t)-1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END);
    const size_t fsize = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> file(fsize + 64, 0);
    if (fread(file.data(), 1, fsize, f) != fsize) return 1;
    fclose(f);
    size_t at = 0, nb = 0, bad = 0;
    while (at + 18 <= fsize && nb < max_blocks) {
        const uint8_t *h = file.data() + at;
        if (h[0] != 0x1f || h[1] != 0x8b) { fprintf(stderr, "not a BGZF block at %zu\n", at); return 1; }
        const size_t bsize = (size_t)(h[16] | h[17] << 8) + 1;
        const uint32_t isize = (uint32_t)h[bsize - 4] | (uint32_t)h[bsize - 3] << 8 | (uint32_t)h[bsize - 2] << 16 | (uint32_t)h[bsize - 1] << 24;
        spl_zblock zb{(uint64_t)at + 18, 0, (uint32_t)(bsize - 26), isize, 0, 0};
        std::vector<uint8_t> want(isize + 1), got(isize + 64, 0xA5);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        inflateInit2(&zs, -15);
        zs.next_in = const_cast<Bytef *>(h + 18); zs.avail_in = (uInt)(bsize - 26); zs.next_out = want.data(); zs.avail_out = isize + 1;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END) { fprintf(stderr, "zlib: block %zu does not inflate\n", nb); return 1; }
        uint32_t status = 99;
        const int er = emul_inflate_blocks(file.data(), &zb, 1, got.data(), &status);
        if (er || status != 0 || memcmp(got.data(), want.data(), isize) != 0 || got[isize] != 0xA5) {
            size_t k = 0;
            while (k < isize && got[k] == want[k]) ++k;
            fprintf(stderr, "block %zu (at %zu, %zu -> %u bytes): emulator %d, status %u, first difference at %zu\n", nb, at, bsize - 26, isize, er, status, k);
            if (++bad > 5) return 1;
        }
        at += bsize;
        nb++;
    }
    printf("%zu blocks, %zu bad\n", nb, bad);
    return bad ? 1 : 0;
}
#endif
