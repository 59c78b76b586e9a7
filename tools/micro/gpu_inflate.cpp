// GPU DEFLATE of a BGZF file against zlib, block by block, with timings (development harness of csrc/spl_inflate.hip).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/gpu_inflate.cpp spliser_amd/csrc/spl_inflate.hip -lz -o /tmp/gpu_inflate
//   /tmp/gpu_inflate file.bam [blocks to check on the host, 0 = all]
#include <hip/hip_runtime.h>
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../spliser_amd/csrc/spl_inflate.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    if (argc < 2) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    fseek(f, 0, SEEK_END);
    const size_t fsize = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> file(fsize + 32, 0);
    if (fread(file.data(), 1, fsize, f) != fsize) return 1;
    fclose(f);
    std::vector<spl_zblock> blocks;
    uint64_t uoff = 0;
    for (size_t off = 0; off + 18 <= fsize;) {
        const uint32_t xlen = file[off + 10] | (file[off + 11] << 8);
        uint32_t bsize = 0;
        for (size_t x = off + 12; x + 4 <= off + 12 + xlen;) {
            const uint32_t slen = file[x + 2] | (file[x + 3] << 8);
            if (file[x] == 'B' && file[x + 1] == 'C' && slen == 2) bsize = (file[x + 4] | (file[x + 5] << 8)) + 1u;
            x += 4 + slen;
        }
        if (!bsize) { fprintf(stderr, "not BGZF at %zu\n", off); return 1; }
        spl_zblock b;
        b.in = off + 12 + xlen;
        b.in_len = bsize - 12 - xlen - 8;
        memcpy(&b.crc, &file[off + bsize - 8], 4);
        memcpy(&b.out_len, &file[off + bsize - 4], 4);
        b.out = uoff;
        b.pad = 0;
        uoff += b.out_len;
        blocks.push_back(b);
        off += bsize;
    }
    const uint32_t n = (uint32_t)blocks.size();
    printf("%s: %.1f MB, %u blocks, %.1f MB inflated\n", argv[1], fsize / 1e6, n, uoff / 1e6);
    uint8_t *d_img = nullptr, *d_out = nullptr;
    spl_zblock *d_blocks = nullptr;
    uint32_t *d_status = nullptr;
    CK(hipMalloc(&d_img, fsize + 32));
    CK(hipMalloc(&d_out, uoff + 64));
    CK(hipMalloc(&d_blocks, sizeof(spl_zblock) * n));
    CK(hipMalloc(&d_status, 4 * n));
    CK(hipMemcpy(d_img, file.data(), fsize + 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_blocks, blocks.data(), sizeof(spl_zblock) * n, hipMemcpyHostToDevice));
    CK(hipMemset(d_status, 0xff, 4 * n));
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        if (spl_dev_launch_inflate(d_img, d_blocks, n, d_out, d_status, nullptr)) return 3;
        CK(hipDeviceSynchronize());
        double t1 = now();
        if (spl_dev_launch_crc32(d_out, d_blocks, n, d_status, nullptr)) return 3;
        CK(hipDeviceSynchronize());
        double t2 = now();
        printf("inflate %.4f s = %.1f GB/s out, %.2f GB/s in; crc32 %.4f s = %.1f GB/s\n", t1 - t0, uoff / 1e9 / (t1 - t0), fsize / 1e9 / (t1 - t0),
               t2 - t1, uoff / 1e9 / (t2 - t1));
    }
    std::vector<uint32_t> status(n);
    CK(hipMemcpy(status.data(), d_status, 4 * n, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (status[i]) { if (bad < 5) printf("block %u: status %u (in_len %u out_len %u)\n", i, status[i], blocks[i].in_len, blocks[i].out_len); ++bad; }
    printf("%zu blocks with a status\n", bad);
    uint32_t check = argc > 2 ? (uint32_t)atoi(argv[2]) : 2000;
    if (check == 0 || check > n) check = n;
    std::vector<uint8_t> got(65536), want(65536);
    size_t diff = 0;
    const uint32_t step = n / check ? n / check : 1;
    for (uint32_t i = 0; i < n; i += step) {
        const spl_zblock &b = blocks[i];
        if (!b.out_len) continue;
        CK(hipMemcpy(got.data(), d_out + b.out, b.out_len, hipMemcpyDeviceToHost));
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        inflateInit2(&zs, -15);
        zs.next_in = file.data() + b.in; zs.avail_in = b.in_len; zs.next_out = want.data(); zs.avail_out = 65536;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || zs.total_out != b.out_len || memcmp(got.data(), want.data(), b.out_len) != 0) {
            if (diff < 5) printf("block %u differs (zlib rc %d, %lu bytes)\n", i, rc, zs.total_out);
            ++diff;
        }
    }
    printf("%zu of the checked blocks differ from zlib\n", diff);
    return (bad || diff) ? 4 : 0;
}
