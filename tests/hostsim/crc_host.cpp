// Test-only host build of spliser_amd/csrc/spl_crc.h (the body of spl_crc32_kernel): tests/test_crc_host.py holds it against zlib.
#include "../../spliser_amd/csrc/spl_crc.h"

namespace {
uint32_t g_t[1024], g_x2n[splcrc::N_X2N];
bool g_built = false;
void build()
{
    if (g_built) return;
    for (uint32_t b = 0; b < 256; ++b) g_t[b] = splcrc::byte_entry(b);
    for (int k = 1; k < 4; ++k)
        for (uint32_t b = 0; b < 256; ++b) {
            const uint32_t c = g_t[(k - 1) * 256 + b];
            g_t[k * 256 + b] = (c >> 8) ^ g_t[c & 0xffu];
        }
    for (int k = 0; k < splcrc::N_X2N; ++k) g_x2n[k] = splcrc::x2n_entry((uint32_t)k);
    g_built = true;
}
} // namespace

extern "C" uint32_t crc_block(const uint8_t *p, uint32_t n, int streams)
{
    build();
    switch (streams) {
    case 1: return splcrc::block<1>(p, n, g_t, g_x2n);
    case 2: return splcrc::block<2>(p, n, g_t, g_x2n);
    case 4: return splcrc::block<4>(p, n, g_t, g_x2n);
    case 8: return splcrc::block<8>(p, n, g_t, g_x2n);
    }
    return 0;
}

extern "C" uint32_t crc_x2n(int k) { build(); return g_x2n[k]; }
extern "C" uint32_t crc_mulmod(uint32_t a, uint32_t b) { return splcrc::mulmod(a, b); }
