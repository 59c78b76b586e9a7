// Test-only host build of spliser_amd/csrc/spl_crc_wave.h (the body of spl_crc32_wave_kernel) under the wave emulator: a wave of 64
// fibers per block; tests/test_crc_host.py holds it against zlib.
#include <vector>

#define WAVE_EMUL_IMPLEMENTATION
#include "wave_emul.h"
#include "../../spliser_amd/csrc/spl_crc_wave.h"

// the CRC32 of n_blocks stretches data[off[k] .. off[k] + len[k]) -> out[k]; < 0: the wave broke a rule of the emulator
extern "C" int crc_wave_blocks(const uint8_t *data, const uint64_t *off, const uint32_t *len, uint32_t n_blocks, uint32_t *out)
{
    static std::vector<uint32_t> t(splcrc::W_TABLE_WORDS + splcrc::W_SCRATCH_WORDS);
    bool ok = wv::run_wave([&]() { splcrc::wave_tables(t.data(), wv::lane(), 64u); });
    if (!ok) return -1;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        ok = wv::run_wave([&]() {
            const uint32_t c = splcrc::wave_block(data + off[b], len[b], t.data(), splcrc::wave_lane_factor(wv::lane()));
            if (wv::lane() == 7u) out[b] = c;
        });
        if (!ok) return -2 - (int)b;
    }
    return 0;
}
