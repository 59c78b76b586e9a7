// wave_emul.h -- spliser_amd/csrc/spl_wave.h on the host: a wave is 64 fibers (ucontext) that take turns, lane 0 first, each
// running until it reaches a primitive; a primitive is a rendezvous of all 64 lanes.  The kernel bodies of spl_inflate_wave.h
// and friends compile against this unchanged, which is how they are tested where there is no GPU.
//
// Stricter than the hardware on purpose: a primitive that is not reached by all 64 lanes, or reached at different places
// (file:line), or a wv::uni() whose lanes disagree, ends the run with a message -- on a GPU those are silent wrong answers.
// Looser in one respect: between two rendezvous the lanes run one after the other, not in lockstep, so an exchange through shared
// memory that lacks its wv::sync() can go unnoticed when the reader happens to run after the writer; the emulator therefore runs
// the lanes in DESCENDING order on every other interval (SPL_WAVE_EMUL_ORDER=0/1 forces one order).
#ifndef WAVE_EMUL_H
#define WAVE_EMUL_H
#define SPL_WAVE_EMUL 1

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <functional>
#include <vector>

#define WV_DEV static inline
#define WV_SHARED_PTR(T) T *

namespace wv {

struct Wave {
    static const int N = 64;
    ucontext_t main_ctx, ctx[N];
    std::vector<char> stacks;
    bool alive[N];
    const char *site[N];
    int line[N];
    int cur = 0;
    uint64_t slot[N];          // what each lane brings to the rendezvous
    uint64_t result[N];        // what it takes away
    std::function<void()> body;
    bool failed = false;
    uint64_t intervals = 0;
};

extern Wave *g_wave;

inline void fiber_entry()
{
    Wave *w = g_wave;
    w->body();
    w->alive[w->cur] = false;
    swapcontext(&w->ctx[w->cur], &w->main_ctx);
}

// all lanes stop here; when the last has arrived, `combine` (run once, by the scheduler) turns slot[] into result[]
inline void rendezvous(const char *file, int line)
{
    Wave *w = g_wave;
    w->site[w->cur] = file;
    w->line[w->cur] = line;
    swapcontext(&w->ctx[w->cur], &w->main_ctx);
}

enum Op { OP_NONE, OP_BALLOT, OP_SHFL, OP_SHFL_UP, OP_UNI, OP_READLANE, OP_SYNC };
struct Pending { Op op; };
extern Op g_op;
extern uint64_t g_arg[64];

// Runs `body` as one wave.  Returns false if the wave broke a rule.
inline bool run_wave(const std::function<void()> &body)
{
    static Wave wave;
    Wave *w = &wave;
    g_wave = w;
    const size_t STACK = 256 * 1024;
    if (w->stacks.size() != STACK * Wave::N) w->stacks.resize(STACK * Wave::N);
    w->body = body;
    w->failed = false;
    for (int l = 0; l < Wave::N; ++l) {
        getcontext(&w->ctx[l]);
        w->ctx[l].uc_stack.ss_sp = w->stacks.data() + STACK * (size_t)l;
        w->ctx[l].uc_stack.ss_size = STACK;
        w->ctx[l].uc_link = &w->main_ctx;
        makecontext(&w->ctx[l], (void (*)())fiber_entry, 0);
        w->alive[l] = true;
        w->site[l] = nullptr;
        w->line[l] = 0;
    }
    static int forced = getenv("SPL_WAVE_EMUL_ORDER") ? atoi(getenv("SPL_WAVE_EMUL_ORDER")) : -1;
    for (;;) {
        const bool descending = forced >= 0 ? forced == 1 : (w->intervals & 1u) != 0;
        w->intervals++;
        g_op = OP_NONE;
        for (int k = 0; k < Wave::N; ++k) {
            const int l = descending ? Wave::N - 1 - k : k;
            if (!w->alive[l]) continue;
            w->cur = l;
            w->site[l] = nullptr;
            swapcontext(&w->main_ctx, &w->ctx[l]);
        }
        int n_alive = 0, first = -1;
        for (int l = 0; l < Wave::N; ++l) if (w->alive[l]) { ++n_alive; if (first < 0) first = l; }
        if (n_alive == 0) break;
        if (n_alive != Wave::N) {
            fprintf(stderr, "wave_emul: %d lanes left the kernel while lane %d waits at %s:%d\n", Wave::N - n_alive, first, w->site[first], w->line[first]);
            w->failed = true;
            break;
        }
        for (int l = 1; l < Wave::N; ++l)
            if (strcmp(w->site[l], w->site[0]) != 0 || w->line[l] != w->line[0]) {
                fprintf(stderr, "wave_emul: lanes at different primitives: lane 0 at %s:%d, lane %d at %s:%d\n", w->site[0], w->line[0], l, w->site[l], w->line[l]);
                w->failed = true;
            }
        if (w->failed) break;
        // combine
        switch (g_op) {
        case OP_BALLOT: {
            uint64_t m = 0;
            for (int l = 0; l < 64; ++l) if (w->slot[l]) m |= 1ull << l;
            for (int l = 0; l < 64; ++l) w->result[l] = m;
            break;
        }
        case OP_SHFL:
            for (int l = 0; l < 64; ++l) w->result[l] = w->slot[g_arg[l] & 63u];
            break;
        case OP_SHFL_UP:
            for (int l = 0; l < 64; ++l) w->result[l] = (uint64_t)l >= g_arg[l] ? w->slot[l - (int)g_arg[l]] : w->slot[l];
            break;
        case OP_UNI:
            for (int l = 1; l < 64; ++l)
                if (w->slot[l] != w->slot[0]) {
                    fprintf(stderr, "wave_emul: wv::uni at %s:%d: lane 0 has %llu, lane %d has %llu\n", w->site[0], w->line[0], (unsigned long long)w->slot[0], l, (unsigned long long)w->slot[l]);
                    w->failed = true;
                }
            for (int l = 0; l < 64; ++l) w->result[l] = w->slot[0];
            break;
        case OP_READLANE:
            for (int l = 1; l < 64; ++l)
                if (g_arg[l] != g_arg[0]) { fprintf(stderr, "wave_emul: wv::readlane at %s:%d with a lane index that is not uniform\n", w->site[0], w->line[0]); w->failed = true; }
            for (int l = 0; l < 64; ++l) w->result[l] = w->slot[g_arg[0] & 63u];
            break;
        default:
            break;
        }
        if (w->failed) break;
    }
    return !w->failed;
}

// (the call site travels into the rendezvous by way of default arguments: no macros, so the names stay ordinary functions)
#define WV_SITE const char *f = __builtin_FILE(), int ln = __builtin_LINE()
inline uint32_t lane() { return (uint32_t)g_wave->cur; }
inline uint64_t ballot(bool p, WV_SITE) { Wave *w = g_wave; w->slot[w->cur] = p; g_op = OP_BALLOT; rendezvous(f, ln); const uint64_t r = w->result[w->cur]; g_op = OP_SYNC; rendezvous(f, -ln); return r; }
inline bool any(bool p, WV_SITE) { return ballot(p, f, ln) != 0ull; }
inline uint32_t shfl(uint32_t v, uint32_t src, WV_SITE) { Wave *w = g_wave; w->slot[w->cur] = v; g_arg[w->cur] = src; g_op = OP_SHFL; rendezvous(f, ln); const uint32_t r = (uint32_t)w->result[w->cur]; g_op = OP_SYNC; rendezvous(f, -ln); return r; }
inline uint32_t shfl_up(uint32_t v, uint32_t d, WV_SITE) { Wave *w = g_wave; w->slot[w->cur] = v; g_arg[w->cur] = d; g_op = OP_SHFL_UP; rendezvous(f, ln); const uint32_t r = (uint32_t)w->result[w->cur]; g_op = OP_SYNC; rendezvous(f, -ln); return r; }
inline uint32_t uni(uint32_t v, WV_SITE) { Wave *w = g_wave; w->slot[w->cur] = v; g_op = OP_UNI; rendezvous(f, ln); const uint32_t r = (uint32_t)w->result[w->cur]; g_op = OP_SYNC; rendezvous(f, -ln); return r; }
inline uint32_t readlane(uint32_t v, uint32_t ln_idx, WV_SITE) { Wave *w = g_wave; w->slot[w->cur] = v; g_arg[w->cur] = ln_idx; g_op = OP_READLANE; rendezvous(f, ln); const uint32_t r = (uint32_t)w->result[w->cur]; g_op = OP_SYNC; rendezvous(f, -ln); return r; }
inline void sync(WV_SITE) { g_op = OP_SYNC; rendezvous(f, ln); }

inline uint32_t lds_max(uint32_t *p, uint32_t v) { const uint32_t o = *p; if (v > o) *p = v; return o; }
inline uint32_t lds_or(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = o | v; return o; }
inline uint32_t popc64(uint64_t m) { return (uint32_t)__builtin_popcountll(m); }
inline uint32_t ffs64(uint64_t m) { return (uint32_t)__builtin_ctzll(m); }
inline uint32_t brev32(uint32_t v)
{
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
    return __builtin_bswap32(v);
}
inline uint32_t scan_add(uint32_t v, WV_SITE)
{
    const uint32_t l = lane();
    for (uint32_t s = 1; s < 64u; s <<= 1) {
        const uint32_t up = shfl_up(v, s, f, ln * 100 + (int)s);
        if (l >= s) v += up;
    }
    return v;
}
inline uint32_t ld32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t ld64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline void ld128(const uint8_t *p, uint64_t &lo, uint64_t &hi) { memcpy(&lo, p, 8); memcpy(&hi, p + 8, 8); }
inline void st128(uint8_t *p, uint64_t lo, uint64_t hi) { memcpy(p, &lo, 8); memcpy(p + 8, &hi, 8); }
inline void lds_ld128(const uint8_t *p, uint64_t &lo, uint64_t &hi) { ld128(p, lo, hi); }
inline void lds_st128(uint8_t *p, uint64_t lo, uint64_t hi) { st128(p, lo, hi); }
inline uint32_t lds_ld8(const uint8_t *p) { return *p; }
inline void mem_ld128(const uint8_t *p, uint64_t &lo, uint64_t &hi) { ld128(p, lo, hi); }
inline void mem_st128(uint8_t *p, uint64_t lo, uint64_t hi) { st128(p, lo, hi); }
inline void settle(uint32_t &) {}
inline void settle64(uint64_t &) {}
struct q128 { uint8_t b[16]; };
inline void mem_ld128_async(const uint8_t *p, q128 &v) { memcpy(v.b, p, 16); }
inline void mem_wait4(q128 &, q128 &, q128 &, q128 &) {}
inline void lds_st128q(uint8_t *p, const q128 &v) { memcpy(p, v.b, 16); }
inline void st16(uint8_t *p, uint32_t v) { const uint16_t x = (uint16_t)v; memcpy(p, &x, 2); }
inline void st32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }
inline void st64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }

} // namespace wv

#ifdef WAVE_EMUL_IMPLEMENTATION
namespace wv {
Wave *g_wave = nullptr;
Op g_op = OP_NONE;
uint64_t g_arg[64];
} // namespace wv
#endif

#endif
