// spl_wave.h -- the handful of wave-level primitives the wave-cooperative kernels are written against (spl_inflate_wave.h, ...).
//
// On the device they are the gfx950 operations themselves (one wave = 64 lanes = one workgroup, so a workgroup barrier is the
// wave's own).  tests/hostsim/wave_emul.h provides the same names on the host, 64 fibers taking turns, so that the kernel
// bodies -- the very same source -- run and are checked on a CPU (tests/test_inflate_wave_host.py): there is no GPU where this
// is built.  Rules the bodies keep, and the emulator enforces: every primitive below is called by all 64 lanes from wave-uniform
// control flow, and data goes from lane to lane through shared memory only across a wv_sync().
#ifndef SPL_WAVE_H
#define SPL_WAVE_H

#ifndef SPL_WAVE_EMUL
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WV_DEV __device__ __forceinline__
#define WV_SHARED_PTR(T) T *

namespace wv {
WV_DEV uint32_t lane() { return threadIdx.x & 63u; }
WV_DEV uint64_t ballot(bool p) { return __ballot(p); }
WV_DEV bool any(bool p) { return __ballot(p) != 0ull; }
WV_DEV uint32_t shfl(uint32_t v, uint32_t src_lane) { return (uint32_t)__shfl((int)v, (int)src_lane, 64); }
WV_DEV uint32_t shfl_up(uint32_t v, uint32_t delta) { return (uint32_t)__shfl_up((int)v, delta, 64); }
// a value every lane holds alike, said to the compiler (scalar registers, scalar branches)
WV_DEV uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
WV_DEV uint32_t readlane(uint32_t v, uint32_t uniform_lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane((int)uniform_lane)); }
// shared memory written before it is readable by every lane after it (the workgroup is one wave: its barrier costs nothing to wait for)
WV_DEV void sync() { __syncthreads(); }
WV_DEV uint32_t lds_max(uint32_t *p, uint32_t v) { return atomicMax(p, v); }
WV_DEV uint32_t lds_or(uint32_t *p, uint32_t v) { return atomicOr(p, v); }
WV_DEV uint32_t popc64(uint64_t m) { return (uint32_t)__popcll(m); }
WV_DEV uint32_t ffs64(uint64_t m) { return (uint32_t)__ffsll((long long)m) - 1u; } // index of the lowest set bit (m != 0)
WV_DEV uint32_t brev32(uint32_t v) { return __brev(v); }
// inclusive prefix sum over the lanes
WV_DEV uint32_t scan_add(uint32_t v)
{
    const uint32_t l = lane();
#pragma unroll
    for (uint32_t s = 1; s < 64u; s <<= 1) {
        const uint32_t up = shfl_up(v, s);
        if (l >= s) v += up;
    }
    return v;
}
// unaligned accesses to global memory (the hardware takes them: the compiler is told the alignment is 1)
WV_DEV uint32_t ld32(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
WV_DEV uint64_t ld64(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
struct alignas(4) u128 { uint32_t w[4]; };
WV_DEV void ld128(const uint8_t *p, uint64_t &lo, uint64_t &hi) { u128 v; __builtin_memcpy(&v, p, 16); lo = (uint64_t)v.w[0] | (uint64_t)v.w[1] << 32; hi = (uint64_t)v.w[2] | (uint64_t)v.w[3] << 32; }
WV_DEV void st128(uint8_t *p, uint64_t lo, uint64_t hi) { u128 v; v.w[0] = (uint32_t)lo; v.w[1] = (uint32_t)(lo >> 32); v.w[2] = (uint32_t)hi; v.w[3] = (uint32_t)(hi >> 32); __builtin_memcpy(p, &v, 16); }
// ... and to shared / to device memory when the compiler cannot tell which of the two a pointer means (it would use FLAT
// instructions, whose every wait is for shared AND device memory: a load that is meant to stay in flight would be waited for)
#define WV_AS3 __attribute__((address_space(3)))
#define WV_AS1 __attribute__((address_space(1)))
struct __attribute__((packed, aligned(1))) pk128 { uint64_t a, b; };
WV_DEV void lds_ld128(const uint8_t *p, uint64_t &lo, uint64_t &hi) { const WV_AS3 pk128 *q = (const WV_AS3 pk128 *)p; lo = q->a; hi = q->b; }
WV_DEV void lds_st128(uint8_t *p, uint64_t lo, uint64_t hi) { WV_AS3 pk128 *q = (WV_AS3 pk128 *)p; q->a = lo; q->b = hi; }
WV_DEV uint32_t lds_ld8(const uint8_t *p) { return *(const WV_AS3 uint8_t *)p; }
WV_DEV void mem_ld128(const uint8_t *p, uint64_t &lo, uint64_t &hi) { const WV_AS1 pk128 *q = (const WV_AS1 pk128 *)p; lo = q->a; hi = q->b; }
WV_DEV void mem_st128(uint8_t *p, uint64_t lo, uint64_t hi) { WV_AS1 pk128 *q = (WV_AS1 pk128 *)p; q->a = lo; q->b = hi; }
// Sixteen bytes from device memory that are NOT waited for where the compiler would (its waits are for everything in flight: a
// load meant to arrive eight turns later would be waited for at the next branch): the load is an instruction the compiler does
// not look into, its registers stay untouched until mem_wait4 has waited for them, and only then are they read.
typedef uint32_t q128 __attribute__((ext_vector_type(4)));
WV_DEV void mem_ld128_async(const uint8_t *p, q128 &v) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); }
WV_DEV void mem_wait4(q128 &a, q128 &b, q128 &c, q128 &d) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory"); }
WV_DEV void lds_st128q(uint8_t *p, const q128 &v) { *(WV_AS3 q128 *)p = v; }
WV_DEV void settle64(uint64_t &v) { asm volatile("" : "+v"(v)); }
// a value that was loaded from memory is there from here on: said before a loop that keeps loads in flight, so that the wait for
// this one does not end up inside it (where it would wait for everything)
WV_DEV void settle(uint32_t &v) { asm volatile("" : "+v"(v)); }
WV_DEV void st16(uint8_t *p, uint32_t v) { const uint16_t x = (uint16_t)v; __builtin_memcpy(p, &x, 2); }
WV_DEV void st32(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
WV_DEV void st64(uint8_t *p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
} // namespace wv
#endif // !SPL_WAVE_EMUL

#endif
