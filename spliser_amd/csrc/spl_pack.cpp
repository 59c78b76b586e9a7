// spl_pack.cpp -- host packer: BAM-native reads -> the chunked, class-partitioned record layout of spl_pack.h.
// Pure host code (no HIP); used by the BAM decoder's consumers and by spl_reads_upload (spl_capi.cpp).
#include "spl_pack.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>

namespace splpack {

namespace {

using splrec::Rec;
using splrec::classify;

// Walks the reads [i0, i1) of a source across its parts.
template <class F>
inline void for_reads(const Source &src, int64_t i0, int64_t i1, F &&f)
{
    if (i0 >= i1) return;
    size_t k = (size_t)(std::upper_bound(src.first.begin(), src.first.end(), i0) - src.first.begin()) - 1;
    int64_t i = i0;
    while (i < i1) {
        const Part &p = src.parts[k];
        const int64_t base = src.first[k], stop = std::min<int64_t>(i1, base + p.n);
        for (; i < stop; ++i) {
            const int64_t j = i - base;
            const uint32_t o0 = p.cig_off[j];
            f(p.pos[j], (uint32_t)p.flag[j], p.cigar + o0, p.cig_off[j + 1] - o0);
        }
        ++k;
    }
}

struct PlanJob { const Source *src; Plan *plan; };

void plan_chunk(size_t c, void *arg)
{
    const PlanJob &job = *(const PlanJob *)arg;
    const Source &src = *job.src;
    ChunkDesc &d = job.plan->chunks[c];
    const int64_t chunk = (int64_t)job.plan->chunk;
    const int64_t i0 = (int64_t)c * chunk, i1 = std::min<int64_t>(src.n_reads, i0 + chunk);
    uint32_t n[SPL_RC_RUNS] = {0, 0, 0, 0}, cost = 0;
    uint64_t wide = 0;
    bool first = true;
    int32_t first_pos = 0;
    Rec r;
    for_reads(src, i0, i1, [&](int32_t pos, uint32_t flag, const uint32_t *ops, uint32_t n_ops) {
        if (first) { first_pos = pos; first = false; }
        classify(pos, flag, ops, n_ops, 0u, r);
        n[r.run]++;
        cost += r.weight;
        wide += r.n_wide;
    });
    for (int k = 0; k < SPL_RC_RUNS; ++k) d.n[k] = (uint16_t)n[k];
    d.first_pos = first_pos;
    d.cost = cost;
    d.wide_off = wide;                       // count for now; prefix sums below
    d.rec_off = spl_run_offset(d.n, 4);      // size for now
}

} // namespace

void parallel_for(size_t n, int n_threads, void (*fn)(size_t, void *), void *arg)
{
    if (n == 0) return;
    // blocks of consecutive items per grab: neighbouring chunks read neighbouring memory
    const size_t grain = std::max<size_t>(1, std::min<size_t>(64, n / (size_t)std::max(1, 4 * n_threads)));
    std::atomic<size_t> next(0);
    auto work = [&]() {
        for (;;) {
            const size_t a = next.fetch_add(grain);
            if (a >= n) break;
            const size_t b = std::min(n, a + grain);
            for (size_t k = a; k < b; ++k) fn(k, arg);
        }
    };
    const int nt = (int)std::min<size_t>((size_t)std::max(1, n_threads), (n + grain - 1) / grain);
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
}

void plan(const Source &src, Plan &out, int n_threads)
{
    const size_t n_chunks = (size_t)((src.n_reads + out.chunk - 1) / out.chunk);
    out.chunks.assign(n_chunks, ChunkDesc());
    PlanJob job{&src, &out};
    parallel_for(n_chunks, n_threads, plan_chunk, &job);
    uint64_t rec = 0, wide = 0;
    for (ChunkDesc &d : out.chunks) {
        const uint64_t bytes = d.rec_off, ops = d.wide_off;
        d.rec_off = rec;
        d.wide_off = wide;
        rec += bytes;
        wide += ops;
    }
    out.rec_bytes = rec;
    out.n_wide = wide;
}

void emit(const Source &src, const Plan &plan, size_t c0, size_t c1, uint8_t *rec_dst, uint32_t *wide_dst)
{
    Rec r;
    for (size_t c = c0; c < c1; ++c) {
        const ChunkDesc &d = plan.chunks[c];
        uint8_t *base = rec_dst + (d.rec_off - plan.chunks[c0].rec_off);
        uint8_t *run[SPL_RC_RUNS];
        for (int k = 0; k < SPL_RC_RUNS; ++k) run[k] = base + spl_run_offset(d.n, k);
        // (the padding between the runs is never read: leave it as it is)
        uint64_t wide_at = d.wide_off;
        uint32_t *wdst = wide_dst + (d.wide_off - plan.chunks[c0].wide_off);
        static const uint32_t rec_size[SPL_RC_RUNS] = {SPL_REC_SIMPLE, SPL_REC_MNM, SPL_REC_M2, SPL_REC_OTHER};
        const int64_t i0 = (int64_t)c * (int64_t)plan.chunk, i1 = std::min<int64_t>(src.n_reads, i0 + (int64_t)plan.chunk);
        for_reads(src, i0, i1, [&](int32_t pos, uint32_t flag, const uint32_t *ops, uint32_t n_ops) {
            classify(pos, flag, ops, n_ops, (uint32_t)wide_at, r);
            memcpy(run[r.run], r.w, rec_size[r.run]);
            run[r.run] += rec_size[r.run];
            if (r.n_wide) {
                memcpy(wdst, ops, sizeof(uint32_t) * r.n_wide);
                wdst += r.n_wide;
                wide_at += r.n_wide;
            }
        });
    }
}

} // namespace splpack
