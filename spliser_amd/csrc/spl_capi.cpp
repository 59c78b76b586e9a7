// spl_capi.cpp -- the C ABI of include/spliser.h over the HIP runtime: contexts, HBM residency of site
// tables and read sets, kernel launches, result download.  No torch, no CPU compute fallback.
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <unistd.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/spliser.h"
#include "spl_bam.h"
#include "spl_inflate.h"
#include "spl_devpack.h"
#include "spl_device.h"
#include "spl_error.h"
#include "spl_pack.h"

// ---- error plumbing -------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

int spl_set_error(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return spl_set_error(SPL_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" const char *spl_last_error(void) { return g_last_error.c_str(); }
extern "C" int spl_abi_version(void) { return SPL_ABI_VERSION; }

// ---- device memory ---------------------------------------------------------------------------------------
// Device memory handed out here is kept by the process when it is given back, for whoever asks next on the same device: fresh
// VRAM costs 30 ms per GB on this stack (cleared on its way out: hipMalloc of the 56 GB a 200 M-read file inflates to takes
// 1.8 s in a new process) and freed VRAM is scrubbed by the copy engines at the next upload's expense (a decode that started
// while 18 GB of the previous one were being scrubbed had its upload run at 4 GB/s instead of 35).  A second `process` call of
// the same process -- `combine`'s samples, a service -- finds its read sets', decoded arrays' and inflated stream's memory
// where the first left it.  Best fit, at most twice what was asked for; buffers below 8 MiB are not worth keeping; the pool
// holds at most half of what was free on the device when it was first asked (SPL_DEV_CACHE_GB overrides, 0 = keep nothing) and
// lets its smallest buffers go first; when the device runs out, everything held is given back before the request fails;
// spl_trim() gives it all back on request (a service between two jobs).  put() waits for the device like the hipFree it stands
// in for.
// (Stream-ordered allocation -- hipMallocAsync / hipFreeAsync -- was tried first: with that pool the BAM decode became flaky
// on this stack, stale reference ids in one run of four.)
// SPL_DEV_POISON=1 (tests): every buffer is filled with 0xA5 when handed out -- fresh device memory is zero on this stack and
// reused memory is not, and nothing may depend on either.
namespace devmem {
struct Held { int device; void *p; size_t bytes; };
static std::mutex &mu() { static std::mutex m; return m; }
static std::vector<Held> &held() { static std::vector<Held> *v = new std::vector<Held>(); return *v; }
static std::vector<Held> &lent() { static std::vector<Held> *v = new std::vector<Held>(); return *v; }
static size_t limit(int device)
{
    static const long long env = []() { const char *e = getenv("SPL_DEV_CACHE_GB"); return e ? (long long)std::max(0, atoi(e)) << 30 : -1LL; }();
    if (env >= 0) return (size_t)env;
    static std::mutex m;
    static std::vector<size_t> per; // by device: half of what was free when the pool was first asked there
    std::lock_guard<std::mutex> lock(m);
    if ((size_t)device >= per.size()) per.resize((size_t)device + 1, 0);
    if (per[(size_t)device] == 0) {
        size_t free_b = 0, total_b = 0;
        int cur = 0;
        const bool have = hipGetDevice(&cur) == hipSuccess;
        if (have && cur != device) (void)hipSetDevice(device);
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)64 << 30; }
        if (have && cur != device) (void)hipSetDevice(cur);
        per[(size_t)device] = std::max<size_t>(free_b / 2, (size_t)1 << 30);
    }
    return per[(size_t)device];
}
static size_t held_bytes(int device)
{
    std::lock_guard<std::mutex> lock(mu());
    size_t n = 0;
    for (const Held &h : held()) if (h.device == device) n += h.bytes;
    return n;
}
static void flush(int device)
{
    std::vector<Held> go;
    {
        std::lock_guard<std::mutex> lock(mu());
        std::vector<Held> &v = held();
        for (size_t k = v.size(); k-- > 0;)
            if (v[k].device == device) { go.push_back(v[k]); v.erase(v.begin() + (long)k); }
    }
    for (const Held &h : go) (void)hipFree(h.p);
}
static hipError_t get(void **out, size_t bytes, char tag = 'x')
{
    static const char *const poison_env = getenv("SPL_DEV_POISON");
    const bool poison = poison_env && (poison_env[0] == '1' || strchr(poison_env, tag));
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    const size_t gran = (size_t)2 << 20;
    bytes = (std::max<size_t>(bytes, 1) + gran - 1) / gran * gran;
    void *p = nullptr;
    size_t cap = bytes;
    {
        std::lock_guard<std::mutex> lock(mu());
        std::vector<Held> &v = held();
        size_t best = v.size();
        for (size_t k = 0; k < v.size(); ++k)
            if (v[k].device == device && v[k].bytes >= bytes && v[k].bytes <= 2 * bytes && (best == v.size() || v[k].bytes < v[best].bytes)) best = k;
        if (best < v.size()) { p = v[best].p; cap = v[best].bytes; v.erase(v.begin() + (long)best); }
    }
    if (!p) {
        static const bool timing = getenv("SPL_DEV_TIMING") != nullptr; // (diagnostic: what the driver took for memory the pool did not have)
        const auto t0 = std::chrono::steady_clock::now();
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); flush(device); e = hipMalloc(&p, bytes); }
        if (e != hipSuccess) return e;
        if (timing)
            fprintf(stderr, "[devmem] hipMalloc of %.1f MB ('%c') on device %d: %.2f ms; the pool holds %.1f MB\n", (double)bytes / 1e6, tag, device,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3, (double)held_bytes(device) / 1e6);
    }
    { std::lock_guard<std::mutex> lock(mu()); lent().push_back(Held{device, p, cap}); }
    if (poison) { // (and done before anybody's stream touches the buffer: a memset on the null stream does not wait for, or hold up, the others)
        e = hipMemset(p, 0xA5, cap);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) return e;
    }
    *out = p;
    return hipSuccess;
}
static void put(void *p)
{
    if (!p) return;
    Held h{-1, p, 0};
    std::vector<Held> go;
    {
        std::lock_guard<std::mutex> lock(mu());
        std::vector<Held> &l = lent();
        for (size_t k = 0; k < l.size(); ++k)
            if (l[k].p == p) { h = l[k]; l.erase(l.begin() + (long)k); break; }
    }
    if (h.device < 0 || h.bytes > limit(h.device)) { (void)hipFree(p); return; }
    if (h.bytes < ((size_t)8 << 20)) {
        // small ones (chunk descriptors, a decode's lists and counters: a few dozen a call) are kept as well, up to a number: a
        // hipFree waits for the device and then takes 0.3 ms -- twelve of them between two shards' counting passes were 4 ms of
        // an idle device (profiles/r04ab_tail.txt)
        std::lock_guard<std::mutex> lock(mu());
        size_t n_small = 0;
        for (const Held &x : held()) if (x.device == h.device && x.bytes < ((size_t)8 << 20)) ++n_small;
        if (n_small >= 256) { (void)hipFree(p); return; }
    }
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    if (have && cur != h.device) (void)hipSetDevice(h.device);
    (void)hipDeviceSynchronize(); // (nothing in flight may still be using it when the next owner gets it)
    if (have && cur != h.device) (void)hipSetDevice(cur);
    {
        std::lock_guard<std::mutex> lock(mu());
        std::vector<Held> &v = held();
        v.push_back(h);
        size_t total = 0;
        for (const Held &x : v) if (x.device == h.device) total += x.bytes;
        const size_t lim = limit(h.device);
        while (total > lim) { // the smallest go first: the large ones are the expensive ones
            size_t s = v.size();
            for (size_t k = 0; k < v.size(); ++k)
                if (v[k].device == h.device && (s == v.size() || v[k].bytes < v[s].bytes)) s = k;
            total -= v[s].bytes;
            go.push_back(v[s]);
            v.erase(v.begin() + (long)s);
        }
    }
    for (const Held &x : go) (void)hipFree(x.p);
}
} // namespace devmem

// Device memory the process keeps for its next call (devmem above) goes back to the driver: device_id, or -1 for every device.
extern "C" int spl_trim(int device_id)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return SPL_OK; }
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    for (int d = 0; d < n; ++d)
        if (device_id < 0 || d == device_id) { (void)hipSetDevice(d); devmem::flush(d); }
    if (have) (void)hipSetDevice(cur);
    return SPL_OK;
}

struct spl_dsites;
struct spl_dreads;
extern "C" int spl_count_algorithmic_bytes(const spl_dsites *ds, const spl_dreads *dr, int64_t *out);

// ---- the library's own stopwatch over its kernels -----------------------------------------------------------------------
// spl_prof_enable(1): from then on every kernel launch of the ingest path (device decode, device packer) and every counting pass
// is bracketed by two events on its stream; spl_prof_report sums them by name.  What bench.py's end-to-end legs quote as their
// `kernels` table: durations measured where the kernels run, with the bytes each is given to work on (algorithmic: the stretch
// of the stream, or the reads, it is responsible for -- not what it happens to fetch).
namespace splprof {
struct Rec { const char *name; int device; hipEvent_t a, b; double bytes; };
static std::atomic<int> g_on{0};
static std::mutex &mu() { static std::mutex m; return m; }
static std::vector<Rec> &recs() { static std::vector<Rec> *v = new std::vector<Rec>(); return *v; }
static std::vector<std::pair<int, hipEvent_t>> &pool() { static std::vector<std::pair<int, hipEvent_t>> *v = new std::vector<std::pair<int, hipEvent_t>>(); return *v; }
static hipEvent_t take(int device)
{
    {
        std::lock_guard<std::mutex> lock(mu());
        std::vector<std::pair<int, hipEvent_t>> &p = pool();
        for (size_t k = 0; k < p.size(); ++k)
            if (p[k].first == device) { hipEvent_t e = p[k].second; p.erase(p.begin() + (long)k); return e; }
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
struct Scope { // (the current device is the stream's)
    const char *name; hipStream_t st; double bytes; hipEvent_t a = nullptr, b = nullptr; int device = 0;
    Scope(const char *n, hipStream_t s, double by) : name(n), st(s), bytes(by)
    {
        if (!g_on.load(std::memory_order_relaxed)) return;
        if (hipGetDevice(&device) != hipSuccess) return;
        a = take(device); b = take(device);
        if (a && b) (void)hipEventRecord(a, st);
    }
    ~Scope()
    {
        if (!a || !b) return;
        (void)hipEventRecord(b, st);
        std::lock_guard<std::mutex> lock(mu());
        recs().push_back(Rec{name, device, a, b, bytes});
    }
};
} // namespace splprof

extern "C" int spl_prof_enable(int on)
{
    std::lock_guard<std::mutex> lock(splprof::mu());
    for (const splprof::Rec &r : splprof::recs()) { splprof::pool().push_back({r.device, r.a}); splprof::pool().push_back({r.device, r.b}); }
    splprof::recs().clear();
    splprof::g_on.store(on ? 1 : 0);
    return SPL_OK;
}

// JSON text: [{"kernel": ..., "calls": n, "ms": sum, "bytes": sum}, ...] (most time first).  Waits for the devices.  Returns the
// length the text needs (it is cut to cap - 1 characters).
extern "C" int spl_prof_report(char *buf, int cap)
{
    std::vector<splprof::Rec> recs;
    { std::lock_guard<std::mutex> lock(splprof::mu()); recs = splprof::recs(); }
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    struct Sum { std::string name; int calls; double ms, bytes; };
    std::vector<Sum> sums;
    for (const splprof::Rec &r : recs) {
        (void)hipSetDevice(r.device);
        float ms = 0;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) { (void)hipGetLastError(); continue; }
        size_t k = 0;
        while (k < sums.size() && sums[k].name != r.name) ++k;
        if (k == sums.size()) sums.push_back(Sum{r.name, 0, 0, 0});
        sums[k].calls++; sums[k].ms += ms; sums[k].bytes += r.bytes;
    }
    if (have) (void)hipSetDevice(cur);
    std::sort(sums.begin(), sums.end(), [](const Sum &a, const Sum &b) { return a.ms > b.ms; });
    std::string out = "[";
    for (size_t k = 0; k < sums.size(); ++k) {
        char line[256];
        snprintf(line, sizeof line, "%s{\"kernel\": \"%s\", \"calls\": %d, \"ms\": %.4f, \"bytes\": %.0f}", k ? ", " : "", sums[k].name.c_str(), sums[k].calls, sums[k].ms, sums[k].bytes);
        out += line;
    }
    out += "]";
    if (buf && cap > 0) { strncpy(buf, out.c_str(), (size_t)cap - 1); buf[cap - 1] = 0; }
    return (int)out.size() + 1;
}

// ---- objects --------------------------------------------------------------------------------------------
struct spl_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // The tail of a counting pass (literal kernel, scan + SSE) runs on a stream of its own, so that the range kernel of the
    // NEXT pass -- of the next shard, sample or step -- starts as soon as this pass's range kernel is done: the tail is a
    // few thousand waves waiting on memory and fits next to it.  Pass n's tail waits for ev_range[n % 4]; pass n's range
    // kernel waits for ev_tail[(n - 2) % 4], the tail that last read the queue buffer and cleared the counter copy it is
    // about to use (three counter copies per site table, two queue buffers per read set).  SPL_TAIL_STREAM=0 when the context
    // is created: one stream.  (A cross-queue dependency still costs ~11 us between two range kernels on this stack, which
    // leaves 4...9 % of the 20 % there is to gain: DESIGN.md section 6.)
    hipStream_t tail = nullptr;
    hipEvent_t ev_range[4] = {nullptr, nullptr, nullptr, nullptr}, ev_tail[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t n_pass = 0;       // counting passes with a tail launched so far
    bool tail_host_wait = false;
    bool tail_pending = false; // the main stream has not been made to wait for the last tail yet
    int32_t *d_err = nullptr;               // error word of launches that are not counting passes (spl_junctions)
    // Read sets reach the device through a ring of page-locked staging buffers: the host packer (spl_pack.h) writes a piece of
    // a segment's records into one of them while the DMA engine drains the others on a stream of their own.  Locked with
    // hipHostRegister: on this stack that costs 0.7 ms per 32 MiB where hipHostMalloc costs 5 (tools/micro/hostmem.cpp), and
    // copies run at 50...55 GB/s from pieces of this size (20 GB/s from 8 MiB pieces).
    struct Stage { char *host = nullptr; size_t bytes = 0; hipEvent_t done = nullptr; bool busy = false, locked = false; };
    size_t stage_mb = 32; // size of a staging buffer (set when the ring is made)
    std::vector<Stage> stage;
    size_t stage_next = 0;
    hipStream_t copy = nullptr;
    hipEvent_t ev_copy = nullptr;
    int pack_threads = 1;
    // SPL_STAGE_TIMING=1 (diagnostic: every piece is waited for, so packing and copying no longer overlap): what the pieces of a
    // read set took to pack and to copy, printed when the read set is finished
    bool stage_timing = false;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    double tm_plan_s = 0, tm_emit_s = 0, tm_copy_ms = 0;
    size_t tm_bytes = 0, tm_pieces = 0;
    int last_grid = 0, last_lds = 0, last_variant = 0;
    struct Junction { int32_t left, right; uint8_t strand; uint32_t count, anchor_left, anchor_right; };
    std::vector<Junction> junctions; // result of the last spl_junctions call, sorted
    // optional per-launch stopwatch around spl_count_kernel alone (bench.py's roofline numerator)
    std::vector<hipEvent_t> k_ev; // pairs
    int k_used = 0;
    bool k_on = false;
    std::vector<hipEvent_t> l_ev; // ... and around the layout kernel (spl_devpack.hip), same switch
    int l_used = 0;
};

struct spl_dsites {
    int64_t n_sites = 0, n_part = 0, n_comp = 0;
    char *slab = nullptr; // one allocation; the pointers below live inside it
    size_t slab_bytes = 0;
    int32_t *pos = nullptr;
    uint8_t *strand = nullptr, *flags = nullptr;
    uint4 *meta = nullptr;
    uint32_t *part_off = nullptr;
    int32_t *part_pos = nullptr, *part_site = nullptr, *comp_pos = nullptr;
    int64_t *alpha = nullptr, *edge_cnt = nullptr;
    uint32_t *bucket = nullptr;
    uint32_t n_buckets = 0;
    int32_t bucket_base = 0, bucket_shift = 0;
    bool has_sse_inputs = false;
    int32_t *diff = nullptr;   // 4 difference arrays of diff_stride int32 over distinct positions (range kernel)
    int32_t *block_sums = nullptr;
    int32_t diff_stride = 0, scan_blocks = 0;
    int32_t n_dpos = 0;        // distinct site positions
    int32_t *dpos_first_row = nullptr; // [n_dpos + 1]
    spl_dbk *dbucket = nullptr; // 32 bp buckets {first dpos, occupancy mask, mask of flagged positions}
    uint32_t n_dbuckets = 0;
    int32_t dbase = 0;         // coordinate of the first (empty) bucket
    int32_t *flag_pos = nullptr; // ends of the junctions that have rivals (sorted; input of the bucket build)
    uint4 *jhash = nullptr;    // junction table (see build_junction_table)
    uint32_t jhash_mask = 0;
    uint4 *jrivals = nullptr;
    // outputs
    uint32_t *beta1 = nullptr, *beta2s = nullptr, *dbl = nullptr; // contiguous: one memset clears all three
    size_t counter_bytes = 0;
    // Everything a counting pass must find zero (counters, difference arrays, error word, queue counters) exists more than
    // once: a pass works on one copy while its literal kernel's idle waves clear another for a later pass (region_clean), so
    // no pass starts with a clearing launch of its own.  The plain members always point into the copy in use.
    // (Three copies, in fact: with the tail of a pass on its own stream the copy that pass n's literal kernel clears is
    // the one pass n + 2 will use -- pass n + 1 may already be running.)
    uint32_t *queue_n = nullptr;
    int32_t *err = nullptr;
    char *region[3] = {nullptr, nullptr, nullptr}; // start of each copy (same layout: beta1, beta2s, dbl, diff, queue counters, error word)
    size_t off_b2 = 0, off_dbl = 0, off_diff = 0, off_ctl = 0; // member offsets inside a copy
    bool region_clean[3] = {false, false, false};
    int cur = 2;                                   // the copy in use (the plain members point into it)
    void point_at(int k)
    {
        cur = k;
        beta1 = (uint32_t *)region[k]; beta2s = (uint32_t *)(region[k] + off_b2); dbl = (uint32_t *)(region[k] + off_dbl);
        diff = (int32_t *)(region[k] + off_diff); queue_n = (uint32_t *)(region[k] + off_ctl);
        err = (int32_t *)(region[k] + off_ctl + 4 * 8 * SPL_COUNTER_STRIDE);
    }
    int64_t *b2_simple = nullptr, *b2_cryptic = nullptr;
    double *b2_weighted = nullptr, *sse = nullptr;
    double *sse_cryptic = nullptr;      // SSE with --beta2Cryptic, written next to `sse` by the fused scan kernel
    bool sse_fused = false;             // the last counting pass computed beta2 / SSE (both settings) already
    const double *sse_view = nullptr;   // what spl_sse_download hands out
};

struct DeviceReads;
struct spl_dreads {
    // A read set = segments (one reference of a BAM file, one caller array, ...) in the chunked layout of spl_pack.h, each with a
    // coordinate shift of its own; the kernels see one flat list of chunk descriptors.  A segment packed on the HOST
    // (add_segment) has a device allocation of its own; segments whose reads are on the device already, BAM-native
    // (add_segment_device), are laid out there when the read set is finished: ONE launch of the layout kernel for all segments
    // that come from the same arrays (a Group), into record slots of one slab.
    struct Segment {
        char *slab = nullptr;       // host-packed: records, then the wide ops
        uint64_t rec_bytes = 0, n_wide = 0;
        int64_t n_reads = 0, n_ops = 0;
        int32_t shift = 0;
        std::vector<splpack::ChunkDesc> chunks; // host-packed
        int group = -1;             // laid out on the device: which group, which of its segments
        size_t group_seg = 0;
    };
    struct Group {
        DeviceReads *src = nullptr;          // (a reference is held: WIDE reads' ops are read from its arrays)
        std::vector<spl_layout_seg> segs;
        uint32_t n_chunks = 0;
        char *slab = nullptr;                // n_chunks record slots
        spl_layout_seg *d_segs = nullptr;    // the segments, then the chunks' descriptors (spl_layout_map_kernel)
        spl_layout_chunk *d_chunks = nullptr;
    };
    std::vector<Segment> segs;
    std::vector<Group> groups;
    int64_t n_reads = 0, n_cigar = 0;
    uint32_t n_chunks = 0;
    uint32_t n_slots = 0;           // slots of the range kernel's grid: 8 * spl_order_per(n_chunks)
    uint32_t chunk_shift = SPL_CHUNK_SHIFT; // reads per chunk of this set (fixed when it is begun: spl_reads_begin_sized)
    // FUSED: every segment of the set lies in ONE set of device arrays, and the counting pass reads those arrays itself
    // (spl_count_ranges_kernel<.., FUSED>: a chunk's records made in LDS, a tile of SPL_TILE_FUSED reads at a time) -- no record slots, no
    // layout kernel.  What needs records in memory (the pair kernel, spl_reads_junctions) lays the set out first (unfuse).
    bool fused = false;
    bool finished = false;
    char *ctl = nullptr;            // chunk descriptors, costs, chunk order, the queues (allocated by spl_reads_finish)
    spl_chunk_meta *meta = nullptr;
    uint32_t *cost = nullptr;
    uint32_t *chunk_order = nullptr; // slot of an XCD share -> chunk, longest first (0xffffffff: none)
    uint32_t *queue = nullptr; // reads the range kernel hands to the literal kernel (the counters are with the site table)
    uint32_t *queue_alt = nullptr;     // ... and the buffer the NEXT pass writes while this pass's literal kernel still reads
    uint32_t *queue_total = nullptr;   // entries the last pass queued (written by its literal kernel)
    uint32_t queue_cap = 0;
    mutable int queue_turn = 0;        // which of the two the next pass takes
    mutable bool queued_pass = false;  // the last counting pass over this read set had a literal kernel
    mutable int64_t tail_pass = -1;    // the context's pass number of the last counting pass over this set that has a tail on the tail stream
};

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// ---- context --------------------------------------------------------------------------------------------
extern "C" int spl_device_count(int *n_out)
{
    if (!n_out) return spl_set_error(SPL_ERR_ARG, "spl_device_count: null output");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *n_out = 0; return spl_set_error(SPL_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *n_out = n;
    return SPL_OK;
}

static int create_ctx(int device_id, void *stream, bool use_given, spl_ctx **out)
{
    if (!out) return spl_set_error(SPL_ERR_ARG, "spl_create: null output");
    *out = nullptr;
    // (Rounds 3-5 asked the runtime for eight hardware queues a priority level here -- GPU_MAX_HW_QUEUES, if nobody had started it
    //  yet -- so that the decode's streams would not share one.  They have levels of their own now (decode_share: Pipe::make), and
    //  with eight the cold command line was slower than with the runtime's four: profiles/r06s_first_call.txt, 6.)
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return spl_set_error(SPL_ERR_NO_DEVICE, "no HIP device visible (%s); libspliser_hip has no CPU fallback",
                             e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return spl_set_error(SPL_ERR_ARG, "device %d out of range (0..%d)", device_id, n - 1);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return spl_set_error(SPL_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code only", device_id,
                             prop.gcnArchName);
    HIP_TRY(hipSetDevice(device_id));
    spl_ctx *c = new (std::nothrow) spl_ctx();
    if (!c) return spl_set_error(SPL_ERR_NOMEM, "out of host memory");
    c->device = device_id;
    if (use_given) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else {
        hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (se != hipSuccess) { delete c; return spl_set_error(SPL_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(se)); }
        c->own_stream = true;
    }
    {
        const char *hw = getenv("SPL_TAIL_HOST_WAIT");
        c->tail_host_wait = hw && hw[0] == '1';
        const char *want_tail = getenv("SPL_TAIL_STREAM");
        if (!(want_tail && want_tail[0] == '0')) {
            bool ok = hipStreamCreateWithFlags(&c->tail, hipStreamNonBlocking) == hipSuccess;
            for (int i = 0; i < 4 && ok; ++i)
                // (no system-scope fences: what these events order is read on this device only)
                ok = hipEventCreate(&c->ev_range[i]) == hipSuccess && // (a kernel's stop event: hipExtLaunchKernelGGL)
                     hipEventCreateWithFlags(&c->ev_tail[i], hipEventDisableTiming | hipEventDisableSystemFence) == hipSuccess;
            if (!ok) { // one stream it is
                if (c->tail) (void)hipStreamDestroy(c->tail);
                c->tail = nullptr;
            }
        }
    }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipMalloc((void **)&c->d_err, sizeof(int32_t)) != hipSuccess) {
        spl_destroy(c);
        return spl_set_error(SPL_ERR_HIP, "context resources could not be created");
    }
    *out = c;
    return SPL_OK;
}

extern "C" int spl_create(int device_id, spl_ctx **out) { return create_ctx(device_id, nullptr, false, out); }
extern "C" int spl_create_on_stream(int device_id, void *hip_stream, spl_ctx **out) { return create_ctx(device_id, hip_stream, true, out); }

static void free_stage(spl_ctx *c);

extern "C" void spl_destroy(spl_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->tail) (void)hipStreamSynchronize(c->tail);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < 4; ++i) {
        if (c->ev_range[i]) (void)hipEventDestroy(c->ev_range[i]);
        if (c->ev_tail[i]) (void)hipEventDestroy(c->ev_tail[i]);
    }
    if (c->tail) (void)hipStreamDestroy(c->tail);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (hipEvent_t e : c->k_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->l_ev) (void)hipEventDestroy(e);
    if (c->d_err) (void)hipFree(c->d_err);
    free_stage(c);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// Everything queued on the main stream after this call runs after the tails of all counting passes launched so far.
static hipError_t join_tail(spl_ctx *c)
{
    if (!c->tail || !c->tail_pending) return hipSuccess;
    c->tail_pending = false;
    return hipStreamWaitEvent(c->stream, c->ev_tail[(c->n_pass - 1) % 4], 0);
}

extern "C" int spl_sync(spl_ctx *c)
{
    if (!c) return spl_set_error(SPL_ERR_ARG, "spl_sync: null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(join_tail(c));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SPL_OK;
}

// Device-side barrier between passes: whatever is launched on this context after the call starts after everything launched
// before it -- the tails of counting passes on the context's second stream included -- has finished.  The host does not wait.
extern "C" int spl_pass_barrier(spl_ctx *c)
{
    if (!c) return spl_set_error(SPL_ERR_ARG, "spl_pass_barrier: null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(join_tail(c));
    return SPL_OK;
}

extern "C" int spl_timer_begin(spl_ctx *c)
{
    if (!c) return spl_set_error(SPL_ERR_ARG, "spl_timer_begin: null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    return SPL_OK;
}

extern "C" int spl_timer_end(spl_ctx *c, float *ms)
{
    if (!c || !ms) return spl_set_error(SPL_ERR_ARG, "spl_timer_end: null argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(join_tail(c));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return SPL_OK;
}

extern "C" int spl_kernel_timing_begin(spl_ctx *c, int max_records)
{
    if (!c || max_records < 0) return spl_set_error(SPL_ERR_ARG, "spl_kernel_timing_begin: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    while ((int)c->k_ev.size() < 2 * max_records) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        c->k_ev.push_back(e);
    }
    while ((int)c->l_ev.size() < 2 * max_records) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        c->l_ev.push_back(e);
    }
    c->k_used = 0;
    c->l_used = 0;
    c->k_on = max_records > 0;
    return SPL_OK;
}

// The layout kernel's launches since spl_kernel_timing_begin (spl_reads_finish, spl_reads_relayout).  Call BEFORE
// spl_kernel_timing_collect, which ends the recording.
extern "C" int spl_layout_timing_collect(spl_ctx *c, float *ms_out, int capacity, int *n_out)
{
    if (!c || !n_out || (capacity > 0 && !ms_out)) return spl_set_error(SPL_ERR_ARG, "spl_layout_timing_collect: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int n = std::min(c->l_used, capacity);
    for (int i = 0; i < n; ++i) HIP_TRY(hipEventElapsedTime(&ms_out[i], c->l_ev[2 * i], c->l_ev[2 * i + 1]));
    *n_out = n;
    return SPL_OK;
}

extern "C" int spl_kernel_timing_collect(spl_ctx *c, float *ms_out, int capacity, int *n_out)
{
    if (!c || !n_out || (capacity > 0 && !ms_out)) return spl_set_error(SPL_ERR_ARG, "spl_kernel_timing_collect: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int n = std::min(c->k_used, capacity);
    for (int i = 0; i < n; ++i) HIP_TRY(hipEventElapsedTime(&ms_out[i], c->k_ev[2 * i], c->k_ev[2 * i + 1]));
    *n_out = n;
    c->k_on = false;
    return SPL_OK;
}

// ---- site table upload ----------------------------------------------------------------------------------
static int validate_sites(const spl_sites *s)
{
    if (!s) return spl_set_error(SPL_ERR_ARG, "null site table");
    if (s->n_sites < 0 || s->n_sites > 0x3ffffff0LL) return spl_set_error(SPL_ERR_ARG, "n_sites out of range");
    if (s->n_sites == 0) return SPL_OK;
    if (!s->pos || !s->strand || !s->part_off || !s->comp_off) return spl_set_error(SPL_ERR_ARG, "site table has null arrays");
    if (s->part_off[0] != 0 || s->comp_off[0] != 0) return spl_set_error(SPL_ERR_ARG, "CSR offsets must start at 0");
    for (int64_t i = 0; i < s->n_sites; ++i) {
        if (i && s->pos[i] < s->pos[i - 1]) return spl_set_error(SPL_ERR_ARG, "site positions must be non-decreasing (row %lld)", (long long)i);
        if (s->part_off[i + 1] < s->part_off[i] || s->comp_off[i + 1] < s->comp_off[i])
            return spl_set_error(SPL_ERR_ARG, "CSR offsets must be non-decreasing (row %lld)", (long long)i);
    }
    if (s->pos[0] < 0 || s->pos[s->n_sites - 1] > SPL_COORD_MAX) return spl_set_error(SPL_ERR_RANGE, "site position outside [0, %d]", SPL_COORD_MAX);
    if (s->part_off[s->n_sites] && !s->part_pos) return spl_set_error(SPL_ERR_ARG, "part_pos is null");
    if (s->comp_off[s->n_sites] && !s->comp_pos) return spl_set_error(SPL_ERR_ARG, "comp_pos is null");
    return SPL_OK;
}

// Junction table: every junction (l, r) that can make checkBam's compSplicing test (SpliSER_v0_1_8.py:494-501) succeed for
// SOME site of the table, with the sites ("rivals") it succeeds for.  The test for site t given a read junction (l, r) is
//     (l in P_t and r in C_t) or (r in P_t and l in C_t)        P_t = partner positions, C_t = competitor positions of t
// so the junctions are exactly the pairs (p, c), p in P_t, c in C_t, smaller coordinate first -- enumerated here from each
// row's own lists.  Nothing else about the table is assumed (partner links need not be mutual, partners and competitors need
// not be rows: the query tables of `combine` qualify), and the table is COMPLETE: a read junction that is not in it has no
// rival anywhere.  The ends of the table's junctions are the "flagged" positions (a bit per position next to the position
// index, see spl_build_dbuckets_kernel): only a read junction with a flagged end is ever looked up.
// Open-addressing hash on (l, r), at most 8 probes; the table grows until every junction is placed.
static void build_junction_table(const spl_sites *s, const std::vector<uint8_t> &flags, const std::vector<int32_t> &dfirst,
                                 const std::vector<int32_t> &row_dpos, std::vector<uint4> &jhash, std::vector<uint4> &jrivals,
                                 std::vector<int32_t> &flag_pos)
{
    const int64_t S = s->n_sites;
    struct Trip { int32_t l, r, t; };
    std::vector<Trip> trips;
    for (int64_t t = 0; t < S; ++t) {
        const uint32_t pa = s->part_off[t], pb = s->part_off[t + 1], ca = s->comp_off[t], cb = s->comp_off[t + 1];
        for (uint32_t e = pa; e < pb; ++e)
            for (uint32_t f = ca; f < cb; ++f) {
                const int32_t p = s->part_pos[e], c = s->comp_pos[f];
                trips.push_back(p <= c ? Trip{p, c, (int32_t)t} : Trip{c, p, (int32_t)t});
            }
    }
    std::sort(trips.begin(), trips.end(), [](const Trip &a, const Trip &b) { return a.l != b.l ? a.l < b.l : (a.r != b.r ? a.r < b.r : a.t < b.t); });
    trips.erase(std::unique(trips.begin(), trips.end(), [](const Trip &a, const Trip &b) { return a.l == b.l && a.r == b.r && a.t == b.t; }), trips.end());
    struct Junc { int32_t l, r; uint32_t off, info; };
    std::vector<Junc> juncs;
    for (size_t i = 0; i < trips.size();) {
        size_t k = i;
        while (k < trips.size() && trips[k].l == trips[i].l && trips[k].r == trips[i].r) ++k;
        Junc j;
        j.l = trips[i].l; j.r = trips[i].r; j.off = (uint32_t)(jrivals.size() / 2); j.info = 0;
        for (size_t x = i; x < k; ++x) {
            const int32_t t = trips[x].t;
            const uint32_t pa = s->part_off[t], pb = s->part_off[t + 1];
            const int32_t tpos = s->pos[t];
            if (tpos == j.l || tpos == j.r) j.info |= SPL_JF_COMPLEX; // alpha read with compSplicing: literal
            const int32_t d = row_dpos[(size_t)t], t0 = dfirst[(size_t)d], t1 = dfirst[(size_t)d + 1];
            if (t1 - t0 > 1) {
                j.info |= SPL_JF_MULTIROW;
                for (int32_t y = t0; y < t1; ++y)
                    if (y != t && (flags[(size_t)y] & 3u) == (flags[(size_t)t] & 3u)) j.info |= SPL_JF_COMPLEX;
            }
            uint32_t edges[2] = {0xffffffffu, 0xffffffffu};
            int ne = 0;
            for (uint32_t e3 = pa; e3 < pb; ++e3)
                if (s->part_pos[e3] == j.l || s->part_pos[e3] == j.r) { if (ne < 2) edges[ne] = e3; ++ne; }
            if (ne > 2) j.info |= SPL_JF_COMPLEX;
            const uint32_t scode = (flags[(size_t)t] & SPL_SF_PLUS) ? 1u : ((flags[(size_t)t] & SPL_SF_MINUS) ? 2u : 0u);
            jrivals.push_back(make_uint4((uint32_t)tpos, (uint32_t)d | (scode << 30), edges[0], edges[1]));
            jrivals.push_back(make_uint4((uint32_t)t, pa, pb - pa, 0u));
        }
        const size_t n = k - i;
        j.info |= (uint32_t)std::min<size_t>(n, SPL_JF_COUNT_MASK);
        if (n > SPL_JF_COUNT_MASK) j.info |= SPL_JF_COMPLEX; // (cannot happen below 16 M rivals of one junction; the literal path would stop short)
        juncs.push_back(j);
        flag_pos.push_back(j.l);
        flag_pos.push_back(j.r);
        i = k;
    }
    std::sort(flag_pos.begin(), flag_pos.end());
    flag_pos.erase(std::unique(flag_pos.begin(), flag_pos.end()), flag_pos.end());
    // a slot is two quads: {l, r, first rival record, count | flags} and a copy of the first rival's first quad, so that
    // the common one-rival junction costs the range kernel one memory trip instead of two
    size_t cap = 16;
    while (cap < 4 * juncs.size() + 1) cap <<= 1; // load <= 1/4
    for (;; cap <<= 1) {
        jhash.assign(2 * cap, make_uint4(0x80000000u, 0, 0, 0));
        bool all = true;
        for (const Junc &j : juncs) {
            uint32_t h = (uint32_t)j.l * 0x9E3779B1u ^ (uint32_t)j.r * 0x85EBCA77u;
            h ^= h >> 15;
            bool placed = false;
            for (uint32_t probe = 0; probe < 8 && !placed; ++probe) { // (the kernels give up after 8 probes)
                uint4 *slot = &jhash[2 * ((h + probe) & (cap - 1))];
                if (slot[0].x != 0x80000000u) continue;
                slot[0] = make_uint4((uint32_t)j.l, (uint32_t)j.r, j.off, j.info);
                slot[1] = jrivals[2 * (size_t)j.off];
                placed = true;
            }
            if (!placed) { all = false; break; }
        }
        if (all) break; // (otherwise: the table must be complete -- twice the slots and again)
    }
    if (getenv("SPL_DEBUG_TABLE")) {
        size_t complex = 0, multirow = 0, many = 0;
        for (const Junc &j : juncs) { complex += (j.info & SPL_JF_COMPLEX) != 0; multirow += (j.info & SPL_JF_MULTIROW) != 0; many += (j.info & SPL_JF_COUNT_MASK) > 4; }
        fprintf(stderr, "[junction table] %zu junctions with rivals (%zu flagged positions), %zu slots, %zu complex, %zu multirow, %zu with > 4 rivals\n",
                juncs.size(), flag_pos.size(), cap, complex, multirow, many);
    }
    if (jrivals.empty()) jrivals.assign(2, make_uint4(0, 0, 0xffffffffu, 0xffffffffu));
}

static int ensure_stage(spl_ctx *c, int n_min = 0);
static int grow_stage(spl_ctx *c, int n);

extern "C" int spl_sites_upload(spl_ctx *c, const spl_sites *s, spl_dsites **out)
{
    if (!c || !out) return spl_set_error(SPL_ERR_ARG, "spl_sites_upload: null argument");
    *out = nullptr;
    int rc = validate_sites(s);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    const bool stamps = getenv("SPL_DEBUG_TABLE") != nullptr;
    auto host_now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = host_now();
    auto stamp = [&](const char *what) { if (stamps) { const double t = host_now(); fprintf(stderr, "[spl_sites_upload] %-28s %.4f s\n", what, t - t_mark); t_mark = t; } };
    const int64_t S = s->n_sites;
    const int64_t P = S ? s->part_off[S] : 0, C = S ? s->comp_off[S] : 0;
    spl_dsites *d = new (std::nothrow) spl_dsites();
    if (!d) return spl_set_error(SPL_ERR_NOMEM, "out of host memory");
    d->n_sites = S; d->n_part = P; d->n_comp = C;
    d->has_sse_inputs = s->alpha && s->edge_cnt && (s->part_site || P == 0);

    // position -> row bucket index: <= 16 buckets per site, buckets at least 16 bp wide
    std::vector<uint32_t> bucket;
    if (S > 0) {
        const int64_t extent = (int64_t)s->pos[S - 1] - (int64_t)s->pos[0] + 1;
        int shift = 4;
        while ((extent >> shift) > 16 * S) ++shift;
        d->bucket_shift = shift;
        d->bucket_base = s->pos[0];
        d->n_buckets = (uint32_t)((extent - 1) >> shift) + 1;
        bucket.resize((size_t)d->n_buckets + 1);
        int64_t row = 0;
        for (uint32_t b = 0; b <= d->n_buckets; ++b) {
            const int64_t start = (int64_t)d->bucket_base + ((int64_t)b << shift);
            while (row < S && (int64_t)s->pos[row] < start) ++row;
            bucket[b] = (uint32_t)row;
        }
        bucket[d->n_buckets] = (uint32_t)S;
    }
    stamp("row buckets");
    std::vector<uint4> meta((size_t)S);
    std::vector<uint8_t> flags((size_t)S);
    for (int64_t i = 0; i < S; ++i) {
        const uint32_t np = s->part_off[i + 1] - s->part_off[i];
        meta[(size_t)i] = make_uint4(s->part_off[i], np, s->comp_off[i], s->comp_off[i + 1] - s->comp_off[i]);
        flags[(size_t)i] = (uint8_t)((s->strand[i] == '+' ? SPL_SF_PLUS : 0u) | (s->strand[i] == '-' ? SPL_SF_MINUS : 0u));
    }
    // distinct positions ("dpos"): rows sharing a position share one index
    std::vector<int32_t> dfirst;
    for (int64_t i = 0; i < S; ++i)
        if (i == 0 || s->pos[i] != s->pos[i - 1]) dfirst.push_back((int32_t)i);
    const int64_t D = (int64_t)dfirst.size();
    dfirst.push_back((int32_t)S);
    d->n_dpos = (int32_t)D;
    std::vector<uint4> jhash, jrivals;
    std::vector<int32_t> flag_pos; // ends of the junctions that have rivals, sorted (not necessarily site positions)
    {
        std::vector<int32_t> row_dpos((size_t)S);
        for (int64_t j = 0; j < D; ++j)
            for (int32_t r = dfirst[(size_t)j]; r < dfirst[(size_t)j + 1]; ++r) row_dpos[(size_t)r] = (int32_t)j;
        if (S > 0) build_junction_table(s, flags, dfirst, row_dpos, jhash, jrivals, flag_pos);
        if (jhash.empty()) { jhash.assign(32, make_uint4(0x80000000u, 0, 0, 0)); jrivals.assign(2, make_uint4(0, 0, 0xffffffffu, 0xffffffffu)); }
    }
    stamp("junction table");
    for (int32_t x : flag_pos)
        if (x < 0 || x > SPL_COORD_MAX) { delete d; return spl_set_error(SPL_ERR_RANGE, "partner / competitor position outside [0, %d]", SPL_COORD_MAX); }
    d->jhash_mask = (uint32_t)(jhash.size() / 2) - 1u;
    if (S > 0) {
        // the position index covers sites and flagged positions alike, with one empty bucket in front of the first and one
        // behind the last of them (first dpos = D there): the kernels only clamp
        const int64_t lo = flag_pos.empty() ? s->pos[0] : std::min<int64_t>(s->pos[0], flag_pos.front());
        const int64_t hi = flag_pos.empty() ? s->pos[S - 1] : std::max<int64_t>(s->pos[S - 1], flag_pos.back());
        d->dbase = lo >= 64 ? (int32_t)(lo - 64) : -64;
        d->n_dbuckets = (uint32_t)((hi - (int64_t)d->dbase) >> 5) + 2;
        // (the bucket entries themselves are built on the device, spl_build_dbuckets_kernel, once the positions are there)
    }
    d->diff_stride = (int32_t)align_up((size_t)D + 2, 64);
    d->scan_blocks = (int32_t)((D + SPL_SCAN_BLOCK - 1) / SPL_SCAN_BLOCK);

    // slab layout
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
    const size_t o_pos = take(4 * S), o_strand = take(S), o_flags = take(S), o_meta = take(16 * S), o_poff = take(4 * (S + 1));
    const size_t o_ppos = take(4 * P), o_psite = take(4 * P), o_cpos = take(4 * C);
    const size_t o_alpha = take(8 * S), o_ecnt = take(8 * P), o_bucket = take(4 * bucket.size());
    const size_t o_dfirst = take(4 * dfirst.size()), o_dbucket = take(sizeof(spl_dbk) * (size_t)d->n_dbuckets), o_fpos = take(4 * flag_pos.size());
    const size_t o_jhash = take(16 * jhash.size()), o_jriv = take(16 * jrivals.size());
    const size_t o_cnt = off;
    const size_t o_b1 = take(4 * S), o_b2 = take(4 * S), o_dbl = take(4 * P);
    const size_t o_diff = take(4 * 4 * (size_t)d->diff_stride);
    const size_t o_ctl = take(4 * 8 * SPL_COUNTER_STRIDE + 256); // 8 queue counters (a cache line each), then the error word
    d->counter_bytes = off - o_cnt;
    const size_t o_alt = take(2 * d->counter_bytes);            // the second and third copy, same layout
    const size_t o_bsum = take(4 * 4 * (size_t)std::max(d->scan_blocks, 1));
    const size_t o_b2s = take(8 * S), o_b2c = take(8 * S), o_b2w = take(8 * S), o_sse = take(8 * S), o_ssec = take(8 * S);
    d->slab_bytes = std::max<size_t>(off, 256);
    hipError_t e = devmem::get((void **)&d->slab, d->slab_bytes, 's');
    if (e != hipSuccess) { delete d; return spl_set_error(SPL_ERR_HIP, "hipMalloc(%zu) for the site table: %s", d->slab_bytes, hipGetErrorString(e)); }
    stamp("hipMalloc");
    d->pos = (int32_t *)(d->slab + o_pos); d->strand = (uint8_t *)(d->slab + o_strand); d->meta = (uint4 *)(d->slab + o_meta);
    d->flags = (uint8_t *)(d->slab + o_flags); d->diff = (int32_t *)(d->slab + o_diff); d->block_sums = (int32_t *)(d->slab + o_bsum);
    d->dpos_first_row = (int32_t *)(d->slab + o_dfirst); d->dbucket = (spl_dbk *)(d->slab + o_dbucket); d->flag_pos = (int32_t *)(d->slab + o_fpos);
    d->jhash = (uint4 *)(d->slab + o_jhash); d->jrivals = (uint4 *)(d->slab + o_jriv);
    d->part_off = (uint32_t *)(d->slab + o_poff); d->part_pos = (int32_t *)(d->slab + o_ppos); d->part_site = (int32_t *)(d->slab + o_psite);
    d->comp_pos = (int32_t *)(d->slab + o_cpos); d->alpha = (int64_t *)(d->slab + o_alpha); d->edge_cnt = (int64_t *)(d->slab + o_ecnt);
    d->bucket = (uint32_t *)(d->slab + o_bucket);
    static_cast<void>(o_b1); // (= o_cnt: beta1 opens a copy)
    d->region[0] = d->slab + o_cnt; d->region[1] = d->slab + o_alt; d->region[2] = d->slab + o_alt + d->counter_bytes;
    d->off_b2 = o_b2 - o_cnt; d->off_dbl = o_dbl - o_cnt; d->off_diff = o_diff - o_cnt; d->off_ctl = o_ctl - o_cnt;
    d->region_clean[0] = d->region_clean[1] = d->region_clean[2] = true; // the upload zeroes all copies
    d->point_at(2);                                                       // (the first pass moves on to copy 0)
    d->b2_simple = (int64_t *)(d->slab + o_b2s); d->b2_cryptic = (int64_t *)(d->slab + o_b2c);
    d->b2_weighted = (double *)(d->slab + o_b2w); d->sse = (double *)(d->slab + o_sse); d->sse_cryptic = (double *)(d->slab + o_ssec);
    d->sse_view = d->sse;

    // The table's arrays go up through the context's page-locked staging ring, packed end to end into its buffers: fifteen
    // arrays, 25 MB for 200 k sites -- as fifteen synchronous copies out of pageable memory that took 8...20 ms (most of the
    // call), this way the copy engine runs at its 50 GB/s while the next buffer is being filled.
    hipError_t r = hipSuccess;
    if (ensure_stage(c) != SPL_OK) r = hipErrorOutOfMemory;
    size_t fill = 0;        // bytes used of the current staging buffer
    bool open_buf = false;  // the current buffer has copies in flight that no event covers yet
    auto next_buffer = [&]() -> hipError_t {
        hipError_t q = hipSuccess;
        if (open_buf) {
            spl_ctx::Stage &cur = c->stage[c->stage_next];
            q = hipEventRecord(cur.done, c->copy);
            cur.busy = true;
            c->stage_next = (c->stage_next + 1) % c->stage.size();
            open_buf = false;
        }
        fill = 0;
        return q;
    };
    auto up = [&](void *dst, const void *src, size_t bytes) -> hipError_t {
        if (!bytes || !src) return hipSuccess;
        const char *from = (const char *)src;
        char *to = (char *)dst;
        while (bytes) {
            spl_ctx::Stage *st = &c->stage[c->stage_next];
            if (fill + 64 > st->bytes) { const hipError_t q = next_buffer(); if (q != hipSuccess) return q; st = &c->stage[c->stage_next]; }
            if (fill == 0 && st->busy) { const hipError_t q = hipEventSynchronize(st->done); st->busy = false; if (q != hipSuccess) return q; }
            const size_t n = std::min(bytes, st->bytes - fill);
            memcpy(st->host + fill, from, n);
            const hipError_t q = hipMemcpyAsync(to, st->host + fill, n, hipMemcpyHostToDevice, c->copy);
            if (q != hipSuccess) return q;
            open_buf = true;
            fill = (fill + n + 63) & ~(size_t)63;
            from += n; to += n; bytes -= n;
        }
        return hipSuccess;
    };
    if (r == hipSuccess) r = up(d->pos, s->pos, 4 * S);
    if (r == hipSuccess) r = up(d->strand, s->strand, S);
    if (r == hipSuccess) r = up(d->meta, meta.data(), 16 * S);
    if (r == hipSuccess) r = up(d->flags, flags.data(), S);
    if (r == hipSuccess) r = up(d->dpos_first_row, dfirst.data(), 4 * dfirst.size());
    if (r == hipSuccess) r = up(d->flag_pos, flag_pos.data(), 4 * flag_pos.size());
    if (r == hipSuccess) r = up(d->jhash, jhash.data(), 16 * jhash.size());
    if (r == hipSuccess) r = up(d->jrivals, jrivals.data(), 16 * jrivals.size());
    if (r == hipSuccess) r = up(d->part_off, s->part_off, S ? 4 * (S + 1) : 0);
    if (r == hipSuccess) r = up(d->part_pos, s->part_pos, 4 * P);
    if (r == hipSuccess) r = up(d->part_site, s->part_site, 4 * P);
    if (r == hipSuccess) r = up(d->comp_pos, s->comp_pos, 4 * C);
    if (r == hipSuccess) r = up(d->alpha, s->alpha, 8 * S);
    if (r == hipSuccess) r = up(d->edge_cnt, s->edge_cnt, 8 * P);
    if (r == hipSuccess) r = up(d->bucket, bucket.data(), 4 * bucket.size());
    if (r == hipSuccess) r = next_buffer();
    if (r == hipSuccess) r = hipStreamSynchronize(c->copy); // (the arrays are on the device before anything is launched on them)
    stamp("copies");
    if (r == hipSuccess) r = hipMemset(d->slab + o_cnt, 0, d->slab_bytes > o_cnt ? d->slab_bytes - o_cnt : 0);
    stamp("memset");
    if (r == hipSuccess && S > 0) {
        r = (hipError_t)spl_dev_launch_build_dbuckets(d->pos, d->dpos_first_row, d->n_dpos, d->flag_pos, (int32_t)flag_pos.size(), d->dbase, d->n_dbuckets, d->dbucket,
                                                      c->stream);
        if (r == hipSuccess) r = hipStreamSynchronize(c->stream);
    }
    stamp("position index kernel");
    if (r != hipSuccess) { devmem::put(d->slab); delete d; return spl_set_error(SPL_ERR_HIP, "site table upload: %s", hipGetErrorString(r)); }
    *out = d;
    return SPL_OK;
}

extern "C" void spl_sites_free(spl_ctx *c, spl_dsites *d)
{
    if (!d) return;
    if (c) (void)hipSetDevice(c->device);
    if (c && c->tail) (void)hipStreamSynchronize(c->tail); // (a tail may still be reading these buffers; hipFree itself waits
    if (c && c->stream) (void)hipStreamSynchronize(c->stream); //  for the device, this makes it independent of that)
    devmem::put(d->slab);
    delete d;
}

// ---- reads upload ---------------------------------------------------------------------------------------
struct PooledStage { int device; spl_ctx::Stage st; };
static std::mutex &stage_pool_mu() { static std::mutex m; return m; }
static std::vector<PooledStage> &stage_pool() { static std::vector<PooledStage> *p = new std::vector<PooledStage>(); return *p; } // (never destroyed: no order-of-exit games with the HIP runtime)

// The staging ring and the copy stream of a context (created at the first upload).
static int ensure_stage(spl_ctx *c, int n_min)
{
    if (!c->stage.empty()) return (int)c->stage.size() < n_min ? grow_stage(c, n_min) : SPL_OK;
    size_t mb = 32;
    int n = 3;
    if (const char *e = getenv("SPL_STAGE_MB")) { const long v = atol(e); if (v >= 1 && v <= 4096) mb = (size_t)v; }
    if (const char *e = getenv("SPL_STAGE_BUFFERS")) { const int v = atoi(e); if (v >= 1 && v <= 16) n = v; }
    {
        int t = (int)std::thread::hardware_concurrency();
        if (const char *e = getenv("SPL_PACK_THREADS")) { const int v = atoi(e); if (v > 0) t = v; }
        else if (t > 32) t = 32;
        c->pack_threads = t > 0 ? t : 1;
    }
    HIP_TRY(hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming));
    if (getenv("SPL_STAGE_TIMING")) {
        HIP_TRY(hipEventCreate(&c->ev_t0));
        HIP_TRY(hipEventCreate(&c->ev_t1));
        c->stage_timing = true;
    }
    c->stage_mb = mb;
    return grow_stage(c, getenv("SPL_STAGE_BUFFERS") ? n : std::max(n, n_min)); // (all the buffers the caller will want in one go: they are made side by side)
}

static size_t stage_bytes_of(size_t mb) { const size_t huge = 2u << 20; return (mb << 20) / huge * huge < huge ? huge : (mb << 20) / huge * huge; }

// n_new page-locked staging buffers of `bytes` for `device`, made side by side, a thread each: touching 32 MiB and locking it is
// 3-8 ms of a new process's first call, six of them one behind the other were 15-50 ms before the file's first byte was on its way
static void make_stage_buffers(int device, size_t bytes, bool want_lock, std::vector<spl_ctx::Stage> &made, std::vector<int> &why)
{
    const size_t huge = 2u << 20;
    const int n_new = (int)made.size();
    auto make = [&](int k) {
        spl_ctx::Stage &st = made[(size_t)k];
        void *p = nullptr;
        if (hipSetDevice(device) != hipSuccess) { why[(size_t)k] = SPL_ERR_HIP; return; }
        if (posix_memalign(&p, huge, bytes) != 0) { why[(size_t)k] = SPL_ERR_NOMEM; return; }
        (void)madvise(p, bytes, MADV_HUGEPAGE);
        memset(p, 0, bytes); // touch: the pages exist before they are locked
        st.host = (char *)p;
        st.bytes = bytes;
        st.locked = want_lock && hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess; // (pageable works too, slower)
        if (hipEventCreateWithFlags(&st.done, hipEventDisableTiming) != hipSuccess) {
            if (st.locked) (void)hipHostUnregister(p);
            free(p);
            st = spl_ctx::Stage();
            why[(size_t)k] = SPL_ERR_HIP;
        }
    };
    std::vector<std::thread> crew;
    for (int k = 1; k < n_new; ++k) crew.emplace_back(make, k);
    if (n_new > 0) make(0);
    for (std::thread &t : crew) t.join();
}

// The staging ring of a context grown to n buffers (never shrunk): page-locked, of the ring's size, from the process's pool of
// such buffers first.
static int grow_stage(spl_ctx *c, int n)
{
    const size_t bytes = stage_bytes_of(c->stage_mb);
    const bool want_lock = !(getenv("SPL_STAGE_PAGEABLE"));
    {   // buffers a destroyed context of this process left behind (same device, same size): page-locking 96 MiB anew costs 12 ms
        std::lock_guard<std::mutex> lock(stage_pool_mu());
        std::vector<PooledStage> &pool = stage_pool();
        for (size_t k = pool.size(); k-- > 0 && (int)c->stage.size() < n;) {
            if (pool[k].device != c->device || pool[k].st.bytes != bytes || pool[k].st.locked != want_lock) continue;
            c->stage.push_back(pool[k].st);
            pool.erase(pool.begin() + (long)k);
        }
    }
    const int n_new = n - (int)c->stage.size();
    if (n_new <= 0) return SPL_OK;
    std::vector<spl_ctx::Stage> made((size_t)n_new);
    std::vector<int> why((size_t)n_new, SPL_OK);
    make_stage_buffers(c->device, bytes, want_lock, made, why);
    int rc = SPL_OK;
    for (int k = 0; k < n_new; ++k) {
        if (why[(size_t)k] == SPL_OK) { c->stage.push_back(made[(size_t)k]); continue; }
        if (rc == SPL_OK) rc = why[(size_t)k] == SPL_ERR_NOMEM ? spl_set_error(SPL_ERR_NOMEM, "out of host memory for the staging buffers") : spl_set_error(SPL_ERR_HIP, "hipEventCreate failed");
    }
    return rc;
}

static void free_stage(spl_ctx *c)
{
    if (c->copy) (void)hipStreamSynchronize(c->copy);
    {   // the buffers (idle now) stay with the process for the next context on this device; a handful at most are kept
        std::lock_guard<std::mutex> lock(stage_pool_mu());
        std::vector<PooledStage> &pool = stage_pool();
        for (spl_ctx::Stage &st : c->stage) {
            st.busy = false;
            if (pool.size() < 8) { pool.push_back(PooledStage{c->device, st}); continue; }
            if (st.done) (void)hipEventDestroy(st.done);
            if (st.locked) (void)hipHostUnregister(st.host);
            free(st.host);
        }
    }
    c->stage.clear();
    if (c->ev_copy) (void)hipEventDestroy(c->ev_copy);
    if (c->ev_t0) (void)hipEventDestroy(c->ev_t0);
    if (c->ev_t1) (void)hipEventDestroy(c->ev_t1);
    if (c->copy) (void)hipStreamDestroy(c->copy);
    c->ev_copy = c->ev_t0 = c->ev_t1 = nullptr;
    c->copy = nullptr;
}

namespace {
struct EmitJob {
    const splpack::Source *src;
    const splpack::Plan *plan;
    size_t c0, c1, per;
    uint8_t *rec;
    uint32_t *wide;
};
void emit_slice(size_t k, void *arg)
{
    const EmitJob &j = *(const EmitJob *)arg;
    const size_t a = j.c0 + k * j.per, b = std::min(j.c1, a + j.per);
    if (a >= b) return;
    const splpack::ChunkDesc &d0 = j.plan->chunks[j.c0], &da = j.plan->chunks[a];
    splpack::emit(*j.src, *j.plan, a, b, j.rec + (da.rec_off - d0.rec_off), j.wide + (da.wide_off - d0.wide_off));
}
} // namespace

// One more segment of a read set under construction: packed on this host's threads piece by piece into the staging ring,
// each piece on its way to the device while the next is packed.  The source arrays are not needed after return.
static int add_segment(spl_ctx *c, spl_dreads *d, const splpack::Source &src, int32_t shift, int64_t max_end)
{
    if (d->finished) return spl_set_error(SPL_ERR_ARG, "the read set is finished: no more segments");
    if (src.n_reads == 0) return SPL_OK;
    if (d->n_reads + src.n_reads > 0xfffffff0LL) return spl_set_error(SPL_ERR_ARG, "n_reads out of range (counters are 32-bit)");
    if (max_end >= 0 && max_end + (int64_t)shift > (int64_t)SPL_COORD_MAX)
        return spl_set_error(SPL_ERR_RANGE, "a read ends beyond coordinate %d once its segment is moved by %d: split the shard (spliser_amd/shard.py)",
                             SPL_COORD_MAX, shift);
    int rc = ensure_stage(c);
    if (rc) return rc;
    auto host_now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tp0 = c->stage_timing ? host_now() : 0.0;
    splpack::Plan plan;
    plan.chunk = 1u << d->chunk_shift;
    splpack::plan(src, plan, c->pack_threads);
    if (c->stage_timing) c->tm_plan_s += host_now() - tp0;
    if (plan.n_wide > 0xfffffff0ull) return spl_set_error(SPL_ERR_ARG, "more than 2^32 CIGAR ops of wide reads in one segment: use more shards");
    if ((uint64_t)d->n_chunks + plan.chunks.size() > (1ull << (32 - d->chunk_shift)))
        return spl_set_error(SPL_ERR_ARG, "too many reads in one read set: use more shards");
    d->segs.emplace_back();
    spl_dreads::Segment &seg = d->segs.back();
    seg.rec_bytes = plan.rec_bytes; seg.n_wide = plan.n_wide; seg.n_reads = src.n_reads; seg.n_ops = src.n_ops; seg.shift = shift;
    const size_t rec_al = align_up((size_t)plan.rec_bytes);
    const size_t slab_bytes = rec_al + 4 * (size_t)plan.n_wide + 256;
    hipError_t e = devmem::get((void **)&seg.slab, slab_bytes, 'r');
    if (e != hipSuccess) { d->segs.pop_back(); return spl_set_error(SPL_ERR_HIP, "hipMalloc(%zu) for a read segment: %s", slab_bytes, hipGetErrorString(e)); }
    const size_t n_chunks = plan.chunks.size();
    auto rec_end = [&](size_t k) { return k + 1 < n_chunks ? plan.chunks[k + 1].rec_off : plan.rec_bytes; };
    auto wide_end = [&](size_t k) { return k + 1 < n_chunks ? plan.chunks[k + 1].wide_off : plan.n_wide; };
    hipError_t q = hipSuccess;
    for (size_t c0 = 0; c0 < n_chunks && q == hipSuccess;) {
        spl_ctx::Stage &st = c->stage[c->stage_next];
        c->stage_next = (c->stage_next + 1) % c->stage.size();
        if (st.busy) { q = hipEventSynchronize(st.done); st.busy = false; if (q != hipSuccess) break; }
        const uint64_t r0 = plan.chunks[c0].rec_off, w0 = plan.chunks[c0].wide_off;
        size_t c1 = c0;
        auto need = [&](size_t k) { return align_up((size_t)(rec_end(k) - r0)) + 4 * (size_t)(wide_end(k) - w0); };
        while (c1 < n_chunks && need(c1) <= st.bytes) ++c1;
        char *host = st.host;
        std::vector<char> big; // a single chunk that does not fit a staging buffer (reads of a million ops): pageable, one at a time
        if (c1 == c0) { c1 = c0 + 1; big.resize(need(c0)); host = big.data(); }
        const size_t rec_span = (size_t)(rec_end(c1 - 1) - r0), wide_span = (size_t)(wide_end(c1 - 1) - w0);
        EmitJob job{&src, &plan, c0, c1, 1, (uint8_t *)host, (uint32_t *)(host + align_up(rec_span))};
        const size_t slices = std::min<size_t>((size_t)c->pack_threads * 4, c1 - c0);
        job.per = (c1 - c0 + slices - 1) / slices;
        const double te0 = c->stage_timing ? host_now() : 0.0;
        splpack::parallel_for(slices, c->pack_threads, emit_slice, &job);
        if (c->stage_timing) { c->tm_emit_s += host_now() - te0; (void)hipEventRecord(c->ev_t0, c->copy); }
        q = hipMemcpyAsync(seg.slab + r0, host, rec_span, hipMemcpyHostToDevice, c->copy);
        if (q == hipSuccess && wide_span)
            q = hipMemcpyAsync(seg.slab + rec_al + 4 * w0, host + align_up(rec_span), 4 * wide_span, hipMemcpyHostToDevice, c->copy);
        if (q == hipSuccess && !big.empty()) q = hipStreamSynchronize(c->copy);
        else if (q == hipSuccess) { q = hipEventRecord(st.done, c->copy); st.busy = true; }
        if (q == hipSuccess && c->stage_timing) {
            float ms = 0;
            (void)hipEventRecord(c->ev_t1, c->copy);
            (void)hipEventSynchronize(c->ev_t1);
            if (hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1) == hipSuccess) { c->tm_copy_ms += ms; c->tm_bytes += rec_span + 4 * wide_span; ++c->tm_pieces; }
        }
        c0 = c1;
    }
    if (q != hipSuccess) {
        (void)hipStreamSynchronize(c->copy);
        devmem::put(seg.slab);
        d->segs.pop_back();
        return spl_set_error(SPL_ERR_HIP, "read segment upload: %s", hipGetErrorString(q));
    }
    seg.chunks.swap(plan.chunks);
    d->n_reads += src.n_reads;
    d->n_cigar += src.n_ops;
    d->n_chunks += (uint32_t)n_chunks;
    return SPL_OK;
}

// ---- BAM decode on the device ----------------------------------------------------------------------------------------
// The file image goes up through the staging ring (one reader thread per buffer, from the page cache), then, a window of the
// inflated stream at a time: inflate + CRC32 (spl_inflate.hip), scan of the window for records (a guess per BGZF block,
// verified as a chain on the host), prefix sums on the host, extraction of POS / FLAG / CIGAR into file-wide arrays -- which
// stay in device memory, owned by the spl_bam (DeviceReads): read sets on this device are laid out from them by kernels
// (spl_devpack.hip, add_segment_device), the host gets copies only when it asks (fetch_device_reads).
namespace {
// Device memory of the decode (devmem above: the file image and the inflated stream are the buffers that made it necessary).
// The decode pipeline's streams (decode_share's Pipe), kept by the process between calls.
namespace pipestreams {
constexpr int NCOPY = 2;
struct Kept { int device, n_copy; hipStream_t a, b, up, cp[NCOPY]; };
static std::mutex &mu() { static std::mutex m; return m; }
static std::vector<Kept> &kept() { static std::vector<Kept> *v = new std::vector<Kept>(); return *v; }
// priority levels in use (SPL_STREAM_PRIORITIES=0: no) and their range on the current device
static bool levels(int *least_out, int *greatest_out)
{
    int least = 0, greatest = 0;
    const char *pe = getenv("SPL_STREAM_PRIORITIES");
    const bool on = !(pe && pe[0] == '0') && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least;
    if (!on) (void)hipGetLastError();
    if (least_out) *least_out = least;
    if (greatest_out) *greatest_out = greatest;
    return on;
}
// k.n_copy copy-kernel streams, the decoding kernel's, the short kernels' and the upload's, on the current device (see Pipe::make)
static hipError_t create(Kept &k)
{
    int least = 0, greatest = 0;
    const char *pe = getenv("SPL_STREAM_PRIORITIES");
    const bool lv = levels(&least, &greatest);
    auto stream_at = [&](hipStream_t *s, int priority) { return lv ? hipStreamCreateWithPriority(s, hipStreamNonBlocking, priority) : hipStreamCreateWithFlags(s, hipStreamNonBlocking); };
    hipError_t e = stream_at(&k.a, greatest);
    if (e == hipSuccess) e = stream_at(&k.b, pe && pe[0] == '2' ? least : greatest);
    for (int q = 0; q < NCOPY && q < k.n_copy && e == hipSuccess; ++q) e = stream_at(&k.cp[q], greatest);
    if (e == hipSuccess && lv && !(pe && pe[0] == '4')) e = stream_at(&k.up, least);
    return e;
}
} // namespace pipestreams

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { devmem::put(p); }
    void release() { devmem::put(p); p = nullptr; }
    hipError_t get(size_t bytes, hipStream_t) { return devmem::get(&p, bytes ? bytes : 16, 'b'); }
    template <class T> T *as() const { return (T *)p; }
};
// A stretch of the file into a staging buffer: pread when the file is open (the page cache's bytes straight into the pinned
// buffer; copying out of the mapping instead faults 4 KiB pages into this process's page tables, a million of them for a
// 3.6 GB file, and every other run of a loop took 0.4 s longer for it), the mapping otherwise.
struct CopyJob { char *dst; const char *src; size_t n, per; int fd; size_t file_off; };
void copy_slice(size_t k, void *arg)
{
    const CopyJob &j = *(const CopyJob *)arg;
    size_t a = k * j.per;
    const size_t b = std::min(j.n, a + j.per);
    while (j.fd >= 0 && a < b) {
        const ssize_t got = pread(j.fd, j.dst + a, b - a, (off_t)(j.file_off + a));
        if (got <= 0) break; // (whatever is left comes from the mapping)
        a += (size_t)got;
    }
    if (a < b) memcpy(j.dst + a, j.src + a, b - a);
}
} // namespace

// What spl_bam_decode_device leaves on the device (and what spl_soa_upload brings up): placed records, BAM-native, and where each
// reference's (segment's) are.  Shared: the file that decoded them holds a reference, and so does every read set that was laid
// out from them (its WIDE reads' ops are read from the cigar array, not copied) -- the arrays go when the last one lets go.
struct DeviceReads {
    int device = 0;
    void *pos = nullptr, *flag = nullptr, *cig_off = nullptr, *cigar = nullptr;
    int64_t n_rec = 0, n_ops = 0;
    std::vector<int64_t> ref_first, ref_n, ref_max, ref_ops;
    std::atomic<int> refs{1};
};
static void free_device_reads(void *h)
{
    DeviceReads *r = (DeviceReads *)h;
    if (!r) return;
    if (r->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    (void)hipSetDevice(r->device);
    devmem::put(r->pos); devmem::put(r->flag); devmem::put(r->cig_off); devmem::put(r->cigar);
    if (have) (void)hipSetDevice(cur);
    delete r;
}

static void *host_array(size_t bytes) // (2 MiB-aligned, huge pages asked for: tens to hundreds of MB that are written once, front to back)
{
    const size_t huge = 2u << 20, size = (std::max<size_t>(bytes, 64) + huge - 1) / huge * huge;
    void *p = nullptr;
    if (posix_memalign(&p, huge, size) != 0) return nullptr;
    (void)madvise(p, size, MADV_HUGEPAGE);
    return p;
}

// Host copies of the reads a device decode left on the device (malloc'ed: the caller owns them), page-locked for the copy only.
// Nobody asks for them while the reads are counted on the device they were decoded on; a reader on the host (spl_bam_reads) or
// a read set on another device does, through the file (spl_bam_set_fetch), once.
static int fetch_device_reads(void *h, int32_t **pos_out, uint16_t **flag_out, uint32_t **cigoff_out, uint32_t **cigar_out)
{
    const DeviceReads *r = (const DeviceReads *)h;
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    const size_t n_rec = (size_t)r->n_rec, n_ops = (size_t)r->n_ops;
    struct Arr { void *host; const void *dev; size_t bytes; bool pinned; } arr[4] = {
        {nullptr, r->pos, 4 * n_rec, false}, {nullptr, r->flag, 2 * n_rec, false}, {nullptr, r->cig_off, 4 * (n_rec + 1), false}, {nullptr, r->cigar, 4 * n_ops, false}};
    bool ok = true;
    for (Arr &a : arr) { a.host = host_array(a.bytes); ok = ok && a.host; }
    hipError_t q = ok ? hipSetDevice(r->device) : hipSuccess;
    hipStream_t st = nullptr;
    if (ok && q == hipSuccess) q = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (ok && q == hipSuccess) {
        // The pages first, on several threads: locking a fresh array faults its pages in one by one on the calling thread -- 0.33 GB
        // of a 20 M-read sample were 50 ms of the 80 this call took (`process --keepReads`, profiles/r06x_keep_reads.txt) --, a touch
        // per 2 MiB from eight threads is the same zeroing side by side.  Then the arrays are locked side by side as well.
        {
            struct Touch { char *p; size_t bytes; };
            std::vector<Touch> spans;
            const size_t step = (size_t)16 << 20;
            for (Arr &a : arr) for (size_t at = 0; at < a.bytes; at += step) spans.push_back(Touch{(char *)a.host + at, std::min(step, a.bytes - at)});
            auto touch = [](size_t k, void *arg) {
                const Touch &t = (*(const std::vector<Touch> *)arg)[k];
                for (size_t at = 0; at < t.bytes; at += 4096) ((volatile char *)t.p)[at] = 0;
            };
            splpack::parallel_for(spans.size(), 8, touch, &spans);
            std::vector<std::thread> lockers;
            for (Arr &a : arr)
                if (a.bytes >= (1u << 20))
                    lockers.emplace_back([&a, r]() { a.pinned = hipSetDevice(r->device) == hipSuccess && hipHostRegister(a.host, a.bytes, hipHostRegisterDefault) == hipSuccess; });
            for (std::thread &t : lockers) t.join();
        }
        for (Arr &a : arr) if (q == hipSuccess && a.bytes) q = hipMemcpyAsync(a.host, a.dev, a.bytes, hipMemcpyDeviceToHost, st);
        if (q == hipSuccess) q = hipStreamSynchronize(st);
        for (Arr &a : arr) if (a.pinned) (void)hipHostUnregister(a.host);
    }
    if (st) (void)hipStreamDestroy(st);
    if (have) (void)hipSetDevice(cur);
    if (!ok || q != hipSuccess) {
        for (Arr &a : arr) free(a.host);
        return ok ? spl_set_error(SPL_ERR_HIP, "decoded reads back to the host: %s", hipGetErrorString(q)) : spl_set_error(SPL_ERR_NOMEM, "out of host memory for the decoded reads");
    }
    *pos_out = (int32_t *)arr[0].host; *flag_out = (uint16_t *)arr[1].host; *cigoff_out = (uint32_t *)arr[2].host; *cigar_out = (uint32_t *)arr[3].host;
    return SPL_OK;
}

// One more segment of a read set from reads that are on the device already, BAM-native (add_segment is the host's version):
// only noted here -- the layout kernel runs once for all such segments when the read set is finished (finish_reads).
static int add_segment_device(spl_ctx *c, spl_dreads *d, DeviceReads *dev, int64_t first, int64_t n_reads, int64_t n_ops, int32_t shift, int64_t max_end)
{
    (void)c;
    if (d->finished) return spl_set_error(SPL_ERR_ARG, "the read set is finished: no more segments");
    if (n_reads == 0) return SPL_OK;
    if (d->n_reads + n_reads > 0xfffffff0LL) return spl_set_error(SPL_ERR_ARG, "n_reads out of range (counters are 32-bit)");
    if (max_end >= 0 && max_end + (int64_t)shift > (int64_t)SPL_COORD_MAX)
        return spl_set_error(SPL_ERR_RANGE, "a read ends beyond coordinate %d once its segment is moved by %d: split the shard (spliser_amd/shard.py)",
                             SPL_COORD_MAX, shift);
    if (n_ops > 0xfffffff0LL) return spl_set_error(SPL_ERR_ARG, "more than 2^32 CIGAR ops in one segment: use more shards");
    if (first < 0 || first + n_reads > dev->n_rec) return spl_set_error(SPL_ERR_ARG, "segment outside the device arrays");
    const uint32_t chunk = 1u << d->chunk_shift;
    const uint32_t n_chunks = spl_layout_seg_chunks(first, n_reads, chunk);
    if ((uint64_t)d->n_chunks + n_chunks > (1ull << (32 - d->chunk_shift))) return spl_set_error(SPL_ERR_ARG, "too many reads in one read set: use more shards");
    size_t g = 0;
    while (g < d->groups.size() && d->groups[g].src != dev) ++g;
    if (g == d->groups.size()) {
        d->groups.emplace_back();
        d->groups[g].src = dev;
        dev->refs.fetch_add(1, std::memory_order_relaxed);
    }
    spl_dreads::Group &grp = d->groups[g];
    spl_layout_seg ls;
    ls.first = first; ls.n_reads = n_reads; ls.chunk0 = d->n_chunks; ls.dev0 = grp.n_chunks; ls.n_chunks = n_chunks; ls.shift = shift;
    grp.segs.push_back(ls);
    grp.n_chunks += n_chunks;
    d->segs.emplace_back();
    spl_dreads::Segment &seg = d->segs.back();
    seg.group = (int)g; seg.group_seg = grp.segs.size() - 1;
    seg.n_reads = n_reads; seg.n_ops = n_ops; seg.shift = shift;
    d->n_reads += n_reads;
    d->n_cigar += n_ops;
    d->n_chunks += n_chunks;
    return SPL_OK;
}

// ---- the decode itself: a share of the file (or all of it) on one device -----------------------------------------------
// A pipeline over WINDOWS of the share's BGZF blocks (a fixed number of blocks each, two buffers of inflated stream):
//   copy stream   the file's bytes, piece by piece through the staging ring (one reader thread per staging buffer)
//   stream A      Huffman decoding of window k (spl_inflate_decode_kernel, a wave per block) as soon as its bytes have arrived
//   stream C      the copies the decoding left (a lane per block)
//   stream B      window k: CRC32, the scan for records, -- host: the chain of record boundaries, prefix sums -- extraction
// The kernels of C and B are lanes waiting for memory (a launch takes as long as its slowest lane's chain of copies, however few
// blocks it has), the one of A is arithmetic: they share the device well.  A record
// that straddles two windows: the bytes from the first block that is not done with to the window's end are copied in front of the
// next window's buffer (its head room), so that scan and extraction see them in one piece; the blocks are not inflated twice.
namespace {
struct ShareOut { DeviceReads *reads = nullptr; int64_t n_all = 0; bool to_host = false; bool published = false; bool more_tokens = false; };
// What the caller of decode_share does with the share's reads, called by decode_share itself as its LAST act before it gives its
// buffers, streams and events back -- which takes 10 ms for a large file, and whoever waits for the file's references need not.
typedef std::function<int(ShareOut &)> Publish;
}
static int decode_share(spl_ctx *c, spl_bam *bam, const spl_bam_share *share, ShareOut &res, const Publish &publish, bool all_token_room = false);

static void fill_whole(spl_bam *bam, spl_bam_share &sh)
{
    sh.block_lo = 0; sh.block_hi = sh.block_own = spl_bam_block_count(bam); sh.tid_lo = 0; sh.tid_hi = spl_bam_n_ref(bam) + 1;
    sh.u_lo = spl_bam_header_end(bam); sh.u_hi = 0; // (u_hi: the stream's end, which decode_share knows from the last block)
}

extern "C" int spl_bam_decode_device(spl_ctx *c, spl_bam *bam, int *on_device_out)
{
    if (!c || !bam) return spl_set_error(SPL_ERR_ARG, "spl_bam_decode_device: null argument");
    if (on_device_out) *on_device_out = 0;
    if (!spl_bam_claim_for_device(bam)) return SPL_OK; // (being decoded already, by whoever asked first: nothing to do here)
    ShareOut res;
    const auto t_call = std::chrono::steady_clock::now();
    double t_pub = 0;
    const Publish adopt = [&](ShareOut &res) -> int {
        int rc = SPL_OK;
        DeviceReads *keep = res.reads;
        const bool eager = getenv("SPL_NO_DEVICE_PACK") != nullptr; // (the round-trip over the host, for A/B: host copies now, nothing kept here)
        if (eager) {
            int32_t *h_pos = nullptr; uint16_t *h_flag = nullptr; uint32_t *h_cigoff = nullptr, *h_cigar = nullptr;
            rc = fetch_device_reads(keep, &h_pos, &h_flag, &h_cigoff, &h_cigar);
            if (rc == SPL_OK) {
                rc = spl_bam_adopt(bam, h_pos, h_flag, h_cigoff, h_cigar, keep->ref_first.data(), keep->ref_n.data(), keep->ref_max.data(), res.n_all);
                if (rc) { free(h_pos); free(h_flag); free(h_cigoff); free(h_cigar); }
            }
            free_device_reads(keep);
        } else {
            spl_bam_set_device_reads(bam, keep, free_device_reads);
            spl_bam_set_fetch(bam, fetch_device_reads);
            rc = spl_bam_adopt(bam, nullptr, nullptr, nullptr, nullptr, keep->ref_first.data(), keep->ref_n.data(), keep->ref_max.data(), res.n_all);
            if (rc) spl_bam_set_device_reads(bam, nullptr, nullptr);
        }
        if (rc == SPL_OK && on_device_out) *on_device_out = 1;
        t_pub = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_call).count();
        return rc;
    };
    int rc = decode_share(c, bam, nullptr, res, adopt);
    if (rc == SPL_OK && res.more_tokens) { res = ShareOut(); rc = decode_share(c, bam, nullptr, res, adopt, true); }
    if (rc == SPL_OK && !res.to_host && res.reads && !res.published) rc = adopt(res); // (SPL_PUBLISH_LATE only: decode_share publishes what it returns)
    // whatever went wrong on the way (device memory, a HIP error, a file this path does not take): the file must not be left
    // without a decoder -- the host threads take it (a no-op when the arrays were adopted); spl_last_error keeps the reason
    (void)spl_bam_device_gives_up(bam);
    if (rc != SPL_OK && getenv("SPL_BAM_TIMING")) fprintf(stderr, "[spl_bam_decode_device] failed (%s): host decoder instead\n", spl_last_error());
    if (getenv("SPL_BAM_TIMING"))
        fprintf(stderr, "[spl_bam_decode_device] the file's references were complete %.4f s after the call, the call returned at %.4f s (buffers, streams and events given back)\n", t_pub,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_call).count());
    return SPL_OK;
}

// Share k of the file's plan (spl_bam_share_plan) on this context's device.  The file must have been reserved for the device
// decoders (spl_bam_reserve_device) and every share must be decoded by somebody: the last one to finish makes the file complete
// -- or, if any share could not be done on its device, hands the whole file to the host threads.
extern "C" int spl_bam_decode_device_share(spl_ctx *c, spl_bam *bam, int k, int *on_device_out)
{
    if (!c || !bam) return spl_set_error(SPL_ERR_ARG, "spl_bam_decode_device_share: null argument");
    if (on_device_out) *on_device_out = 0;
    spl_bam_share sh;
    int rc = spl_bam_share_get(bam, k, &sh);
    if (rc) return rc;
    ShareOut res;
    const Publish report = [&](ShareOut &res) -> int { // (a share that is done: reported at once, the last one to report completes the file)
        spl_bam_set_fetch(bam, fetch_device_reads);
        const int rc2 = spl_bam_share_done(bam, k, res.reads, free_device_reads, res.reads->ref_first.data(), res.reads->ref_n.data(), res.reads->ref_max.data(), res.n_all, 0);
        if (rc2 == SPL_OK && on_device_out) *on_device_out = 1;
        return rc2;
    };
    rc = decode_share(c, bam, &sh, res, report);
    if (rc == SPL_OK && res.more_tokens) { res = ShareOut(); rc = decode_share(c, bam, &sh, res, report, true); }
    if (res.published) return rc;
    const bool failed = rc != SPL_OK || res.to_host || !res.reads;
    if (!failed) return report(res); // (SPL_PUBLISH_LATE only: decode_share publishes what it returns)
    if (getenv("SPL_BAM_TIMING")) fprintf(stderr, "[spl_bam_decode_device] share %d not done on its device (%s)\n", k, rc ? spl_last_error() : "handed to the host");
    if (res.reads) { free_device_reads(res.reads); res.reads = nullptr; }
    return spl_bam_share_done(bam, k, nullptr, free_device_reads, nullptr, nullptr, nullptr, res.n_all, 1);
}

static int decode_share(spl_ctx *c, spl_bam *bam, const spl_bam_share *share, ShareOut &res, const Publish &publish, bool all_token_room)
{
    HIP_TRY(hipSetDevice(c->device));
    const bool timing = getenv("SPL_BAM_TIMING") != nullptr;
    auto host_now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = host_now();
    auto to_host = [&](const char *why) { // not a file for this path: the host threads take it (and find the words for what is wrong with it)
        if (timing) fprintf(stderr, "[spl_bam_decode_device] handing the file to the host decoder: %s\n", why);
        spl_bam_note_decline(bam, why);
        res.to_host = true;
        return SPL_OK;
    };
    size_t fsize = 0;
    const uint8_t *image = spl_bam_image(bam, &fsize);
    const int n_ref = spl_bam_n_ref(bam);
    int rc = SPL_OK;
    // ---- everything the streams touch is declared before them: what is declared last goes first, and that is the guard that waits
    constexpr int NBUF = 4, NCOPY = pipestreams::NCOPY; // (at most: n_buf buffers and n_copy streams for the copying kernels are used, below)
    DevBuf d_image, d_stream[NBUF], d_zwork[NBUF], d_blocks0, d_status0, d_recs, d_blocks, d_status, d_scan, d_recoff, d_opoff, d_pos, d_flag, d_cigoff, d_cigar, d_tid, d_maxend, d_bounds, d_nbounds;
    std::vector<spl_zblock> blocks, blocks0; // (blocks0: the first window's, for its early launch)
    std::unique_ptr<uint32_t[]> status;
    std::unique_ptr<spl_bscan[]> scan;
    std::unique_ptr<uint64_t[]> rec_off, op_off;
    std::vector<unsigned long long> maxend;
    std::vector<uint64_t> bounds;
    uint32_t n_bounds = 0;
    struct Pipe { // stream A, stream B, their events; waits for everything on the way out, whichever way that is
        spl_ctx *c;
        hipStream_t a = nullptr, b = nullptr, cp[NCOPY] = {}, up2 = nullptr; // (up2: a second stream for the file's pieces, beside the context's copy stream)
        hipEvent_t k1[NBUF] = {}, freed[NBUF] = {}, setup = nullptr; // (k1: a token buffer's decoding is done; freed: a stream buffer's records are extracted)
        std::vector<hipEvent_t> piece;
        std::vector<hipEvent_t> dec; // window w's Huffman decoding is done: nobody reads its pieces of the file image again (their slots of the ring are free)
        std::vector<hipEvent_t> k2;  // window w's copying kernel is done: its bytes are there, its token buffer is free
        explicit Pipe(spl_ctx *ctx) : c(ctx) {}
        bool keep = false;
        int n_copy_made = 0;
        using Kept = pipestreams::Kept;
        hipStream_t up = nullptr; // the file's pieces: a stream of this pipeline's own at the low level (a queue of its own again); without levels the context's copy stream
        hipStream_t upload() const { return up ? up : c->copy; }
        static std::mutex &kept_mu() { return pipestreams::mu(); }
        static std::vector<Kept> &kept() { return pipestreams::kept(); }
        // (only the streams that will be used: the runtime deals streams out to a few hardware queues, and one more stream --
        //  made, never used -- put the decoding and the copying kernels behind each other: 0.75 s instead of 0.43 for a 14 GB file)
        // Which hardware queue a stream gets is the runtime's choice -- the least used of four per priority level, whatever else
        // the process has made streams for -- and two of this pipeline's streams on one queue are one behind the other: the
        // first call of a process that already had a context took 0.70 s where its later calls took 0.36, windows of 39 ms
        // instead of 15, the same as every call with GPU_MAX_HW_QUEUES=2 (profiles/r06s_first_call.txt).  Priority levels have
        // queues of their own: the three kernel streams are made at the high level -- nothing else of this library is, so each
        // gets a queue -- and the file's pieces go up on a stream of the pipeline's own at the low level (its barrier packets wait
        // for DMA: nobody's kernels may stand behind them, the counting passes of chromosomes that are complete included).  The
        // process keeps the four for its next call on the device (~Pipe): made anew after they had been given back they were seen
        // to share a queue again.  SPL_STREAM_PRIORITIES=0: all at the normal level and made per call, as before (2, 3, 4: the
        // variants measured in the profile).
        hipError_t make(size_t n_pieces, int n_copy_streams, bool second_upload)
        {
            const char *pe = getenv("SPL_STREAM_PRIORITIES");
            hipError_t e = hipSuccess;
            keep = pipestreams::levels(nullptr, nullptr) && !(pe && pe[0] == '3');
            if (keep) { // the streams an earlier call of this process left (and their queues with them: see ~Pipe)
                std::lock_guard<std::mutex> lock(kept_mu());
                std::vector<Kept> &v = kept();
                for (size_t k = 0; k < v.size(); ++k)
                    if (v[k].device == c->device && v[k].n_copy == n_copy_streams) { a = v[k].a; b = v[k].b; up = v[k].up; for (int q = 0; q < NCOPY; ++q) cp[q] = v[k].cp[q]; v.erase(v.begin() + (long)k); break; }
            }
            n_copy_made = n_copy_streams;
            if (!a) {
                Kept made{c->device, n_copy_streams, nullptr, nullptr, nullptr, {}};
                e = pipestreams::create(made);
                a = made.a; b = made.b; up = made.up;
                for (int q = 0; q < NCOPY; ++q) cp[q] = made.cp[q];
            }
            if (e == hipSuccess && second_upload) e = hipStreamCreateWithFlags(&up2, hipStreamNonBlocking);
            for (int k = 0; k < NBUF && e == hipSuccess; ++k) {
                e = hipEventCreateWithFlags(&k1[k], hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&freed[k], hipEventDisableTiming);
            }
            if (e == hipSuccess) e = hipEventCreateWithFlags(&setup, hipEventDisableTiming);
            piece.assign(n_pieces, nullptr); // (a piece's event is made by the reader that sends the piece, just before it is recorded: 430 of them
                                             //  made here were 35-50 ms of a new process's call before its first byte was on its way)
            dec.assign(2, nullptr); // (the first two windows' are needed before the number of windows is known)
            k2.assign(2, nullptr);
            for (int k = 0; k < 2 && e == hipSuccess; ++k) {
                e = hipEventCreateWithFlags(&dec[(size_t)k], hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&k2[(size_t)k], hipEventDisableTiming);
            }
            return e;
        }
        hipError_t make_windows(size_t n_win)
        {
            hipError_t e = hipSuccess;
            while (dec.size() < n_win && e == hipSuccess) { dec.push_back(nullptr); e = hipEventCreateWithFlags(&dec.back(), hipEventDisableTiming); }
            while (k2.size() < n_win && e == hipSuccess) { k2.push_back(nullptr); e = hipEventCreateWithFlags(&k2.back(), hipEventDisableTiming); }
            return e;
        }
        ~Pipe()
        {
            if (a) (void)hipStreamSynchronize(a);
            for (int k = 0; k < NCOPY; ++k) if (cp[k]) (void)hipStreamSynchronize(cp[k]);
            if (b) (void)hipStreamSynchronize(b);
            if (c->copy) (void)hipStreamSynchronize(c->copy);
            if (up) (void)hipStreamSynchronize(up);
            if (up2) (void)hipStreamSynchronize(up2);
            for (hipEvent_t e : piece) if (e) (void)hipEventDestroy(e);
            for (hipEvent_t e : dec) if (e) (void)hipEventDestroy(e);
            for (hipEvent_t e : k2) if (e) (void)hipEventDestroy(e);
            for (int k = 0; k < NBUF; ++k) { if (k1[k]) (void)hipEventDestroy(k1[k]); if (freed[k]) (void)hipEventDestroy(freed[k]); }
            if (setup) (void)hipEventDestroy(setup);
            if (up2) (void)hipStreamDestroy(up2);
            // The three streams stay with the process for its next call on this device (idle: everything on them has been waited
            // for above): the queues the runtime gave them the first time were each its own, and a stream made anew after they
            // had been given back was seen to share one (windows of 25 ms instead of 15 from the second call on).
            if (keep && a && b && cp[0]) {
                std::lock_guard<std::mutex> lock(kept_mu());
                if (kept().size() < 16) {
                    Kept k{c->device, n_copy_made, a, b, up, {}};
                    for (int q = 0; q < NCOPY; ++q) k.cp[q] = cp[q];
                    kept().push_back(k);
                    return;
                }
            }
            if (a) (void)hipStreamDestroy(a);
            for (int k = 0; k < NCOPY; ++k) if (cp[k]) (void)hipStreamDestroy(cp[k]);
            if (b) (void)hipStreamDestroy(b);
            if (up) (void)hipStreamDestroy(up);
        }
    } pipe(c);
    // ---- the share: which blocks, which bytes of the file
    int walk_rc = SPL_OK;
    double t_walk = 0;
    spl_bam_share sh;
    size_t byte_lo = 0, byte_hi = fsize;
    if (share) {
        sh = *share; // (its plan walked the directory)
        spl_bam_block_info b0, b1;
        spl_bam_block_get(bam, (size_t)sh.block_lo, &b0);
        spl_bam_block_get(bam, (size_t)sh.block_hi - 1, &b1);
        byte_lo = (size_t)b0.data_off;
        byte_hi = std::min(fsize, (size_t)b1.data_off + b1.data_len + 8);
    }
    const size_t n_bytes = byte_hi - byte_lo;
    size_t free_b = 0, total_b = 0;
    // (SPL_DEV_FREE_LIMIT_MB: the tests' way to a device with little memory -- every look at the free memory below sees at most this much)
    const size_t free_limit = getenv("SPL_DEV_FREE_LIMIT_MB") ? (size_t)std::max(1L, atol(getenv("SPL_DEV_FREE_LIMIT_MB"))) << 20 : (size_t)-1;
    const double slack = free_limit == (size_t)-1 ? (double)((size_t)1 << 30) : (double)((size_t)8 << 20);
    size_t used_b = 0; // what this call has taken so far (only the tests' limit needs to be told)
    auto look_at_free = [&]() -> hipError_t {
        const hipError_t e = hipMemGetInfo(&free_b, &total_b);
        free_b += devmem::held_bytes(c->device); // (given back before a request fails)
        if (free_limit != (size_t)-1) free_b = std::min(free_b, free_limit > used_b ? free_limit - used_b : (size_t)0);
        return e;
    };
    HIP_TRY(look_at_free());
    double t_stage = 0, t_image = 0, t_pipe = 0, t_early_bufs = 0, t_early = 0; // (SPL_BAM_TIMING: the way to the first window, seconds after the call began)
    // The file's bytes: page cache -> staging buffer -> device.  One reader thread per staging buffer: it preads its pieces (a
    // piece = what a buffer holds, dealt round-robin), sends each on its way itself and records the piece's event behind it.
    // (a reader fills its buffer from the page cache at 8-10 GB/s before the copy engine takes 0.6 ms to empty it: three readers
    //  bring 25-40 GB/s, and since the kernels got through a window in 16 ms a large file waited for its bytes: six for those)
    int n_copy = 1; // streams the copying kernels take turns on (1: one window's copies behind the other's)
    if (const char *e = getenv("SPL_INFLATE_COPY_STREAMS")) n_copy = std::min(NCOPY, std::max(1, atoi(e)));
    const bool two_up = getenv("SPL_UPLOAD_STREAMS") && atoi(getenv("SPL_UPLOAD_STREAMS")) >= 2; // (the file's pieces on two streams in turn)
    rc = ensure_stage(c, n_bytes >= ((size_t)4 << 30) && !getenv("SPL_STAGE_BUFFERS") ? 6 : 0);
    if (rc) return rc;
    const size_t n_stage = c->stage.size();
    t_stage = host_now() - t_begin;
    // On the device the file exists as a RING of pieces, not whole: only the Huffman decoding reads it, a window of blocks at a
    // time, so a piece's slot is given to the piece R further on as soon as the last window that reads it has been decoded --
    // device memory for about four windows' worth of the file (3.4 GB of a 14 GB file's) instead of all of it, whatever the
    // file's size; fresh device memory is what a new process's first call waits for (6-15 ms a gigabyte).  A block's data is
    // less than 64 KiB: with every piece goes the beginning of the next (TAIL), so that a block that begins in a piece is in one
    // piece of memory in that piece's slot.
    const size_t TAIL = (size_t)65536 + 4096; // (and what the decoder may read beyond a block's end, SPL_Z_IMAGE_PAD)
    const size_t slot = c->stage[0].bytes;    // a staging buffer holds a piece and its tail
    if (slot < 4 * TAIL) return to_host("staging buffers too small for the file image's pieces");
    const size_t piece = slot - TAIL;
    const size_t n_pieces = (n_bytes + piece - 1) / piece;
    size_t win_blocks = (size_t)49152;
    if (const char *e = getenv("SPL_INFLATE_WINDOW_BLOCKS")) win_blocks = (size_t)std::max(2, atoi(e));
    if (!share) spl_bam_walk_some(bam, (size_t)5 << 28); // (1.25 GB of the directory, 3 ms: 49 152 blocks of a file that compresses 4x are 0.8 GB; a share's plan has walked all of it)
    size_t ring = n_pieces; // slots
    size_t win_bytes_cap = (size_t)-1; // a window ends where its blocks' bytes in the file would exceed this (so that four windows fit the ring)
    double per_block_file = 65536.0;
    uint32_t max_block_file = 0; // the most bytes of file any block known so far has
    {
        const size_t n_known = spl_bam_block_count(bam);
        const size_t b_lo = share ? (size_t)share->block_lo : 0, b_hi = share ? (size_t)share->block_hi : n_known;
        double per_block = 65536.0; // bytes of file per block, from the blocks known so far
        if (b_hi > b_lo + 1) {
            spl_bam_block_info f0, f1;
            spl_bam_block_get(bam, b_lo, &f0);
            spl_bam_block_get(bam, b_hi - 1, &f1);
            per_block = (double)(f1.data_off - f0.data_off) / (double)(b_hi - 1 - b_lo);
        }
        per_block_file = per_block;
        for (size_t i = b_lo; i < b_hi; ++i) { // (the largest block so far: the token room below goes by it)
            spl_bam_block_info bi;
            spl_bam_block_get(bam, i, &bi);
            max_block_file = std::max<uint32_t>(max_block_file, bi.data_len);
        }
        const double est_win = per_block * (double)win_blocks;
        size_t want = (size_t)(4.2 * est_win / (double)piece) + 4;
        if (const char *e = getenv("SPL_IMAGE_RING_PIECES")) want = (size_t)std::max(4, atoi(e)); // (tests: a ring of a few pieces)
        if (want < n_pieces) {
            ring = want;
            win_bytes_cap = (ring - 3) * piece / 4;
        }
    }
    // Token room per block (decoding kernel -> copying kernel).  The worst case is 82 048 bytes -- five bytes of tokens for four
    // of output -- and what BAM blocks take is 1.3 to 1.8 times their compressed size (29 KB where a block deflates to 16.6 KB):
    // two token buffers at the worst case are 8 GB of the 26 GB a new process's first call waited for.  So: 2.25 times the
    // LARGEST compressed block the directory holds by now + 4 KB, and a block that needs more says so (SPL_Z_TOKENS) -- the share
    // is then decoded again with all of it: zlib's Z_HUFFMAN_ONLY streams in the tests do, and until round 6, when the room
    // went by the MEAN block (2.5 times it), so did every call on the bench's own htslib-shaped file -- its blocks deflate to
    // 7.6 KB on average and to 15-18 KB at most, and one block in the first window always needed more than 23 KB: the call began
    // twice, 0.37 s instead of 0.32 (profiles/r06G_decode_five_waves.txt).  SPL_Z_ALL_TOKEN_ROOM=1: always all.
    uint32_t tok_stride = SPL_Z_TOKEN_STRIDE;
    if (!all_token_room && !getenv("SPL_Z_ALL_TOKEN_ROOM")) {
        const double want = getenv("SPL_Z_ROOM_BY_MEAN") ? 2.5 * per_block_file + 4096.0 : 2.25 * (double)std::max<uint32_t>(max_block_file, (uint32_t)per_block_file) + 4096.0;
        tok_stride = (uint32_t)std::min<double>((double)SPL_Z_TOKEN_STRIDE, std::max(8192.0, want));
        if (const char *e = getenv("SPL_Z_TOKEN_ROOM")) tok_stride = (uint32_t)std::max(512, atoi(e)); // (tests: a room most blocks do not fit)
        tok_stride = (tok_stride + 63u) & ~63u;
        if (tok_stride > SPL_Z_TOKEN_STRIDE) tok_stride = SPL_Z_TOKEN_STRIDE;
    }
    // which decoding kernel: the denser one for files whose blocks deflate well (spl_inflate.hip)
    const uint32_t k1_flags = per_block_file <= 12288.0 ? SPL_Z_LAUNCH_DENSE : 0u;
    if (timing) fprintf(stderr, "[spl_bam_decode_device] device %d: %u bytes of token room a block (%.0f bytes of file a block, %u at most)%s\n", c->device, tok_stride, per_block_file,
                        max_block_file, k1_flags ? ", the denser decoding kernel" : "");
    {   // a first look, before anything is allocated: the ring, two windows of a stream that is at most 64 KiB a block with their
        // token room, a fifth of the inflated stream for what is extracted; the exact sizes are checked again where they are known
        const double blocks_all = (double)n_bytes / std::max(per_block_file, 28.0) + 1.0, blocks_most = std::min((double)win_blocks, blocks_all);
        if ((double)ring * (double)slot + 0.2 * blocks_all * 65536.0 + 2.0 * blocks_most * (65536.0 + (double)tok_stride) + slack > (double)free_b)
            return to_host("not enough device memory");
    }
    HIP_TRY(d_image.get(ring * slot + SPL_Z_IMAGE_PAD, c->copy));
    used_b += ring * slot;
    t_image = host_now() - t_begin;
    if (timing) fprintf(stderr, "[spl_bam_decode_device] device %d: the file's %.1f MB as %zu pieces of %.1f MB, %zu slots on the device (%.1f MB)%s\n", c->device, n_bytes / 1e6, n_pieces,
                        piece / 1e6, ring, ring * slot / 1e6, ring < n_pieces ? "" : ": all of it");
    HIP_TRY(pipe.make(n_pieces, n_copy, two_up));
    t_pipe = host_now() - t_begin;
    // The extracted arrays' first size, BEFORE the first window has been scanned.  They used to be sized from that scan: four or five
    // requests for gigabytes on the host's loop between the first window and the second -- and on a box in use a request costs
    // 27-28 ms a gigabyte, the driver clearing what it hands out (profiles/r06s_first_call.txt, 7; nothing where the memory is the
    // process's own, kept from an earlier call).  So the host looks at a few blocks itself (spl_bam_sample_density: records and
    // CIGAR ops per inflated byte at three places), a quarter is added, the arrays are asked for with the windows' buffers ahead
    // of the first launch, and make_room below finds the room there -- or grows it, as before, where the sample was wrong.
    uint64_t cap_rec = 0, cap_ops = 0, n_rec = 0, n_ops = 0;
    auto early_room = [&](size_t b_lo, size_t b_hi, double inflated_bytes) -> int {
        if (cap_rec || getenv("SPL_NO_EARLY_ROOM")) return SPL_OK;
        uint64_t s_rec = 0, s_ops = 0, s_bytes = 0;
        if (!spl_bam_sample_density(bam, b_lo, b_hi, &s_rec, &s_ops, &s_bytes)) return SPL_OK;
        const double scale = 1.25 * inflated_bytes / (double)s_bytes;
        // (ops a read differ from gene to gene more than bytes a record do: three places of a 20 M-read file said 1.5 where the
        //  file had 2.15, and its arrays grew behind the last window -- the smaller the file, the more room its guess gets)
        const double scale_ops = (inflated_bytes < 16e9 ? 1.6 : 1.3) * inflated_bytes / (double)s_bytes;
        const uint64_t want_rec = (uint64_t)std::min((double)s_rec * scale, inflated_bytes / 36.0) + 1024;
        const uint64_t want_ops = (uint64_t)std::min((double)s_ops * scale_ops, inflated_bytes / 4.0) + 1024;
        HIP_TRY(look_at_free());
        if ((double)(14 * want_rec + 4 * want_ops) > (double)free_b / 3.0) return SPL_OK; // (a guess must not be what the windows' buffers go without)
        HIP_TRY(d_pos.get(4 * want_rec, pipe.b));
        HIP_TRY(d_flag.get(2 * want_rec, pipe.b));
        HIP_TRY(d_cigoff.get(4 * (want_rec + 1), pipe.b));
        HIP_TRY(d_cigar.get(4 * want_ops, pipe.b));
        HIP_TRY(d_tid.get(4 * want_rec, pipe.b));
        HIP_TRY(hipMemsetAsync(d_cigoff.p, 0, 4, pipe.b));
        used_b += (size_t)(14 * want_rec + 4 * want_ops);
        cap_rec = want_rec;
        cap_ops = want_ops;
        if (timing) fprintf(stderr, "[spl_bam_decode_device] device %d: room for %llu records and %llu CIGAR ops from %llu records in %.1f KB on the host\n", c->device,
                            (unsigned long long)want_rec, (unsigned long long)want_ops, (unsigned long long)s_rec, (double)s_bytes / 1e3);
        return SPL_OK;
    };
    std::vector<hipError_t> errs(n_stage, hipSuccess);
    std::vector<std::atomic<int>> sent(n_pieces); // piece k's copy and event are in the copy stream's queue (or will never be: errs)
    for (auto &f : sent) f.store(0, std::memory_order_relaxed);
    std::atomic<int> reader_failed(0);
    // what the readers need to know before they may overwrite a slot: which window reads a piece last (known when the windows
    // are: windows_known), and that this window's decoding kernel is on its stream (launched_pub), with its event behind it
    std::vector<uint32_t> piece_last_win(n_pieces, 0);
    std::atomic<int> windows_known(0), stop_upload(0);
    std::atomic<size_t> launched_pub(0);
    std::mutex up_mu;               // (the readers sleep on this while they wait for a slot, the host loop while it waits for a piece:
    std::condition_variable up_cv;  //  threads that spin instead eat the CPU time the process is granted, and the others with it)
    auto up_wake = [&]() { { std::lock_guard<std::mutex> lock(up_mu); } up_cv.notify_all(); };
    const int fd = spl_bam_fd(bam);
    char *const d_img = d_image.as<char>();
    if (two_up || pipe.up) HIP_TRY(hipStreamSynchronize(c->copy)); // (what was put on the copy stream for the image so far is done before the other stream writes into it)
    auto reader = [&](size_t t) {
        if (hipSetDevice(c->device) != hipSuccess) errs[t] = hipErrorInvalidDevice;
        spl_ctx::Stage &st = c->stage[t];
        for (size_t k = t; k < n_pieces; k += n_stage) {
            hipStream_t up = two_up && (t & 1u) ? pipe.up2 : pipe.upload();
            if (errs[t] == hipSuccess && k >= ring) { // the slot's last piece must have been read by everybody who reads it
                size_t w = 0;
                {
                    std::unique_lock<std::mutex> lock(up_mu);
                    up_cv.wait(lock, [&]() { return windows_known.load(std::memory_order_acquire) || stop_upload.load(std::memory_order_acquire); });
                    w = windows_known.load(std::memory_order_acquire) ? piece_last_win[k - ring] : 0;
                    up_cv.wait(lock, [&]() { return launched_pub.load(std::memory_order_acquire) > w || stop_upload.load(std::memory_order_acquire); });
                }
                if (stop_upload.load(std::memory_order_acquire)) errs[t] = hipErrorNotReady; // (the call is on its way out: nobody waits for this piece)
                else errs[t] = hipStreamWaitEvent(up, pipe.dec[w], 0);
            }
            if (errs[t] == hipSuccess && st.busy) { errs[t] = hipEventSynchronize(st.done); st.busy = false; }
            if (errs[t] == hipSuccess) {
                const size_t off = k * piece, n = std::min(piece + TAIL, n_bytes - off);
                CopyJob job{st.host, (const char *)image + byte_lo + off, n, n, fd, byte_lo + off};
                copy_slice(0, &job);
                errs[t] = hipMemcpyAsync(d_img + (k % ring) * slot, st.host, n, hipMemcpyHostToDevice, up);
                if (errs[t] == hipSuccess) { errs[t] = hipEventRecord(st.done, up); st.busy = true; }
                if (errs[t] == hipSuccess) errs[t] = hipEventCreateWithFlags(&pipe.piece[k], hipEventDisableTiming); // (piece k is this reader's alone; whoever waits for it looks after sent[k])
                if (errs[t] == hipSuccess) errs[t] = hipEventRecord(pipe.piece[k], up);
            }
            if (errs[t] != hipSuccess) reader_failed.store(1, std::memory_order_release);
            sent[k].store(1, std::memory_order_release);
            up_wake();
        }
    };
    auto wait_sent = [&](size_t q) { // piece q's copy is on its stream (or never will be: reader_failed)
        if (sent[q].load(std::memory_order_acquire)) return;
        std::unique_lock<std::mutex> lock(up_mu);
        up_cv.wait(lock, [&]() { return sent[q].load(std::memory_order_acquire) != 0; });
    };
    // where a block's data lies in the ring (off: its offset in this share's stretch of the file), and the windows' ends
    auto ring_at = [&](uint64_t file_off) { const size_t o = (size_t)file_off - byte_lo, pc = o / piece; return (uint64_t)((pc % ring) * slot + (o - pc * piece)); };
    auto piece_of = [&](uint64_t file_off) { return ((size_t)file_off - byte_lo) / piece; };
    std::vector<uint64_t> foff; // the blocks' offsets in the file (spl_zblock.in is where they lie in the ring)
    size_t blocks_listed = 0;   // blocks and foff hold the whole file's list already (made by the walker thread)
    struct Crew { // (declared behind everything its threads touch: joined before any of that goes away, on every way out)
        std::vector<std::thread> threads;
        std::atomic<int> *stop = nullptr;
        void join() { for (std::thread &t : threads) if (t.joinable()) t.join(); }
        std::function<void()> wake;
        ~Crew() { if (stop) stop->store(1, std::memory_order_release); if (wake) wake(); join(); } // (a reader waiting for a slot is told that nobody will free it)
    } crew;
    crew.stop = &stop_upload;
    crew.wake = up_wake;
    // ---- the whole file: its first window is on its streams before the directory is complete.  The directory of the file's first
    // gigabyte takes 3 ms, the rest of a 14 GB file 25: the first window's decoding and copying kernels (23 ms) run beside that.
    // They get a list of blocks and status words of their own (the file's are allocated when their number is known).
    const uint64_t HEAD = (uint64_t)8 << 20; // room in front of a window's bytes for what the window before left unfinished
    const uint8_t *const image0 = d_image.as<uint8_t>(); // (a block's `in` is its place in the ring)
    size_t pieces_waited = 0;
    size_t early = 0;  // blocks of the first window if it has been launched already (0: not)
    size_t early2 = 0; // where the second window ends if ITS decoding kernel has been launched as well (0: not)
    struct Joiner { std::thread t; ~Joiner() { if (t.joinable()) t.join(); } } walker; // (the rest of the directory, walked beside the second window's launch)
    // (the readers start when the device memory the call can size by now has been asked for -- where the driver clears what it
    //  hands out the process's other HIP calls wait for it anyway; SPL_READERS_FIRST=1: the readers first, as until round 6)
    bool readers_started = false;
    auto start_readers = [&]() {
        if (readers_started) return;
        readers_started = true;
        for (size_t t = 0; t < n_stage; ++t) crew.threads.emplace_back(reader, t);
    };
    if (getenv("SPL_READERS_FIRST")) start_readers();
    // where a window that begins with block b0 ends: win_blocks blocks on, or where its bytes in the file would not fit a quarter of the ring
    auto window_end = [&](const std::vector<uint64_t> &off, size_t b0, size_t n_all) {
        size_t b1 = std::min(n_all, b0 + win_blocks);
        if (win_bytes_cap != (size_t)-1 && off[b1 - 1] - off[b0] > win_bytes_cap) {
            size_t lo2 = b0 + 1, hi2 = b1; // the first b with off[b - 1] - off[b0] > cap is where it must end at the latest
            while (lo2 < hi2) { const size_t mid = lo2 + (hi2 - lo2) / 2; if (off[mid - 1] - off[b0] > win_bytes_cap) hi2 = mid; else lo2 = mid + 1; }
            b1 = std::max(b0 + 1, lo2 - 1);
        }
        return b1;
    };
    if (!share) // the rest of the directory -- 25-30 ms of a 14 GB file -- and the list of all its blocks (9 ms) on a thread of their own, from here: beside the first windows' uploads and launches
        walker.t = std::thread([&]() {
            const double w0 = host_now();
            walk_rc = spl_bam_walk_all(bam);
            t_walk = host_now() - w0;
            if (walk_rc) return;
            const size_t n_all = spl_bam_block_count(bam);
            if (n_all == 0 || n_all > 0xfffffff0ull) return;
            blocks.resize(n_all);
            foff.resize(n_all);
            for (size_t i = 0; i < n_all; ++i) {
                spl_bam_block_info bi;
                spl_bam_block_get(bam, i, &bi);
                foff[i] = bi.data_off;
                blocks[i].in = ring_at(bi.data_off); blocks[i].out = bi.uoff; blocks[i].in_len = bi.data_len; blocks[i].out_len = bi.isize; blocks[i].crc = bi.crc; blocks[i].pad = 0;
            }
            blocks_listed = n_all;
        });
    if (!share && !getenv("SPL_INFLATE_NO_EARLY")) {
        const size_t n_known = spl_bam_block_count(bam);
        if (n_known >= win_blocks || (n_known >= 2 && spl_bam_walk_complete(bam))) {
            // (two windows' worth of the list when the directory so far holds them: the second window's decoding kernel goes out early too)
            int early_zw = 2;
            if (const char *e = getenv("SPL_INFLATE_TOKEN_BUFFERS")) early_zw = atoi(e);
            const bool two = n_known >= 2 * win_blocks && early_zw >= 2 && !getenv("SPL_INFLATE_ONE_EARLY");
            std::vector<uint64_t> off0(two ? 2 * win_blocks : std::min(win_blocks, n_known));
            blocks0.resize(off0.size());
            for (size_t i = 0; i < off0.size(); ++i) {
                spl_bam_block_info bi;
                spl_bam_block_get(bam, i, &bi);
                off0[i] = bi.data_off;
                blocks0[i].in = ring_at(bi.data_off); blocks0[i].out = bi.uoff; blocks0[i].in_len = bi.data_len; blocks0[i].out_len = bi.isize; blocks0[i].crc = bi.crc; blocks0[i].pad = 0;
            }
            const size_t b1 = window_end(off0, 0, off0.size()), b_most = std::min(win_blocks, n_known); // (b_most: what any window of this file can have -- the buffers serve later windows too)
            const size_t b2 = two ? window_end(off0, b1, off0.size()) : b1; // (where the second window ends)
            blocks0.resize(b2);
            const size_t work0 = spl_dev_inflate_work_bytes2((uint32_t)b_most, tok_stride);
            HIP_TRY(look_at_free());
            if ((double)HEAD + (double)b_most * 65536.0 + (double)work0 + 2.0 * slack < (double)free_b) {
                HIP_TRY(d_stream[0].get(HEAD + (uint64_t)b_most * 65536u + 256, c->copy)); // (no block inflates to more than 64 KiB)
                HIP_TRY(d_zwork[0].get(work0, c->copy));
                used_b += (size_t)HEAD + b_most * 65536u + work0;
                // (the second window's buffers and the extracted arrays too while the device is idle -- see early_room)
                HIP_TRY(look_at_free());
                const bool more_windows = two || (double)n_bytes / std::max(per_block_file, 28.0) > 1.02 * (double)b_most; // (as far as the directory so far can tell)
                if (more_windows && (double)HEAD + (double)b_most * 65536.0 + (double)work0 + 2.0 * slack < (double)free_b && !getenv("SPL_NO_EARLY_ROOM")) {
                    HIP_TRY(d_stream[1].get(HEAD + (uint64_t)b_most * 65536u + 256, c->copy));
                    HIP_TRY(d_zwork[1].get(work0, c->copy));
                    used_b += (size_t)HEAD + b_most * 65536u + work0;
                }
                {
                    spl_bam_block_info last;
                    spl_bam_block_get(bam, n_known - 1, &last);
                    const double ratio = (double)(last.uoff + last.isize) / (double)std::max<uint64_t>(last.data_off + last.data_len, 1);
                    rc = early_room(0, n_known, ratio * (double)n_bytes);
                    if (rc) return rc;
                }
                start_readers();
                t_early_bufs = host_now() - t_begin;
                HIP_TRY(d_blocks0.get(sizeof(spl_zblock) * b2, c->copy));
                HIP_TRY(d_status0.get(4 * b2, c->copy));
                HIP_TRY(hipMemcpyAsync(d_blocks0.p, blocks0.data(), sizeof(spl_zblock) * b2, hipMemcpyHostToDevice, pipe.a));
                HIP_TRY(hipMemsetAsync(d_status0.p, 0xff, 4 * b2, pipe.a));
                const size_t need = std::min(n_pieces, piece_of(off0[b1 - 1]) + 1); // (the piece its last block begins in holds all of it)
                for (; pieces_waited < need; ++pieces_waited) {
                    wait_sent(pieces_waited);
                    if (reader_failed.load(std::memory_order_acquire)) { for (hipError_t e : errs) HIP_TRY(e); }
                    HIP_TRY(hipStreamWaitEvent(pipe.a, pipe.piece[pieces_waited], 0));
                }
                uint8_t *const stream0 = d_stream[0].as<uint8_t>() + HEAD - blocks0[0].out;
                const double w_in = (double)(off0[b1 - 1] + blocks0[b1 - 1].in_len - off0[0]), w_out = (double)(blocks0[b1 - 1].out + blocks0[b1 - 1].out_len - blocks0[0].out);
                {
                    splprof::Scope p("spl_inflate_decode_kernel", pipe.a, w_in + w_out);
                    HIP_TRY((hipError_t)spl_dev_launch_inflate_decode3(image0, d_blocks0.as<spl_zblock>(), (uint32_t)b1, d_status0.as<uint32_t>(), d_zwork[0].p, tok_stride, k1_flags, pipe.a));
                }
                HIP_TRY(hipEventRecord(pipe.k1[0], pipe.a));
                HIP_TRY(hipEventRecord(pipe.dec[0], pipe.a));
                HIP_TRY(hipStreamWaitEvent(pipe.cp[0], pipe.k1[0], 0));
                {
                    splprof::Scope p("spl_inflate_copy_kernel", pipe.cp[0], w_out);
                    HIP_TRY((hipError_t)spl_dev_launch_inflate_copy2(d_blocks0.as<spl_zblock>(), (uint32_t)b1, stream0, d_status0.as<uint32_t>(), d_zwork[0].p, tok_stride, pipe.cp[0]));
                }
                HIP_TRY(hipEventRecord(pipe.k2[0], pipe.cp[0]));
                early = b1;
                t_early = host_now() - t_begin;
                // The second window's Huffman decoding behind the first's, before the directory is complete: the walk of the rest
                // (25-30 ms of a large file, on a thread of its own from here) and the list of all blocks (10 ms) kept the host away
                // from its loop until the first window's kernels had long finished -- 30 ms of an idle device at the start of a
                // call whose clock is the decoding kernels one behind the other (profiles/r04ap_second_window_early.txt).
                // Its copying kernel, CRC32 and the rest are the loop's (they need the second byte buffer and the file's lists).
                if (b2 > b1) HIP_TRY(look_at_free());
                if (b2 > b1 && (double)work0 + 2.0 * slack < (double)free_b) {
                    if (!d_zwork[1].p) { HIP_TRY(d_zwork[1].get(work0, c->copy)); used_b += work0; }
                    const size_t need2 = std::min(n_pieces, piece_of(off0[b2 - 1]) + 1);
                    for (; pieces_waited < need2; ++pieces_waited) {
                        wait_sent(pieces_waited);
                        if (reader_failed.load(std::memory_order_acquire)) { for (hipError_t e : errs) HIP_TRY(e); }
                        HIP_TRY(hipStreamWaitEvent(pipe.a, pipe.piece[pieces_waited], 0));
                    }
                    {
                        splprof::Scope p("spl_inflate_decode_kernel", pipe.a, (double)(off0[b2 - 1] + blocks0[b2 - 1].in_len - off0[b1]) + (double)(blocks0[b2 - 1].out + blocks0[b2 - 1].out_len - blocks0[b1].out));
                        HIP_TRY((hipError_t)spl_dev_launch_inflate_decode3(image0, d_blocks0.as<spl_zblock>() + b1, (uint32_t)(b2 - b1), d_status0.as<uint32_t>() + b1, d_zwork[1].p, tok_stride, k1_flags, pipe.a));
                    }
                    HIP_TRY(hipEventRecord(pipe.k1[1], pipe.a));
                    HIP_TRY(hipEventRecord(pipe.dec[1], pipe.a));
                    early2 = b2;
                }
            }
        }
    }
    if (!share) start_readers(); // (a share's: behind its buffers, below -- its directory is complete and the rest takes a millisecond)
    if (!share) {
        if (walker.t.joinable()) walker.t.join();
        else { const double w0 = host_now(); walk_rc = spl_bam_walk_all(bam); t_walk = host_now() - w0; } // (what the early part left)
        if (timing) fprintf(stderr, "[spl_bam_decode_device] (the directory walk took %.4f s%s)\n", t_walk, early2 ? ", beside the second window's launch" : "");
        if (walk_rc) return to_host("block directory");
        fill_whole(bam, sh);
    }
    const double t_walked = host_now() - t_begin;
    const size_t n_blocks_file = spl_bam_block_count(bam);
    const size_t lo = (size_t)sh.block_lo, hi = (size_t)sh.block_hi, n_blocks = hi - lo;
    if (n_blocks == 0 || n_blocks > 0xfffffff0ull) return to_host("no blocks");
    const bool last_share = hi == n_blocks_file, first_share = lo == 0;
    // The share's OWN blocks: the records that begin in them are its records.  The blocks behind them (a share that is not the
    // file's last) are inflated for the end of its last record and otherwise left alone -- the next share's device has them too.
    const size_t n_own = (size_t)sh.block_own - lo;
    if (n_own == 0 || n_own > n_blocks) return to_host("a share without blocks of its own");
    if (!(blocks_listed == n_blocks && lo == 0)) { // (not listed beside the walk: a share, a file whose second window did not go out early)
        blocks.resize(n_blocks);
        foff.resize(n_blocks);
        for (size_t i = 0; i < n_blocks; ++i) {
            spl_bam_block_info bi;
            spl_bam_block_get(bam, lo + i, &bi);
            foff[i] = bi.data_off;
            blocks[i].in = ring_at(bi.data_off); blocks[i].out = bi.uoff; blocks[i].in_len = bi.data_len; blocks[i].out_len = bi.isize; blocks[i].crc = bi.crc; blocks[i].pad = 0;
        }
    }
    const double t_blocks = host_now() - t_begin;
    const uint64_t stream_begin = blocks[0].out, stream_len = blocks[n_blocks - 1].out + blocks[n_blocks - 1].out_len; // (of the share; offsets are the file's)
    // ---- windows
    win_blocks = std::min(win_blocks, n_blocks);
    // (Short first windows -- an eighth, a quarter, a half -- were tried to get the copying kernel started earlier, and cost
    // more than they bring: a launch of the copying kernel takes as long as one lane needs for its block, 9-12 ms however few
    // blocks it has, so three short launches are 20 ms of that kernel's stream for less than one window's worth of blocks.)
    std::vector<size_t> win_at(1, 0);
    while (win_at.back() < n_blocks) win_at.push_back(window_end(foff, win_at.back(), n_blocks));
    const size_t n_win = win_at.size() - 1;
    if (early && early != win_at[1]) return to_host("the first window changed under the decoder"); // (cannot happen: the directory only grows)
    if (early2 && (n_win < 2 || early2 != win_at[2])) return to_host("the second window changed under the decoder");
    HIP_TRY(pipe.make_windows(n_win));
    for (size_t w = 0; w < n_win; ++w) // (a piece is read last by the window of the last block that begins in it; pieces without one: by the window before)
        for (size_t pc = piece_of(foff[win_at[w]]), pe = piece_of(foff[win_at[w + 1] - 1]); pc <= pe && pc < n_pieces; ++pc) piece_last_win[pc] = (uint32_t)w;
    const size_t n_early = early2 ? 2 : early ? 1 : 0; // windows whose decoding kernel is out already: they work on the early list of blocks and its status words
    launched_pub.store(n_early, std::memory_order_release);
    windows_known.store(1, std::memory_order_release);
    up_wake();
    uint64_t win_cap = 0;
    for (size_t k = 0; k < n_win; ++k) {
        const size_t b0 = win_at[k], b1 = win_at[k + 1];
        win_cap = std::max(win_cap, blocks[b1 - 1].out + blocks[b1 - 1].out_len - blocks[b0].out);
    }
    size_t most_blocks = 0;
    for (size_t k = 0; k < n_win; ++k) most_blocks = std::max(most_blocks, win_at[k + 1] - win_at[k]);
    const size_t work_bytes = spl_dev_inflate_work_bytes2((uint32_t)most_blocks, tok_stride);
    HIP_TRY(look_at_free());
    // Windows in flight.  A window has two buffers, each free again when its reader is done: the TOKENS (decoding kernel ->
    // copying kernel) when the window's copying kernel has run, the inflated BYTES (copying kernel -> CRC32, scan, extraction)
    // when its records are extracted.  n_zw of the first, n_buf of the second, and each kernel waits for its own buffer only: the
    // decoding of window k + n_zw begins when the copying of window k is done, the copying of window k + n_buf when window k's
    // records are out.  (Until round 4 a window had one slot for both, and with two slots the decoding of window k + 2 stood
    // behind the extraction of window k: decode, copy, CRC32, scan, extract of one window in a row, 29 ms for two windows of work
    // on an htslib-shaped file; profiles/r04x_window_dependencies.txt.)  Fresh device memory costs 6-15 ms a gigabyte and a token
    // buffer has four of them, a byte buffer three: a process that decodes one file (the command line) stays with two of each.
    int want_buf = 2, want_zw = 2;
    if (const char *e = getenv("SPL_INFLATE_BUFFERS")) want_buf = std::min(NBUF, std::max(1, atoi(e)));
    if (const char *e = getenv("SPL_INFLATE_TOKEN_BUFFERS")) want_zw = std::min(NBUF, std::max(1, atoi(e)));
    const int n_buf = (int)std::min<size_t>((size_t)want_buf, n_win), n_zw = (int)std::min<size_t>((size_t)want_zw, n_win);
    if ((double)n_buf * ((double)win_cap + (double)HEAD) + (double)n_zw * (double)work_bytes + (double)(stream_len - stream_begin) * 0.2 + slack > (double)free_b)
        return to_host("not enough device memory for the inflated stream");
    for (int k = 0; k < std::max(n_buf, n_zw); ++k) {
        if (k == 0 && early) continue; // (the first window has its buffers, large enough for any)
        if (k < n_buf && !d_stream[k].p) HIP_TRY(d_stream[k].get(HEAD + win_cap + 256, c->copy)); // (the early ones are large enough for any window)
        if (k < n_zw && !d_zwork[k].p) HIP_TRY(d_zwork[k].get(work_bytes, c->copy));
    }
    rc = early_room(lo, lo + n_own, (double)(blocks[n_own - 1].out + blocks[n_own - 1].out_len - stream_begin)); // (a share, or a file whose first window did not go out early: nothing runs yet)
    if (rc) return rc;
    start_readers();
    // where the placed records of a window's blocks begin (scan -> extraction, one window at a time on stream B): room for a window's
    // blocks and what an 8 MB carry can hold of ordinary ones; a window with more blocks than that is extracted by walking
    const size_t recs_blocks = std::min(n_own, win_blocks + (size_t)32768);
    HIP_TRY(d_recs.get(2 * (size_t)SPL_BS_REC_CAP * recs_blocks, c->copy));
    HIP_TRY(d_blocks.get(sizeof(spl_zblock) * n_blocks, c->copy));
    HIP_TRY(d_status.get(4 * n_blocks, c->copy));
    HIP_TRY(d_scan.get(sizeof(spl_bscan) * n_blocks, c->copy));
    HIP_TRY(d_recoff.get(8 * (n_blocks + 1), c->copy));
    HIP_TRY(d_opoff.get(8 * (n_blocks + 1), c->copy));
    HIP_TRY(d_maxend.get(8 * (size_t)std::max(n_ref, 1), c->copy));
    const uint32_t cap = (uint32_t)std::max(n_ref, 1) * 4u + 64u;
    HIP_TRY(d_bounds.get(16 * (size_t)cap, c->copy));
    HIP_TRY(d_nbounds.get(4, c->copy));
    const double t_bufs = host_now() - t_begin;
    status.reset(new (std::nothrow) uint32_t[n_blocks]); // (four arrays the device fills window by window: not zeroed first, that was 10 ms of a large file's start)
    scan.reset(new (std::nothrow) spl_bscan[n_blocks]);
    rec_off.reset(new (std::nothrow) uint64_t[n_blocks + 1]);
    op_off.reset(new (std::nothrow) uint64_t[n_blocks + 1]);
    if (!status || !scan || !rec_off || !op_off) return spl_set_error(SPL_ERR_NOMEM, "out of host memory");
    rec_off[0] = op_off[0] = 0;
    HIP_TRY(hipMemcpyAsync(d_blocks.p, blocks.data(), sizeof(spl_zblock) * n_blocks, hipMemcpyHostToDevice, pipe.b));
    HIP_TRY(hipMemsetAsync(d_status.p, 0xff, 4 * n_blocks, pipe.b));
    HIP_TRY(hipMemsetAsync(d_maxend.p, 0, 8 * (size_t)std::max(n_ref, 1), pipe.b));
    HIP_TRY(hipMemsetAsync(d_nbounds.p, 0, 4, pipe.b));
    HIP_TRY(hipEventRecord(pipe.setup, pipe.b));
    // (what the early windows' kernels said about their blocks goes to its place among the file's when they have said it: the loop, behind the window's copying kernel)
    HIP_TRY(hipStreamWaitEvent(pipe.a, pipe.setup, 0));
    for (int k = 0; k < n_copy; ++k) HIP_TRY(hipStreamWaitEvent(pipe.cp[k], pipe.setup, 0));
    const uint64_t H = spl_bam_header_end(bam);
    if (first_share && H > stream_len) return to_host("no BAM header");
    size_t first = 0; // the first block that holds more than BAM header
    if (first_share)
        while (first < n_blocks && blocks[first].out + blocks[first].out_len <= H && !(blocks[first].out + blocks[first].out_len == H && first + 1 == n_blocks)) ++first;
    // the extracted arrays: as large as the first window says the share will need and a fifth more, larger when that was wrong
    auto make_room = [&](uint64_t need_rec, uint64_t need_ops, double part_done) -> int {
        if (need_rec <= cap_rec && need_ops <= cap_ops && cap_rec) return SPL_OK;
        const double scale = 1.2 / std::max(part_done, 1e-6); // (a fifth more than the windows so far say: growing later means fresh memory and moving what is there)
        if (timing && cap_rec) fprintf(stderr, "[spl_bam_decode_device] device %d: the extracted arrays grow (%llu records and %llu ops needed %.0f %% into the stretch, room for %llu and %llu)\n",
                                       c->device, (unsigned long long)need_rec, (unsigned long long)need_ops, 100.0 * part_done, (unsigned long long)cap_rec, (unsigned long long)cap_ops);
        const uint64_t want_rec = std::max<uint64_t>(need_rec, (uint64_t)((double)need_rec * scale)) + 1024;
        const uint64_t want_ops = std::max<uint64_t>(need_ops, (uint64_t)((double)need_ops * scale)) + 1024;
        DevBuf pos2, flag2, cigoff2, cigar2, tid2;
        HIP_TRY(pos2.get(4 * want_rec, pipe.b));
        HIP_TRY(flag2.get(2 * want_rec, pipe.b));
        HIP_TRY(cigoff2.get(4 * (want_rec + 1), pipe.b));
        HIP_TRY(cigar2.get(4 * want_ops, pipe.b));
        HIP_TRY(tid2.get(4 * want_rec, pipe.b));
        if (cap_rec) { // what the windows so far have left (n_rec records, n_ops ops) moves
            HIP_TRY(hipMemcpyAsync(pos2.p, d_pos.p, 4 * n_rec, hipMemcpyDeviceToDevice, pipe.b));
            HIP_TRY(hipMemcpyAsync(flag2.p, d_flag.p, 2 * n_rec, hipMemcpyDeviceToDevice, pipe.b));
            HIP_TRY(hipMemcpyAsync(cigoff2.p, d_cigoff.p, 4 * (n_rec + 1), hipMemcpyDeviceToDevice, pipe.b));
            HIP_TRY(hipMemcpyAsync(cigar2.p, d_cigar.p, 4 * n_ops, hipMemcpyDeviceToDevice, pipe.b));
            HIP_TRY(hipMemcpyAsync(tid2.p, d_tid.p, 4 * n_rec, hipMemcpyDeviceToDevice, pipe.b));
            HIP_TRY(hipStreamSynchronize(pipe.b));
        } else {
            HIP_TRY(hipMemsetAsync(cigoff2.p, 0, 4, pipe.b));
        }
        std::swap(d_pos.p, pos2.p); std::swap(d_flag.p, flag2.p); std::swap(d_cigoff.p, cigoff2.p); std::swap(d_cigar.p, cigar2.p); std::swap(d_tid.p, tid2.p);
        cap_rec = want_rec;
        cap_ops = want_ops;
        return SPL_OK;
    };
    auto win_range = [&](size_t k, size_t &b0, size_t &b1) { b0 = win_at[k]; b1 = win_at[k + 1]; };
    auto stream0_of = [&](size_t k) { size_t b0, b1; win_range(k, b0, b1); return d_stream[k % (size_t)n_buf].as<uint8_t>() + HEAD - blocks[b0].out; }; // (indexed with offsets into the whole stream)
    // the Huffman decoding of window k on stream A: behind the pieces of the file it reads and behind the copying kernel that read its token buffer last
    auto launch_k1 = [&](size_t k, bool wait, bool &launched_it) -> int {
        size_t b0, b1;
        win_range(k, b0, b1);
        launched_it = false;
        const size_t need = std::min(n_pieces, piece_of(foff[b1 - 1]) + 1); // (the piece its last block begins in holds all of it)
        if (!wait) // (a window ahead of the one the host is at: only if its bytes are on their way already)
            for (size_t q = pieces_waited; q < need; ++q) if (!sent[q].load(std::memory_order_acquire)) return SPL_OK;
        for (; pieces_waited < need; ++pieces_waited) {
            wait_sent(pieces_waited);
            if (reader_failed.load(std::memory_order_acquire)) { for (hipError_t e : errs) HIP_TRY(e); }
            HIP_TRY(hipStreamWaitEvent(pipe.a, pipe.piece[pieces_waited], 0));
        }
        if (k >= (size_t)n_zw) HIP_TRY(hipStreamWaitEvent(pipe.a, pipe.k2[k - (size_t)n_zw], 0));
        const double w_in = (double)(foff[b1 - 1] + blocks[b1 - 1].in_len - foff[b0]), w_out = (double)(blocks[b1 - 1].out + blocks[b1 - 1].out_len - blocks[b0].out);
        {
            splprof::Scope p("spl_inflate_decode_kernel", pipe.a, w_in + w_out);
            HIP_TRY((hipError_t)spl_dev_launch_inflate_decode3(image0, d_blocks.as<spl_zblock>() + b0, (uint32_t)(b1 - b0), d_status.as<uint32_t>() + b0, d_zwork[k % (size_t)n_zw].p, tok_stride, k1_flags, pipe.a));
        }
        HIP_TRY(hipEventRecord(pipe.k1[k % (size_t)n_zw], pipe.a));
        HIP_TRY(hipEventRecord(pipe.dec[k], pipe.a));
        launched_pub.store(k + 1, std::memory_order_release); // (the readers may give this window's pieces' slots to later pieces, behind that event)
        up_wake();
        launched_it = true;
        return SPL_OK;
    };
    // its copying kernel: behind the decoding and behind the extraction of the window that had its byte buffer
    auto launch_k2 = [&](size_t k) -> int {
        size_t b0, b1;
        win_range(k, b0, b1);
        hipStream_t cs = pipe.cp[k % (size_t)n_copy];
        HIP_TRY(hipStreamWaitEvent(cs, pipe.k1[k % (size_t)n_zw], 0));
        if (k >= (size_t)n_buf) HIP_TRY(hipStreamWaitEvent(cs, pipe.freed[k % (size_t)n_buf], 0));
        const bool from_early = k < n_early; // (its decoding kernel left the blocks' status in the early list's words)
        {
            splprof::Scope p("spl_inflate_copy_kernel", cs, (double)(blocks[b1 - 1].out + blocks[b1 - 1].out_len - blocks[b0].out));
            HIP_TRY((hipError_t)spl_dev_launch_inflate_copy2((from_early ? d_blocks0 : d_blocks).as<spl_zblock>() + b0, (uint32_t)(b1 - b0), stream0_of(k),
                                                             (from_early ? d_status0 : d_status).as<uint32_t>() + b0, d_zwork[k % (size_t)n_zw].p, tok_stride, cs));
        }
        HIP_TRY(hipEventRecord(pipe.k2[k], cs));
        return SPL_OK;
    };
    int64_t n_all = 0;
    // The chain of record boundaries: from the end of the BAM header, or (a later share) from the first record that begins in the
    // share's first block -- which the plan found on the host, and which the share in front of this one must arrive at: every
    // share's chain ends exactly where the next one's begins (checked below), so every record is somebody's and nobody's twice.
    uint64_t expect = first_share ? H : sh.u_lo;
    const bool expect_known = true;
    int32_t last_tid = -1;
    size_t carry = 0; // the first block whose records are not all extracted yet
    size_t launched = n_early, copying = early ? 1 : 0; // windows whose decoding / whose copying kernel has been put on its stream
    std::vector<double> t_win; // (SPL_BAM_TIMING: when each window's scan was back on the host)
    const double t_setup = host_now() - t_begin;
    for (size_t k = 0; k < n_win; ++k) {
        size_t b0, b1;
        win_range(k, b0, b1);
        if (spl_bam_cancelled(bam)) return to_host("the file is being closed");
        // Whatever can be put on its stream now.  A copying kernel: its window's decoding is there, and the extraction of the window that
        // had its byte buffer is on stream B (windows k .. k + n_buf - 1).  A decoding kernel: the copying kernel that reads its token
        // buffer before it is on its stream (n_zw windows back), and its pieces of the file are on their way (window k's are waited for).
        for (;;) {
            bool progress = false;
            if (copying < launched && copying < k + (size_t)n_buf) {
                rc = launch_k2(copying);
                if (rc) return rc;
                ++copying;
                progress = true;
            }
            if (launched < n_win && launched < copying + (size_t)n_zw) {
                bool did = false;
                rc = launch_k1(launched, launched == k, did);
                if (rc) return rc;
                if (did) { ++launched; progress = true; }
            }
            if (!progress) break;
        }
        const size_t slot = k % (size_t)n_buf;
        const bool more = b1 < n_blocks; // (behind the last window nothing follows: a share's tail blocks hold the end of its last record)
        const uint32_t nb = (uint32_t)(b1 - b0);
        const uint64_t win_end = blocks[b1 - 1].out + blocks[b1 - 1].out_len;
        uint8_t *const stream0 = stream0_of(k);
        const size_t s0 = k == 0 ? b0 : carry; // scan and extraction begin with what the window before left
        const size_t b1s = std::min(b1, n_own); // ... and end with the share's own blocks
        HIP_TRY(hipStreamWaitEvent(pipe.b, pipe.k2[k], 0));
        if (k < n_early) HIP_TRY(hipMemcpyAsync(d_status.as<uint32_t>() + b0, d_status0.as<uint32_t>() + b0, 4 * (size_t)nb, hipMemcpyDeviceToDevice, pipe.b)); // (an early window's status words: to their place among the file's)
        const double win_out = (double)(win_end - blocks[b0].out);
        { splprof::Scope p("spl_crc32_kernel", pipe.b, win_out); HIP_TRY((hipError_t)spl_dev_launch_crc32(stream0, d_blocks.as<spl_zblock>() + b0, nb, d_status.as<uint32_t>() + b0, pipe.b)); }
        const bool with_recs = b1s > s0 && b1s - s0 <= recs_blocks && !getenv("SPL_EXTRACT_WALK");
        if (b1s > s0) {
            splprof::Scope p("spl_bam_scan_kernel", pipe.b, (double)(blocks[b1s - 1].out + blocks[b1s - 1].out_len - blocks[s0].out));
            HIP_TRY((hipError_t)spl_dev_launch_bam_scan(stream0, win_end, H, n_ref, 0, n_ref + 1, d_blocks.as<spl_zblock>() + s0, (uint32_t)(b1s - s0), d_scan.as<spl_bscan>() + s0,
                                                        more ? 1 : 0, with_recs ? d_recs.as<uint16_t>() : nullptr, pipe.b));
        }
        HIP_TRY(hipMemcpyAsync(status.get() + b0, d_status.as<uint32_t>() + b0, 4 * (size_t)nb, hipMemcpyDeviceToHost, pipe.b));
        if (b1s > s0) HIP_TRY(hipMemcpyAsync(scan.get() + s0, d_scan.as<spl_bscan>() + s0, sizeof(spl_bscan) * (b1s - s0), hipMemcpyDeviceToHost, pipe.b));
        HIP_TRY(hipStreamSynchronize(pipe.b));
        if (timing) t_win.push_back(host_now() - t_begin);
        for (size_t i = b0; i < b1; ++i)
            if (status[i] == SPL_Z_TOKENS && tok_stride < SPL_Z_TOKEN_STRIDE) { // (a block wants more token room than this file's were given: again, with all of it)
                if (timing) fprintf(stderr, "[spl_bam_decode_device] block %zu needs more than %u bytes of token room: the share again with %u\n", i, tok_stride, SPL_Z_TOKEN_STRIDE);
                res.more_tokens = true;
                return SPL_OK;
            }
        for (size_t i = b0; i < b1; ++i)
            if (status[i] != SPL_Z_OK) return to_host("a block did not inflate (or its CRC32 is wrong)");
        // Which of the window's blocks are done with: all whose records end inside it.  A block near the window's end may have
        // looked for its first record, or walked its last one, into bytes that are not there yet: it is told by its flag, or
        // -- within reach of the end -- by anything being wrong with it, and is looked at again with the next window.
        const uint64_t reach = b1s > s0 ? std::min<uint64_t>((win_end - blocks[s0].out) / 2, ((uint64_t)5 << 18)) : 0;
        size_t b_done = std::max(b1s, s0);
        for (size_t b = s0; b < b1s; ++b) {
            spl_bscan &sc = scan[b];
            if (first_share && b < first) { // (BAM header only: nothing to extract)
                sc.n_placed = 0;
                rec_off[b + 1] = rec_off[b];
                op_off[b + 1] = op_off[b];
                continue;
            }
            const char *wrong = nullptr;
            if (sc.flags & SPL_BS_CORRUPT) wrong = "a record contradicts itself";
            else if (sc.flags & SPL_BS_NO_START) wrong = "no record boundary found near a block";
            else if (sc.flags & SPL_BS_NEEDS_HOST) wrong = "a CIGAR parked in a CG tag";
            else if (sc.flags & SPL_BS_UNSORTED) wrong = "not sorted by reference";
            else if (expect_known && sc.start != expect) wrong = "a guessed record boundary did not hold";
            else if (sc.n_placed && sc.tid_first < last_tid) wrong = "not sorted by reference";
            if ((sc.flags & SPL_BS_INCOMPLETE) || (wrong && more && blocks[b].out + reach >= win_end)) { b_done = b; break; }
            if (wrong) return to_host(wrong);
            if (sc.n_placed) last_tid = sc.tid_last;
            expect = sc.reached;
            n_all += sc.n_all;
            rec_off[b + 1] = rec_off[b] + sc.n_placed;
            op_off[b + 1] = op_off[b] + sc.n_ops;
        }
        if (b_done < b1s && b1 == n_blocks) return to_host(last_share ? "the file ends inside a record" : "a record runs past the blocks behind a share");
        if (b_done < b1s && win_end - blocks[b_done].out > HEAD) return to_host("a record larger than the room between two windows");
        if (op_off[b_done] > 0xfffffff0ull) return to_host("more than 2^32 CIGAR operations");
        if (b_done > s0) {
            rc = make_room(rec_off[b_done], op_off[b_done], (double)(blocks[b_done - 1].out + blocks[b_done - 1].out_len - stream_begin) / (double)std::max<uint64_t>(stream_len - stream_begin, 1));
            if (rc) return rc;
            n_rec = rec_off[b_done];
            n_ops = op_off[b_done];
            const size_t nd = b_done - s0;
            HIP_TRY(hipMemcpyAsync(d_recoff.as<uint64_t>() + s0, rec_off.get() + s0, 8 * nd, hipMemcpyHostToDevice, pipe.b));
            HIP_TRY(hipMemcpyAsync(d_opoff.as<uint64_t>() + s0, op_off.get() + s0, 8 * nd, hipMemcpyHostToDevice, pipe.b));
            HIP_TRY(hipMemcpyAsync(d_scan.as<spl_bscan>() + s0, scan.get() + s0, sizeof(spl_bscan) * nd, hipMemcpyHostToDevice, pipe.b));
            splprof::Scope p("spl_bam_extract_kernel", pipe.b, (double)(blocks[b_done - 1].out + blocks[b_done - 1].out_len - blocks[s0].out));
            HIP_TRY((hipError_t)spl_dev_launch_bam_extract(stream0, win_end, n_ref, 0, n_ref + 1, d_blocks.as<spl_zblock>() + s0, (uint32_t)nd, d_scan.as<spl_bscan>() + s0,
                                                           d_recoff.as<uint64_t>() + s0, d_opoff.as<uint64_t>() + s0, d_pos.as<int32_t>(), d_flag.as<uint16_t>(),
                                                           d_cigoff.as<uint32_t>(), d_cigar.as<uint32_t>(), d_tid.as<int32_t>(), d_maxend.as<unsigned long long>(),
                                                           with_recs ? d_recs.as<uint16_t>() : nullptr, pipe.b));
        }
        if (b_done < b1s && k + 1 < n_win) // what is left of this window: in front of the next one's bytes
            HIP_TRY(hipMemcpyAsync(stream0_of(k + 1) + blocks[b_done].out, stream0 + blocks[b_done].out, (size_t)(win_end - blocks[b_done].out), hipMemcpyDeviceToDevice, pipe.b));
        HIP_TRY(hipEventRecord(pipe.freed[slot], pipe.b));
        carry = b_done;
    }
    if (carry < n_own) return to_host("a share's last blocks were never done with");
    // every record that begins in the share's own blocks has been walked: the walk must have arrived where the next share's begins
    if (!last_share && expect != sh.u_hi) return to_host("a share's records do not end where the next share's begin");
    if (last_share && expect_known && expect != stream_len) return to_host("the file ends inside a record");
    if (!cap_rec) { rc = make_room(0, 0, 1.0); if (rc) return rc; } // (a share without a record of its own)
    HIP_TRY((hipError_t)spl_dev_launch_bam_bounds(d_tid.as<int32_t>(), d_cigoff.as<uint32_t>(), n_rec, d_bounds.as<uint64_t>(), d_nbounds.as<uint32_t>(), cap, pipe.b));
    maxend.assign((size_t)std::max(n_ref, 1), 0);
    bounds.assign(2 * (size_t)cap, 0);
    HIP_TRY(hipMemcpyAsync(maxend.data(), d_maxend.p, 8 * maxend.size(), hipMemcpyDeviceToHost, pipe.b));
    HIP_TRY(hipMemcpyAsync(bounds.data(), d_bounds.p, 16 * (size_t)cap, hipMemcpyDeviceToHost, pipe.b));
    HIP_TRY(hipMemcpyAsync(&n_bounds, d_nbounds.p, 4, hipMemcpyDeviceToHost, pipe.b));
    HIP_TRY(hipStreamSynchronize(pipe.b));
    crew.join();
    for (hipError_t e : errs) HIP_TRY(e);
    if (n_bounds > cap) return to_host("not sorted by reference");
    struct Run { uint64_t first; int32_t tid; uint32_t op; };
    std::vector<Run> runs;
    for (uint32_t k = 0; k < n_bounds; ++k) runs.push_back(Run{bounds[2 * k], (int32_t)(uint32_t)bounds[2 * k + 1], (uint32_t)(bounds[2 * k + 1] >> 32)});
    std::sort(runs.begin(), runs.end(), [](const Run &a, const Run &b) { return a.first < b.first; });
    const size_t nr = (size_t)std::max(n_ref, 1);
    // The records stay where they are, BAM-native in device memory: a read set on this device is laid out from them by kernels
    // (spl_devpack.hip), and the host gets copies only if somebody asks the file for them (fetch_device_reads).
    DeviceReads *keep = new (std::nothrow) DeviceReads();
    if (!keep) return spl_set_error(SPL_ERR_NOMEM, "out of host memory");
    keep->device = c->device;
    keep->n_rec = (int64_t)n_rec; keep->n_ops = (int64_t)n_ops;
    keep->ref_first.assign(nr, 0); keep->ref_n.assign(nr, 0); keep->ref_max.assign(nr, 0); keep->ref_ops.assign(nr, 0);
    for (size_t k = 0; k < runs.size(); ++k) {
        const int32_t t = runs[k].tid;
        if (t < 0 || t >= n_ref || (k && t <= runs[k - 1].tid)) { delete keep; return to_host("not sorted by reference"); }
        const bool last = k + 1 == runs.size();
        keep->ref_first[(size_t)t] = (int64_t)runs[k].first;
        keep->ref_n[(size_t)t] = (int64_t)((last ? n_rec : runs[k + 1].first) - runs[k].first);
        keep->ref_ops[(size_t)t] = (int64_t)((last ? n_ops : (uint64_t)runs[k + 1].op) - runs[k].op);
        keep->ref_max[(size_t)t] = (int64_t)maxend[(size_t)t];
    }
    keep->pos = d_pos.p; keep->flag = d_flag.p; keep->cig_off = d_cigoff.p; keep->cigar = d_cigar.p;
    d_pos.p = d_flag.p = d_cigoff.p = d_cigar.p = nullptr; // (the caller owns them from here)
    res.reads = keep;
    res.n_all = n_all;
    if (timing) fprintf(stderr, "[spl_bam_decode_device] device %d: blocks %zu..%zu, %.1f MB -> %.1f MB inflated in %zu window%s, %llu placed records of %lld: %.4f s\n", c->device, lo, hi,
                        n_bytes / 1e6, (stream_len - stream_begin) / 1e6, n_win, n_win == 1 ? "" : "s", (unsigned long long)n_rec, (long long)n_all, host_now() - t_begin);
    if (timing) {
        fprintf(stderr, "[spl_bam_decode_device] device %d: staging buffers at %.4f s, the ring's memory %.4f, streams and events %.4f, the first window's buffers %.4f, its kernels on their streams %.4f\n",
                c->device, t_stage, t_image, t_pipe, t_early_bufs, t_early);
        fprintf(stderr, "[spl_bam_decode_device] device %d: directory at %.4f s, block list %.4f, buffers %.4f, set up %.4f; windows scanned at", c->device, t_walked, t_blocks, t_bufs, t_setup);
        for (double t : t_win) fprintf(stderr, " %.3f", t);
        fprintf(stderr, " s\n");
    }
    // The decode's device buffers go back to the process's pool BEFORE the waiters are told (nothing is in flight: a moment's work):
    // what they allocate next -- read sets, their control blocks -- finds them there, as it did when this function's return came
    // first (a control block of 1 GB that had to come fresh from the driver cost 9 ms of the call's last 20).  Streams, events,
    // the readers' threads and the host's lists go afterwards, on this thread's own time.
    if (getenv("SPL_PUBLISH_LATE")) return SPL_OK; // (A/B: the caller tells the waiters when everything of this function's has gone, as until round 4)
    d_image.release();
    for (int k = 0; k < NBUF; ++k) { d_stream[k].release(); d_zwork[k].release(); }
    d_blocks0.release(); d_status0.release(); d_recs.release(); d_blocks.release(); d_status.release(); d_scan.release(); d_recoff.release(); d_opoff.release();
    d_tid.release(); d_maxend.release(); d_bounds.release(); d_nbounds.release();
    res.published = true;
    const int told = publish(res);
    // ... and not at once: 500 events and five streams destroyed are 10 ms of the HIP runtime's locks, which the thread that was
    // told above needs now -- for the layout kernels, the chunk order's upload, the counting launches (its 9 ms took 17 beside
    // this thread's clearing up; profiles/r04ab_tail.txt).  This thread stands aside until the file is being closed, or 0.1 s.
    spl_bam_linger(bam, 0.1);
    return told;
}

// What a read set's device segments need between the arrays and the counters, on the context's main stream, nothing for the host
// to wait for: per group the chunks' descriptors and cost estimates (from their numbers of reads and ops), the chunk order of the
// range kernel from the costs (host-packed chunks' are up already), then the layout kernel per group -- the records and the
// chunks' spl_chunk_meta -- and the range kernel can follow it directly.
static int launch_layout(spl_ctx *c, spl_dreads *d)
{
    const uint32_t chunk = 1u << d->chunk_shift;
    std::vector<spl_layout_params> lps;
    for (spl_dreads::Group &g : d->groups) {
        spl_layout_params lp;
        lp.src = spl_devreads{(const int32_t *)g.src->pos, (const uint16_t *)g.src->flag, (const uint32_t *)g.src->cig_off, (const uint32_t *)g.src->cigar};
        lp.n_rec = g.src->n_rec; lp.n_ops = g.src->n_ops;
        lp.chunks = g.d_chunks; lp.rec_base = (uint8_t *)g.slab; lp.meta = d->meta;
        // (the chunks' descriptors hold offsets read from the arrays: made anew whenever the layout runs)
        const int rc0 = spl_dev_launch_layout_map(&lp.src, g.d_segs, (uint32_t)g.segs.size(), g.n_chunks, chunk, g.d_chunks, d->cost, c->stream);
        if (rc0) return spl_set_error(SPL_ERR_HIP, "layout map kernel launch: %s", hipGetErrorString((hipError_t)rc0));
        lps.push_back(lp);
    }
    if (d->n_chunks) {
        splprof::Scope prof("spl_chunk_order_kernel", c->stream, 8.0 * (double)d->n_chunks);
        const int rc = spl_dev_launch_chunk_order(d->cost, d->n_chunks, chunk, d->chunk_order, c->stream);
        if (rc) return spl_set_error(SPL_ERR_HIP, "chunk order kernel launch: %s", hipGetErrorString((hipError_t)rc));
    }
    size_t gi = 0;
    if (d->fused) return SPL_OK; // (the counting pass makes its records itself)
    for (spl_dreads::Group &g : d->groups) {
        const spl_layout_params &lp = lps[gi++];
        int64_t n_reads = 0, n_ops = 0;
        for (const spl_layout_seg &ls : g.segs) n_reads += ls.n_reads;
        for (const spl_dreads::Segment &seg : d->segs) if (seg.group >= 0 && &d->groups[(size_t)seg.group] == &g) n_ops += seg.n_ops;
        const bool timed = c->k_on && (size_t)(2 * c->l_used + 1) < c->l_ev.size();
        int rc;
        { // (bytes: the arrays read once, 10 bytes a read and 4 an op)
            splprof::Scope prof("spl_layout_kernel", c->stream, 10.0 * (double)n_reads + 4.0 * (double)n_ops);
            rc = spl_dev_launch_layout(&lp, g.n_chunks, chunk, c->stream, timed ? (void *)c->l_ev[2 * c->l_used] : nullptr, timed ? (void *)c->l_ev[2 * c->l_used + 1] : nullptr);
        }
        if (timed) c->l_used++;
        if (rc) return spl_set_error(SPL_ERR_HIP, "layout kernel launch: %s", hipGetErrorString((hipError_t)rc));
    }
    return SPL_OK;
}

// SPL_FUSED=0: every read set gets its records in memory (the layout kernel), as before round 5's fused pass.
static bool fused_wanted()
{
    const char *e = getenv("SPL_FUSED");
    return !e || atoi(e) != 0;
}

// The flat chunk list of a read set, the chunk order of the range kernel and the queues.
static int finish_reads(spl_ctx *c, spl_dreads *d)
{
    if (d->finished) return SPL_OK;
    if (c->stage_timing && c->tm_pieces) {
        fprintf(stderr, "[spl_reads] %lld reads: classify %.4f s, write records %.4f s (host, %d threads); H2D %zu pieces, %.1f MB, %.3f ms in the copies = %.1f GB/s\n",
                (long long)d->n_reads, c->tm_plan_s, c->tm_emit_s, c->pack_threads, c->tm_pieces, c->tm_bytes / 1e6, c->tm_copy_ms,
                c->tm_copy_ms > 0 ? c->tm_bytes / 1e6 / c->tm_copy_ms : 0.0);
        c->tm_plan_s = c->tm_emit_s = c->tm_copy_ms = 0;
        c->tm_bytes = c->tm_pieces = 0;
    }
    if (d->n_cigar > 0xfffffff0LL) return spl_set_error(SPL_ERR_ARG, "more than 2^32 CIGAR ops in one read set: use more shards");
    auto host_now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tf0 = host_now();
    bool all_device = !d->segs.empty();
    for (const spl_dreads::Segment &seg : d->segs) all_device = all_device && seg.group >= 0;
    d->fused = fused_wanted() && all_device && d->groups.size() == 1;
    const size_t n = d->n_chunks;
    const uint32_t per = spl_order_per((uint32_t)n);
    d->n_slots = 8u * per;
    // literal queue: one region per XCD share (workgroup index & 7), each big enough for all of that share's chunks
    const size_t shard_cap = (size_t)per << d->chunk_shift;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
    const size_t o_meta = take(sizeof(spl_chunk_meta) * std::max<size_t>(n, 1)), o_cost = take(4 * std::max<size_t>(n, 1)), o_order = take(4 * std::max<size_t>(d->n_slots, 1));
    const size_t o_total = take(4);
    const size_t o_queue = take(4 * 8 * shard_cap), o_queue_alt = take(c->tail ? 4 * 8 * shard_cap : 0);
    hipError_t e = devmem::get((void **)&d->ctl, std::max<size_t>(off, 256), 'c');
    if (e != hipSuccess) return spl_set_error(SPL_ERR_HIP, "hipMalloc(%zu) for the read set: %s", off, hipGetErrorString(e));
    d->meta = (spl_chunk_meta *)(d->ctl + o_meta);
    d->cost = (uint32_t *)(d->ctl + o_cost);
    d->chunk_order = (uint32_t *)(d->ctl + o_order);
    d->queue_total = (uint32_t *)(d->ctl + o_total);
    d->queue = (uint32_t *)(d->ctl + o_queue);
    d->queue_alt = c->tail ? (uint32_t *)(d->ctl + o_queue_alt) : d->queue;
    d->queue_cap = (uint32_t)shard_cap;
    const double tf_alloc = host_now() - tf0;
    // host-packed segments: their chunks' descriptors and costs go up (stretches of the flat lists)
    hipError_t q = hipSuccess;
    std::vector<spl_chunk_meta> meta;
    std::vector<uint32_t> cost;
    bool host_segs = false;
    {
        size_t k = 0;
        for (const spl_dreads::Segment &seg : d->segs) {
            if (seg.group >= 0) { k += d->groups[(size_t)seg.group].segs[seg.group_seg].n_chunks; continue; }
            if (seg.chunks.empty()) continue;
            if (!host_segs) { meta.resize(n); cost.resize(n); host_segs = true; }
            const uint64_t rec_al = align_up((size_t)seg.rec_bytes);
            const size_t k0 = k;
            for (const splpack::ChunkDesc &cd : seg.chunks) {
                spl_chunk_meta &m = meta[k];
                m.rec = (uint64_t)(uintptr_t)seg.slab + cd.rec_off;
                m.wide = (uint64_t)(uintptr_t)seg.slab + rec_al;
                m.shift = seg.shift;
                m.first_pos = cd.first_pos;
                for (int r = 0; r < SPL_RC_RUNS; ++r) m.n[r] = cd.n[r];
                cost[k++] = cd.cost;
            }
            if (c->copy == nullptr) { int rc = ensure_stage(c); if (rc) return rc; }
            if (q == hipSuccess) q = hipMemcpyAsync(d->meta + k0, meta.data() + k0, sizeof(spl_chunk_meta) * (k - k0), hipMemcpyHostToDevice, c->copy);
            if (q == hipSuccess) q = hipMemcpyAsync(d->cost + k0, cost.data() + k0, 4 * (k - k0), hipMemcpyHostToDevice, c->copy);
        }
    }
    // (everything the host-packed segments queued on the copy stream is over with this; the vectors above die at return)
    if (q == hipSuccess && c->copy && (host_segs || !c->stage.empty())) q = hipStreamSynchronize(c->copy);
    for (spl_ctx::Stage &st : c->stage) st.busy = false;
    if (q != hipSuccess) return spl_set_error(SPL_ERR_HIP, "read set upload: %s", hipGetErrorString(q));
    // segments on the device: record slots, the segment list and the chunk -> segment map of every group
    const uint32_t chunk = 1u << d->chunk_shift;
    for (spl_dreads::Group &g : d->groups) {
        const size_t seg_bytes = align_up(sizeof(spl_layout_seg) * g.segs.size());
        e = d->fused ? hipSuccess : devmem::get((void **)&g.slab, SPL_LAYOUT_SLOT(chunk) * (size_t)g.n_chunks + 256, 'R');
        if (e == hipSuccess) e = devmem::get((void **)&g.d_segs, seg_bytes + sizeof(spl_layout_chunk) * (size_t)g.n_chunks + 16, 'd');
        if (e != hipSuccess) return spl_set_error(SPL_ERR_HIP, "hipMalloc for the device layout of %u chunks: %s", g.n_chunks, hipGetErrorString(e));
        g.d_chunks = (spl_layout_chunk *)((char *)g.d_segs + seg_bytes);
        HIP_TRY(hipMemcpyAsync(g.d_segs, g.segs.data(), sizeof(spl_layout_seg) * g.segs.size(), hipMemcpyHostToDevice, c->stream)); // (g.segs lives as long as the read set)
    }
    const int rc = launch_layout(c, d);
    if (rc) return rc;
    if (c->stage_timing)
        fprintf(stderr, "[spl_reads_finish] %zu chunks (%zu group(s) laid out on the device): control block (%.1f MB) at %.4f s, launched %.4f\n", n, d->groups.size(), off / 1e6,
                tf_alloc, host_now() - tf0);
    d->finished = true;
    return SPL_OK;
}

// A fused read set gets records in memory after all: the chunk size it would have had, record slots, the layout kernel.
static int unfuse(spl_ctx *c, spl_dreads *d)
{
    if (!d->fused) return SPL_OK;
    // (the callers hold the set through a pointer to const -- a counting pass does not change what a read set IS -- and this gives
    //  it records all the same: a set is either fused or laid out, never half of it.  Should a slot not be had, or the layout not
    //  be launched, the slots that were got go back and the set stays fused: a later default pass counts it as before.)
    const uint32_t chunk = 1u << d->chunk_shift;
    int rc = SPL_OK;
    for (spl_dreads::Group &g : d->groups) {
        const hipError_t e = devmem::get((void **)&g.slab, SPL_LAYOUT_SLOT(chunk) * (size_t)g.n_chunks + 256, 'R');
        if (e != hipSuccess) { rc = spl_set_error(SPL_ERR_HIP, "hipMalloc for the device layout of %u chunks: %s", g.n_chunks, hipGetErrorString(e)); break; }
    }
    if (rc == SPL_OK) {
        d->fused = false;
        rc = launch_layout(c, d);
    }
    if (rc != SPL_OK) {
        d->fused = true;
        for (spl_dreads::Group &g : d->groups)
            if (g.slab) { devmem::put(g.slab); g.slab = nullptr; }
    }
    return rc;
}

// The layout again, into the same record slots (bench.py's step: BAM-native arrays -> records -> counters, every step).
extern "C" int spl_reads_relayout(spl_ctx *c, spl_dreads *d)
{
    if (!c || !d) return spl_set_error(SPL_ERR_ARG, "spl_reads_relayout: null argument");
    if (!d->finished) return spl_set_error(SPL_ERR_ARG, "spl_reads_relayout: the read set is not finished (spl_reads_finish)");
    if (d->groups.empty()) return spl_set_error(SPL_ERR_ARG, "spl_reads_relayout: no segment of this read set was laid out on the device");
    HIP_TRY(hipSetDevice(c->device));
    // The literal kernel of the last pass over THIS set reads the records this rewrites: wait for that tail, not for all of them
    // (another shard's tail may run beside this layout).  Pass q's range kernel waited for tail q - 2, so a tail three passes
    // back and more is over as far as the main stream is concerned.
    if (c->tail && d->tail_pass >= 0 && (uint64_t)d->tail_pass + 3 > c->n_pass) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_tail[d->tail_pass % 4], 0));
    return launch_layout(c, d);
}

static int source_of(const spl_reads *r, splpack::Source &src, int64_t *max_end, const char *who)
{
    if (r->n_reads < 0) return spl_set_error(SPL_ERR_ARG, "%s: negative n_reads", who);
    if (r->n_reads && (!r->pos || !r->flag || !r->cig_off)) return spl_set_error(SPL_ERR_ARG, "read set has null arrays");
    if (r->n_reads && r->cig_off[0] != 0) return spl_set_error(SPL_ERR_ARG, "cig_off[0] must be 0");
    const int64_t g = r->n_reads ? r->cig_off[r->n_reads] : 0;
    if (g && !r->cigar) return spl_set_error(SPL_ERR_ARG, "cigar is null");
    if (r->n_reads) src.add(splpack::Part{r->pos, r->flag, r->cig_off, r->cigar, r->n_reads});
    *max_end = -1; // not known: the general path of the kernels checks every read it cannot vouch for (see add_segment for shifts)
    return SPL_OK;
}

extern "C" int spl_pack_host(const spl_reads *reads, int n_threads, int64_t *n_chunks_out, int64_t *rec_bytes_out, int64_t *n_wide_out,
                             void *chunk_desc, void *rec, uint32_t *wide)
{
    if (!reads || !n_chunks_out || !rec_bytes_out || !n_wide_out) return spl_set_error(SPL_ERR_ARG, "spl_pack_host: null argument");
    splpack::Source src;
    int64_t max_end;
    int rc = source_of(reads, src, &max_end, "spl_pack_host");
    if (rc) return rc;
    splpack::Plan plan;
    splpack::plan(src, plan, n_threads > 0 ? n_threads : 1);
    *n_chunks_out = (int64_t)plan.chunks.size();
    *rec_bytes_out = (int64_t)plan.rec_bytes;
    *n_wide_out = (int64_t)plan.n_wide;
    static_assert(sizeof(splpack::ChunkDesc) == 32, "chunk descriptors are handed out as 32-byte records");
    if (chunk_desc && !plan.chunks.empty()) memcpy(chunk_desc, plan.chunks.data(), sizeof(splpack::ChunkDesc) * plan.chunks.size());
    if (rec && !plan.chunks.empty()) {
        std::vector<uint32_t> no_wide((size_t)plan.n_wide + 1);
        memset(rec, 0, (size_t)plan.rec_bytes); // (the padding between runs, so that two packings compare equal)
        splpack::emit(src, plan, 0, plan.chunks.size(), (uint8_t *)rec, wide ? wide : no_wide.data());
    }
    return SPL_OK;
}

extern "C" int spl_reads_begin_sized(spl_ctx *c, int64_t expected_reads, spl_dreads **out)
{
    if (!c || !out) return spl_set_error(SPL_ERR_ARG, "spl_reads_begin: null argument");
    *out = new (std::nothrow) spl_dreads();
    if (!*out) return spl_set_error(SPL_ERR_NOMEM, "out of host memory");
    // large sets in chunks of twice the size (spl_pack.h); SPL_FORCE_CHUNK=<reads per chunk> overrides (tests)
    bool big = expected_reads >= SPL_BIG_SET_READS;
    if (const char *e = getenv("SPL_FORCE_CHUNK")) big = atol(e) == SPL_CHUNK_BIG;
    (*out)->chunk_shift = big ? SPL_CHUNK_BIG_SHIFT : SPL_CHUNK_SHIFT;
    return SPL_OK;
}

extern "C" int spl_reads_begin(spl_ctx *c, spl_dreads **out) { return spl_reads_begin_sized(c, 0, out); }

// End (1-based, inclusive) of the last base any read of a caller's arrays covers; needed when the segment is moved.
static int64_t max_end_of(const spl_reads *r)
{
    int64_t best = 0;
    for (int64_t i = 0; i < r->n_reads; ++i) {
        int64_t len = 0;
        for (uint32_t k = r->cig_off[i]; k < r->cig_off[i + 1]; ++k) {
            const uint32_t code = r->cigar[k] & 15u;
            if (code == 0 || code == 2 || code == 3 || code == 7 || code == 8) len += r->cigar[k] >> 4;
        }
        const int64_t e = (int64_t)r->pos[i] + (len > 0 ? len : 1) - 1;
        if (e > best && e <= (int64_t)SPL_COORD_MAX) best = e; // (reads out of range as they stand: the kernels report them)
    }
    return best;
}

static int upload_native(spl_ctx *c, int n_seg, const spl_reads *segs, DeviceReads **out, const int64_t *max_end = nullptr);

extern "C" int spl_reads_add(spl_ctx *c, spl_dreads *d, const spl_reads *reads, int32_t pos_shift) { return spl_reads_add2(c, d, reads, pos_shift, -1); }

// ... known_max_end: the last base (1-based) any of the reads covers, where the caller knows it (< 0: looked for here when the
// segment is moved, a pass over every CIGAR)
extern "C" int spl_reads_add2(spl_ctx *c, spl_dreads *d, const spl_reads *reads, int32_t pos_shift, int64_t known_max_end)
{
    if (!c || !d || !reads) return spl_set_error(SPL_ERR_ARG, "spl_reads_add: null argument");
    HIP_TRY(hipSetDevice(c->device));
    splpack::Source src;
    int64_t max_end;
    int rc = source_of(reads, src, &max_end, "spl_reads_add");
    if (rc) return rc;
    // SPL_RAW_UPLOAD=1 (A/B): the arrays go up as they are and the layout kernel makes the records (spl_devpack.hip), as for a BAM
    // decoded on the device.  Not the default: a hand-over is bound by what crosses PCIe and by the copies into the staging
    // buffers, and the records the host's threads pack on the way are two thirds of the arrays' bytes (20 M reads: 13.7 ms
    // packed against 18.6 ms as they are, one box, profiles/r05R_pcie_rate.txt).
    static const bool raw_upload = getenv("SPL_RAW_UPLOAD") != nullptr;
    if (raw_upload && reads->n_reads >= 4096) {
        DeviceReads *dev = nullptr;
        const int64_t no_end = 0;
        rc = upload_native(c, 1, reads, &dev, pos_shift == 0 ? &no_end : (known_max_end >= 0 ? &known_max_end : nullptr));
        if (rc) return rc;
        rc = add_segment_device(c, d, dev, 0, dev->n_rec, dev->n_ops, pos_shift, pos_shift != 0 ? dev->ref_max[0] : -1);
        free_device_reads(dev); // (the read set holds them now)
        return rc;
    }
    if (pos_shift != 0) max_end = known_max_end >= 0 ? known_max_end : max_end_of(reads);
    return add_segment(c, d, src, pos_shift, max_end);
}

extern "C" int spl_reads_add_bam(spl_ctx *c, spl_dreads *d, spl_bam *bam, int tid, int32_t pos_shift)
{
    if (!c || !d || !bam) return spl_set_error(SPL_ERR_ARG, "spl_reads_add_bam: null argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = spl_bam_wait_ref(bam, tid, nullptr, nullptr);
    if (rc) return rc;
    if (DeviceReads *dev = (DeviceReads *)spl_bam_device_reads(bam, tid)) {
        // decoded on this device, and the arrays are still there: laid out there, nothing crosses PCIe
        if (dev->device == c->device && tid >= 0 && (size_t)tid < dev->ref_n.size()) {
            const size_t t = (size_t)tid;
            return add_segment_device(c, d, dev, dev->ref_first[t], dev->ref_n[t], dev->ref_ops[t], pos_shift, dev->ref_n[t] ? dev->ref_max[t] : -1);
        }
    }
    splpack::Source src;
    int64_t max_end = 0;
    rc = spl_bam_source(bam, tid, &src, &max_end); // (the views are what the host packer reads; reads that stayed on another device come to the host here)
    if (rc) return rc;
    return add_segment(c, d, src, pos_shift, max_end);
}

// The records of reference `tid` that SHARE `share` of a decode in shares left on its device (spl_bam_decode_device_share): a
// reference cut by a share boundary is counted piece by piece, each piece where it was decoded, against the reference's whole
// site table -- the counters of checkBam only ever add one per read (SpliSER_v0_1_8.py:519-559), so the pieces' counters add up.
// The context must be on the share's device.
extern "C" int spl_reads_add_bam_share(spl_ctx *c, spl_dreads *d, spl_bam *bam, int share, int tid, int32_t pos_shift)
{
    if (!c || !d || !bam) return spl_set_error(SPL_ERR_ARG, "spl_reads_add_bam_share: null argument");
    HIP_TRY(hipSetDevice(c->device));
    DeviceReads *dev = (DeviceReads *)spl_bam_share_reads(bam, share);
    if (!dev) return spl_set_error(SPL_ERR_ARG, "spl_reads_add_bam_share: share %d has nothing on a device", share);
    if (dev->device != c->device) return spl_set_error(SPL_ERR_ARG, "spl_reads_add_bam_share: share %d was decoded on device %d, the context is on device %d", share, dev->device, c->device);
    if (tid < 0 || (size_t)tid >= dev->ref_n.size()) return spl_set_error(SPL_ERR_ARG, "tid %d out of range", tid);
    const size_t t = (size_t)tid;
    return add_segment_device(c, d, dev, dev->ref_first[t], dev->ref_n[t], dev->ref_ops[t], pos_shift, dev->ref_n[t] ? dev->ref_max[t] : -1);
}

// ---- BAM-native reads resident on the device, from a caller's arrays ------------------------------------------------
struct spl_dsoa { DeviceReads *reads = nullptr; };

namespace {
struct MaxEndJob { const spl_reads *r; int64_t per; std::vector<int64_t> *best; };
void max_end_slice(size_t k, void *arg)
{
    const MaxEndJob &j = *(const MaxEndJob *)arg;
    const spl_reads *r = j.r;
    const int64_t a = (int64_t)k * j.per, b = std::min(r->n_reads, a + j.per);
    int64_t best = 0;
    for (int64_t i = a; i < b; ++i) {
        int64_t len = 0;
        for (uint32_t q = r->cig_off[i]; q < r->cig_off[i + 1]; ++q) {
            const uint32_t code = r->cigar[q] & 15u;
            if (code == 0 || code == 2 || code == 3 || code == 7 || code == 8) len += r->cigar[q] >> 4;
        }
        const int64_t e = (int64_t)r->pos[i] + (len > 0 ? len : 1) - 1;
        if (e > best && e <= (int64_t)SPL_COORD_MAX) best = e; // (reads out of range as they stand: the kernels report them)
    }
    (*j.best)[k] = best;
}
} // namespace

// n_seg host segments as they are -> one set of device arrays (DeviceReads, one reference held by the caller).
static int upload_native(spl_ctx *c, int n_seg, const spl_reads *segs, DeviceReads **out, const int64_t *max_end)
{
    *out = nullptr;
    HIP_TRY(hipSetDevice(c->device));
    int64_t n_rec = 0, n_ops = 0;
    for (int k = 0; k < n_seg; ++k) {
        const spl_reads &r = segs[k];
        if (r.n_reads < 0) return spl_set_error(SPL_ERR_ARG, "negative n_reads");
        if (r.n_reads && (!r.pos || !r.flag || !r.cig_off)) return spl_set_error(SPL_ERR_ARG, "read set has null arrays");
        if (r.n_reads && r.cig_off[0] != 0) return spl_set_error(SPL_ERR_ARG, "cig_off[0] must be 0");
        if (r.n_reads && r.cig_off[r.n_reads] && !r.cigar) return spl_set_error(SPL_ERR_ARG, "cigar is null");
        n_rec += r.n_reads;
        n_ops += r.n_reads ? (int64_t)r.cig_off[r.n_reads] : 0;
    }
    if (n_ops > 0xfffffff0LL) return spl_set_error(SPL_ERR_ARG, "more than 2^32 CIGAR ops: use more shards");
    DeviceReads *dev = new (std::nothrow) DeviceReads();
    if (!dev) return spl_set_error(SPL_ERR_NOMEM, "out of host memory");
    dev->device = c->device;
    dev->n_rec = n_rec; dev->n_ops = n_ops;
    const size_t ns = (size_t)std::max(n_seg, 1);
    dev->ref_first.assign(ns, 0); dev->ref_n.assign(ns, 0); dev->ref_max.assign(ns, 0); dev->ref_ops.assign(ns, 0);
    hipError_t q = devmem::get(&dev->pos, 4 * (size_t)n_rec + 64, 'b');
    if (q == hipSuccess) q = devmem::get(&dev->flag, 2 * (size_t)n_rec + 64, 'b');
    if (q == hipSuccess) q = devmem::get(&dev->cig_off, 4 * ((size_t)n_rec + 1) + 64, 'b');
    if (q == hipSuccess) q = devmem::get(&dev->cigar, 4 * (size_t)n_ops + 64, 'b');
    // The arrays go up as they are, through the context's ring of page-locked staging buffers: a piece is copied into a buffer by
    // the packing threads side by side (the CIGAR offsets moved behind the ops of the segments before theirs on the way), and is
    // on the copy stream while the next piece is copied -- a caller's pageable array straight into hipMemcpy is a third of that.
    int rc = q == hipSuccess ? ensure_stage(c) : SPL_OK;
    if (rc) { free_device_reads(dev); return rc; }
    struct Piece { char *dst; const char *src; size_t bytes; uint32_t add; bool offsets; size_t per; };
    auto slice = [](size_t k, void *arg) {
        const Piece &pc = *(const Piece *)arg;
        const size_t a = k * pc.per, b = std::min(pc.bytes, a + pc.per);
        if (a >= b) return;
        if (!pc.offsets || pc.add == 0u) { memcpy(pc.dst + a, pc.src + a, b - a); return; }
        const uint32_t *in = (const uint32_t *)(pc.src + a);
        uint32_t *out = (uint32_t *)(pc.dst + a);
        for (size_t i = 0; i < (b - a) / 4; ++i) out[i] = in[i] + pc.add;
    };
    auto send = [&](void *dst_dev, const void *src, size_t bytes, uint32_t add, bool offsets) {
        for (size_t done = 0; done < bytes && q == hipSuccess;) {
            spl_ctx::Stage &st = c->stage[c->stage_next];
            c->stage_next = (c->stage_next + 1) % c->stage.size();
            if (st.busy) { q = hipEventSynchronize(st.done); st.busy = false; if (q != hipSuccess) break; }
            const size_t n = std::min(bytes - done, st.bytes / 64 * 64);
            Piece pc{st.host, (const char *)src + done, n, add, offsets, 0};
            const size_t slices = std::max<size_t>(1, std::min<size_t>((size_t)c->pack_threads * 2, n >> 16));
            pc.per = ((n + slices - 1) / slices + 63) / 64 * 64;
            splpack::parallel_for(slices, c->pack_threads, slice, &pc);
            q = hipMemcpyAsync((char *)dst_dev + done, st.host, n, hipMemcpyHostToDevice, c->copy);
            if (q == hipSuccess) { q = hipEventRecord(st.done, c->copy); st.busy = true; }
            done += n;
        }
    };
    int64_t at = 0, op_at = 0;
    for (int k = 0; k < n_seg && q == hipSuccess; ++k) {
        const spl_reads &r = segs[k];
        const int64_t g = r.n_reads ? (int64_t)r.cig_off[r.n_reads] : 0;
        dev->ref_first[(size_t)k] = at; dev->ref_n[(size_t)k] = r.n_reads; dev->ref_ops[(size_t)k] = g;
        if (r.n_reads) {
            if (max_end && max_end[k] >= 0) dev->ref_max[(size_t)k] = max_end[k]; // (the caller knows: a decoder's arrays, a kept file's)
            else {
                const size_t slices = (size_t)std::min<int64_t>(64, (r.n_reads + 65535) / 65536);
                std::vector<int64_t> best(slices, 0);
                MaxEndJob job{&r, (r.n_reads + (int64_t)slices - 1) / (int64_t)slices, &best};
                splpack::parallel_for(slices, std::max(1, c->pack_threads), max_end_slice, &job);
                dev->ref_max[(size_t)k] = *std::max_element(best.begin(), best.end());
            }
            send((int32_t *)dev->pos + at, r.pos, 4 * (size_t)r.n_reads, 0u, false);
            send((uint16_t *)dev->flag + at, r.flag, 2 * (size_t)r.n_reads, 0u, false);
            if (g) send((uint32_t *)dev->cigar + op_at, r.cigar, 4 * (size_t)g, 0u, false);
            send((uint32_t *)dev->cig_off + at, r.cig_off, 4 * (size_t)r.n_reads, (uint32_t)op_at, true);
        }
        at += r.n_reads;
        op_at += g;
    }
    const uint32_t last = (uint32_t)op_at;
    if (q == hipSuccess) q = hipMemcpyAsync((uint32_t *)dev->cig_off + at, &last, 4, hipMemcpyHostToDevice, c->copy);
    if (q == hipSuccess) q = hipStreamSynchronize(c->copy);
    for (spl_ctx::Stage &st : c->stage) st.busy = false;
    if (q != hipSuccess) {
        free_device_reads(dev);
        return spl_set_error(SPL_ERR_HIP, "reads to the device: %s", hipGetErrorString(q));
    }
    *out = dev;
    return SPL_OK;
}

extern "C" int spl_soa_upload(spl_ctx *c, int n_seg, const spl_reads *segs, spl_dsoa **out) { return spl_soa_upload2(c, n_seg, segs, nullptr, out); }

// ... max_end[k] (or null; an entry < 0 = not known): the last base any read of segment k covers, 1-based, where the caller knows
// it (a decoder does, a file of kept reads does): saves a pass over every CIGAR on the host
extern "C" int spl_soa_upload2(spl_ctx *c, int n_seg, const spl_reads *segs, const int64_t *max_end, spl_dsoa **out)
{
    if (!c || !out || n_seg < 0 || (n_seg && !segs)) return spl_set_error(SPL_ERR_ARG, "spl_soa_upload: null argument");
    *out = nullptr;
    DeviceReads *dev = nullptr;
    const int rc = upload_native(c, n_seg, segs, &dev, max_end);
    if (rc) return rc;
    spl_dsoa *h = new (std::nothrow) spl_dsoa();
    if (!h) { free_device_reads(dev); return spl_set_error(SPL_ERR_NOMEM, "out of host memory"); }
    h->reads = dev;
    *out = h;
    return SPL_OK;
}

extern "C" void spl_soa_free(spl_ctx *c, spl_dsoa *soa)
{
    if (!soa) return;
    if (c) (void)hipSetDevice(c->device);
    free_device_reads(soa->reads); // (read sets laid out from the arrays keep them alive)
    delete soa;
}

extern "C" int spl_reads_add_soa(spl_ctx *c, spl_dreads *d, spl_dsoa *soa, int seg, int32_t pos_shift)
{
    if (!c || !d || !soa) return spl_set_error(SPL_ERR_ARG, "spl_reads_add_soa: null argument");
    DeviceReads *dev = soa->reads;
    if (seg < 0 || (size_t)seg >= dev->ref_n.size()) return spl_set_error(SPL_ERR_ARG, "spl_reads_add_soa: no segment %d", seg);
    if (dev->device != c->device) return spl_set_error(SPL_ERR_ARG, "spl_reads_add_soa: the arrays are on device %d, the context on %d", dev->device, c->device);
    HIP_TRY(hipSetDevice(c->device));
    const size_t t = (size_t)seg;
    return add_segment_device(c, d, dev, dev->ref_first[t], dev->ref_n[t], dev->ref_ops[t], pos_shift, dev->ref_n[t] && pos_shift != 0 ? dev->ref_max[t] : -1);
}

// What the layout kernel moves for this read set: the BAM-native arrays it reads (10 bytes a read and 4 an op: every input
// once) and the records it writes (the chunks' record areas as laid out, padding between runs included).  Waits for the layout.
extern "C" int spl_reads_layout_bytes(spl_ctx *c, const spl_dreads *d, int64_t *soa_bytes_out, int64_t *record_bytes_out)
{
    if (!c || !d || !soa_bytes_out || !record_bytes_out) return spl_set_error(SPL_ERR_ARG, "spl_reads_layout_bytes: null argument");
    if (!d->finished) return spl_set_error(SPL_ERR_ARG, "spl_reads_layout_bytes: the read set is not finished (spl_reads_finish)");
    HIP_TRY(hipSetDevice(c->device));
    int64_t soa = 0, rec = 0;
    std::vector<spl_chunk_meta> meta(d->n_chunks);
    if (d->fused) { // no records are written: what the counting pass reads is the arrays
        for (const spl_dreads::Segment &seg : d->segs) soa += 10 * seg.n_reads + 4 * seg.n_ops;
        *soa_bytes_out = soa;
        *record_bytes_out = 0;
        return SPL_OK;
    }
    if (d->n_chunks) {
        HIP_TRY(hipMemcpyAsync(meta.data(), d->meta, sizeof(spl_chunk_meta) * d->n_chunks, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    size_t k = 0;
    for (const spl_dreads::Segment &seg : d->segs) {
        const size_t nc = seg.group >= 0 ? d->groups[(size_t)seg.group].segs[seg.group_seg].n_chunks : seg.chunks.size();
        if (seg.group >= 0) {
            soa += 10 * seg.n_reads + 4 * seg.n_ops;
            for (size_t j = 0; j < nc; ++j) rec += (int64_t)spl_run_offset(meta[k + j].n, 4);
        }
        k += nc;
    }
    *soa_bytes_out = soa;
    *record_bytes_out = rec;
    return SPL_OK;
}

extern "C" int spl_reads_finish(spl_ctx *c, spl_dreads *d)
{
    if (!c || !d) return spl_set_error(SPL_ERR_ARG, "spl_reads_finish: null argument");
    HIP_TRY(hipSetDevice(c->device));
    return finish_reads(c, d);
}

extern "C" void spl_reads_free(spl_ctx *c, spl_dreads *d)
{
    if (!d) return;
    if (c) (void)hipSetDevice(c->device);
    if (c && c->copy) (void)hipStreamSynchronize(c->copy);
    if (c && c->tail) (void)hipStreamSynchronize(c->tail); // (a tail may still be reading these buffers; hipFree itself waits
    if (c && c->stream) (void)hipStreamSynchronize(c->stream); //  for the device, this makes it independent of that)
    for (spl_dreads::Segment &seg : d->segs) devmem::put(seg.slab);
    for (spl_dreads::Group &g : d->groups) { devmem::put(g.slab); devmem::put(g.d_segs); free_device_reads(g.src); }
    devmem::put(d->ctl);
    delete d;
}

extern "C" int spl_reads_upload_segments(spl_ctx *c, int n_seg, const spl_reads *segs, const int32_t *pos_shift, spl_dreads **out)
{
    if (!c || !out || n_seg < 0 || (n_seg && (!segs || !pos_shift)))
        return spl_set_error(SPL_ERR_ARG, "spl_reads_upload_segments: null argument");
    *out = nullptr;
    spl_dreads *d = nullptr;
    int64_t total = 0;
    for (int k = 0; k < n_seg; ++k) total += segs[k].n_reads > 0 ? segs[k].n_reads : 0;
    int rc = spl_reads_begin_sized(c, total, &d);
    for (int k = 0; k < n_seg && rc == SPL_OK; ++k) rc = spl_reads_add(c, d, &segs[k], pos_shift[k]);
    if (rc == SPL_OK) rc = spl_reads_finish(c, d);
    if (rc != SPL_OK) { spl_reads_free(c, d); return rc; }
    *out = d;
    return SPL_OK;
}

extern "C" int spl_reads_upload(spl_ctx *c, const spl_reads *r, spl_dreads **out)
{
    if (!c || !r || !out) return spl_set_error(SPL_ERR_ARG, "spl_reads_upload: null argument");
    const int32_t zero = 0;
    return spl_reads_upload_segments(c, 1, r, &zero, out);
}

// ---- launches -------------------------------------------------------------------------------------------
extern "C" int spl_count_launch(spl_ctx *c, spl_dsites *ds, const spl_dreads *dr, const spl_opts *o)
{
    if (!c || !ds || !dr || !o) return spl_set_error(SPL_ERR_ARG, "spl_count_launch: null argument");
    if (o->stranded < 0 || o->stranded > 2)
        return spl_set_error(SPL_ERR_ARG, "stranded must be 0 (unstranded), 1 (fr) or 2 (rf); the reference raises "
                                          "UnboundLocalError for any other strandedType");
    HIP_TRY(hipSetDevice(c->device));
    // counters, difference arrays, error word and queue counters start from zero: the copy the previous pass cleared on the
    // side, or -- first pass after a pair-kernel pass -- one clearing launch
    // range kernel whenever the table allows it; the literal pair kernel otherwise or on request
    const int variant = (o->flags & SPL_OPT_PAIR_KERNEL) ? 1 : ((o->flags & SPL_OPT_WAVE_AGGREGATION) ? 2 : 0);
    const bool piped = c->tail != nullptr && variant != 1 && ds->region_clean[(ds->cur + 1) % 3];
    if (piped) {
        // the tail of the pass before the last one is the last thing that read the queue buffer and wrote the counter copy
        // this pass takes (tails run in order on their stream): a wait that is over long before it is asked for
        // The wait packet in the main queue costs 6 us between two range kernels.  SPL_TAIL_HOST_WAIT=1: the HOST waits
        // instead (the call then blocks while more than two passes are in flight) -- 4 % faster in bench.py, and one
        // millisecond of host jitter is one millisecond of idle GPU (1 run in 12 lost 25 % that way): not the default.
        if (c->n_pass >= 2) {
            if (c->tail_host_wait) HIP_TRY(hipEventSynchronize(c->ev_tail[(c->n_pass - 2) % 4]));
            else HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_tail[(c->n_pass - 2) % 4], 0));
        }
    } else
        HIP_TRY(join_tail(c)); // (pair kernel, first pass after one, or one stream: everything in order on the main stream)
    const int next = (ds->cur + 1) % 3;
    if (ds->region_clean[next]) {
        ds->point_at(next);
        ds->region_clean[next] = false;
    } else if (int rc0 = spl_dev_launch_clear(ds->beta1, ds->counter_bytes, c->stream)) // (in place: the copy in use)
        return spl_set_error(SPL_ERR_HIP, "clear kernel launch: %s", hipGetErrorString((hipError_t)rc0));
    if (!dr->finished) return spl_set_error(SPL_ERR_ARG, "spl_count_launch: the read set is not finished (spl_reads_finish)");
    if (dr->fused && variant != 0) { const int rc0 = unfuse(c, const_cast<spl_dreads *>(dr)); if (rc0) return rc0; } // (the pair kernel and the merging variant read records)
    uint32_t *const queue = dr->queue_turn ? dr->queue_alt : dr->queue;
    dr->queue_turn ^= 1;
    dr->queued_pass = false;
    ds->sse_fused = false;
    spl_count_params p;
    memset(&p, 0, sizeof(p));
    p.n_reads = dr->n_reads;
    p.n_chunks = dr->n_chunks;
    p.chunk_meta = dr->meta;
    p.n_sites = (int32_t)ds->n_sites;
    p.site_pos = ds->pos; p.site_strand = ds->strand; p.site_flags = ds->flags; p.site_meta = ds->meta;
    p.part_pos = ds->part_pos; p.part_site = ds->part_site; p.comp_pos = ds->comp_pos;
    p.diff = ds->diff; p.diff_stride = ds->diff_stride;
    p.dbucket = ds->dbucket; p.n_dbuckets = ds->n_dbuckets; p.dbase = ds->dbase; p.n_dpos = ds->n_dpos;
    p.dpos_first_row = ds->dpos_first_row;
    p.jhash = ds->jhash; p.jhash_mask = ds->jhash_mask; p.jrivals = ds->jrivals;
    spl_hot_params h;
    memset(&h, 0, sizeof(h));
    h.n_chunks = dr->n_slots; h.chunk_shift = dr->chunk_shift; h.chunk_meta = dr->meta; h.chunk_order = dr->chunk_order; h.part_pos = ds->part_pos;
    h.dbucket = p.dbucket; h.n_dbuckets = p.n_dbuckets; h.dbase = p.dbase; h.n_dpos = p.n_dpos;
    h.stranded = o->stranded; h.diff = p.diff; h.diff_stride = p.diff_stride;
    h.queue = queue; h.queue_n = ds->queue_n; h.err = ds->err;
    h.queue_cap = dr->queue_cap;
    if (dr->fused) {
        const spl_dreads::Group &g = dr->groups[0];
        h.cells = g.d_chunks;
        h.src = spl_devreads{(const int32_t *)g.src->pos, (const uint16_t *)g.src->flag, (const uint32_t *)g.src->cig_off, (const uint32_t *)g.src->cigar};
        h.src_n_rec = g.src->n_rec; h.src_n_ops = g.src->n_ops;
    }
    h.jhash = ds->jhash; h.jhash_mask = ds->jhash_mask; h.jrivals = ds->jrivals; h.dbl = ds->dbl; h.combine_mode = o->combine_mode ? 1 : 0;

    p.bucket = ds->bucket; p.n_buckets = ds->n_buckets; p.bucket_base = ds->bucket_base; p.bucket_shift = ds->bucket_shift;
    p.stranded = o->stranded; p.combine_mode = o->combine_mode ? 1 : 0;
    p.beta1 = ds->beta1; p.beta2s_reads = ds->beta2s; p.dbl = ds->dbl; p.err = ds->err;
    int grid = 0, lds = 0;
    const bool timed = c->k_on && (size_t)(2 * c->k_used + 1) < c->k_ev.size();
    // (with the tail on its own stream that stream waits for the range kernel's OWN stop event: a marker packet behind the
    //  kernel -- hipEventRecord -- holds the main queue up for ~18 us when another queue depends on it)
    hipEvent_t ev_stop = timed ? c->k_ev[2 * c->k_used + 1] : (piped ? c->ev_range[c->n_pass % 4] : nullptr);
    int64_t alg = 0;
    if (splprof::g_on.load(std::memory_order_relaxed)) (void)spl_count_algorithmic_bytes(ds, dr, &alg);
    int rc;
    { splprof::Scope prof("spl_count_ranges_kernel", c->stream, (double)alg); rc = spl_dev_launch_count(&p, &h, variant, c->stream, &grid, &lds, timed ? (void *)c->k_ev[2 * c->k_used] : nullptr, (void *)ev_stop); }
    if (timed) c->k_used++;
    c->last_grid = grid;
    c->last_lds = lds;
    c->last_variant = variant;
    if (rc != 0) return spl_set_error(SPL_ERR_HIP, "count kernel launch: %s", hipGetErrorString((hipError_t)rc));
    if (variant != 1 && grid > 0) { // queued reads through the literal kernel, then difference arrays -> counters
        hipStream_t ts = c->stream;
        if (piped) {
            HIP_TRY(hipStreamWaitEvent(c->tail, ev_stop, 0));
            ts = c->tail;
        }
        // the copy to clear: with the tail on its own stream the one after next (the next pass may be running by then),
        // otherwise the next one
        const int to_clear = piped ? (ds->cur + 2) % 3 : (ds->cur + 1) % 3;
        spl_queue_params lq;
        lq.queue = queue; lq.queue_n = ds->queue_n; lq.queue_cap = h.queue_cap; lq.chunk_shift = dr->chunk_shift; lq.queue_total = dr->queue_total;
        lq.clear_region = (uint4 *)ds->region[to_clear]; lq.clear_n16 = ds->counter_bytes / 16;
        lq.cells = h.cells; lq.src = h.src;
        lq.diff = ds->diff; lq.block_sums = ds->block_sums; lq.diff_stride = ds->diff_stride; lq.n_dpos = ds->n_dpos;
        lq.scan_blocks = ds->scan_blocks; lq.scan_arrays = o->stranded ? 4 : 2;
        rc = spl_dev_launch_literal(&p, &lq, ts);
        if (rc != 0) return spl_set_error(SPL_ERR_HIP, "literal kernel launch: %s", hipGetErrorString((hipError_t)rc));
        ds->region_clean[to_clear] = true;
        dr->queued_pass = true;
        spl_scan_params q;
        memset(&q, 0, sizeof(q));
        q.n_dpos = ds->n_dpos; q.dpos_first_row = ds->dpos_first_row;
        q.n_arrays = o->stranded ? 4 : 2; q.diff_stride = ds->diff_stride;
        q.n_blocks = ds->scan_blocks; q.diff = ds->diff; q.block_sums = ds->block_sums; q.site_flags = ds->flags;
        q.beta1 = ds->beta1; q.beta2s_reads = ds->beta2s;
        q.with_sse = ds->has_sse_inputs ? 1 : 0;
        if (q.with_sse) {
            q.sse.n_sites = ds->n_sites; q.sse.site_pos = ds->pos; q.sse.part_off = ds->part_off; q.sse.part_pos = ds->part_pos;
            q.sse.part_site = ds->part_site; q.sse.alpha = ds->alpha; q.sse.edge_cnt = ds->edge_cnt; q.sse.dbl = ds->dbl;
            q.sse.beta2_simple = ds->b2_simple; q.sse.beta2_cryptic = ds->b2_cryptic; q.sse.beta2_weighted = ds->b2_weighted;
            q.sse.sse = ds->sse; q.sse_with_cryptic = ds->sse_cryptic;
        }
        rc = spl_dev_launch_scan(&q, ts);
        if (rc != 0) return spl_set_error(SPL_ERR_HIP, "scan kernel launch: %s", hipGetErrorString((hipError_t)rc));
        ds->sse_fused = q.with_sse != 0;
        if (piped) {
            HIP_TRY(hipEventRecord(c->ev_tail[c->n_pass % 4], c->tail));
            dr->tail_pass = (int64_t)c->n_pass;
            c->n_pass++;
            c->tail_pending = true;
        }
    }
    return SPL_OK;
}

extern "C" int spl_sse_launch(spl_ctx *c, spl_dsites *ds, int cryptic)
{
    if (!c || !ds) return spl_set_error(SPL_ERR_ARG, "spl_sse_launch: null argument");
    if (!ds->has_sse_inputs) return spl_set_error(SPL_ERR_ARG, "spl_sse needs sites->alpha, sites->edge_cnt and sites->part_site");
    HIP_TRY(hipSetDevice(c->device));
    if (ds->sse_fused) { // the counting pass on these very counters computed both settings: pick one, nothing to launch
        ds->sse_view = cryptic ? ds->sse_cryptic : ds->sse;
        return SPL_OK;
    }
    ds->sse_view = ds->sse;
    HIP_TRY(join_tail(c));
    spl_sse_params p;
    memset(&p, 0, sizeof(p));
    p.n_sites = ds->n_sites; p.site_pos = ds->pos; p.part_off = ds->part_off; p.part_pos = ds->part_pos; p.part_site = ds->part_site;
    p.alpha = ds->alpha; p.edge_cnt = ds->edge_cnt; p.beta1 = ds->beta1; p.beta2s_reads = ds->beta2s; p.dbl = ds->dbl;
    p.cryptic = cryptic ? 1 : 0;
    p.beta2_simple = ds->b2_simple; p.beta2_cryptic = ds->b2_cryptic; p.beta2_weighted = ds->b2_weighted; p.sse = ds->sse;
    int rc = spl_dev_launch_sse(&p, c->stream);
    if (rc != 0) return spl_set_error(SPL_ERR_HIP, "spl_sse_kernel launch: %s", hipGetErrorString((hipError_t)rc));
    return SPL_OK;
}

static int check_device_error(spl_ctx *c, const spl_dsites *ds)
{
    int32_t err = 0;
    HIP_TRY(join_tail(c));
    HIP_TRY(hipMemcpyAsync(&err, ds->err, sizeof(err), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (err & SPL_DEV_ERR_TABLE) return spl_set_error(SPL_ERR_HIP, "a wave's list of reads overflowed in the range kernel (internal error: results discarded)");
    if (err & SPL_DEV_ERR_RANGE)
        return spl_set_error(SPL_ERR_RANGE, "a read starts below 0 or ends beyond coordinate %d: split the shard (spliser_amd/shard.py)", SPL_COORD_MAX);
    return SPL_OK;
}

extern "C" int spl_counters_download(spl_ctx *c, const spl_dsites *ds, uint32_t *beta1, uint32_t *b2s, uint32_t *dbl)
{
    if (!c || !ds) return spl_set_error(SPL_ERR_ARG, "spl_counters_download: null argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(join_tail(c));
    if (beta1 && ds->n_sites) HIP_TRY(hipMemcpyAsync(beta1, ds->beta1, 4 * ds->n_sites, hipMemcpyDeviceToHost, c->stream));
    if (b2s && ds->n_sites) HIP_TRY(hipMemcpyAsync(b2s, ds->beta2s, 4 * ds->n_sites, hipMemcpyDeviceToHost, c->stream));
    if (dbl && ds->n_part) HIP_TRY(hipMemcpyAsync(dbl, ds->dbl, 4 * ds->n_part, hipMemcpyDeviceToHost, c->stream));
    return check_device_error(c, ds);
}

extern "C" int spl_sse_download(spl_ctx *c, const spl_dsites *ds, int64_t *b2s, int64_t *b2c, double *b2w, double *sse)
{
    if (!c || !ds) return spl_set_error(SPL_ERR_ARG, "spl_sse_download: null argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = 8 * (size_t)ds->n_sites;
    HIP_TRY(join_tail(c));
    if (n) {
        if (b2s) HIP_TRY(hipMemcpyAsync(b2s, ds->b2_simple, n, hipMemcpyDeviceToHost, c->stream));
        if (b2c) HIP_TRY(hipMemcpyAsync(b2c, ds->b2_cryptic, n, hipMemcpyDeviceToHost, c->stream));
        if (b2w) HIP_TRY(hipMemcpyAsync(b2w, ds->b2_weighted, n, hipMemcpyDeviceToHost, c->stream));
        if (sse) HIP_TRY(hipMemcpyAsync(sse, ds->sse_view, n, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SPL_OK;
}

extern "C" int spl_count_algorithmic_bytes(const spl_dsites *ds, const spl_dreads *dr, int64_t *out)
{
    if (!ds || !dr || !out) return spl_set_error(SPL_ERR_ARG, "spl_count_algorithmic_bytes: null argument");
    // every input once, every output once.
    const int64_t R = dr->n_reads, G = dr->n_cigar, S = ds->n_sites, P = ds->n_part, C = ds->n_comp;
    *out = R * (4 + 2 + 4) + 4 * G + S * (4 + 1 + 8) + 4 * (P + C) + S * 8 + 4 * P;
    return SPL_OK;
}

// ---- junction table of a read set ---------------------------------------------------------
extern "C" int spl_junctions(spl_ctx *c, const spl_dreads *dr, int stranded, int32_t min_anchor, int32_t min_intron, int32_t max_intron,
                             int64_t *n_out)
{
    if (!c || !dr || !n_out) return spl_set_error(SPL_ERR_ARG, "spl_junctions: null argument");
    if (min_anchor < 0 || min_intron < 0 || max_intron < 0) return spl_set_error(SPL_ERR_ARG, "spl_junctions: negative filter value");
    if (stranded < 0 || stranded > 2) return spl_set_error(SPL_ERR_ARG, "stranded must be 0, 1 (fr) or 2 (rf)");
    HIP_TRY(hipSetDevice(c->device));
    c->junctions.clear();
    *n_out = 0;
    // an N op needs an aligned op on both sides: fewer than half of all ops are junctions, so a table with one slot per op
    // (rounded up to a power of two) stays under half full
    uint64_t slots = 1024;
    while (slots < (uint64_t)dr->n_cigar) slots <<= 1;
    if (slots > 0x80000000ull) return spl_set_error(SPL_ERR_ARG, "read set too large for one junction table: use more shards");
    char *buf = nullptr;
    const size_t bytes = (size_t)slots * (8 + 12) * 2 + 256;
    hipError_t e = hipMalloc((void **)&buf, bytes);
    if (e != hipSuccess) return spl_set_error(SPL_ERR_HIP, "hipMalloc(%zu) for the junction table: %s", bytes, hipGetErrorString(e));
    unsigned long long *keys = (unsigned long long *)buf, *out_keys = keys + slots;
    uint32_t *vals = (uint32_t *)(out_keys + slots), *out_vals = vals + 3 * slots, *n_dev = out_vals + 3 * slots;
    if (!dr->finished) { (void)hipFree(buf); return spl_set_error(SPL_ERR_ARG, "spl_junctions: the read set is not finished (spl_reads_finish)"); }
    if (dr->fused) { const int rc0 = unfuse(c, const_cast<spl_dreads *>(dr)); if (rc0) { (void)hipFree(buf); return rc0; } } // (the junction kernel reads records)
    int rc = spl_dev_launch_junctions(dr->meta, dr->n_chunks, stranded, (uint32_t)min_anchor, (uint32_t)min_intron,
                                      (uint32_t)max_intron, keys, vals, (uint32_t)slots, out_keys,
                                      out_vals, n_dev, c->d_err, c->stream);
    uint32_t n = 0;
    int32_t err = 0;
    hipError_t q = rc ? (hipError_t)rc : hipMemcpyAsync(&n, n_dev, 4, hipMemcpyDeviceToHost, c->stream);
    if (q == hipSuccess) q = hipMemcpyAsync(&err, c->d_err, 4, hipMemcpyDeviceToHost, c->stream);
    if (q == hipSuccess) q = hipStreamSynchronize(c->stream);
    std::vector<unsigned long long> hk;
    std::vector<uint32_t> hv;
    if (q == hipSuccess && n) {
        hk.resize(n);
        hv.resize(3 * (size_t)n);
        q = hipMemcpy(hk.data(), out_keys, 8 * (size_t)n, hipMemcpyDeviceToHost);
        if (q == hipSuccess) q = hipMemcpy(hv.data(), out_vals, 12 * (size_t)n, hipMemcpyDeviceToHost);
    }
    (void)hipFree(buf);
    if (q != hipSuccess) return spl_set_error(SPL_ERR_HIP, "junction table: %s", hipGetErrorString(q));
    if (err & SPL_DEV_ERR_RANGE) return spl_set_error(SPL_ERR_RANGE, "a read ends beyond coordinate %d: split the shard", SPL_COORD_MAX);
    if (err & SPL_DEV_ERR_TABLE) return spl_set_error(SPL_ERR_HIP, "junction table overflow (internal error)");
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hk[a] < hk[b]; }); // left, right, strand
    c->junctions.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        const unsigned long long k = hk[order[i]];
        spl_ctx::Junction &j = c->junctions[i];
        j.left = (int32_t)(k >> 32);
        j.right = (int32_t)((k & 0xffffffffull) >> 1);
        j.strand = stranded ? ((k & 1ull) ? (uint8_t)'-' : (uint8_t)'+') : (uint8_t)'?';
        j.count = hv[3 * (size_t)order[i]];
        j.anchor_left = hv[3 * (size_t)order[i] + 1];
        j.anchor_right = hv[3 * (size_t)order[i] + 2];
    }
    *n_out = n;
    return SPL_OK;
}

extern "C" int spl_junctions_get(const spl_ctx *c, int32_t *left, int32_t *right, uint8_t *strand, uint32_t *count, uint32_t *anchor_left,
                                 uint32_t *anchor_right)
{
    if (!c) return spl_set_error(SPL_ERR_ARG, "spl_junctions_get: null context");
    for (size_t i = 0; i < c->junctions.size(); ++i) {
        const spl_ctx::Junction &j = c->junctions[i];
        if (left) left[i] = j.left;
        if (right) right[i] = j.right;
        if (strand) strand[i] = j.strand;
        if (count) count[i] = j.count;
        if (anchor_left) anchor_left[i] = j.anchor_left;
        if (anchor_right) anchor_right[i] = j.anchor_right;
    }
    return SPL_OK;
}

extern "C" int spl_literal_queue_size(spl_ctx *c, const spl_dreads *dr, int64_t *n_out)
{
    if (!c || !dr || !n_out) return spl_set_error(SPL_ERR_ARG, "spl_literal_queue_size: null argument");
    HIP_TRY(hipSetDevice(c->device));
    *n_out = 0;
    if (!dr->queued_pass || !dr->queue_total) return SPL_OK;
    uint32_t total = 0; // (a word of the read set's own, written by the literal kernel of its last pass)
    HIP_TRY(join_tail(c));
    HIP_TRY(hipMemcpyAsync(&total, dr->queue_total, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *n_out = total;
    return SPL_OK;
}

extern "C" int spl_last_launch_info(const spl_ctx *c, int32_t *grid, int32_t *block, int32_t *lds)
{
    if (!c) return spl_set_error(SPL_ERR_ARG, "spl_last_launch_info: null context");
    if (grid) *grid = c->last_grid;
    if (block) *block = SPL_BLOCK;
    if (lds) *lds = c->last_lds;
    return SPL_OK;
}

// ---- one-shot entry points ------------------------------------------------------------------------------
extern "C" int spl_count(spl_ctx *c, const spl_sites *s, const spl_reads *r, const spl_opts *o, uint32_t *beta1, uint32_t *b2s, uint32_t *dbl)
{
    if (!c || !s || !r || !o) return spl_set_error(SPL_ERR_ARG, "spl_count: null argument");
    spl_dsites *ds = nullptr;
    spl_dreads *dr = nullptr;
    int rc = spl_sites_upload(c, s, &ds);
    if (rc == SPL_OK) rc = spl_reads_upload(c, r, &dr);
    if (rc == SPL_OK) rc = spl_count_launch(c, ds, dr, o);
    if (rc == SPL_OK) rc = spl_counters_download(c, ds, beta1, b2s, dbl);
    spl_reads_free(c, dr);
    spl_sites_free(c, ds);
    return rc;
}

extern "C" int spl_sse(spl_ctx *c, const spl_sites *s, const uint32_t *beta1, const uint32_t *b2s_reads, const uint32_t *dbl, int cryptic,
                       int64_t *b2s, int64_t *b2c, double *b2w, double *sse)
{
    if (!c || !s) return spl_set_error(SPL_ERR_ARG, "spl_sse: null argument");
    if (s->n_sites && (!beta1 || !b2s_reads)) return spl_set_error(SPL_ERR_ARG, "spl_sse: null counters");
    spl_dsites *ds = nullptr;
    int rc = spl_sites_upload(c, s, &ds);
    if (rc != SPL_OK) return rc;
    hipError_t e = hipSuccess;
    if (ds->n_sites) {
        e = hipMemcpyAsync(ds->beta1, beta1, 4 * ds->n_sites, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(ds->beta2s, b2s_reads, 4 * ds->n_sites, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && ds->n_part && dbl) e = hipMemcpyAsync(ds->dbl, dbl, 4 * ds->n_part, hipMemcpyHostToDevice, c->stream);
    }
    if (e != hipSuccess) rc = spl_set_error(SPL_ERR_HIP, "spl_sse upload: %s", hipGetErrorString(e));
    if (rc == SPL_OK) rc = spl_sse_launch(c, ds, cryptic);
    if (rc == SPL_OK) rc = spl_sse_download(c, ds, b2s, b2c, b2w, sse);
    spl_sites_free(c, ds);
    return rc;
}
