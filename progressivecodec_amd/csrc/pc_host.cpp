// pc_host.cpp -- host side of libpcodec: the rANS entropy coder, pmf->CDF quantiser, a small
// thread pool for per-image streams, error strings.
//
// The coder reproduces the reference's byte streams exactly (cpp_exts/rans/rans_interface.cpp,
// third_party/ryg_rans/rans64.h) but is organised differently: symbols are walked backwards and
// encoded straight into the output words (no intermediate record vector), and the decoder finds
// the symbol by binary search over the CDF row instead of a linear scan.
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/pc_math.h"
#include "../../include/pcodec.h"
#include "pc_host.h"

namespace {
constexpr uint64_t kRansL = 1ull << 31;   // rans64.h:59
constexpr int kPrecision = 16;            // rans_interface.cpp:40
constexpr int kBypassBits = 4;            // rans_interface.cpp:42
constexpr int kBypassMax = 15;            // rans_interface.cpp:43

struct Writer {
    uint32_t* begin;
    uint32_t* ptr;   // grows downwards
    bool overflow = false;
    inline void put(uint32_t w) { if (ptr == begin) { overflow = true; return; } *--ptr = w; }
};

inline void enc_put(uint64_t& x, Writer& wr, uint32_t start, uint32_t freq)          // Rans64EncPut, rans64.h:77-93
{
    const uint64_t x_max = ((kRansL >> kPrecision) << 32) * freq;
    if (x >= x_max) { wr.put((uint32_t)x); x >>= 32; }
    x = ((x / freq) << kPrecision) + (x % freq) + start;
}
inline void enc_put_bits(uint64_t& x, Writer& wr, uint32_t val)                       // Rans64EncPutBits, rans_interface.cpp:60-78
{
    const uint32_t freq = 1u << (16 - kBypassBits);
    const uint64_t x_max = ((kRansL >> 16) << 32) * freq;
    if (x >= x_max) { wr.put((uint32_t)x); x >>= 32; }
    x = (x << kBypassBits) | val;
}
}  // namespace

extern "C" size_t pc_rans_bound(size_t n)
{
    // worst case per symbol: 1 table symbol (<= 16 bits... one 32-bit word at most every other put) plus
    // bypass: count nibbles (<= 1 + 8/15 rounds) and 8 value nibbles -> well under 3 words; + 2 state words.
    return 4 * (3 * n + 4);
}

extern "C" int pc_rans_encode_with_indexes(const int32_t* symbols, const int32_t* indexes, size_t n,
                                           const int32_t* cdfs, int n_cdf, int cdf_stride,
                                           const int32_t* cdf_sizes, const int32_t* offsets,
                                           uint8_t* out, size_t out_cap, size_t* out_len)
{
    if ((!symbols || !indexes) && n) return PC_ERR_ARG;
    if (!cdfs || !cdf_sizes || !offsets || !out || !out_len || n_cdf <= 0 || cdf_stride <= 0) return PC_ERR_ARG;
    const size_t cap_words = out_cap / 4;
    if (cap_words < 2) return PC_ERR_BUFFER;
    // Encode into the tail of `out` (aligned down to 4 bytes), then move to the front.
    uint32_t* base = reinterpret_cast<uint32_t*>(out);
    if (reinterpret_cast<uintptr_t>(out) & 3) return PC_ERR_ARG;
    Writer wr{base, base + cap_words};
    uint64_t x = kRansL;                                                              // Rans64EncInit
    for (size_t ii = n; ii-- > 0;) {
        const int32_t ci = indexes[ii];
        if (ci < 0 || ci >= n_cdf) return PC_ERR_INDEX;
        const int32_t len = cdf_sizes[ci];
        if (len < 2 || len > cdf_stride) return PC_ERR_CDF;
        const int32_t* cdf = cdfs + (size_t)ci * cdf_stride;
        const int32_t max_value = len - 2;                                            // :115
        // (all of this in 64 bits: the reference's int32 arithmetic overflows -- undefined behaviour -- for symbols near INT32_MIN/MAX,
        // rans_interface.cpp:119-133; a symbol whose escape code does not fit the 8 nibbles the decoder reads back is refused instead)
        const int64_t v64 = (int64_t)symbols[ii] - (int64_t)offsets[ci];              // :119
        int32_t value = 0;
        uint32_t raw = 0;
        bool escaped = false;
        if (v64 < 0) { const int64_t r = -2 * v64 - 1; if (r > 0xffffffffll) return PC_ERR_ARG; raw = (uint32_t)r; value = max_value; escaped = true; }
        else if (v64 >= max_value) { const int64_t r = 2 * (v64 - max_value); if (r > 0xffffffffll) return PC_ERR_ARG; raw = (uint32_t)r; value = max_value; escaped = true; }
        else value = (int32_t)v64;
        if (escaped) {
            // forward order is: symbol, count nibbles (15,15,...,rest), value nibbles (LSB first)   :138-162
            // -> encode in exact reverse
            int32_t n_bypass = 0;
            while (n_bypass < 8 && (raw >> (n_bypass * kBypassBits)) != 0) ++n_bypass;
            for (int32_t j = n_bypass - 1; j >= 0; --j) enc_put_bits(x, wr, (raw >> (j * kBypassBits)) & kBypassMax);
            const int32_t full = n_bypass / kBypassMax, rest = n_bypass % kBypassMax;
            enc_put_bits(x, wr, (uint32_t)rest);
            for (int32_t j = 0; j < full; ++j) enc_put_bits(x, wr, kBypassMax);
        }
        const uint32_t start = (uint32_t)cdf[value], freq = (uint32_t)cdf[value + 1] - (uint32_t)cdf[value];
        if (freq == 0 || start > 65535u || freq > 65536u - start) return PC_ERR_CDF;     // (a row that is not an increasing CDF up to 2^16)
        enc_put(x, wr, start, freq);
    }
    wr.put((uint32_t)(x >> 32));                                                      // Rans64EncFlush, rans64.h:96-103
    wr.put((uint32_t)x);
    if (wr.overflow) return PC_ERR_BUFFER;
    const size_t nwords = (size_t)((base + cap_words) - wr.ptr);
    std::memmove(out, wr.ptr, nwords * 4);
    *out_len = nwords * 4;
    return PC_OK;
}

namespace {
// decoder core: state (x, p = next word to read) lives with the caller so that a stream can be decoded in several calls
int rans_decode_core(const uint8_t* encoded, size_t encoded_len, uint64_t& x, size_t& p, const int32_t* indexes, size_t n,
                     const int32_t* cdfs, int n_cdf, int cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets, int32_t* out)
{
    const size_t nw = encoded_len / 4;
    auto word = [&](size_t i) { uint32_t w; std::memcpy(&w, encoded + 4 * i, 4); return w; };
    bool trunc = false;
    auto renorm = [&]() { if (x < kRansL) { if (p >= nw) { trunc = true; return; } x = (x << 32) | word(p++); } };
    auto get_bits = [&]() -> int32_t {                                                 // Rans64DecGetBits, rans_interface.cpp:80-96
        const int32_t v = (int32_t)(x & ((1u << kBypassBits) - 1));
        x >>= kBypassBits;
        renorm();
        return v;
    };
    for (size_t i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        if (ci < 0 || ci >= n_cdf) return PC_ERR_INDEX;
        const int32_t len = cdf_sizes[ci];
        if (len < 2 || len > cdf_stride) return PC_ERR_CDF;
        const int32_t* cdf = cdfs + (size_t)ci * cdf_stride;
        const int32_t max_value = len - 2;
        const uint32_t cf = (uint32_t)(x & 0xFFFFu);                                   // Rans64DecGet
        // last s in [0, len-2] with cdf[s] <= cf  (== find_if(first > cf) - 1, rans_interface.cpp:238-241)
        int32_t lo = 0, hi = len - 1;
        while (hi - lo > 1) { const int32_t mid = (lo + hi) >> 1; if ((uint32_t)cdf[mid] <= cf) lo = mid; else hi = mid; }
        const int32_t s = lo;
        const uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)cdf[s + 1] - start;
        if (start > cf || freq == 0 || freq > 65536u) return PC_ERR_CDF;               // (the row is not an increasing CDF: the reference asserts, :48-57)
        x = (uint64_t)freq * (x >> kPrecision) + (x & 0xFFFFu) - start;               // Rans64DecAdvance, rans64.h:126-142
        renorm();
        uint32_t value = (uint32_t)s;                                                   // (unsigned: a corrupt stream may carry any 32-bit escape code)
        if (s == max_value) {                                                          // :247-269
            int32_t val = get_bits();
            int64_t n_bypass = val;
            while (val == kBypassMax) { val = get_bits(); n_bypass += val; if (trunc) return PC_ERR_TRUNCATED; }
            uint32_t raw = 0;
            for (int64_t j = 0; j < n_bypass; ++j) { val = get_bits(); if (j < 8) raw |= (uint32_t)val << (j * kBypassBits); if (trunc) return PC_ERR_TRUNCATED; }
            value = raw >> 1;
            if (raw & 1) value = 0u - value - 1u; else value += (uint32_t)max_value;
        }
        if (trunc) return PC_ERR_TRUNCATED;
        out[i] = (int32_t)(value + (uint32_t)offsets[ci]);
    }
    return PC_OK;
}
}  // namespace

extern "C" int pc_rans_decode_with_indexes(const uint8_t* encoded, size_t encoded_len,
                                           const int32_t* indexes, size_t n,
                                           const int32_t* cdfs, int n_cdf, int cdf_stride,
                                           const int32_t* cdf_sizes, const int32_t* offsets,
                                           int32_t* out)
{
    if (!encoded || (!indexes && n) || !cdfs || !cdf_sizes || !offsets || (!out && n)) return PC_ERR_ARG;
    if (encoded_len < 8) return PC_ERR_TRUNCATED;
    uint32_t w0, w1;
    std::memcpy(&w0, encoded, 4); std::memcpy(&w1, encoded + 4, 4);
    uint64_t x = (uint64_t)w0 | ((uint64_t)w1 << 32);                                  // Rans64DecInit, rans64.h:107-115
    size_t p = 2;
    return rans_decode_core(encoded, encoded_len, x, p, indexes, n, cdfs, n_cdf, cdf_stride, cdf_sizes, offsets, out);
}

extern "C" int pc_rans_decode_stream(const uint8_t* encoded, size_t encoded_len, uint64_t* state,
                                     const int32_t* indexes, size_t n,
                                     const int32_t* cdfs, int n_cdf, int cdf_stride,
                                     const int32_t* cdf_sizes, const int32_t* offsets, int32_t* out)
{
    if (!encoded || !state || (!indexes && n) || !cdfs || !cdf_sizes || !offsets || (!out && n)) return PC_ERR_ARG;
    if (encoded_len < 8) return PC_ERR_TRUNCATED;
    if (state[1] == 0) {                                                               // RansDecoder::set_stream, rans_interface.cpp:277-283
        uint32_t w0, w1;
        std::memcpy(&w0, encoded, 4); std::memcpy(&w1, encoded + 4, 4);
        state[0] = (uint64_t)w0 | ((uint64_t)w1 << 32);
        state[1] = 2;
    }
    uint64_t x = state[0];
    size_t p = (size_t)state[1];
    if (p < 2 || p > encoded_len / 4) return PC_ERR_ARG;
    const int r = rans_decode_core(encoded, encoded_len, x, p, indexes, n, cdfs, n_cdf, cdf_stride, cdf_sizes, offsets, out);
    if (r == PC_OK) { state[0] = x; state[1] = (uint64_t)p; }
    return r;
}

extern "C" int pc_pmf_to_quantized_cdf(const float* pmf, int n, int precision, uint32_t* cdf)
{
    if (!pmf || !cdf || n <= 0 || precision < 1 || precision > 16) return PC_ERR_ARG;
    cdf[0] = 0;
    for (int i = 0; i < n; ++i) {
        const float p = pmf[i];
        if (!(p >= 0.0f) || !std::isfinite(p) || p > 65535.0f) return PC_ERR_CDF;      // (beyond that the float -> uint32 conversion below is undefined)
        cdf[i + 1] = (uint32_t)std::round(p * (float)(1 << precision));               // ops.cpp:20-21
    }
    uint32_t total = 0;
    for (int i = 0; i <= n; ++i) total += cdf[i];                                     // :23 (uint32 accumulate)
    if (total == 0) return PC_ERR_CDF;
    for (int i = 0; i <= n; ++i) cdf[i] = (uint32_t)((((uint64_t)1 << precision) * cdf[i]) / total);   // :25-28
    for (int i = 1; i <= n; ++i) cdf[i] += cdf[i - 1];                                // :30
    cdf[n] = 1u << precision;                                                         // :31
    for (int i = 0; i < n; ++i) {                                                     // :33-58
        if (cdf[i] != cdf[i + 1]) continue;
        uint32_t best_freq = ~0u;
        int best = -1;
        for (int j = 0; j < n; ++j) {
            const uint32_t f = cdf[j + 1] - cdf[j];
            if (f > 1 && f < best_freq) { best_freq = f; best = j; }
        }
        if (best < 0) return PC_ERR_CDF;
        if (best < i) { for (int j = best + 1; j <= i; ++j) cdf[j]--; }
        else { for (int j = i + 1; j <= best; ++j) cdf[j]++; }
    }
    return PC_OK;
}

// ------------------------------------------------------------------------------------------
// thread pool: parallel_for over independent streams
// ------------------------------------------------------------------------------------------
namespace pc {

// Several callers may be inside parallel_for at once (the lanes of a decoder; an encoder and a decoder object of the same process):
// every call is a job in a shared list, the workers take items from whichever job has the fewest left (a decoder's 32 short streams
// are not queued behind an encoder's 640), the caller works on its own job and returns when its last item is done.
struct ThreadPool::Impl {
    struct Job {
        const std::function<void(size_t)>* fn;
        size_t n, next = 0;                 // next: guarded by mu
        std::atomic<size_t> done{0};
    };
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<Job*> jobs;                 // jobs that still have unclaimed items
    bool stop = false;

    // claim one item of the job with the fewest unclaimed items (mu held); false: nothing to do
    bool claim(Job*& job, size_t& item)
    {
        Job* best = nullptr;
        for (Job* j : jobs) if (!best || j->n - j->next < best->n - best->next) best = j;
        if (!best) return false;
        job = best; item = best->next++;
        if (best->next == best->n) jobs.erase(std::find(jobs.begin(), jobs.end(), best));
        return true;
    }
    void finish(Job* job)                   // after running one item; `job` may be gone as soon as done reaches n
    {
        const size_t n = job->n;
        if (job->done.fetch_add(1, std::memory_order_acq_rel) + 1 == n) { std::lock_guard<std::mutex> lk(mu); cv_done.notify_all(); }
    }
};

// CPUs this process may really use: the affinity mask, cut by the cgroup CPU quota.  On a multi-GPU node every rank (one per GPU) has
// its own pool: the CPUs are divided evenly among the local ranks (LOCAL_WORLD_SIZE / LOCAL_RANK as torchrun exports them, or
// PC_LOCAL_WORLD_SIZE / PC_LOCAL_RANK), each rank takes its own contiguous slice of the allowed CPUs and pins its workers there,
// so eight ranks do not pile 8 x 16 unpinned threads onto the same cores.
static std::vector<int> allowed_cpus()
{
    std::vector<int> cpus;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0)
        for (int c = 0; c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &set)) cpus.push_back(c);
    if (cpus.empty()) { const unsigned hc = std::thread::hardware_concurrency(); for (unsigned c = 0; c < (hc ? hc : 4); ++c) cpus.push_back((int)c); }
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {          // cgroup v2 quota: "max 100000" or "<quota> <period>"
        char q[32] = {0};
        long period = 0;
        if (std::fscanf(f, "%31s %ld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
            const long n = std::max(1L, std::atol(q) / period);
            if ((size_t)n < cpus.size()) cpus.resize((size_t)n);
        }
        std::fclose(f);
    }
    return cpus;
}

static int env_int(const char* a, const char* b, int dflt)
{
    for (const char* name : {a, b}) if (const char* v = name ? std::getenv(name) : nullptr) { const int x = std::atoi(v); if (x >= 0) return x; }
    return dflt;
}

extern "C" int pc_host_pool_plan(int* n_threads, int* first_cpu, int* n_allowed)
{
    // the plan the default pool uses (exported for tests / bench.py's report)
    const std::vector<int> cpus = allowed_cpus();
    const int lws = std::max(1, env_int("PC_LOCAL_WORLD_SIZE", "LOCAL_WORLD_SIZE", 1));
    const int lr = std::min(lws - 1, std::max(0, env_int("PC_LOCAL_RANK", "LOCAL_RANK", 0)));
    int share = std::max(1, (int)cpus.size() / lws);
    int n = std::min(share, 16);                                         // one codec serves one GPU: ~16 cores is its share of an 8-GPU node
    if (const char* e = std::getenv("PC_HOST_THREADS")) { const int v = std::atoi(e); if (v > 0) n = v; }
    if (n_threads) *n_threads = n;
    if (first_cpu) *first_cpu = cpus[((size_t)lr * share) % cpus.size()];
    if (n_allowed) *n_allowed = (int)cpus.size();
    return PC_OK;
}

ThreadPool::ThreadPool(int n_threads) : impl_(new Impl)
{
    std::vector<int> pin;                                  // CPUs of this rank's slice (empty: no pinning)
    if (n_threads <= 0) {
        const std::vector<int> cpus = allowed_cpus();
        const int lws = std::max(1, env_int("PC_LOCAL_WORLD_SIZE", "LOCAL_WORLD_SIZE", 1));
        const int lr = std::min(lws - 1, std::max(0, env_int("PC_LOCAL_RANK", "LOCAL_RANK", 0)));
        const int share = std::max(1, (int)cpus.size() / lws);
        n_threads = std::min(share, 16);
        if (const char* e = std::getenv("PC_HOST_THREADS")) { const int v = std::atoi(e); if (v > 0) n_threads = v; }
        const char* np = std::getenv("PC_HOST_NO_PIN");
        if (lws > 1 && !(np && std::atoi(np)))
            for (int t = 0; t < share; ++t) pin.push_back(cpus[((size_t)lr * share + t) % cpus.size()]);
    }
    n_ = n_threads;
    for (int t = 0; t < n_threads - 1; ++t) {
        impl_->workers.emplace_back([this, pin, t] {
            if (!pin.empty()) {                            // worker t of this rank stays on one CPU of the rank's slice
                cpu_set_t set;
                CPU_ZERO(&set);
                CPU_SET(pin[(size_t)(t + 1) % pin.size()], &set);
                (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
            }
            Impl& s = *impl_;
            for (;;) {
                Impl::Job* job = nullptr;
                size_t item = 0;
                {
                    std::unique_lock<std::mutex> lk(s.mu);
                    s.cv_work.wait(lk, [&] { return s.stop || !s.jobs.empty(); });
                    if (s.stop) return;
                    if (!s.claim(job, item)) continue;
                }
                (*job->fn)(item);
                s.finish(job);
            }
        });
    }
}

ThreadPool::~ThreadPool()
{
    { std::lock_guard<std::mutex> lk(impl_->mu); impl_->stop = true; }
    impl_->cv_work.notify_all();
    for (auto& w : impl_->workers) w.join();
    delete impl_;
}

void ThreadPool::parallel_for(size_t n, const std::function<void(size_t)>& fn)
{
    Impl& s = *impl_;
    if (n == 0) return;
    if (s.workers.empty() || n == 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
    Impl::Job job;
    job.fn = &fn; job.n = n;
    { std::lock_guard<std::mutex> lk(s.mu); s.jobs.push_back(&job); }
    s.cv_work.notify_all();
    for (;;) {                                             // the caller works on its own job only
        size_t item;
        {
            std::lock_guard<std::mutex> lk(s.mu);
            if (job.next >= job.n) break;
            item = job.next++;
            if (job.next == job.n) s.jobs.erase(std::find(s.jobs.begin(), s.jobs.end(), &job));
        }
        fn(item);
        job.done.fetch_add(1, std::memory_order_acq_rel);
    }
    std::unique_lock<std::mutex> lk(s.mu);
    s.cv_done.wait(lk, [&] { return job.done.load(std::memory_order_acquire) == n; });
}

ThreadPool& default_pool()
{
    static ThreadPool pool(0);
    return pool;
}

}  // namespace pc

namespace pc {

// Start table of the fast decoder: for row r and the high byte `hi` of the 16-bit cumulative frequency, ONE 64-bit entry holding the
// largest s (<= len-2) with row[s] <= hi << 8 together with row[s] and row[s+1]:  s | row[s] << 16 | row[s+1] << 32.  A symbol whose
// interval contains the whole bucket -- the common case on the peaked rows that carry most symbols -- is resolved by that one load;
// the decoder steps forward through the row only when cf >= row[s+1].  (Round 3 kept s alone and loaded row[s+1], row[s] behind it: two
// dependent loads on the chain that IS the per-stream decode rate.)
void build_decode_lut(const int32_t* cdf, int n, int stride, const int32_t* len, uint64_t* lut)
{
    for (int r = 0; r < n; ++r) {
        const int32_t* row = cdf + (size_t)r * stride;
        int s = 0;
        for (int hi = 0; hi < 256; ++hi) {
            const int32_t v = hi << 8;
            while (s + 1 < len[r] - 1 && row[s + 1] <= v) ++s;         // largest s <= len-2 with row[s] <= v
            lut[(size_t)r * 256 + hi] = (uint64_t)(uint16_t)s | ((uint64_t)(uint32_t)row[s] << 16) | ((uint64_t)(uint32_t)row[s + 1] << 32);
        }
    }
}

namespace {
struct DecState {
    const uint8_t* enc; size_t nw, p; uint64_t x; int err;
    void init(const uint8_t* e, size_t len)
    {
        enc = e; nw = len / 4; p = 2; err = len < 8 ? PC_ERR_TRUNCATED : PC_OK; x = 0;
        if (!err) { uint32_t w0, w1; std::memcpy(&w0, e, 4); std::memcpy(&w1, e + 4, 4); x = (uint64_t)w0 | ((uint64_t)w1 << 32); }
    }
    inline void renorm()
    {
        if (x < kRansL) {
            if (p >= nw) { err = PC_ERR_TRUNCATED; return; }
            uint32_t w; std::memcpy(&w, enc + 4 * p, 4); ++p;
            x = (x << 32) | w;
        }
    }
    inline int32_t bits() { const int32_t v = (int32_t)(x & ((1u << kBypassBits) - 1)); x >>= kBypassBits; renorm(); return v; }
    inline int32_t symbol(int ci, const DecTables& t)
    {
        const int32_t len = t.len[ci];
        const int32_t* cdf = t.cdf + (size_t)ci * t.stride;
        const uint32_t cf = (uint32_t)(x & 0xFFFFu);
        const uint64_t e = t.lut[(size_t)ci * 256 + (cf >> 8)];
        int32_t s = (int32_t)(e & 0xffffu);
        uint32_t start = (uint32_t)(e >> 16) & 0xffffu, next = (uint32_t)(e >> 32);
        if (__builtin_expect(next <= cf, 0)) {                            // the bucket holds a boundary below cf: walk the row
            do { ++s; } while ((uint32_t)cdf[s + 1] <= cf);               // cdf[len-1] = 65536 > cf: stops at s <= len-2
            start = (uint32_t)cdf[s]; next = (uint32_t)cdf[s + 1];
        }
        x = (uint64_t)(next - start) * (x >> kPrecision) + cf - start;
        renorm();
        uint32_t value = (uint32_t)s;
        if (__builtin_expect(s == len - 2, 0)) {                          // bypass, rans_interface.cpp:247-269
            int32_t val = bits();
            int64_t nb = val;
            while (val == kBypassMax && !err) { val = bits(); nb += val; }
            uint32_t raw = 0;
            for (int64_t j = 0; j < nb && !err; ++j) { val = bits(); if (j < 8) raw |= (uint32_t)val << (j * kBypassBits); }
            value = raw >> 1;
            if (raw & 1) value = 0u - value - 1u; else value += (uint32_t)(len - 2);
        }
        return (int32_t)(value + (uint32_t)t.off[ci]);
    }
};
}  // namespace

int rans_decode_u8_batch(const uint8_t* const* encoded, const size_t* encoded_lens, size_t n_streams, const uint8_t* indexes, size_t n,
                         const DecTables& t, int32_t* out, int n_threads)
{
    if (!encoded || !encoded_lens || (!indexes && n) || !t.cdf || !t.lut || (!out && n)) return PC_ERR_ARG;
    for (size_t s = 0; s < n_streams; ++s) if (!encoded[s] && encoded_lens[s]) return PC_ERR_ARG;
    std::atomic<int> rc{PC_OK};
    const size_t n_pairs = (n_streams + 1) / 2;
    auto job = [&](size_t pr) {
        const size_t a = 2 * pr, b = 2 * pr + 1 < n_streams ? 2 * pr + 1 : a;       // a lone last stream is paired with itself (decoded once)
        DecState A, B;
        A.init(encoded[a], encoded_lens[a]);
        B.init(encoded[b], encoded_lens[b]);
        const uint8_t *ia = indexes + a * n, *ib = indexes + b * n;
        int32_t *oa = out + a * n, *ob = out + b * n;
        if (A.err || B.err) { rc = PC_ERR_TRUNCATED; return; }
        const bool two = b != a;
        for (size_t i = 0; i < n; ++i) {
            const int ca = ia[i], cb = ib[i];
            if (ca >= t.n || cb >= t.n) { rc = PC_ERR_INDEX; return; }
            oa[i] = A.symbol(ca, t);
            if (two) ob[i] = B.symbol(cb, t);
            if (A.err | B.err) { rc = PC_ERR_TRUNCATED; return; }
        }
    };
    if (n_threads == 1) for (size_t j = 0; j < n_pairs; ++j) job(j);
    else default_pool().parallel_for(n_pairs, job);
    return rc;
}

}  // namespace pc

extern "C" int pc_rans_encode_batch(const int32_t* symbols, const int32_t* indexes, size_t n_streams, size_t n,
                                    const int32_t* cdfs, int n_cdf, int cdf_stride,
                                    const int32_t* cdf_sizes, const int32_t* offsets,
                                    uint8_t* out, size_t out_stride, size_t* out_lens, int n_threads)
{
    if (!out || !out_lens || (out_stride & 3) || ((!symbols || !indexes) && n && n_streams)) return PC_ERR_ARG;
    std::atomic<int> rc{PC_OK};
    auto job = [&](size_t s) {
        const int r = pc_rans_encode_with_indexes(symbols + s * n, indexes + s * n, n, cdfs, n_cdf, cdf_stride, cdf_sizes,
                                                  offsets, out + s * out_stride, out_stride, &out_lens[s]);
        if (r != PC_OK) rc = r;
    };
    if (n_threads == 1) for (size_t s = 0; s < n_streams; ++s) job(s);
    else pc::default_pool().parallel_for(n_streams, job);
    return rc;
}

extern "C" int pc_rans_decode_batch(const uint8_t* const* encoded, const size_t* encoded_lens, size_t n_streams,
                                    const int32_t* indexes, size_t n,
                                    const int32_t* cdfs, int n_cdf, int cdf_stride,
                                    const int32_t* cdf_sizes, const int32_t* offsets,
                                    int32_t* out, int n_threads)
{
    if (!encoded || !encoded_lens || ((!indexes || !out) && n && n_streams)) return PC_ERR_ARG;
    for (size_t s = 0; s < n_streams; ++s) if (!encoded[s]) return PC_ERR_ARG;
    std::atomic<int> rc{PC_OK};
    auto job = [&](size_t s) {
        const int r = pc_rans_decode_with_indexes(encoded[s], encoded_lens[s], indexes + s * n, n, cdfs, n_cdf, cdf_stride,
                                                  cdf_sizes, offsets, out + s * n);
        if (r != PC_OK) rc = r;
    };
    if (n_threads == 1) for (size_t s = 0; s < n_streams; ++s) job(s);
    else pc::default_pool().parallel_for(n_streams, job);
    return rc;
}

extern "C" int pc_rans_decode_batch_u8(const uint8_t* const* encoded, const size_t* encoded_lens, size_t n_streams,
                                       const uint8_t* indexes, size_t n,
                                       const int32_t* cdfs, int n_cdf, int cdf_stride,
                                       const int32_t* cdf_sizes, const int32_t* offsets,
                                       int32_t* out, int n_threads)
{
    if (!cdfs || !cdf_sizes || !offsets || n_cdf <= 0 || n_cdf > 256 || cdf_stride < 2) return PC_ERR_ARG;
    for (int i = 0; i < n_cdf; ++i) {                                // the start table needs well-formed rows (rans_interface.cpp:48-57)
        const int32_t* row = cdfs + (size_t)i * cdf_stride;
        if (cdf_sizes[i] < 2 || cdf_sizes[i] > cdf_stride || row[0] != 0 || row[cdf_sizes[i] - 1] != (1 << 16)) return PC_ERR_CDF;
        for (int j = 0; j + 1 < cdf_sizes[i]; ++j) if (row[j + 1] <= row[j]) return PC_ERR_CDF;
    }
    std::vector<uint64_t> lut((size_t)n_cdf * 256);
    pc::build_decode_lut(cdfs, n_cdf, cdf_stride, cdf_sizes, lut.data());
    return pc::rans_decode_u8_batch(encoded, encoded_lens, n_streams, indexes, n, pc::DecTables{cdfs, n_cdf, cdf_stride, cdf_sizes, offsets, lut.data()}, out,
                                    n_threads);
}

extern "C" const char* pc_version(void) { return "progressivecodec_amd 0.3 (gfx950), numeric contract 0x00020001"; }
extern "C" uint32_t pc_contract_id(void) { return PC_NUMERIC_CONTRACT_ID; }

extern "C" const char* pc_strerror(int code)
{
    switch (code) {
    case PC_OK: return "ok";
    case PC_ERR_ARG: return "invalid argument or unsupported shape";
    case PC_ERR_INDEX: return "CDF index out of range";
    case PC_ERR_BUFFER: return "output buffer too small";
    case PC_ERR_TRUNCATED: return "truncated bitstream";
    case PC_ERR_CDF: return "malformed CDF / pmf";
    case PC_ERR_HIP: return "HIP runtime error";
    case PC_ERR_NOMEM: return "out of memory";
    case PC_ERR_STATE: return "object not ready (tables or weights missing; run update()/finalize first)";
    case PC_ERR_MISSING: return "state_dict tensor missing or of the wrong shape";
    default: return "unknown error";
    }
}
