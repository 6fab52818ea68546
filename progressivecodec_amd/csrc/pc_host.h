// pc_host.h -- host utilities of libpcodec (thread pool).
#ifndef PC_HOST_H
#define PC_HOST_H
#include <cstddef>
#include <cstdint>
#include <functional>

namespace pc {
class ThreadPool {
public:
    explicit ThreadPool(int n_threads);
    ~ThreadPool();
    void parallel_for(size_t n, const std::function<void(size_t)>& fn);
    int size() const { return n_; }
private:
    struct Impl;
    Impl* impl_;
    int n_;
};
ThreadPool& default_pool();

// The decoder's fast path (pc_codec.hip): byte indexes (the device packs them: four times less D2H traffic per slice), a 256-entry
// start table per CDF row instead of a binary search, and two streams decoded in lock step by one thread (two independent rANS
// states give the core something to overlap the dependent loads of one with).  Same symbols as pc_rans_decode_with_indexes.
struct DecTables {
    const int32_t* cdf; int n, stride; const int32_t* len; const int32_t* off;
    const uint64_t* lut;          // [n][256]: s | cdf[s] << 16 | cdf[s+1] << 32 for the largest s with cdf[s] <= (hi << 8), built by build_decode_lut
};
void build_decode_lut(const int32_t* cdf, int n, int stride, const int32_t* len, uint64_t* lut);
int rans_decode_u8_batch(const uint8_t* const* encoded, const size_t* encoded_lens, size_t n_streams, const uint8_t* indexes, size_t n,
                         const DecTables& t, int32_t* out, int n_threads);
}  // namespace pc
#endif
