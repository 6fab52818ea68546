// pc_host.h -- host utilities of libpcodec (thread pool).
#ifndef PC_HOST_H
#define PC_HOST_H
#include <cstddef>
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
}  // namespace pc
#endif
