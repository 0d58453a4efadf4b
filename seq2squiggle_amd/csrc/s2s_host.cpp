// s2s_host.cpp -- host-side (no GPU) helpers of the predict path's back end, part of libs2s_hip.so.
//
// s2s_blow5_pack frames and compresses one batch of BLOW5 records on worker threads: the job pyslow5's
// write_record_batch(threads = cpu_count) does for the reference (signal_io.py:167-171).  The Python writer builds the small
// per-record field bytes (ids, calibration, auxiliary fields -- signal_io.py:143-161) and hands over the packed int16 samples
// as they came off the GPU; nothing here runs under the interpreter lock.
#include "../../include/s2s_hip.h"

#include <dlfcn.h>
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace {

// ---- optional codecs, bound at first use (neither ships a header in this image)
struct Zstd {
    size_t (*compress)(void*, size_t, const void*, size_t, int) = nullptr;
    size_t (*bound)(size_t) = nullptr;
    unsigned (*is_error)(size_t) = nullptr;
    bool ok = false;
    Zstd() {
        void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        compress = reinterpret_cast<decltype(compress)>(dlsym(h, "ZSTD_compress"));
        bound = reinterpret_cast<decltype(bound)>(dlsym(h, "ZSTD_compressBound"));
        is_error = reinterpret_cast<decltype(is_error)>(dlsym(h, "ZSTD_isError"));
        ok = compress && bound && is_error;
    }
};
const Zstd& zstd() { static Zstd z; return z; }

// libdeflate writes the same zlib container (RFC 1950) about three times faster than zlib's own level 1
struct Deflate {
    void* (*alloc)(int) = nullptr;
    size_t (*zlib_compress)(void*, const void*, size_t, void*, size_t) = nullptr;
    size_t (*zlib_bound)(void*, size_t) = nullptr;
    void (*free_)(void*) = nullptr;
    bool ok = false;
    Deflate() {
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = reinterpret_cast<decltype(alloc)>(dlsym(h, "libdeflate_alloc_compressor"));
        zlib_compress = reinterpret_cast<decltype(zlib_compress)>(dlsym(h, "libdeflate_zlib_compress"));
        zlib_bound = reinterpret_cast<decltype(zlib_bound)>(dlsym(h, "libdeflate_zlib_compress_bound"));
        free_ = reinterpret_cast<decltype(free_)>(dlsym(h, "libdeflate_free_compressor"));
        ok = alloc && zlib_compress && zlib_bound && free_;
    }
};
const Deflate& deflate() { static Deflate d; return d; }

// ---- a process-wide pool of worker threads; run(n, fn) calls fn(i, worker) for i in [0, n) and returns when all are done
class Pool {
  public:
    explicit Pool(int n) {
        for (int w = 0; w < n; ++w) workers_.emplace_back([this, w] { loop(w); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    int size() const { return (int)workers_.size(); }
    void run(int n, const std::function<void(int, int)>& fn) {
        std::unique_lock<std::mutex> l(m_);
        fn_ = &fn; n_ = n; next_ = 0; left_ = n; ++gen_;
        cv_.notify_all();
        // every task done AND every worker that joined this generation back out of its task loop: none of them may
        // touch `fn` or the task counter of the next call
        done_.wait(l, [this] { return left_ == 0 && active_ == 0; });
        fn_ = nullptr;
    }

  private:
    void loop(int w) {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(int, int)>* fn;
            int n;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return stop_ || (gen_ != seen && fn_); });
                if (stop_) return;
                seen = gen_;
                fn = fn_;
                n = n_;
                ++active_;
            }
            int finished = 0;
            for (;;) {
                const int i = next_.fetch_add(1);
                if (i >= n) break;
                (*fn)(i, w);
                ++finished;
            }
            std::lock_guard<std::mutex> l(m_);
            left_ -= finished;
            --active_;
            if (left_ == 0 && active_ == 0) done_.notify_all();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int, int)>* fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, left_ = 0, active_ = 0;
    unsigned long gen_ = 0;
    bool stop_ = false;
};

std::mutex g_pool_mutex;          // one batch at a time (the writers call from a single writer thread anyway)
Pool* g_pool = nullptr;
std::vector<std::vector<uint8_t>> g_scratch;   // per worker: the uncompressed body of the record it is on (kept between calls:
                                               // fresh pages every batch cost more than the compression of a small one)

}  // namespace

extern "C" {

int64_t s2s_blow5_pack_bound(int64_t body_bytes_total, int32_t n_records) {
    if (body_bytes_total < 0 || n_records < 0) return S2S_ERR_ARG;
    // per record: u64 size + the worst case of the three codecs (zstd's bound is the largest: n + n/256 + 64)
    return body_bytes_total + body_bytes_total / 128 + (int64_t)n_records * (8 + 1024);
}

int64_t s2s_blow5_pack(const uint8_t* prefix, const int64_t* prefix_offs, const uint8_t* suffix, const int64_t* suffix_offs,
                       const uint8_t* signal, const int64_t* signal_offs, int32_t n, int32_t method, int32_t level,
                       int32_t threads, uint8_t* out, int64_t capacity) {
    if (n < 0 || threads < 1 || (n > 0 && (!prefix || !prefix_offs || !suffix || !suffix_offs || !signal || !signal_offs || !out)))
        return S2S_ERR_ARG;
    if (method < 0 || method > 2) return S2S_ERR_ARG;
    if (method == 2 && !zstd().ok) return S2S_ERR_ARG;
    if (n == 0) return 0;
    std::lock_guard<std::mutex> guard(g_pool_mutex);
    if (!g_pool || g_pool->size() < threads) { delete g_pool; g_pool = new Pool(threads); }   // grows, never shrinks
    const int workers = g_pool->size();

    // slot i of `out`: room for record i's worst case, so that workers never wait for each other; compacted afterwards
    std::vector<int64_t> slot(n + 1), size(n);
    slot[0] = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t body = (prefix_offs[i + 1] - prefix_offs[i]) + (signal_offs[i + 1] - signal_offs[i]) + (suffix_offs[i + 1] - suffix_offs[i]);
        slot[i + 1] = slot[i] + 8 + body + body / 128 + 1024;
    }
    if (slot[n] > capacity) return S2S_ERR_ARG;
    std::atomic<int> failed{0};
    if ((int)g_scratch.size() < workers) g_scratch.resize(workers);
    std::vector<std::vector<uint8_t>>& scratch = g_scratch;
    std::vector<void*> defl(workers, nullptr);
    const bool use_deflate = method == 1 && deflate().ok;
    g_pool->run(n, [&](int i, int w) {
        const int64_t np = prefix_offs[i + 1] - prefix_offs[i], ns = signal_offs[i + 1] - signal_offs[i], nx = suffix_offs[i + 1] - suffix_offs[i];
        const int64_t body = np + ns + nx;
        uint8_t* dst = out + slot[i] + 8;
        const int64_t room = slot[i + 1] - slot[i] - 8;
        int64_t written = -1;
        if (method == 0) {
            std::memcpy(dst, prefix + prefix_offs[i], np);
            std::memcpy(dst + np, signal + signal_offs[i], ns);
            std::memcpy(dst + np + ns, suffix + suffix_offs[i], nx);
            written = body;
        } else {
            std::vector<uint8_t>& b = scratch[w];
            if ((int64_t)b.size() < body) b.resize(body);
            std::memcpy(b.data(), prefix + prefix_offs[i], np);
            std::memcpy(b.data() + np, signal + signal_offs[i], ns);
            std::memcpy(b.data() + np + ns, suffix + suffix_offs[i], nx);
            if (use_deflate) {
                if (!defl[w]) defl[w] = deflate().alloc(level < 1 ? 1 : level);
                const size_t r = defl[w] ? deflate().zlib_compress(defl[w], b.data(), body, dst, room) : 0;
                if (r) written = (int64_t)r;
            } else if (method == 1) {
                uLongf dl = (uLongf)room;
                if (compress2(dst, &dl, b.data(), (uLong)body, level) == Z_OK) written = (int64_t)dl;
            } else {
                const size_t r = zstd().compress(dst, room, b.data(), body, level);
                if (!zstd().is_error(r)) written = (int64_t)r;
            }
        }
        if (written < 0) { failed = 1; written = 0; }
        size[i] = written;
        const uint64_t sz = (uint64_t)written;
        std::memcpy(out + slot[i], &sz, 8);                     // (little-endian host)
    });
    for (void* d : defl)
        if (d) deflate().free_(d);
    if (failed) return S2S_ERR_HIP;
    int64_t pos = 0;
    for (int i = 0; i < n; ++i) {                               // close the gaps, in record order
        const int64_t len = 8 + size[i];
        if (pos != slot[i]) std::memmove(out + pos, out + slot[i], len);
        pos += len;
    }
    return pos;
}

int64_t s2s_compress_rows(const uint8_t* in, const int64_t* in_offs, int32_t n, int32_t method, int32_t level, int32_t threads,
                          uint8_t* out, int64_t capacity, int64_t* out_offs) {
    if (n < 0 || threads < 1 || !out_offs || (n > 0 && (!in || !in_offs || !out))) return S2S_ERR_ARG;
    if (method != 1 && method != 2) return S2S_ERR_ARG;
    if (method == 2 && !zstd().ok) return S2S_ERR_ARG;
    out_offs[0] = 0;
    if (n == 0) return 0;
    std::lock_guard<std::mutex> guard(g_pool_mutex);
    if (!g_pool || g_pool->size() < threads) { delete g_pool; g_pool = new Pool(threads); }
    const int workers = g_pool->size();
    std::vector<int64_t> slot(n + 1), size(n);
    slot[0] = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t len = in_offs[i + 1] - in_offs[i];
        if (len < 0) return S2S_ERR_ARG;
        slot[i + 1] = slot[i] + len + len / 128 + 1024;
    }
    if (slot[n] > capacity) return S2S_ERR_ARG;
    std::atomic<int> failed{0};
    std::vector<void*> defl(workers, nullptr);
    const bool use_deflate = method == 1 && deflate().ok;
    g_pool->run(n, [&](int i, int w) {
        const uint8_t* src = in + in_offs[i];
        const int64_t len = in_offs[i + 1] - in_offs[i], room = slot[i + 1] - slot[i];
        uint8_t* dst = out + slot[i];
        int64_t written = -1;
        if (use_deflate) {
            if (!defl[w]) defl[w] = deflate().alloc(level < 1 ? 1 : level);
            const size_t r = defl[w] ? deflate().zlib_compress(defl[w], src, len, dst, room) : 0;
            if (r) written = (int64_t)r;
        } else if (method == 1) {
            uLongf dl = (uLongf)room;
            if (compress2(dst, &dl, src, (uLong)len, level) == Z_OK) written = (int64_t)dl;
        } else {
            const size_t r = zstd().compress(dst, room, src, len, level);
            if (!zstd().is_error(r)) written = (int64_t)r;
        }
        if (written < 0) { failed = 1; written = 0; }
        size[i] = written;
    });
    for (void* d : defl)
        if (d) deflate().free_(d);
    if (failed) return S2S_ERR_HIP;
    int64_t pos = 0;
    for (int i = 0; i < n; ++i) {                               // close the gaps, in row order
        if (pos != slot[i]) std::memmove(out + pos, out + slot[i], size[i]);
        pos += size[i];
        out_offs[i + 1] = pos;
    }
    return pos;
}

}  // extern "C"
