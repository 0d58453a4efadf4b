// s2s_host.cpp -- host-side (no GPU) helpers of the predict path's back end, part of libs2s_hip.so.
//
// s2s_blow5_pack frames and compresses one batch of BLOW5 records on worker threads: the job pyslow5's
// write_record_batch(threads = cpu_count) does for the reference (signal_io.py:167-171).  The Python writer builds the small
// per-record field bytes (ids, calibration, auxiliary fields -- signal_io.py:143-161) and hands over the packed int16 samples
// as they came off the GPU; nothing here runs under the interpreter lock.
#include "../../include/s2s_hip.h"

#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
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

// ---- Huffman-only deflate (RFC 1951, dynamic blocks, literals only): the encoder of method 3.
// A nanopore signal is noise on top of a level: LZ77 finds next to nothing in it (libdeflate level 1: 0.69-0.77 of the raw size,
// Huffman coding of the bytes alone: 0.71-0.78), but the match search is what a deflate encoder spends its time in.  One piece =
// one dynamic block with its own code + an empty stored block that pads to a byte boundary (and carries the final bit of the
// last piece), so pieces are encoded independently and laid back to back.

// Code lengths (<= max_len, a complete code) for freq[0..n): symbols with freq 0 get length 0; at least two symbols get a code.
void huff_lengths(const uint32_t* freq_in, int n, int max_len, uint8_t* len_out) {
    uint32_t freq[288];
    int order[288], used = 0;
    for (int i = 0; i < n; ++i) { freq[i] = freq_in[i]; len_out[i] = 0; }
    for (int i = 0; i < n && used < 2; ++i) used += freq[i] != 0;
    for (int i = 0; used < 2; ++i)                       // a decoder wants a complete code: give a second symbol a leaf
        if (!freq[i]) { freq[i] = 1; ++used; }
    used = 0;
    for (int i = 0; i < n; ++i)
        if (freq[i]) order[used++] = i;
    std::sort(order, order + used, [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
    // two-queue Huffman: leaves in ascending order, internal nodes are created in ascending order as well
    uint64_t w[576];
    int parent[576];
    for (int i = 0; i < used; ++i) w[i] = freq[order[i]];
    int leaf = 0, inner = used, made = used;
    auto take = [&]() { return (leaf < used && (inner >= made || w[leaf] <= w[inner])) ? leaf++ : inner++; };
    while (made < 2 * used - 1) {
        const int a = take(), b = take();
        w[made] = w[a] + w[b];
        parent[a] = parent[b] = made;
        ++made;
    }
    int count[64] = {0};
    for (int i = 0; i < used; ++i) {
        int d = 0;
        for (int v = i; v != made - 1; v = parent[v]) ++d;
        ++count[d < 63 ? d : 63];
    }
    // enforce max_len: everything deeper moves up to max_len, then lengths are traded until the Kraft sum is exactly one
    for (int d = max_len + 1; d < 64; ++d) { count[max_len] += count[d]; count[d] = 0; }
    uint64_t total = 0;
    for (int d = 1; d <= max_len; ++d) total += (uint64_t)count[d] << (max_len - d);
    while (total > (1ull << max_len)) {
        --count[max_len];
        for (int d = max_len - 1; d > 0; --d)
            if (count[d]) { --count[d]; count[d + 1] += 2; break; }
        --total;
    }
    int at = 0;                                          // rarest symbols take the longest codes
    for (int d = max_len; d >= 1; --d)
        for (int c = 0; c < count[d]; ++c) len_out[order[at++]] = (uint8_t)d;
}

// canonical codes of the lengths, bit-reversed (deflate writes Huffman codes starting from their most significant bit)
void huff_codes(const uint8_t* len, int n, uint16_t* code) {
    int bl_count[16] = {0}, next[16];
    for (int i = 0; i < n; ++i) ++bl_count[len[i]];
    bl_count[0] = 0;
    int c = 0;
    for (int b = 1; b < 16; ++b) { c = (c + bl_count[b - 1]) << 1; next[b] = c; }
    for (int i = 0; i < n; ++i) {
        if (!len[i]) { code[i] = 0; continue; }
        unsigned v = (unsigned)next[len[i]]++, r = 0;
        for (int b = 0; b < len[i]; ++b) { r = (r << 1) | (v & 1); v >>= 1; }
        code[i] = (uint16_t)r;
    }
}

struct BitOut {
    uint8_t* p;
    uint64_t acc = 0;
    int n = 0;
    explicit BitOut(uint8_t* dst) : p(dst) {}
    inline void put(uint32_t v, int bits) {              // bits <= 30; the buffer holds < 32 bits on entry
        acc |= (uint64_t)v << n;
        n += bits;
        if (n >= 32) { std::memcpy(p, &acc, 4); p += 4; acc >>= 32; n -= 32; }      // (little-endian host)
    }
    // Four codes of <= 15 bits each behind < 8 pending bits: one unconditional 8-byte store, the pointer moves by the whole bytes
    // (the destination has slack behind the stream: see the bound at huff_deflate_piece).
    inline void put4(uint32_t a, int la, uint32_t b, int lb, uint32_t c, int lc, uint32_t d, int ld) {
        if (n + la + lb + lc + ld >= 64) {               // four long codes in a row (rare symbols): the 64-bit word would overflow
            put(a | b << la, la + lb);
            put(c | d << lc, lc + ld);
            align_pending();
            return;
        }
        uint64_t v = acc | (uint64_t)a << n;
        int m = n + la;
        v |= (uint64_t)b << m; m += lb;
        v |= (uint64_t)c << m; m += lc;
        v |= (uint64_t)d << m; m += ld;                  // m < 64
        std::memcpy(p, &v, 8);
        p += m >> 3;
        acc = v >> (m & ~7);
        n = m & 7;
    }
    inline void align_pending() {                        // bring the buffer below 8 pending bits (put4's entry condition)
        while (n >= 8) { *p++ = (uint8_t)acc; acc >>= 8; n -= 8; }
    }
    uint8_t* finish() {                                  // pad to a byte boundary
        while (n > 0) { *p++ = (uint8_t)acc; acc >>= 8; n -= 8; }
        n = 0; acc = 0;
        return p;
    }
};

struct Segs {                                            // bytes [lo, hi) of the concatenation of up to three buffers
    const uint8_t* p[3];
    int64_t n[3];
    template <class F> void each(int64_t lo, int64_t hi, F&& f) const {
        int64_t base = 0;
        for (int s = 0; s < 3; ++s) {
            const int64_t a = lo > base ? lo : base, b = hi < base + n[s] ? hi : base + n[s];
            if (b > a) f(p[s] + (a - base), b - a);
            base += n[s];
        }
    }
};

// One piece -> dst (room >= len + len / 128 + 512): [dynamic block | stored blocks when those are smaller][empty stored block,
// final bit = last].  Returns the bytes written; *adler = Adler-32 of the piece's bytes.
int64_t huff_deflate_piece(const Segs& in, int64_t lo, int64_t hi, bool last, uint8_t* dst, uint32_t* adler) {
    uint32_t hist[8][256];                               // eight tables: int16 samples put near-equal bytes two apart
    std::memset(hist, 0, sizeof hist);
    uint32_t ad = 1;
    in.each(lo, hi, [&](const uint8_t* s, int64_t m) {
        ad = (uint32_t)adler32(ad, s, (uInt)m);
        int64_t i = 0;
        for (; i + 8 <= m; i += 8) {
            uint64_t w;
            std::memcpy(&w, s + i, 8);
            ++hist[0][w & 255]; ++hist[1][(w >> 8) & 255]; ++hist[2][(w >> 16) & 255]; ++hist[3][(w >> 24) & 255];
            ++hist[4][(w >> 32) & 255]; ++hist[5][(w >> 40) & 255]; ++hist[6][(w >> 48) & 255]; ++hist[7][w >> 56];
        }
        for (; i < m; ++i) ++hist[0][s[i]];
    });
    *adler = ad;
    uint32_t freq[257];
    for (int i = 0; i < 256; ++i) {
        freq[i] = 0;
        for (int k = 0; k < 8; ++k) freq[i] += hist[k][i];
    }
    freq[256] = 1;                                       // end of block
    uint8_t len[259];
    huff_lengths(freq, 257, 15, len);
    len[257] = len[258] = 1;                             // two distance codes of one bit, never used (what zlib sends as well)
    uint32_t clfreq[19] = {0};
    for (int i = 0; i < 259; ++i) ++clfreq[len[i]];
    uint8_t cllen[19];
    huff_lengths(clfreq, 16, 7, cllen);
    cllen[16] = cllen[17] = cllen[18] = 0;               // the run-length symbols are not used: 259 lengths cost ~130 bytes
    uint16_t code[257], clcode[19];
    huff_codes(len, 257, code);
    huff_codes(cllen, 19, clcode);
    int64_t bits = 3 + 5 + 5 + 4 + 19 * 3;
    for (int i = 0; i < 259; ++i) bits += cllen[len[i]];
    for (int i = 0; i < 257; ++i) bits += (int64_t)freq[i] * len[i];
    const int64_t n = hi - lo;
    uint8_t* p = dst;
    if ((bits + 7) / 8 >= n + 5 * ((n + 65534) / 65535)) {
        // incompressible: stored blocks (byte-aligned by construction)
        int64_t at = lo;
        while (at < hi) {
            const int64_t m = hi - at < 65535 ? hi - at : 65535;
            *p++ = 0;
            *p++ = (uint8_t)m; *p++ = (uint8_t)(m >> 8); *p++ = (uint8_t)~m; *p++ = (uint8_t)(~m >> 8);
            in.each(at, at + m, [&](const uint8_t* s, int64_t k) { std::memcpy(p, s, k); p += k; });
            at += m;
        }
    } else {
        BitOut out(p);
        out.put(0, 1);                                   // not the final block: the empty stored block below carries that bit
        out.put(2, 2);                                   // dynamic Huffman codes
        out.put(0, 5);                                   // HLIT: 257 literal/length codes
        out.put(1, 5);                                   // HDIST: 2 distance codes
        out.put(15, 4);                                  // HCLEN: all 19 code length codes
        static const uint8_t cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        for (int i = 0; i < 19; ++i) out.put(cllen[cl_order[i]], 3);
        for (int i = 0; i < 259; ++i) out.put(clcode[len[i]], cllen[len[i]]);
        uint32_t tab[256];                               // code | length << 16
        for (int i = 0; i < 256; ++i) tab[i] = code[i] | (uint32_t)len[i] << 16;
        in.each(lo, hi, [&](const uint8_t* s, int64_t m) {
            int64_t i = 0;
            out.align_pending();
            for (; i + 4 <= m; i += 4) {                 // four codes (<= 15 bits each) per store
                const uint32_t t = tab[s[i]], u = tab[s[i + 1]], v = tab[s[i + 2]], w = tab[s[i + 3]];
                out.put4(t & 0xFFFF, (int)(t >> 16), u & 0xFFFF, (int)(u >> 16), v & 0xFFFF, (int)(v >> 16), w & 0xFFFF, (int)(w >> 16));
            }
            for (; i < m; ++i) { const uint32_t t = tab[s[i]]; out.put(t & 0xFFFF, (int)(t >> 16)); }
        });
        out.put(code[256], len[256]);
        out.put(last ? 1 : 0, 1);                        // the empty stored block: final bit, type 00, pad, LEN 0, NLEN ~0
        out.put(0, 2);
        p = out.finish();
        *p++ = 0; *p++ = 0; *p++ = 0xFF; *p++ = 0xFF;
        return p - dst;
    }
    *p++ = last ? 1 : 0;                                 // after stored blocks the stream is byte-aligned already
    *p++ = 0; *p++ = 0; *p++ = 0xFF; *p++ = 0xFF;
    return p - dst;
}

}  // namespace

extern "C" {

int64_t s2s_blow5_pack_bound(int64_t body_bytes_total, int32_t n_records) {
    if (body_bytes_total < 0 || n_records < 0) return S2S_ERR_ARG;
    // per record: u64 size + the worst case of the three codecs (zstd's bound is the largest: n + n/256 + 64); a long zlib
    // record's pieces each end in a flush marker
    return body_bytes_total + body_bytes_total / 64 + (int64_t)n_records * (8 + 1024) + 4096;
}

int64_t s2s_blow5_pack(const uint8_t* prefix, const int64_t* prefix_offs, const uint8_t* suffix, const int64_t* suffix_offs,
                       const uint8_t* signal, const int64_t* signal_offs, int32_t n, int32_t method, int32_t level,
                       int32_t threads, uint8_t* out, int64_t capacity) {
    if (n < 0 || threads < 1 || (n > 0 && (!prefix || !prefix_offs || !suffix || !suffix_offs || !signal || !signal_offs || !out)))
        return S2S_ERR_ARG;
    if (method < 0 || method > 3) return S2S_ERR_ARG;
    if (method == 2 && !zstd().ok) return S2S_ERR_CODEC;
    if (n == 0) return 0;
    std::lock_guard<std::mutex> guard(g_pool_mutex);
    if (!g_pool || g_pool->size() < threads) { delete g_pool; g_pool = new Pool(threads); }   // grows, never shrinks
    const int workers = g_pool->size();
    const int zlevel = level < 1 ? 1 : (level > 9 ? 9 : level);

    // One task per record -- except that a batch is only done when its longest record is: with method 3 a record is deflated
    // as independent pieces of at most PIECE bytes (huff_deflate_piece); header + pieces + Adler-32 of the whole body are one
    // ordinary zlib stream (RFC 1950).
    constexpr int64_t PIECE = 128 << 10;
    struct Task { int rec; int64_t lo, hi; bool piece, last; int64_t slot, room, size; uint32_t adler; };
    std::vector<Task> tasks;
    std::vector<int> first_task(n + 1);
    int64_t need = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t body = (prefix_offs[i + 1] - prefix_offs[i]) + (signal_offs[i + 1] - signal_offs[i]) + (suffix_offs[i + 1] - suffix_offs[i]);
        first_task[i] = (int)tasks.size();
        if (method == 3) {
            const int64_t k = body > 0 ? (body + PIECE - 1) / PIECE : 1, len = body > 0 ? (body + k - 1) / k : 1;
            for (int64_t lo = 0; lo < body || lo == 0; lo += len) {
                const int64_t hi = lo + len < body ? lo + len : body;
                const int64_t room = (hi - lo) + (hi - lo) / 128 + 512;
                tasks.push_back({i, lo, hi, true, hi == body, need, room, 0, 0});
                need += room;
            }
        } else {
            const int64_t room = body + body / 128 + 1024;
            tasks.push_back({i, 0, body, false, true, need, room, 0, 0});
            need += room;
        }
    }
    first_task[n] = (int)tasks.size();
    static std::vector<uint8_t> staged;            // compressed pieces before they are laid back to back (kept between calls)
    if ((int64_t)staged.size() < need) staged.resize(need + need / 4);
    std::atomic<int> failed{0};
    if ((int)g_scratch.size() < workers) g_scratch.resize(workers);
    std::vector<std::vector<uint8_t>>& scratch = g_scratch;
    std::vector<void*> defl(workers, nullptr);
    const bool use_deflate = method == 1 && deflate().ok;
    g_pool->run((int)tasks.size(), [&](int t, int w) {
        Task& T = tasks[t];
        const int i = T.rec;
        const int64_t np = prefix_offs[i + 1] - prefix_offs[i], ns = signal_offs[i + 1] - signal_offs[i], nx = suffix_offs[i + 1] - suffix_offs[i];
        const int64_t body = np + ns + nx;
        uint8_t* dst = staged.data() + T.slot;
        int64_t written = -1;
        if (T.piece) {
            const Segs in = {{prefix + prefix_offs[i], signal + signal_offs[i], suffix + suffix_offs[i]}, {np, ns, nx}};
            written = huff_deflate_piece(in, T.lo, T.hi, T.last, dst, &T.adler);
            if (written > T.room) written = -1;
        } else if (method == 0) {
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
                if (!defl[w]) defl[w] = deflate().alloc(zlevel);
                const size_t r = defl[w] ? deflate().zlib_compress(defl[w], b.data(), body, dst, T.room) : 0;
                if (r) written = (int64_t)r;
            } else if (method == 1) {
                uLongf dl = (uLongf)T.room;
                if (compress2(dst, &dl, b.data(), (uLong)body, level) == Z_OK) written = (int64_t)dl;
            } else {
                const size_t r = zstd().compress(dst, T.room, b.data(), body, level);
                if (!zstd().is_error(r)) written = (int64_t)r;
            }
        }
        if (written < 0) { failed = 1; written = 0; }
        T.size = written;
    });
    for (void* d : defl)
        if (d) deflate().free_(d);
    if (failed) return S2S_ERR_CODEC;

    // where every record lands in `out`, then the copies (again on the workers: 10+ MB per batch)
    std::vector<int64_t> at(n + 1);
    at[0] = 0;
    for (int i = 0; i < n; ++i) {
        int64_t len = 0;
        for (int t = first_task[i]; t < first_task[i + 1]; ++t) len += tasks[t].size;
        if (tasks[first_task[i]].piece) len += 2 + 4;
        at[i + 1] = at[i] + 8 + len;
    }
    if (at[n] > capacity) return S2S_ERR_ARG;
    g_pool->run(n, [&](int i, int) {
        uint8_t* dst = out + at[i];
        const uint64_t sz = (uint64_t)(at[i + 1] - at[i] - 8);
        std::memcpy(dst, &sz, 8);                               // (little-endian host)
        dst += 8;
        const bool split = tasks[first_task[i]].piece;
        uint32_t ad = 1;
        if (split) { *dst++ = 0x78; *dst++ = 0x01; }             // CMF/FLG: deflate, 32 K window, fastest, check bits
        for (int t = first_task[i]; t < first_task[i + 1]; ++t) {
            const Task& T = tasks[t];
            std::memcpy(dst, staged.data() + T.slot, T.size);
            dst += T.size;
            if (split) ad = t == first_task[i] ? T.adler : (uint32_t)adler32_combine(ad, T.adler, (z_off_t)(T.hi - T.lo));
        }
        if (split) { dst[0] = (uint8_t)(ad >> 24); dst[1] = (uint8_t)(ad >> 16); dst[2] = (uint8_t)(ad >> 8); dst[3] = (uint8_t)ad; }
    });
    return at[n];
}

int64_t s2s_compress_rows(const uint8_t* in, const int64_t* in_offs, int32_t n, int32_t method, int32_t level, int32_t threads,
                          uint8_t* out, int64_t capacity, int64_t* out_offs) {
    if (n < 0 || threads < 1 || !out_offs || (n > 0 && (!in || !in_offs || !out))) return S2S_ERR_ARG;
    if (method != 1 && method != 2) return S2S_ERR_ARG;
    if (method == 2 && !zstd().ok) return S2S_ERR_CODEC;
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
    if (failed) return S2S_ERR_CODEC;
    int64_t pos = 0;
    for (int i = 0; i < n; ++i) {                               // close the gaps, in row order
        if (pos != slot[i]) std::memmove(out + pos, out + slot[i], size[i]);
        pos += size[i];
        out_offs[i + 1] = pos;
    }
    return pos;
}

}  // extern "C"

// ================================================================================ read-sampler replay
// The reference samples reads one after the other from Python's global `random` (Mersenne Twister): per attempt a start
// position (random.randint), a strand (random.choice, DNA only), and per N of an accepted read a replacement base; the read
// length comes from scipy with a per-(read, retry) seed (utils.py:311-331, 415-479).  Every draw consumes a data-dependent
// number of generator words, so a rank that owns reads lo..hi of a sharded run can only find the generator state of read lo by
// replaying the draws of reads 0..lo-1 -- 4.3 us per read in the interpreter, a few tens of ns here.  Nothing is built: the
// function walks the draws, returns the accepted reads' lengths and leaves the generator where the next read starts.
namespace {

struct Mt19937 {                      // CPython's _randommodule.c generator: state = random.getstate()[1] (624 words + index)
    uint32_t* mt;
    uint32_t idx;
    uint32_t next() {
        if (idx >= 624) {
            int kk;
            for (kk = 0; kk < 624 - 397; ++kk) {
                const uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
                mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            for (; kk < 623; ++kk) {
                const uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
                mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            const uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
            mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    // random._randbelow_with_getrandbits(n), 0 < n < 2^32: k = n.bit_length(); redraw getrandbits(k) until < n
    uint64_t below(uint64_t n) {
        int k = 0;
        for (uint64_t t = n; t; t >>= 1) ++k;
        for (;;) {
            const uint64_t r = (uint64_t)(next() >> (32 - k));
            if (r < n) return r;
        }
    }
};

inline uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// draw_expon_dis(mean = r, seed): scipy's expon.rvs(loc, scale, size = 1, random_state = seed) builds np.random.RandomState(seed)
// (init_genrand) and takes loc + scale * -log(1 - random_sample()); then int(x * r / 7106), clipped to [1, total_len]
// (reference utils.py:325-331).  Only the generator's first two outputs are needed: state words 0-2, 397, 398.
struct MtWords { uint32_t w0, w1, w2, w397, w398; };

#pragma clang fp contract(off)
int64_t expon_from_words(const MtWords& w, double r, int64_t total_len) {
    auto twist = [](uint32_t a, uint32_t b, uint32_t c) {
        const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
        return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    };
    const uint32_t a = mt_temper(twist(w.w0, w.w1, w.w397)) >> 5, b = mt_temper(twist(w.w1, w.w2, w.w398)) >> 6;
    const double u = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    const double e = -std::log(1.0 - u);
    const double loc = 213.98910256668592, scale = 6972.5319847131141, fitted_mean = 7106.0;
    double v = scale * e;
    v = loc + v;
    v = v * r;
    v = v / fitted_mean;
    int64_t n = (int64_t)v;                                  // numpy .astype(int): truncation
    if (n < 1) n = 1;
    if (n > total_len) n = total_len;
    return n;
}

int64_t expon_length(uint32_t seed, double r, int64_t total_len) {
    uint32_t x = seed;
    MtWords w{seed, 0, 0, 0, 0};
    for (uint32_t i = 1; i < 399; ++i) {
        x = 1812433253u * (x ^ (x >> 30)) + i;
        if (i == 1) w.w1 = x; else if (i == 2) w.w2 = x; else if (i == 397) w.w397 = x; else if (i == 398) w.w398 = x;
    }
    return expon_from_words(w, r, total_len);
}

// ---- the other two length laws (reference utils.py:311-322): scipy's gamma.rvs / beta.rvs on a fresh np.random.RandomState(seed)
// are numpy's LEGACY samplers (legacy-distributions.c): standard_gamma by Marsaglia-Tsang on legacy_gauss (the polar method, which
// keeps its second variate for the next call -- also across the two gammas of a beta) and 53-bit doubles of MT19937.  They consume
// a variable number of generator outputs, so the whole state is seeded (623 multiplies, ~1 us per (read, try)).
struct NpLegacy {
    uint32_t mt[624];
    int pos = 624;
    bool has_gauss = false;
    double gauss = 0.0;
    explicit NpLegacy(uint32_t seed) {
        mt[0] = seed;
        for (uint32_t i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i;
    }
    uint32_t next() {
        if (pos >= 624) {
            Mt19937 g{mt, 624};
            (void)g.next();                                       // regenerates the block (and tempers word 0, which we redo below)
            pos = 0;
        }
        return mt_temper(mt[pos++]);
    }
    double dbl() {
        const uint32_t a = next() >> 5, b = next() >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
    double normal() {
        if (has_gauss) { has_gauss = false; const double t = gauss; gauss = 0.0; return t; }
        double f, x1, x2, r2;
        do {
            x1 = 2.0 * dbl() - 1.0;
            x2 = 2.0 * dbl() - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        f = std::sqrt(-2.0 * std::log(r2) / r2);
        gauss = f * x1;
        has_gauss = true;
        return f * x2;
    }
    double std_gamma(double shape) {                              // shape > 1 (both laws)
        const double b = shape - 1.0 / 3.0, c = 1.0 / std::sqrt(9.0 * b);
        for (;;) {
            double X, V;
            do {
                X = normal();
                V = 1.0 + c * X;
            } while (V <= 0.0);
            V = V * V * V;
            const double U = dbl();
            if (U < 1.0 - 0.0331 * (X * X) * (X * X)) return b * V;
            if (std::log(U) < 0.5 * X * X + b * (1.0 - V + std::log(V))) return b * V;
        }
    }
};

// law 0: expon (above); 1: gamma.rvs(6.3693711, loc = 0.53834893) * r / 4.39; 2: beta.rvs(1.778, 7.892, loc = 316.758,
// scale = 34191.257) * r / 6615 -- scipy returns vals * scale + loc; then int(), clipped to [1, total_len]
int64_t law_length(int32_t law, uint32_t seed, double r, int64_t total_len) {
    if (law == 0) return expon_length(seed, r, total_len);
    NpLegacy g(seed);
    double v, fitted;
    if (law == 1) {
        v = g.std_gamma(6.3693711);
        v = v * 1.0 + 0.53834893;
        fitted = 4.39;
    } else {
        const double ga = g.std_gamma(1.778), gb = g.std_gamma(7.892);
        v = ga / (ga + gb);
        v = v * 34191.257;
        v = v + 316.758;
        fitted = 6615.0;
    }
    v = v * r;
    v = v / fitted;
    int64_t n = (int64_t)v;
    if (n < 1) n = 1;
    if (n > total_len) n = total_len;
    return n;
}

// The seeding recurrence is a serial chain of 398 multiplies per seed (0.5 us): eight seeds at a time fill the lanes of one AVX2
// register (the first tries of eight consecutive reads; retries stay scalar).  Same integer arithmetic, lane by lane.
#define S2S_MT_WORDS8_BODY                                                                          \
    uint32_t x[8];                                                                                  \
    for (int l = 0; l < 8; ++l) { x[l] = seeds[l]; out[l].w0 = seeds[l]; }                          \
    for (uint32_t i = 1; i < 399; ++i) {                                                            \
        for (int l = 0; l < 8; ++l) x[l] = 1812433253u * (x[l] ^ (x[l] >> 30)) + i;                 \
        if (i == 1) { for (int l = 0; l < 8; ++l) out[l].w1 = x[l]; }                               \
        else if (i == 2) { for (int l = 0; l < 8; ++l) out[l].w2 = x[l]; }                          \
        else if (i == 397) { for (int l = 0; l < 8; ++l) out[l].w397 = x[l]; }                      \
        else if (i == 398) { for (int l = 0; l < 8; ++l) out[l].w398 = x[l]; }                      \
    }
__attribute__((target("avx2"))) void mt_words8_avx2(const uint32_t* seeds, MtWords* out) { S2S_MT_WORDS8_BODY }
void mt_words8_plain(const uint32_t* seeds, MtWords* out) { S2S_MT_WORDS8_BODY }
void mt_words8(const uint32_t* seeds, MtWords* out) {
    static const bool avx2 = __builtin_cpu_supports("avx2") && !std::getenv("S2S_NO_AVX2");      // (the variable: tests of the plain path)
    if (avx2) mt_words8_avx2(seeds, out); else mt_words8_plain(seeds, out);
}

}  // namespace

extern "C" int64_t s2s_sampler_replay_law(uint32_t* mt_state, const int64_t* contig_ends, int32_t n_contigs,
                                          const int64_t* const* n_pos, const int64_t* n_pos_count, int64_t num_seqs, int64_t first_read_i, int64_t r,
                                          uint64_t seed, int64_t total_len, int32_t is_dna, int32_t min_read_len, int32_t max_retries,
                                          int64_t stop_after, int32_t law, int32_t* out_lengths, int64_t* out_next_read_i) {
    if (!mt_state || !contig_ends || n_contigs < 1 || num_seqs < 0 || first_read_i < 0 || r <= 0 || max_retries < 1 || !out_next_read_i ||
        law < 0 || law > 2)
        return S2S_ERR_ARG;
    const int64_t genome = contig_ends[n_contigs - 1];
    if (genome < 1 || genome >= (1ll << 31) || mt_state[624] > 624) return S2S_ERR_ARG;
    if (seed + (uint64_t)num_seqs * (uint64_t)(max_retries + 1) >= (1ull << 32)) return S2S_ERR_ARG;   // (the scipy seed must not wrap)
    Mt19937 g{mt_state, mt_state[624]};
    int64_t accepted = 0, read_i = first_read_i;
    int64_t first_try[8], first_base = -1;                    // lengths of the first tries of reads first_base .. first_base + 7
    for (; read_i < num_seqs && (stop_after < 0 || accepted < stop_after); ++read_i) {
        if (law == 0 && (first_base < 0 || read_i >= first_base + 8)) {
            uint32_t seeds[8];
            MtWords w[8];
            for (int l = 0; l < 8; ++l) seeds[l] = (uint32_t)(seed + (uint64_t)(read_i + l) * (uint64_t)(max_retries + 1));
            mt_words8(seeds, w);
            for (int l = 0; l < 8; ++l) first_try[l] = expon_from_words(w[l], (double)r, total_len);
            first_base = read_i;
        }
        for (int retry = 0; retry < max_retries; ++retry) {
            const int64_t pos = (int64_t)g.below((uint64_t)genome);                 // random.randint(0, total_genome_len - 1)
            const int64_t* ce = std::upper_bound(contig_ends, contig_ends + n_contigs, pos);   // bisect_right
            const int where = (int)(ce - contig_ends);
            const int64_t start = where ? contig_ends[where - 1] : 0, offset = pos - start, clen = contig_ends[where] - start;
            const int64_t length = (retry == 0 && law == 0) ? first_try[read_i - first_base]
                                              : law_length(law, (uint32_t)(seed + (uint64_t)read_i * (uint64_t)(max_retries + 1) + (uint64_t)retry),
                                                           (double)r, total_len);
            const int64_t got = std::min(length, clen - offset);                     // genome[offset : offset + length]
            if (is_dna) (void)g.below(2);                                           // random.choice("+-")
            if (is_dna && got != length) continue;                                  // read_check: end-of-contig rejection
            if (got < min_read_len) continue;
            int64_t n_count = 0;
            if (n_pos && n_pos[where]) {                                            // read.count("N"): sorted N positions of the contig
                const int64_t* p0 = n_pos[where], *p1 = p0 + n_pos_count[where];
                n_count = std::lower_bound(p0, p1, offset + got) - std::lower_bound(p0, p1, offset);
                if ((double)n_count > 0.1 * (double)length) continue;
            }
            for (int64_t i = 0; i < n_count; ++i) (void)g.below(4);                 // fill_unknown_bases: one choice("ACGT") per N
            if (out_lengths) out_lengths[accepted] = (int32_t)got;
            ++accepted;
            break;
        }
    }
    mt_state[624] = g.idx;
    *out_next_read_i = read_i;
    return accepted;
}

extern "C" int64_t s2s_sampler_replay(uint32_t* mt_state, const int64_t* contig_ends, int32_t n_contigs,
                                      const int64_t* const* n_pos, const int64_t* n_pos_count, int64_t num_seqs, int64_t first_read_i, int64_t r,
                                      uint64_t seed, int64_t total_len, int32_t is_dna, int32_t min_read_len, int32_t max_retries,
                                      int64_t stop_after, int32_t* out_lengths, int64_t* out_next_read_i) {
    return s2s_sampler_replay_law(mt_state, contig_ends, n_contigs, n_pos, n_pos_count, num_seqs, first_read_i, r, seed, total_len, is_dna,
                                  min_read_len, max_retries, stop_after, 0, out_lengths, out_next_read_i);
}

// One read length of law 0 / 1 / 2 (expon / gamma / beta) for a scipy seed: the test hook of the three laws.
extern "C" int64_t s2s_length_law(int32_t law, uint32_t seed, double r, int64_t total_len) {
    if (law < 0 || law > 2 || !(r > 0) || total_len < 1) return S2S_ERR_ARG;
    return law_length(law, seed, r, total_len);
}

// ---- FASTA text -> cleaned sequences (replaces the line loop of utils.read_fasta + process_genome, reference utils.py:290-308,
// 594-597, for plain FASTA files): every rank of a sharded run parses the whole reference before its first kernel.
namespace {
inline bool fa_blank(uint8_t c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }
// Bytes the text-mode line loop treats differently from this byte-level parser: anything >= 0x80 (the loop decodes UTF-8 -- a name
// keeps its characters, an invalid sequence raises -- and U+0085 / U+00A0 are blanks for str.split) and 0x1c-0x1f (blanks for
// str.split / str.strip as well).  A file that holds one is left to the line loop.
inline bool odd_bytes(const uint8_t* data, int64_t n) {
    unsigned any = 0;
    for (int64_t k = 0; k < n; ++k) any |= (unsigned)(data[k] >= 0x80) | (unsigned)((uint8_t)(data[k] - 0x1c) < 4);
    return any != 0;
}
}  // namespace

// Number of records ('>' in column 0, as pysam / the line loop of utils.read_fasta take it); -2 when the first non-empty line
// starts with '@' (FASTQ: s2s_fastq_clean); -3 when a line holds a carriage return that is not part of its line end (a line
// break of its own under Python's universal newlines: left to the caller's line loop; found by tools/fuzz_host.py).
extern "C" int64_t s2s_fasta_count(const uint8_t* data, int64_t n) {
    if (!data || n < 0) return S2S_ERR_ARG;
    int64_t i = 0, recs = 0;
    bool first = true;
    if (odd_bytes(data, n)) return -3;
    const bool has_cr = n > 0 && std::memchr(data, '\r', (size_t)n) != nullptr;      // (Unix files: one pass, no per-line check)
    while (i < n) {
        const uint8_t* nl = static_cast<const uint8_t*>(std::memchr(data + i, '\n', (size_t)(n - i)));
        int64_t end = nl ? nl - data : n;
        const int64_t next = end + 1;
        if (end > i && data[end - 1] == '\r') --end;                                 // "\r\n" is ONE line end; a second CR in front of it is a line of its own
        if (has_cr && end > i && std::memchr(data + i, '\r', (size_t)(end - i))) return -3;   // a lone CR inside (or at the end of) a line
        if (end > i) {
            if (first && data[i] == '@') return -2;
            first = false;
            recs += data[i] == '>';
        }
        i = next;
    }
    return recs;
}

// out receives the records' sequences back to back: line ends removed, every line stripped of blanks at both ends (as the line
// loop does), optionally upper-cased with everything but ACGT mapped to N (map_acgtn = 1: process_genome).  seq_offs[r],
// seq_offs[r+1] delimit record r in out; name_span[2r], name_span[2r+1] delimit its name (the header's first token) in data.
// Lines before the first header are ignored.  Returns the number of records (<= max_records, else -1).
extern "C" int64_t s2s_fasta_clean(const uint8_t* data, int64_t n, int32_t map_acgtn, uint8_t* out, int64_t* seq_offs,
                                   int64_t* name_span, int64_t max_records) {
    if (!data || n < 0 || !out || !seq_offs || !name_span || max_records < 0) return S2S_ERR_ARG;
    uint8_t tab[256];
    for (int c = 0; c < 256; ++c) {
        uint8_t u = (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : (uint8_t)c;
        tab[c] = map_acgtn ? ((u == 'A' || u == 'C' || u == 'G' || u == 'T') ? u : (uint8_t)'N') : (uint8_t)c;
    }
    int64_t i = 0, recs = 0, o = 0;
    while (i < n) {
        const uint8_t* nl = static_cast<const uint8_t*>(std::memchr(data + i, '\n', (size_t)(n - i)));
        int64_t end = nl ? nl - data : n;
        const int64_t next = end + 1;
        while (end > i && data[end - 1] == '\r') --end;
        if (end > i) {
            if (data[i] == '>') {
                if (recs == max_records) return -1;
                seq_offs[recs] = o;
                int64_t s = i + 1;
                while (s < end && fa_blank(data[s])) ++s;
                int64_t e = s;
                while (e < end && !fa_blank(data[e])) ++e;
                name_span[2 * recs] = s;
                name_span[2 * recs + 1] = e;
                ++recs;
            } else if (recs > 0) {
                int64_t a2 = i, b2 = end;
                while (a2 < b2 && fa_blank(data[a2])) ++a2;
                while (b2 > a2 && fa_blank(data[b2 - 1])) --b2;
                if (map_acgtn) {
                    for (int64_t k = a2; k < b2; ++k) out[o + (k - a2)] = tab[data[k]];
                } else {
                    std::memcpy(out + o, data + a2, (size_t)(b2 - a2));
                }
                o += b2 - a2;
            }
        }
        i = next;
    }
    seq_offs[recs] = o;
    return recs;
}

// FASTQ text (four-line records, as the line loop of utils.read_fasta takes them: blank lines are skipped in front of a header only,
// the sequence line is kept as it stands minus its line end, the third line must start with '+', the fourth is skipped) -> the
// sequences back to back in `out` (>= n bytes; map_acgtn as in s2s_fasta_clean), seq_offs [records + 1], name_span [2 * records]
// (the header's first token inside data).  Returns the number of records; -1 when there are more than max_records; -2 for
// anything the line loop must judge itself (no '@' where a header should be, a missing '+', a truncated last record, a lone
// carriage return inside a line -- a line break for Python's universal newlines).  max_records = 0 with out == NULL only counts.
extern "C" int64_t s2s_fastq_clean(const uint8_t* data, int64_t n, int32_t map_acgtn, uint8_t* out, int64_t* seq_offs,
                                   int64_t* name_span, int64_t max_records) {
    if (!data || n < 0 || max_records < 0) return S2S_ERR_ARG;
    const bool count_only = out == nullptr;
    if (!count_only && (!seq_offs || !name_span)) return S2S_ERR_ARG;
    uint8_t tab[256];
    for (int c = 0; c < 256; ++c) {
        uint8_t u = (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : (uint8_t)c;
        tab[c] = map_acgtn ? ((u == 'A' || u == 'C' || u == 'G' || u == 'T') ? u : (uint8_t)'N') : (uint8_t)c;
    }
    int64_t i = 0, recs = 0, o = 0;
    bool bad = false;
    if (odd_bytes(data, n)) return -2;
    auto line = [&](int64_t& b, int64_t& e) {                // next line [b, e) without its line end; false at the end of the data
        if (i >= n) return false;
        const uint8_t* nl = static_cast<const uint8_t*>(std::memchr(data + i, '\n', (size_t)(n - i)));
        e = nl ? nl - data : n;
        b = i;
        i = e + 1;
        if (e > b && data[e - 1] == '\r') --e;                 // one CR belongs to the line end; any other is a line break of its own
        if (e > b && std::memchr(data + b, '\r', (size_t)(e - b))) bad = true;
        return true;
    };
    int64_t b, e;
    while (line(b, e)) {
        if (bad) return -2;
        if (e == b) continue;
        if (data[b] != '@') return -2;
        int64_t s = b + 1;
        while (s < e && fa_blank(data[s])) ++s;
        int64_t t = s;
        while (t < e && !fa_blank(data[t])) ++t;
        int64_t sb, se, pb, pe, qb, qe;
        if (!line(sb, se) || !line(pb, pe) || !line(qb, qe) || bad) return -2;
        if (pe == pb || data[pb] != '+') return -2;
        if (!count_only) {
            if (recs == max_records) return -1;
            seq_offs[recs] = o;
            name_span[2 * recs] = s;
            name_span[2 * recs + 1] = t;
            if (map_acgtn) {
                for (int64_t k = sb; k < se; ++k) out[o + (k - sb)] = tab[data[k]];
            } else {
                std::memcpy(out + o, data + sb, (size_t)(se - sb));
            }
            o += se - sb;
        }
        ++recs;
    }
    if (!count_only) seq_offs[recs] = o;
    return recs;
}

// ---- rank-shard merge (`predict --gpus N`, `merge-shards`): the reference writes ONE file (inference.py:65-79), a sharded run
// writes one per rank, and joining them must not cost more than the ranks' parallel phase did.  Both helpers work on file
// descriptors; nothing passes through the interpreter or (copy_file_range) through user space.
#include <cerrno>
#include <climits>
#include <fcntl.h>
#include <map>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/vfs.h>
#include <unistd.h>

namespace {

int64_t copy_one(int src, int64_t src_off, int dst, int64_t dst_off, int64_t len, std::vector<uint8_t>& bounce, bool& in_kernel) {
    while (len > 0) {
        if (in_kernel) {
            off_t so = (off_t)src_off, d_o = (off_t)dst_off;
            const ssize_t r = copy_file_range(src, &so, dst, &d_o, (size_t)std::min<int64_t>(len, 1 << 30), 0);
            if (r > 0) { src_off += r; dst_off += r; len -= r; continue; }
            if (r == 0) return -EIO;                                 // source shorter than the caller said
            if (errno == EINTR) continue;
            if (errno != EXDEV && errno != ENOSYS && errno != EINVAL && errno != EOPNOTSUPP && errno != EPERM) return -errno;
            in_kernel = false;                                      // another file system / an old kernel: through a bounce buffer
        }
        if (bounce.empty()) bounce.resize(8 << 20);
        const ssize_t r = pread(src, bounce.data(), (size_t)std::min<int64_t>(len, (int64_t)bounce.size()), (off_t)src_off);
        if (r == 0) return -EIO;
        if (r < 0) { if (errno == EINTR) continue; return -errno; }
        for (ssize_t done = 0; done < r;) {
            const ssize_t w = pwrite(dst, bounce.data() + done, (size_t)(r - done), (off_t)(dst_off + done));
            if (w < 0) { if (errno == EINTR) continue; return -errno; }
            done += w;
        }
        src_off += r; dst_off += r; len -= r;
    }
    return 0;
}

}  // namespace

// Two engines.
//  * descriptors (engine 0): copy_file_range, in the kernel.  One destination file has ONE fast writer this way: buffered writes
//    take its inode lock, so 2-8 threads into the SAME file are slower than one (3.2-4.1 against 6.5 GB/s on the MI355X box's tmpfs,
//    profiles/r05/fs_write_probe_shm.txt), and what a lone writer spends its time on is allocating the destination's pages.
//  * preallocate + mapped fill (engine 1, tmpfs destinations): posix_fallocate() the destination ranges first -- allocation WITHOUT
//    data runs at 18.6 GB/s there -- then `threads` threads each map-populate a 16-MiB piece of the destination (MADV_POPULATE_WRITE:
//    one call instead of a trap per page) and pread() the source straight into it: the kernel copies from the source's page cache
//    into pages that already exist -- no inode lock, no allocation.  8.0 GB/s against 5.7 GB/s for engine 0 on the same box
//    (profiles/r05/merge_bench_shm.txt; memcpy from a source mapping instead of pread: 6.1; without the populate call: 6.7; with the
//    next range's fallocate running beside the copy: 4.0 -- it slows the faults down).  The space is reserved before the first store,
//    so a full file system is an error code (the fallocate fails, engine 0 takes over and reports ENOSPC), never a SIGBUS.  On a
//    disk file system's page cache the same engine LOSES (3.7-4.5 against 7-10 GB/s: fallocate is a block allocation there), so
//    engine 1 is only taken for tmpfs.  (Round 5's first mapped engine, commit 1b15c95, stored into pages that did NOT exist yet and
//    lost everywhere: the page allocation under the faults was the bound.)
namespace {
struct Span { int64_t lo = INT64_MAX, hi = 0; uint8_t* base = nullptr; size_t bytes = 0; };      // the hull of a file's ranges, and its mapping

bool map_spans(std::map<int, Span>& spans) {
    for (auto& kv : spans) {
        Span& sp = kv.second;
        sp.lo &= ~(int64_t)4095;                                     // (mmap wants a page-aligned file offset)
        sp.bytes = (size_t)(sp.hi - sp.lo);
        void* m = mmap(nullptr, sp.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, kv.first, (off_t)sp.lo);
        if (m == MAP_FAILED) { sp.base = nullptr; return false; }
        sp.base = static_cast<uint8_t*>(m);
    }
    return true;
}
void unmap_spans(std::map<int, Span>& spans) {
    for (auto& kv : spans)
        if (kv.second.base) munmap(kv.second.base, kv.second.bytes);
}
}  // namespace

extern "C" int64_t s2s_copy_ranges(int32_t n, const int32_t* src_fd, const int64_t* src_off, const int32_t* dst_fd,
                                   const int64_t* dst_off, const int64_t* len, int32_t threads, int32_t engine) {
    if (n < 0 || threads < 1 || engine < 0 || engine > 2 || (n > 0 && (!src_fd || !src_off || !dst_fd || !dst_off || !len))) return S2S_ERR_ARG;
    // pieces of <= 16 MiB, so that one long range does not leave the other threads idle
    struct Piece { int src, dst; int64_t so, d_o, len; };
    std::vector<Piece> pieces;
    const int64_t cut = 16ll << 20;
    int64_t total = 0;
    std::map<int, Span> dsts;
    std::map<int, int64_t> src_end;
    for (int i = 0; i < n; ++i) {
        if (len[i] < 0 || src_off[i] < 0 || dst_off[i] < 0) return S2S_ERR_ARG;
        if (len[i] == 0) continue;
        for (int64_t o = 0; o < len[i]; o += cut)
            pieces.push_back({src_fd[i], dst_fd[i], src_off[i] + o, dst_off[i] + o, std::min(cut, len[i] - o)});
        total += len[i];
        src_end[src_fd[i]] = std::max(src_end[src_fd[i]], src_off[i] + len[i]);
        Span& b = dsts[dst_fd[i]];
        b.lo = std::min(b.lo, dst_off[i]); b.hi = std::max(b.hi, dst_off[i] + len[i]);
    }
    if (pieces.empty()) return 0;
    bool mapped = engine >= 1;
    if (mapped)
        for (auto& kv : dsts) {
            struct stat st;
            struct statfs fs;
            if (src_end.count(kv.first) || fstat(kv.first, &st) != 0 || !S_ISREG(st.st_mode)) { mapped = false; break; }   // (a file copied onto itself: descriptors)
            if (engine == 1 && (fstatfs(kv.first, &fs) != 0 || fs.f_type != 0x01021994)) { mapped = false; break; }       // TMPFS_MAGIC (engine 2: any file system, A/B)
            if (posix_fallocate(kv.first, (off_t)kv.second.lo, (off_t)(kv.second.hi - kv.second.lo)) != 0) { mapped = false; break; }
        }
    if (mapped) {
        mapped = map_spans(dsts);
        if (!mapped) unmap_spans(dsts);
    }
    // every piece's destination mapping is looked up HERE, on one thread: the workers only read the vector (std::map::operator[] is
    // a non-const member, and concurrent calls of it are a data race on paper even when every key exists)
    std::vector<const Span*> span_of(pieces.size(), nullptr);
    if (mapped)
        for (size_t i = 0; i < pieces.size(); ++i) span_of[i] = &dsts.find(pieces[i].dst)->second;
    std::atomic<size_t> next{0};
    std::atomic<int64_t> err{0};
    auto work = [&] {
        std::vector<uint8_t> bounce;
        bool in_kernel = true;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= pieces.size() || err.load()) return;
            const Piece& p = pieces[i];
            if (mapped) {
                const Span& b = *span_of[i];
                uint8_t* to = b.base + (p.d_o - b.lo);
#ifdef MADV_POPULATE_WRITE
                const uintptr_t lo = (uintptr_t)to & ~(uintptr_t)4095, hi = ((uintptr_t)to + (uintptr_t)p.len + 4095) & ~(uintptr_t)4095;
                (void)madvise((void*)lo, hi - lo, MADV_POPULATE_WRITE);        // (Linux 5.14+; refused elsewhere: plain faults)
#endif
                int64_t done = 0;
                while (done < p.len) {                                       // a source shorter than its range ends the call with -EIO, no fault
                    const ssize_t r = pread(p.src, to + done, (size_t)(p.len - done), (off_t)(p.so + done));
                    if (r <= 0) { if (r < 0 && errno == EINTR) continue; err = r < 0 ? -errno : -EIO; break; }
                    done += r;
                }
                continue;
            }
            const int64_t r = copy_one(p.src, p.so, p.dst, p.d_o, p.len, bounce, in_kernel);
            if (r < 0) err = r;
        }
    };
    const int workers = (int)std::min<size_t>((size_t)(mapped ? threads : 1), std::max<size_t>(pieces.size(), 1));   // (descriptors: one writer)
    std::vector<std::thread> pool;
    for (int w = 1; w < workers; ++w) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    if (mapped) unmap_spans(dsts);
    return err.load() < 0 ? err.load() : total;
}

extern "C" int64_t s2s_blow5_scan(int32_t fd, int64_t begin, int64_t end) {
    if (fd < 0 || begin < 0 || end < begin) return S2S_ERR_ARG;
    int64_t pos = begin, n = 0;
    while (pos < end) {
        uint64_t size;
        if (end - pos < 8 || pread(fd, &size, 8, (off_t)pos) != 8) return -2;
        if (size > (uint64_t)(end - pos - 8)) return -2;                // a record runs past the end-of-file marker: truncated shard
        pos += 8 + (int64_t)size;
        ++n;
    }
    return n;
}

// The same walk over a file that is still being written (the live join tails the rank files while the ranks run): complete records
// only -- a size prefix or a body that reaches past `limit` (the file's size at the moment of the call) ends the walk without an
// error -- and at most max_records of them.  *out_end = the byte offset behind the last complete record.
extern "C" int64_t s2s_blow5_scan_upto(int32_t fd, int64_t begin, int64_t limit, int64_t max_records, int64_t* out_end) {
    if (fd < 0 || begin < 0 || !out_end) return S2S_ERR_ARG;
    int64_t pos = begin, n = 0;
    while (n < max_records && limit - pos >= 8) {
        uint64_t size;
        if (pread(fd, &size, 8, (off_t)pos) != 8) break;
        if (size > (uint64_t)(limit - pos - 8)) break;                 // body not (yet) whole: the writer is in the middle of it, or this is the end marker
        pos += 8 + (int64_t)size;
        ++n;
    }
    *out_end = pos;
    return n;
}

