// sdf_expand_host.h -- k_expand on the HOST: a slab's 16-byte triangle records -> the ordered float64 soup, on `workers` threads.
//
// What a drop-in caller of `f.generate()` waits for is the (3T, 3) float64 ndarray (reference sdf/core.py:58-60, 131-141), and until
// round 5 most of that wait was the copy of the 72-byte triangles over PCIe (212 MB at 512^3: 3.7 of 4.4 ms).  The soup is
// `float64(local float32) * scale + offset` of the batch's marching-cubes output, and the exchange slab (sdf_slab.h) already holds
// that output as 16-byte records + one transform per work item: 47 MB.  So the records travel, and the host does what k_expand does
// on the device -- the same operation on the same operands, hence the same bits -- with the threads the reference's `workers=`
// argument names (sdf/core.py:87, 131), block by block while the later records are still on the link.
//
// Pure host C++ (no HIP): tests/test_slab_host.py builds it for the CPU and holds it to the NumPy restatement (sdf_amd/slabcodec.py).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "sdf_slab.h"

namespace sdfhost {

struct ExpandJob {
    // the slab's parts, on the host (pinned staging): inclusive triangle prefix per work item (flag bits above bit 61), transform per
    // work item (offset[3], scale[3]), the records, the raw area
    const unsigned long long *prefix = nullptr;
    const double *xf = nullptr;
    const Tri16 *recs = nullptr;
    const float *raw = nullptr;
    long long raw_cap = 0, n_items = 0, n_tris = 0;
    double *out = nullptr;                 // 9 doubles per triangle
    long long block = 8192;                // triangles per unit of work
    std::atomic<long long> avail{0};       // records [0, avail) have arrived (published by the thread that watches the copies)
    std::atomic<long long> next{0};        // the next block to take
    std::atomic<long long> done{0};        // blocks finished
    std::atomic<int> abort{0};             // the copies failed: leave
    long long nblocks() const { return (n_tris + block - 1) / block; }
    // diagnostics (SDF_REC_TRACE): per block, when its records were there for the thread that took it and when it was written, in
    // microseconds since `t_origin`
    float *trace = nullptr;
    std::chrono::steady_clock::time_point t_origin;
};

static inline void cpu_relax() {
#if defined(__x86_64__)
    _mm_pause();
#endif
}
// The soup is written once, front to back, and read by somebody else much later: stores that go around the caches (movnti, eight
// bytes each; consecutive ones fill a write-combining buffer, i.e. whole lines) spare the memory system the read of every line before
// it is overwritten -- with 8 and more threads the expansion is bound by the memory the ndarray lies in (r06k: 32 threads write the
// 212 MB of the 512^3 job no faster than 12, ~ 85 GB/s, with ordinary stores).
static inline void store_soup(double *p, double v) {
#if defined(__x86_64__)
    long long b;
    std::memcpy(&b, &v, 8);
    _mm_stream_si64(reinterpret_cast<long long *>(p), b);
#else
    *p = v;
#endif
}
static inline void store_fence() {
#if defined(__x86_64__)
    _mm_sfence();
#endif
}

// triangles [t0, t1) of the job: `out[9 t + e] = double(local[e]) * scale[e % 3] + offset[e % 3]` (k_expand, sdf_plain.hip)
static inline void expand_range(const ExpandJob &j, long long t0, long long t1) {
    const unsigned long long VAL = (1ull << 62) - 1ull;
    // the work item of t0: the smallest i whose inclusive prefix exceeds t0
    long long lo = 0, hi = j.n_items - 1;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if ((long long)(j.prefix[mid] & VAL) > t0) hi = mid; else lo = mid + 1;
    }
    long long item = lo;
    long long t = t0;
    while (t < t1) {
        while (item + 1 < j.n_items && (long long)(j.prefix[item] & VAL) <= t) item++;     // (items without triangles are passed over)
        const long long end = std::min<long long>(t1, item + 1 < j.n_items ? (long long)(j.prefix[item] & VAL) : t1);
        const double *xf = j.xf + 6 * item;
        const double of[3] = {xf[0], xf[1], xf[2]}, sc[3] = {xf[3], xf[4], xf[5]};
        // a lattice coordinate c (0 .. 64) of axis a lands on double(float(c)) * scale + offset: a table per axis instead of two
        // conversions, a product and a sum per coordinate (six of a triangle's nine coordinates are lattice coordinates)
        double lat[3][66];
        const long long span = end - t;
        const int nl = span >= 24 ? 66 : 0;
        for (int a = 0; a < 3 && nl; a++)
            for (int c = 0; c < nl; c++) lat[a][c] = (double)(float)c * sc[a] + of[a];
        for (; t < end; t++) {
            const Tri16 r = j.recs[t];
            double *o = j.out + 9 * t;
            if (r.code & TRI16_RAW || !nl) {
                float f9[9];
                if (r.code & TRI16_RAW) {
                    unsigned ri;
                    std::memcpy(&ri, &r.f[0], 4);
                    const float *src = j.raw + ((long long)ri < j.raw_cap ? (long long)ri : 0ll) * 9;
                    for (int q = 0; q < 9; q++) f9[q] = src[q];
                } else {
                    slab_decode16(r, f9);
                }
                for (int q = 0; q < 9; q++) store_soup(o + q, (double)f9[q] * sc[q % 3] + of[q % 3]);
                continue;
            }
            const unsigned c0 = r.code & 63u, c1 = (r.code >> 6) & 63u, c2 = (r.code >> 12) & 63u;
            for (int k = 0; k < 3; k++) {
                const unsigned v = (r.code >> (18 + 4 * k)) & 15u, frac = v & 3u, b1 = (v >> 2) & 1u, b2 = (v >> 3) & 1u;
                // slab_decode16: the float sits on axis `frac`; the other two axes ascending take the offsets b1, b2
                const double fv = (double)r.f[k] * sc[frac == 3u ? 0 : frac] + of[frac == 3u ? 0 : frac];
                const double x = lat[0][c0 + b1];                               // (axis 0 is the first "other" axis unless it holds the float)
                const double y = lat[1][c1 + (frac == 0u ? b1 : b2)];           // (axis 1: first other axis when the float is on 0, else second)
                const double z = lat[2][c2 + b2];                               // (axis 2 is the second other axis unless it holds the float)
                store_soup(o + 3 * k, frac == 0u ? fv : x);
                store_soup(o + 3 * k + 1, frac == 1u ? fv : y);
                store_soup(o + 3 * k + 2, frac == 2u ? fv : z);
            }
        }
    }
    store_fence();      // (the stores above are weakly ordered: they are globally visible before the block is reported done)
}

// one participant: takes blocks until none is left; a block whose records have not arrived yet is waited for
static inline void expand_work(ExpandJob &j) {
    const long long nb = j.nblocks();
    for (;;) {
        const long long b = j.next.fetch_add(1, std::memory_order_relaxed);
        if (b >= nb) return;
        const long long t0 = b * j.block, t1 = std::min(j.n_tris, t0 + j.block);
        unsigned spins = 0;
        while (j.avail.load(std::memory_order_acquire) < t1) {
            if (j.abort.load(std::memory_order_relaxed)) { j.done.fetch_add(1, std::memory_order_release); return; }
            if (++spins < 4096) cpu_relax(); else std::this_thread::yield();
        }
        if (j.trace) j.trace[2 * b] = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - j.t_origin).count();
        expand_range(j, t0, t1);
        if (j.trace) j.trace[2 * b + 1] = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - j.t_origin).count();
        j.done.fetch_add(1, std::memory_order_release);
    }
}

// A small pool of threads that sleep between jobs (a job is a millisecond: creating threads per call would cost as much).
class Pool {
  public:
    static Pool &get() { static Pool p; return p; }
    // `n` helpers start on `j`; the caller publishes j.avail, may take part itself (expand_work) and ends with wait(j)
    void start(ExpandJob &j, int n) {
        std::unique_lock<std::mutex> lk(mu_);
        while ((int)threads_.size() < n) { const int id = (int)threads_.size(); threads_.emplace_back([this, id] { loop(id); }); }
        job_ = &j; want_ = n; running_ = n; gen_++;
        lk.unlock();
        cv_.notify_all();
    }
    void wait(ExpandJob &j) {
        const long long nb = j.nblocks();
        unsigned spins = 0;
        while (j.done.load(std::memory_order_acquire) < nb && !j.abort.load(std::memory_order_relaxed)) { if (++spins < 4096) cpu_relax(); else std::this_thread::yield(); }
        std::unique_lock<std::mutex> lk(mu_);       // (the helpers have let go of the job before it goes out of scope)
        idle_.wait(lk, [this] { return running_ == 0; });
        job_ = nullptr;
    }
    ~Pool() {
        { std::lock_guard<std::mutex> lk(mu_); quit_ = true; gen_++; }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }

  private:
    void loop(int id) {
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [&] { return gen_ != seen; });
            seen = gen_;
            if (quit_) return;
            if (id >= want_ || !job_) continue;
            ExpandJob *j = job_;
            lk.unlock();
            expand_work(*j);
            lk.lock();
            if (--running_ == 0) idle_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, idle_;
    std::vector<std::thread> threads_;
    ExpandJob *job_ = nullptr;
    int want_ = 0, running_ = 0;
    unsigned long long gen_ = 0;
    bool quit_ = false;
};

}  // namespace sdfhost
