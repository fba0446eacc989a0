// micro-benchmark (host side of a GPU box): the threaded record expansion (csrc/sdf_expand_host.h) on records that are all there, 1 .. 64 threads
#include <chrono>
#include <cstdio>
#include <vector>
#include <random>
#include "sdf_expand_host.h"
int main(int argc, char **argv) {
    const long long n_items = 1744, per = 1690, n = n_items * per;
    std::vector<unsigned long long> prefix(n_items);
    std::vector<double> xf(n_items * 6, 0.01);
    for (long long i = 0; i < n_items; i++) prefix[i] = (i + 1) * per | (2ull << 62);
    std::vector<Tri16> rec(n);
    std::mt19937 g(1);
    for (long long t = 0; t < n; t++) { rec[t].code = (g() & 0x3FFFFFFFu); for (int k = 0; k < 3; k++) { unsigned v = (rec[t].code >> (18 + 4 * k)) & 3u; if (v == 3u) rec[t].code &= ~(1u << (18 + 4 * k)); rec[t].f[k] = (float)(g() & 0xffff) / 65536.f; } rec[t].code &= ~((32u) | (32u << 6) | (32u << 12)); }
    std::vector<float> raw(9);
    double *out = (double *)aligned_alloc(4096, n * 72);
    memset(out, 0, n * 72);
    for (int threads : {1, 2, 4, 8, 16, 32, 48, 64}) {
        for (int rep = 0; rep < 5; rep++) {
            sdfhost::ExpandJob j;
            j.prefix = prefix.data(); j.xf = xf.data(); j.recs = rec.data(); j.raw = raw.data(); j.raw_cap = 1; j.n_items = n_items; j.n_tris = n; j.out = out;
            auto t0 = std::chrono::steady_clock::now();
            sdfhost::Pool &p = sdfhost::Pool::get();
            p.start(j, threads - 1);
            j.avail.store(n);
            sdfhost::expand_work(j);
            p.wait(j);
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (rep >= 3) printf("%d threads: %.2f ms  (%.1f ns/tri/thread, %.1f GB/s written)\n", threads, ms, ms * 1e6 * threads / n, n * 72 / ms * 1e-6);
        }
    }
}
