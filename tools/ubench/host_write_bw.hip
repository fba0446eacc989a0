// micro-benchmark (GPU box, host side): how fast N host threads write a 212 MB result block, depending on where the block lies --
// hipHostMalloc'ed (pinned: on the NUMA node next to the GPU) or ordinary memory first touched by the writing threads -- and on the
// kind of store (ordinary / movnti).  Decides where sdf_mesh_emit_host_workers' output should live.
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/mman.h>
#include <sched.h>
#include <unistd.h>
#include <sys/syscall.h>
#include <fstream>
#include <string>

static std::vector<int> node_cpus(int node) {   // /sys/devices/system/node/nodeN/cpulist: "0-63,128-191"
    std::vector<int> v;
    std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
    std::string s; std::getline(f, s);
    size_t i = 0;
    while (i < s.size()) {
        size_t j = s.find(',', i); if (j == std::string::npos) j = s.size();
        std::string part = s.substr(i, j - i);
        size_t d = part.find('-');
        int a = std::stoi(part.substr(0, d)), b = d == std::string::npos ? a : std::stoi(part.substr(d + 1));
        for (int c = a; c <= b; c++) v.push_back(c);
        i = j + 1;
    }
    return v;
}
static int g_pin_node = -1, g_spread = 0;   // g_spread: thread i -> cpu (i * 8) % n + (i * 8) / n of the node's list (one CCD after the other gets a thread)
static double run(char *p, size_t n, int threads, bool nt, int reps) {
    std::vector<int> cpus = g_pin_node >= 0 ? node_cpus(g_pin_node) : std::vector<int>();
    double best = 1e30;
    for (int r = 0; r < reps; r++) {
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int i = 0; i < threads; i++)
            th.emplace_back([=] {
                if (!cpus.empty()) { const size_t nc = cpus.size() > 64 ? 64 : cpus.size(); const size_t k = g_spread ? ((size_t)i * 8) % nc + ((size_t)i * 8) / nc : (size_t)i % cpus.size(); cpu_set_t cs; CPU_ZERO(&cs); CPU_SET(cpus[k % cpus.size()], &cs); sched_setaffinity(0, sizeof cs, &cs); }
                // interleaved blocks of 576 KB like the expansion's (8192 triangles x 72 B)
                const size_t blk = 8192 * 72, nblk = (n + blk - 1) / blk;
                for (size_t b = i; b < nblk; b += threads) {
                    long long *q = (long long *)(p + b * blk);
                    const size_t cnt = (b * blk + blk <= n ? blk : n - b * blk) / 8;
                    if (nt) { for (size_t k = 0; k < cnt; k++) _mm_stream_si64(q + k, (long long)k); _mm_sfence(); }
                    else for (size_t k = 0; k < cnt; k++) q[k] = (long long)k;
                }
            });
        for (auto &t : th) t.join();
        best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    return best;
}

int main() {
    const size_t n = 2945152ull * 72;
    char *pinned = nullptr;
    (void)hipHostMalloc((void **)&pinned, n, hipHostMallocDefault);
    char *plain = (char *)aligned_alloc(1 << 21, (n + (1 << 21) - 1) & ~(size_t)((1 << 21) - 1));
    char *huge = (char *)mmap(nullptr, (n + (1 << 21) - 1) & ~(size_t)((1 << 21) - 1), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(huge, n, MADV_HUGEPAGE);
    char *numa_pinned = nullptr;
    (void)hipHostMalloc((void **)&numa_pinned, n, hipHostMallocNumaUser);
    printf("hardware threads %u\n", std::thread::hardware_concurrency());
    for (int threads : {1, 8, 16, 32, 64}) {
        // first touch of the plain blocks by these threads (timed: what a fresh ndarray costs)
        double t_first = 0;
        if (threads == 32) { t_first = run(plain, n, threads, false, 1); printf("first touch of fresh ordinary memory, 32 threads: %.2f ms\n", t_first); t_first = run(huge, n, threads, false, 1); printf("first touch, MADV_HUGEPAGE, 32 threads: %.2f ms\n", t_first); }
        printf("%2d threads: pinned %.2f / nt %.2f ms | ordinary %.2f / nt %.2f | hugepage-advised %.2f / nt %.2f | pinned NumaUser %.2f / nt %.2f\n", threads,
               run(pinned, n, threads, false, 4), run(pinned, n, threads, true, 4), run(plain, n, threads, false, 4), run(plain, n, threads, true, 4),
               run(huge, n, threads, false, 4), run(huge, n, threads, true, 4), numa_pinned ? run(numa_pinned, n, threads, false, 4) : -1.0, numa_pinned ? run(numa_pinned, n, threads, true, 4) : -1.0);
    }
    // which node does the pinned block lie on, and what do threads bound to either node achieve on it?
    { int node = -1; long rc = syscall(SYS_get_mempolicy, &node, nullptr, 0, pinned, 3 /* MPOL_F_NODE | MPOL_F_ADDR */); printf("get_mempolicy(pinned): rc %ld node %d\n", rc, node); }
    for (int node : {0, 1}) {
        g_pin_node = node;
        for (int threads : {8, 16, 32})
            printf("threads bound to node %d, %2d threads: pinned %.2f / nt %.2f ms | ordinary %.2f / nt %.2f\n", node, threads, run(pinned, n, threads, false, 4), run(pinned, n, threads, true, 4),
                   run(plain, n, threads, false, 4), run(plain, n, threads, true, 4));
    }
    g_spread = 1;
    for (int node : {0, 1}) {
        g_pin_node = node;
        for (int threads : {8, 16, 32})
            printf("threads bound to node %d ONE PER CCD FIRST, %2d threads: pinned %.2f / nt %.2f ms | ordinary %.2f / nt %.2f\n", node, threads, run(pinned, n, threads, false, 4), run(pinned, n, threads, true, 4),
                   run(plain, n, threads, false, 4), run(plain, n, threads, true, 4));
    }
    g_spread = 0;
    g_pin_node = -1;
    // the D2H side: 47 MB of records into pinned memory while nothing else runs
    char *d = nullptr; (void)hipMalloc((void **)&d, 48 << 20);
    for (int r = 0; r < 3; r++) {
        auto t0 = std::chrono::steady_clock::now();
        (void)hipMemcpy(pinned, d, 47122432, hipMemcpyDeviceToHost);
        printf("D2H 47 MB into pinned: %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    return 0;
}
