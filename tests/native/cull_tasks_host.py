"""Host build of k_cull's body, THE DEVICE SOURCE ITSELF: `cull_tasks` and `cull_sample` are cut out of
sdf_amd/csrc/sdf_device.h as text, the handful of device idioms in them is replaced by host stand-ins (threadIdx ->
a thread-local index, __syncthreads -> a barrier of std::threads, the DPP block scans -> scans through a shared
array), and a workgroup is played by BLOCK host threads.  Test infrastructure only (tests/test_cull_host.py): what the
interval levels decide, which units are listed and how a lane finds its sample can be checked without a GPU."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

PRELUDE = r'''
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>
static thread_local int g_simt_tid = 0;
static int sdf_host_simt_tid() { return g_simt_tid; }
#define SDF_HOST_SIMT 1
#include "sdf_interval.h"
using namespace sdfk;
using std::min; using std::max;
#ifndef SDF_UNROLL
#define SDF_UNROLL
#endif
namespace {
struct Barrier {           // (sense-reversing; the threads of one emulated workgroup)
    std::atomic<int> count{0}, generation{0};
    int n = 1;
    void wait() {
        const int g = generation.load();
        if (count.fetch_add(1) == n - 1) { count.store(0); generation.fetch_add(1); }
        else while (generation.load() == g) std::this_thread::yield();
    }
};
Barrier g_bar;
int g_scan_tmp[1024];
#define HOST_SYNC() g_bar.wait()
#define HOST_TID g_simt_tid
template <int BLOCK> int host_scan(int v, int *wave_sums, int &total) {
    g_scan_tmp[HOST_TID] = v;
    HOST_SYNC();
    int base = 0, tot = 0;
    for (int i = 0; i < BLOCK; i++) { if (i < HOST_TID) base += g_scan_tmp[i]; tot += g_scan_tmp[i]; }
    total = tot;
    HOST_SYNC();
    return base;
}
template <int BLOCK> int host_count(bool p, int *wave_sums, int &total) { return host_scan<BLOCK>(p ? 1 : 0, wave_sums, total); }
'''

POSTLUDE = r'''
template <int BLOCK> static int run_block(const uint32_t *code, const double *consts, int n_instr, int n_p, int n_d, int lx, int ly, int lz,
                                          const double *axes99, unsigned char *record, int *ntl_out, int levels) {
    const int per_thread = (6 * n_p + 2 * n_d) * 8;
    std::vector<unsigned char> scratch(CULL_SCRATCH + 64, 0xA5);            // (garbage, like LDS)
    std::vector<double> ia_state((size_t)BLOCK * (6 * n_p + 2 * n_d) + 8);
    std::vector<double> axes(axes99, axes99 + 99);
    int wave_sums[16];
    std::vector<int> ret(BLOCK, -12345);
    g_bar.n = BLOCK;
    g_bar.count.store(0);
    std::vector<std::thread> th;
    for (int t = 0; t < BLOCK; t++)
        th.emplace_back([&, t] {
            g_simt_tid = t;
            ret[t] = cull_tasks<BLOCK, true, true>(code, consts, n_instr, lx, ly, lz, axes.data(), ia_state.data(), BLOCK * per_thread,
                                                   scratch.data(), wave_sums, n_p, n_d, nullptr, levels);
        });
    for (auto &x : th) x.join();
    for (int t = 1; t < BLOCK; t++) if (ret[t] != ret[0]) return 2;          // (the return value is workgroup-uniform)
    *ntl_out = ret[0];
    memcpy(record, scratch.data(), CULL_RECORD);
    return 0;
}
extern "C" int cull_record_bytes() { return CULL_RECORD; }
extern "C" int cull_layout(int *out) { out[0] = CULL_ULIST; out[1] = CULL_SSTATE; out[2] = CULL_UNIT_CAP; out[3] = CULL_COLINFO; return 0; }
extern "C" int cull_host(int block, const uint32_t *code, const double *consts, int n_instr, int n_p, int n_d, int lx, int ly, int lz,
                         const double *axes99, unsigned char *record, int *ntl_out, int levels) {
    if (n_p < 1) n_p = 1;
    if (n_d < 1) n_d = 1;
    switch (block) {
    case 64: return run_block<64>(code, consts, n_instr, n_p, n_d, lx, ly, lz, axes99, record, ntl_out, levels);
    case 128: return run_block<128>(code, consts, n_instr, n_p, n_d, lx, ly, lz, axes99, record, ntl_out, levels);
    case 256: return run_block<256>(code, consts, n_instr, n_p, n_d, lx, ly, lz, axes99, record, ntl_out, levels);
    }
    return 1;
}
// the sample of (task, lane) as k_mesh finds it: out[0..2] = ix, iy, iz; returns 1 if the lane has a sample
extern "C" int cull_sample_host(const unsigned short *units, int task, int lane, int lx, int ly, int lz, int *out) {
    return cull_sample(units, task, lane, lx, ly, lz, out[0], out[1], out[2]) ? 1 : 0;
}
}  // namespace
'''


def generate(path):
    src = open(os.path.join(ROOT, 'sdf_amd', 'csrc', 'sdf_device.h')).read()
    i = src.index('enum { CULL_UNIT_CAP')
    j = src.index('// Stores of the soup.  Marking them non-temporal')
    body = src[i:j]
    for a, b in (('__device__ __forceinline__', 'static inline'), ('threadIdx.x', 'HOST_TID'), ('__syncthreads()', 'HOST_SYNC()'),
                 ('block_exclusive_count<BLOCK>', 'host_count<BLOCK>'), ('block_exclusive_scan<BLOCK>', 'host_scan<BLOCK>'),
                 ('__popc(', '__builtin_popcount('), ('__ffs(', '__builtin_ffs('), ('clock64()', '0LL')):
        body = body.replace(a, b)
    assert not re.search(r'__builtin_amdgcn|__shfl|__ballot', body), 'a device idiom the host stand-ins do not cover'
    # k_mesh's loop that sets the sign bits of the samples of decided sub-groups, as a host function over all `tid`
    a, b = src.index('// <cull-sign-fill>'), src.index('// </cull-sign-fill>')
    fill = src[src.index('\n', a) + 1:b]
    fill_fn = '''
static inline int fast_div(int i, float inv_d) { return (int)(((float)i + 0.5f) * inv_d); }
static inline void atomicOr(unsigned long long *p, unsigned long long v) { *p |= v; }
extern "C" int cull_sign_fill_host(const unsigned *sstate, int lx, int ly, int lz, unsigned long long *bits) {
    constexpr int BLOCK = 1024;
    const int c0 = lx - 1, c1 = ly - 1, c2 = lz - 1;
    for (int tid = 0; tid < BLOCK; tid++) {
''' + fill + '''
    }
    return 0;
}
'''
    # k_mesh's sampling loop over the listed units, with a stand-in for the tape interpreter (the value of a sample encodes
    # its coordinates) and all 1024 threads of a workgroup one after the other
    a, b = src.index('// <sample-loop>'), src.index('// </sample-loop>')
    loop = src[src.index('\n', a) + 1:b].replace('run_tape<T, FULL, NP, ND, NS>(wcode, consts, px, py, pz)', 'host_field(px, py, pz)')
    assert 'host_field' in loop
    loop_fn = '''
extern "C" int cull_sample_loop_host(const unsigned char *record, int lx, int ly, int lz, const double *axes, float *vol, unsigned long long *bits,
                                     int sparse_tile, float *smp) {
    typedef double T;
    const bool sparse = sparse_tile != 0;
    constexpr int NS = 3, BLOCK = 1024, NWAVE = BLOCK / 64;
    struct V { T v[NS]; };
    auto host_field = [](const V &x, const V &y, const V &z) { V r; for (int k = 0; k < NS; k++) r.v[k] = x.v[k] + 64.0 * y.v[k] + 4096.0 * z.v[k] - 70000.0; return r; };
    struct { int lz; bool sample(int, int, int &, int &, int &) const { return false; } } tt{lz};
    const unsigned *list = reinterpret_cast<const unsigned *>(record);
    const unsigned short *units = reinterpret_cast<const unsigned short *>(reinterpret_cast<const unsigned char *>(list) + CULL_ULIST);   // (as in k_mesh)
    const int n = (int)reinterpret_cast<const unsigned short *>(list)[0];
    if (n == 0xFFFF) return 1;
    const bool culled = true;
    const int ntl = (n + 7) >> 3, lyz = ly * lz;
    for (int tid = 0; tid < BLOCK; tid++) {
        const int wave = tid >> 6, lane = tid & 63;
''' + loop + '''
    }
    return 0;
}
'''
    # k_mesh's view of a tile (dense / sparse): the struct as it stands in the device source
    a, b = src.index('struct TileView {'), src.index('__device__ __forceinline__ void mc_vertex_view(')
    view = src[a:b].replace('__device__ __forceinline__', 'inline').replace('__popc(', '__builtin_popcount(').replace('#pragma unroll', '')
    view_fn = view + '''
extern "C" float tile_at_host(const float *smp, const unsigned *colinfo, int ix, int iy, int iz) {
    const TileView vw{smp, colinfo, 0, 0, true};
    return vw.at(ix, iy, iz);
}
'''
    # (`extern "C"` functions inside an anonymous namespace keep C linkage)
    open(path, 'w').write(PRELUDE + body + fill_fn + loop_fn + view_fn + POSTLUDE)


def build(workdir):
    import ctypes
    gen = os.path.join(workdir, 'cull_tasks_host_gen.hip')
    so = os.path.join(workdir, 'libcull_host.so')
    generate(gen)
    subprocess.check_call([HIPCC, '--offload-host-only', '-O1', '-std=c++17', '-ffp-contract=off', '-w', '-fPIC', '-shared', '-pthread',
                           '-I', os.path.join(ROOT, 'sdf_amd', 'csrc'), '-o', so, gen])
    lib = ctypes.CDLL(so)
    lib.cull_host.restype = ctypes.c_int
    lib.cull_host.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.cull_sample_host.restype = ctypes.c_int
    lib.cull_sample_host.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
    lib.cull_layout.argtypes = [ctypes.c_void_p]
    lib.cull_sample_loop_host.restype = ctypes.c_int
    lib.cull_sample_loop_host.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_int, ctypes.c_void_p]
    lib.tile_at_host.restype = ctypes.c_float
    lib.tile_at_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.cull_sign_fill_host.restype = ctypes.c_int
    lib.cull_sign_fill_host.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return lib
