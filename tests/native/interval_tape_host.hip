// Host build of the interval run of a whole tape (sdf_amd/csrc/sdf_interval.h, ia_run_tape without the pruning
// decisions): tests/test_interval_host.py loads it with ctypes, runs the tapes of the value fixtures over random
// boxes and checks the CPU checker's point values against the intervals.  Test infrastructure only.
#include <cstdint>
#include <cstring>

#include "sdf_interval.h"

using namespace sdfk;

// boxes: n x 6 doubles (x.lo x.hi y.lo y.hi z.lo z.hi); out: n x 2 doubles (lo, hi).  Returns 0.
extern "C" int ia_tape_boxes(const uint32_t *code, const double *consts, int n_instr, int n_p, int n_d,
                             const double *boxes, long long n, double *out) {
    double state[6 * 8 + 2 * 8];
    if (n_p < 1) n_p = 1;
    if (n_d < 1) n_d = 1;
    if (n_p > 8 || n_d > 8) return 1;
    for (long long i = 0; i < n; i++) {
        const double *b = boxes + 6 * i;
        std::memset(state, 0, sizeof state);
        const Ival v = ia_run_tape<false, true, true>(code, consts, nullptr, nullptr, n_instr, Ival{b[0], b[1]}, Ival{b[2], b[3]}, Ival{b[4], b[5]},
                                                      true, IaShared{state, n_p, 1}, n_d, nullptr);
        out[2 * i] = v.lo; out[2 * i + 1] = v.hi;
    }
    return 0;
}
