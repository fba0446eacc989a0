// sdf_mesh_inst.hip -- instantiations of the fused sample+march kernel for ONE family.
// Built twice (see build.sh): -DMESH_T=double -DMESH_FULL=0|1 -DMESH_NAME=... (FULL: the tape uses the trigonometric ops),
// so the families compile in parallel.  (float32 sampling of the meshing path was removed in round 5.)
#include "sdf_device.h"

namespace sdfk {

template <int NP, int ND, int NS, int BLOCK, bool TWOPASS = false>
static int launch_one(int grid, size_t lds, hipStream_t stream, const uint32_t *code, const MESH_T *consts, const MeshArgs &a) {
    auto fn = k_mesh<MESH_T, (MESH_FULL != 0), NP, ND, NS, BLOCK, TWOPASS>;
    static size_t lds_set[16] = {};      // per device: the dynamic-LDS limit this instantiation was last raised to
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || lds_set[dev] < lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0 && dev < 16) lds_set[dev] = lds;
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(BLOCK), lds, stream, code, consts, a);
    return (int)hipGetLastError();
}

// Register-file variants (NP point slots, ND distance slots).  Fewer slots = fewer live VGPRs around the interpreter: the host
// picks the SMALLEST variant that holds the tape's slots (exact fits for the models at hand: example / gearlike (1,1), blobby /
// knurling (2,2), weave (4,2), pawn (2,4)).  ONE launch shape per register file and scheme, 1024 threads x NS samples per lane:
//   * one pass, files (1,1), (2,2), (2,4): THREE samples per lane since r06ah -- a third less of the interpreter's scalar stream (fetch,
//     decode, dispatch: as many scalar as vector instructions per tape instruction at two samples) and 48 instead of 32 tasks per
//     sampling pass; 54 - 112 spilled vector registers instead of 5 - 26 do not matter (as r06ad had shown for weave).  One box,
//     alternating, three repeats (profiles/r06ai_three_samples.json): example 2^27 k_mesh 0.186 -> 0.180 ms, 0.2201 -> 0.2149 ms per
//     sustained step; gearlike 2^30 1.126 -> 1.083; knurling 2^27 1.368 -> 1.322; blobby 2^30 0.572 -> 0.560; pawn 2^27 0.261 -> 0.255.
//     FOUR samples per lane (204 spills): - 1.6 % only;
//   * the long tapes' two-pass scheme and the (4,2), (4,4) files: two samples per lane (weave 2^33 with three: + 6.5 %, r06aj; with four
//     at 512 threads the same as two, r06ad; with one 43 % slower, r06f; profiles/r02b_shapes.txt: 47.5 ms at 1024 x 2 against 59.5 at
//     1024 x 1 and 64.7 at 512 x 2);
//   * the 8-slot file does not fit two samples per lane at 1024 threads and runs 1024 x 1.
SDF_DECLARE_MESH_LAUNCH(MESH_NAME, MESH_T) {
    (void)shape;
    if (twopass) {   // sample + classify only (k_scan_items and k_emit2 follow)
        switch (slots) {
        case 0: return launch_one<1, 1, 2, 1024, true>(grid, lds, stream, code, consts, a);
        case 1: return launch_one<2, 2, 2, 1024, true>(grid, lds, stream, code, consts, a);
        case 2: return launch_one<4, 2, 2, 1024, true>(grid, lds, stream, code, consts, a);
        case 3: return launch_one<2, 4, 2, 1024, true>(grid, lds, stream, code, consts, a);
        case 4: return launch_one<4, 4, 2, 1024, true>(grid, lds, stream, code, consts, a);
        default: return launch_one<8, 8, 1, 1024, true>(grid, lds, stream, code, consts, a);
        }
    }
    switch (slots) {
    case 0: return launch_one<1, 1, 3, 1024>(grid, lds, stream, code, consts, a);
    case 1: return launch_one<2, 2, 3, 1024>(grid, lds, stream, code, consts, a);
    case 2: return launch_one<4, 2, 2, 1024>(grid, lds, stream, code, consts, a);
    case 3: return launch_one<2, 4, 3, 1024>(grid, lds, stream, code, consts, a);
    case 4: return launch_one<4, 4, 2, 1024>(grid, lds, stream, code, consts, a);
    default: return launch_one<8, 8, 1, 1024>(grid, lds, stream, code, consts, a);
    }
}

}  // namespace sdfk
