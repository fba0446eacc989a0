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
// knurling (2,2), weave (4,2), pawn (2,4)).  ONE launch shape per register file -- 1024 threads x 2 samples per lane, the
// measured optimum wherever it exists (profiles/r02b_shapes.txt: weave 2^33 47.5 ms vs 59.5 ms at 1024 x 1 and 64.7 ms at
// 512 x 2, although the 4-slot variants spill at 128 VGPRs); the 8-slot file does not fit two samples per lane at 1024 threads
// and runs 1024 x 1.  (The other shapes were instantiated for tuning until round 5: 12 kernels nothing selected.)
SDF_DECLARE_MESH_LAUNCH(MESH_NAME, MESH_T) {
    if (twopass) {   // sample + classify only (k_scan_items and k_emit2 follow): the default shape of each register file
        switch (slots) {
        case 0: return launch_one<1, 1, 2, 1024, true>(grid, lds, stream, code, consts, a);
        case 1: return launch_one<2, 2, 2, 1024, true>(grid, lds, stream, code, consts, a);
        case 2: return launch_one<4, 2, 2, 1024, true>(grid, lds, stream, code, consts, a);   // (one sample per lane: 8 spilled registers instead of 184 and 43 % slower, profiles/r06f_weave_ns1.json)
        case 3: return launch_one<2, 4, 2, 1024, true>(grid, lds, stream, code, consts, a);
        case 4: return launch_one<4, 4, 2, 1024, true>(grid, lds, stream, code, consts, a);
        default: return launch_one<8, 8, 1, 1024, true>(grid, lds, stream, code, consts, a);
        }
    }
    (void)shape;
    switch (slots) {
    case 0: return launch_one<1, 1, 2, 1024>(grid, lds, stream, code, consts, a);
    case 1: return launch_one<2, 2, 2, 1024>(grid, lds, stream, code, consts, a);
    case 2: return launch_one<4, 2, 2, 1024>(grid, lds, stream, code, consts, a);
    case 3: return launch_one<2, 4, 2, 1024>(grid, lds, stream, code, consts, a);
    case 4: return launch_one<4, 4, 2, 1024>(grid, lds, stream, code, consts, a);
    default: return launch_one<8, 8, 1, 1024>(grid, lds, stream, code, consts, a);
    }
}

}  // namespace sdfk
