// sdf_mesh_inst.hip -- instantiations of the fused sample+march kernel for ONE (T, FULL) family.
// Built four times (see build.sh): -DMESH_T=double|float -DMESH_FULL=0|1 -DMESH_NAME=...
// so the families compile in parallel.
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

// Register-file variants (NP point slots, ND distance slots) and launch shapes (threads x samples per lane).  Fewer
// slots = fewer live VGPRs around the interpreter: the host picks the SMALLEST variant that holds the tape's slots
// (exact fits for the models at hand: example / gearlike (1,1), blobby / knurling (2,2), weave (4,2), pawn (2,4)).
// Only the shapes the host can select are instantiated: 1024 x 2 is the measured optimum wherever it exists
// (profiles/r02b_shapes.txt: weave 2^33 47.5 ms vs 59.5 ms at 1024 x 1 and 64.7 ms at 512 x 2, although the 4-slot
// variants spill at 128 VGPRs); the 8-slot file does not fit two samples per lane at 1024 threads at all.
template <int NP, int ND>
static int launch_shape3(int shape, int grid, size_t lds, hipStream_t stream, const uint32_t *code, const MESH_T *consts, const MeshArgs &a) {
    switch (shape) {
    case 0: return launch_one<NP, ND, 1, 1024>(grid, lds, stream, code, consts, a);
    case 1: return launch_one<NP, ND, 2, 512>(grid, lds, stream, code, consts, a);
    default: return launch_one<NP, ND, 2, 1024>(grid, lds, stream, code, consts, a);
    }
}

SDF_DECLARE_MESH_LAUNCH(MESH_NAME, MESH_T) {
    if (twopass) {   // sample + classify only (k_scan_items and k_emit2 follow): the default shape of each register file
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
    case 0: return launch_one<1, 1, 2, 1024>(grid, lds, stream, code, consts, a);
    case 1: return launch_shape3<2, 2>(shape, grid, lds, stream, code, consts, a);
    case 2: return launch_one<4, 2, 2, 1024>(grid, lds, stream, code, consts, a);
    case 3: return launch_one<2, 4, 2, 1024>(grid, lds, stream, code, consts, a);
    case 4: return launch_shape3<4, 4>(shape, grid, lds, stream, code, consts, a);
    default: return shape == 1 ? launch_one<8, 8, 2, 512>(grid, lds, stream, code, consts, a)
                               : launch_one<8, 8, 1, 1024>(grid, lds, stream, code, consts, a);
    }
}

}  // namespace sdfk
