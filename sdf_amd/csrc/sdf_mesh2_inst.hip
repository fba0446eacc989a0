// sdf_mesh2_inst.hip -- instantiations of k_mesh2 (sdf_mesh2.h: the fused sample+march kernel as two workgroups of 512 threads
// per compute unit) for ONE family.  Built twice (build.sh): -DMESH_FULL=0|1 -DMESH_NAME=...
#include "sdf_mesh2.h"

namespace sdfk {

template <int NP, int ND, int NS>
static int launch_one2(int grid, size_t lds, hipStream_t stream, const uint32_t *code, const double *consts, const MeshArgs &a) {
    auto fn = k_mesh2<double, (MESH_FULL != 0), NP, ND, NS>;
    static size_t lds_set[16] = {};      // per device: the dynamic-LDS limit this instantiation was last raised to
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || lds_set[dev] < lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0 && dev < 16) lds_set[dev] = lds;
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(M2_BLOCK), lds, stream, code, consts, a);
    return (int)hipGetLastError();
}

// the register files whose interpreters hold 128 vector registers without spilling much: (1,1) example / gearlike, (2,2) blobby /
// knurling, (2,4) pawn; tapes that need more slots stay with k_mesh (-1)
SDF_DECLARE_MESH2_LAUNCH(MESH_NAME, double) {
    switch (slots) {
    case 0: return launch_one2<1, 1, 2>(grid, lds, stream, code, consts, a);
    case 1: return launch_one2<2, 2, 2>(grid, lds, stream, code, consts, a);
    case 3: return launch_one2<2, 4, 2>(grid, lds, stream, code, consts, a);
    default: return -1;
    }
}

}  // namespace sdfk
