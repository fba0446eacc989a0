// sdf_mesh_inst.hip -- instantiations of the fused sample+march kernel for ONE (T, FULL) family.
// Built four times (see build.sh): -DMESH_T=double|float -DMESH_FULL=0|1 -DMESH_NAME=...
// so the families compile in parallel.
#include "sdf_device.h"

namespace sdfk {

template <int NP, int ND, int NS, int BLOCK>
static int launch_one(int grid, size_t lds, hipStream_t stream, const uint32_t *code, const MESH_T *consts, const MeshArgs &a) {
    auto fn = k_mesh<MESH_T, (MESH_FULL != 0), NP, ND, NS, BLOCK>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(BLOCK), lds, stream, code, consts, a);
    return (int)hipGetLastError();
}

template <int NP, int ND>
static int launch_shape(int shape, int grid, size_t lds, hipStream_t stream, const uint32_t *code, const MESH_T *consts, const MeshArgs &a) {
    switch (shape) {
    case 0: return launch_one<NP, ND, 1, 1024>(grid, lds, stream, code, consts, a);
    case 1: return launch_one<NP, ND, 2, 512>(grid, lds, stream, code, consts, a);
    default: return launch_one<NP, ND, 2, 1024>(grid, lds, stream, code, consts, a);
    }
}

SDF_DECLARE_MESH_LAUNCH(MESH_NAME, MESH_T) {
    switch (slots) {
    case 0: return launch_shape<2, 2>(shape, grid, lds, stream, code, consts, a);
    case 1: return launch_shape<4, 4>(shape, grid, lds, stream, code, consts, a);
    default: return launch_shape<8, 8>(shape, grid, lds, stream, code, consts, a);
    }
}

}  // namespace sdfk
