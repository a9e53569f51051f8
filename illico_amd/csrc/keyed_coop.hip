// The COOP form of k_ovr_partition_packed (kernels_csc_ovr.h: long blocks dealt over all the wavefronts of a gene's workgroup) in a
// translation unit of its own: the plain form in keyed_u32.hip / keyed_u64.hip keeps the code it had.
#include "keyed_driver.h"

template <typename KeyT>
int launch_ovr_partition_packed_coop(illico_ctx *c, const OvrPartPackedParams &Q, int nb) {
    auto kern = k_ovr_partition_packed<KeyT, true>;
    HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ovrp_lds_bytes()));
    hipLaunchKernelGGL(kern, dim3(nb), dim3(OVRP_NT), ovrp_lds_bytes(), c->stream, Q);
    HIPCHK(c, hipGetLastError());
    return ILLICO_OK;
}
template int launch_ovr_partition_packed_coop<u32>(illico_ctx *, const OvrPartPackedParams &, int);
#ifndef ILLICO_DEV_F32_ONLY
template int launch_ovr_partition_packed_coop<u64>(illico_ctx *, const OvrPartPackedParams &, int);
#endif
