// dense-input drivers and kernels for int32_t values (u32 keys)
#include "dense_driver.h"
template int run_fused_ovo<int32_t>(illico_ctx *, const void *, int64_t, int64_t, int, int, int, const OutPlanes &, int64_t, std::vector<u32> &, int, bool, int64_t, const u32 *);
template int run_dense_t<int32_t, u32>(illico_ctx *, const void *, int, int64_t, int64_t, int64_t, int64_t, int, int, const OutPlanes &);
template int run_leftovers<int32_t, u32>(illico_ctx *, const void *, int, int64_t, int64_t, int64_t, int64_t, int, int, const OutPlanes &, const u32 *, bool, const int *);
