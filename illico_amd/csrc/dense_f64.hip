// dense-input drivers and kernels for double values (u64 keys)
#include "dense_driver.h"
template int run_fused_ovo<double>(illico_ctx *, const void *, int64_t, int64_t, int, int, int, const OutPlanes &, int64_t, std::vector<u32> &, int, bool, int64_t, const u32 *);
template int run_dense_t<double, u64>(illico_ctx *, const void *, int, int64_t, int64_t, int64_t, int64_t, int, int, const OutPlanes &);
template int run_leftovers<double, u64>(illico_ctx *, const void *, int, int64_t, int64_t, int64_t, int64_t, int, int, const OutPlanes &, const u32 *, bool, const int *);
