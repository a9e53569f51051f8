// CSC / CSR drivers and kernels for float values (u32 keys), int32 and int64 indices
#include "sparse_driver.h"
template int run_sparse_t<float, int32_t, u32>(illico_ctx *, bool, const void *, const void *, const void *, int, int64_t, int64_t, int64_t, int64_t, int, int, const OutPlanes &, bool, bool, bool, bool);
#ifndef ILLICO_DEV_F32_ONLY
template int run_sparse_t<float, int64_t, u32>(illico_ctx *, bool, const void *, const void *, const void *, int, int64_t, int64_t, int64_t, int64_t, int, int, const OutPlanes &, bool, bool, bool, bool);
#endif
