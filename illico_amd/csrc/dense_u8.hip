// the fused single-pass kernels on byte windows (what k_csr_densify writes for count-valued CSR input; the host-window pipeline's
// narrow uploads)
#define ILLICO_DENSE_U8_UNIT
#include "dense_driver.h"
template int run_fused_ovo<uint8_t>(illico_ctx *, const void *, int64_t, int64_t, int, int, int, const OutPlanes &, int64_t, std::vector<u32> &, int, bool, int64_t, const u32 *);
