// Host-side drivers of the CSC / CSR entry points.
#pragma once
extern "C" int illico_run_csc(illico_ctx *c, const void *data, int dtype, const void *indices, const void *indptr,
                              int idx_dtype, int64_t n_rows, int64_t n_cols, int64_t col_lb, int64_t col_ub, int flags,
                              int alternative, double *out_p, double *out_u, double *out_fc, int64_t out_ld) {
    int rc = check_common(c, n_rows, n_cols, col_lb, col_ub, alternative, out_p, out_u, out_fc, out_ld);
    if (rc) return rc;
    return fail(c, ILLICO_ERR_UNSUPPORTED, "CSC path not built yet");
}
extern "C" int illico_run_csr(illico_ctx *c, const void *data, int dtype, const void *indices, const void *indptr,
                              int idx_dtype, int64_t n_rows, int64_t n_cols, int64_t col_lb, int64_t col_ub, int flags,
                              int alternative, double *out_p, double *out_u, double *out_fc, int64_t out_ld) {
    int rc = check_common(c, n_rows, n_cols, col_lb, col_ub, alternative, out_p, out_u, out_fc, out_ld);
    if (rc) return rc;
    return fail(c, ILLICO_ERR_UNSUPPORTED, "CSR path not built yet");
}
extern "C" int illico_csr_indices_sorted(illico_ctx *c, const void *indices, const void *indptr, int idx_dtype,
                                         int64_t n_rows, int flags, int *out_sorted) {
    return fail(c, ILLICO_ERR_UNSUPPORTED, "not built yet");
}
