// Host-side launch of the OVR per-gene kernel (included by illico_hip.hip after the context helpers).
#pragma once

template <typename KeyT, bool SPARSE, bool OVO, int NT>
static int launch_ovr_gene_nt(illico_ctx *c, const OvrParams &P) {
    size_t lds = ovr_lds_bytes(P.G, SPARSE, OVO, NT);
    if (lds > kMaxLds)
        return fail(c, ILLICO_ERR_UNSUPPORTED, "%d groups do not fit the LDS accumulators of the global-sort route in this build", P.G);
    auto kern = k_ovr_gene<KeyT, SPARSE, OVO, NT>;
    HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        ProfScope ps(c, KID_OVR_SCAN);
        hipLaunchKernelGGL(kern, dim3(P.n_genes), dim3(NT), lds, c->stream, P);
        HIPCHK(c, hipGetLastError());
    }
    return ILLICO_OK;
}

// workgroup size of the per-gene sort kernel: several smaller workgroups per CU overlap each other's barriers
template <typename KeyT, bool SPARSE, bool OVO = false>
static int launch_ovr_gene(illico_ctx *c, const OvrParams &P) {
    switch (c->ovr_threads) {
    case 1024: return launch_ovr_gene_nt<KeyT, SPARSE, OVO, 1024>(c, P);
    case 512: return launch_ovr_gene_nt<KeyT, SPARSE, OVO, 512>(c, P);
    default: return launch_ovr_gene_nt<KeyT, SPARSE, OVO, 256>(c, P);
    }
}

static int launch_gene_totals(illico_ctx *c, const double *ssum, int G, int nb, double *gtot) {
    ProfScope ps(c, KID_GENE_TOTALS);
    hipLaunchKernelGGL(k_gene_totals, dim3((nb + 255) / 256), dim3(256), 0, c->stream, ssum, G, nb, gtot);
    HIPCHK(c, hipGetLastError());
    return ILLICO_OK;
}

template <typename KeyT>
static int run_ovr_dense_batch(illico_ctx *c, KeyT *Xt, int64_t stride, int nb, int N, int dtype, int flags,
                               long long *s2u, u64 *stie, double *ssum, double *gtot) {
    void *v;
    int rc;
    if ((rc = get_scratch(c, "ovr_kb", (size_t)nb * stride * sizeof(KeyT), &v))) return rc;
    void *kb = v;
    if ((rc = get_scratch(c, "ovr_va", (size_t)nb * stride * 4, &v))) return rc;
    u32 *va = (u32 *)v;
    if ((rc = get_scratch(c, "ovr_vb", (size_t)nb * stride * 4, &v))) return rc;
    u32 *vb = (u32 *)v;
    OvrParams P;
    P.keys_a = Xt; P.keys_b = kb; P.vals_a = va; P.vals_b = vb;
    P.code_by_pos = c->d_code_by_pos; P.seg_ptr = nullptr; P.stride = stride; P.pos_ptr = c->d_posptr;
    P.counts = c->d_counts; P.G = (int)c->n_groups; P.n_genes = nb; P.dt = dtype;
    P.is_log1p = (flags & ILLICO_FLAG_LOG1P) ? 1 : 0; P.n_cells = N; P.ref = -1; P.gene_flags = nullptr;
    P.out_2u = s2u; P.out_tie = stie; P.out_sum = ssum;
    if ((rc = launch_ovr_gene<KeyT, false>(c, P))) return rc;
    return launch_gene_totals(c, ssum, (int)c->n_groups, nb, gtot);
}
