// Host-side launch of the OVR per-gene kernel (included by illico_hip.hip after the context helpers).
#pragma once

constexpr int kOvrThreads = 256; // several small workgroups per CU overlap each other's barriers (1024 measured the same)

// Per-group accumulators in LDS when they fit, else in HBM (one [3*G] u64 block per gene of the batch).
template <typename KeyT, bool SPARSE, bool OVO = false>
static int launch_ovr_gene(illico_ctx *c, OvrParams P) {
    constexpr bool DC = !SPARSE && !OVO; // dense OVR: compact the zeros away inside the kernel
    const bool accg = ovr_lds_bytes(P.G, SPARSE || DC, OVO, kOvrThreads, false) > kMaxLds;
    const size_t lds = ovr_lds_bytes(P.G, SPARSE || DC, OVO, kOvrThreads, accg);
    P.acc_global = nullptr;
    if (accg) {
        void *v;
        int rc = get_scratch(c, "ovr_acc", (size_t)P.n_genes * 3 * P.G * 8, &v);
        if (rc) return rc;
        P.acc_global = (u64 *)v;
    }
    ProfScope ps(c, KID_OVR_SCAN);
    if (accg) {
        auto kern = k_ovr_gene<KeyT, SPARSE, OVO, kOvrThreads, true, DC>;
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(P.n_genes), dim3(kOvrThreads), lds, c->stream, P);
    } else {
        auto kern = k_ovr_gene<KeyT, SPARSE, OVO, kOvrThreads, false, DC>;
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(P.n_genes), dim3(kOvrThreads), lds, c->stream, P);
    }
    HIPCHK(c, hipGetLastError());
    return ILLICO_OK;
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
