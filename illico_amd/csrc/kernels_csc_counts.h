// CSC, count-valued genes, small groups: one pass over a gene's stored entries builds the per-group value histograms in
// LDS -- h[group][value] -- with ONE non-returning LDS atomic per entry; a sweep with one thread per group then turns
// histograms into the statistics:
//   OVO:  S2 = sum_{c>=1} tB[c] (2 zA + 2 cumA[c] + tA[c]) + zB zA,      tie = T_A + sum_{c>=1} tB (3 tA (tA+tB) + tB^2 - 1) + (t0^3 - t0)
//   OVR:  r2 = sum_{c>=1} tB[c] (2 n0 + 2 cum[c] + t[c] + 1) + zB (n0 + 1),  tie = sum_{c>=1} (t^3 - t) + (n0^3 - n0)
// (zA, zB, n0 = implicit zeros of the reference / the group / the column, t0 = zA + zB) -- the same integers the sort-based
// CSC kernels produce (kernels_csc_gene.h, kernels_ovr.h; sparse_ovo.py:58-85, sparse_ovr.py:70-83), without sorting,
// regrouping or searching anything.  Nothing but the CSC arrays is read from HBM: 1.9 GB at C3.
//
// Takes a gene only if every stored value is an integer in [1, RT) (stored zeros are dropped: they are zeros) and
// -- host-checked -- at most CSCC_MAX_BIG ranked groups have more than 255 cells (those get 32-bit cells); other genes set
// fallback[gene] = 1 and go to the general CSC routes.
//
// Two cell layouts (word-major: word w of group g = h[w * G + g], so lanes = consecutive groups read consecutive words in
// the sweep and the increments of random groups spread over all banks):
//  * MIXED: 8 bits for the values 1 .. 7, 4 bits for 8 .. 63 -> 36 bytes per group, 72 KB for 2000 groups, so that TWO
//    workgroups fit a CU and one gene's entry loop overlaps the other's zeroing / sweep / stores.  A 4-bit cell that
//    overflows (16 or more cells of one group with the same value >= 8) carries into its neighbour: the cells of the gene then
//    add up to fewer entries than were counted (a carry can only lose entries), the gene sets fallback[gene] = 2 and the host
//    sends it through the 8-bit form.  words 0, 1: the 8-bit cells of values 0 .. 7; words 2 .. 8: eight 4-bit cells each.
//  * 8-bit cells for every value: RT bytes per group (128 KB for 2000 groups at RT = 64: one workgroup per CU).
//
// What bounds the entry loop is the texture addresser: one codes[row] gather per entry.  A column's rows ascend, so the 64
// gathers of a wavefront fall into a few cache lines -- half as many with 16-bit codes (codes16): 0.80 -> 0.65 ms at C3.
// History at C3 (tools/micro/cscc_bench.hip, nnz 2.4e8): 8-bit cells, 1024 threads, sweep fully unrolled (43 spilled VGPRs)
// 1.51 ms -> sweep rolled, no spills 1.05 -> 16 entries per thread in flight 1.01 -> mixed cells, 2 x 512 threads per CU 0.79
// -> 16-bit codes 0.65 ms.  Phases of the 1.01 ms form: zeroing + launch 0.15, entry loop 0.64, sweep + stores 0.2.
#pragma once
#include <type_traits>
#include "common.h"
#include "kernels_finalize.h"

#define CSCC_NT 512
#ifndef CSCC_UL
#define CSCC_UL 16 // entries per thread and round
#endif
// the forms the library launches (tools/micro/cscc_bench.hip times the others against them)
#define CSCC_WT true
#define CSCC_LEAN true
#define CSCC_PUTB(OVRF) false // (a branch-free entry: 0.58 -> 0.60 ms at C3, the branches skip more than they cost)
#define CSCC_RT 64 // widest table (values 1 .. 63); the 32-value form is used when 64 bytes per group do not fit LDS

struct CscCountsParams {
    const void *data, *indices, *indptr; // CSC arrays (device); stored entry k lives at data[k - kshift], indices[k - kshift]
    long long kshift;
    long long col0;                      // first gene of the batch (contiguous batches)
    const int *gene_cols;                // or: the batch's genes as a column list (absolute indices); nullptr = contiguous
    int nb;
    const u16 *codes16;                  // [n_cells] group code per cell as 16-bit values (C16); else `indices` already holds group codes
    const int *counts;                   // [G]
    int G, ref;                          // ref == -1: OVR.  G: the groups THIS launch holds tables for (a window of the groups, below)
    int g_lo, G_total;                   // groups [g_lo, g_lo + G) of G_total: more groups than LDS holds tables for are taken window by window,
                                         // one launch each (entries of other groups only feed the selected histogram); codes, counts, big_slot,
                                         // ref and the statistics are indexed by the group's number among all G_total
    long long n_cells;
    const signed char *big_slot;         // [G] -1, or the row of the group in the 32-bit table (groups of more than 255 cells)
    u32 *fallback;                       // [nb] set to 1 for genes this kernel cannot take
    long long *out_2u;
    u64 *out_tie;
    double *out_sum;
    double *gene_total;                  // OVR: [nb] the column's value sum (what k_gene_totals would add up from out_sum), or nullptr
    int pack16;                          // statistics as 16 bytes per test: out_2u = value sum << 40 | 2U (40 bits, two's complement: -2 = the OVO
                                         // reference row), out_tie; out_sum unused.  Host-checked: no group beyond 255 cells (sums < 2^24, 2U < 2^39)
    int tie_f64;                         // OVR: out_tie = the bits of the float64 tie sum of the reference's sparse path (tie_f64_sparse)
    const u32 *verdict;                  // deferred calls: {non-integers, -, samples} of k_sample_noncount_cols, looked at on the device
                                         // (more than 2 % non-integers: not a count matrix, every gene is left to the general routes); or nullptr
};

#define CSCM_WPG 9 // words per group of the mixed layout
// cells + slot bytes; rt = 0: the mixed layout
static inline size_t cscc_lds_bytes(int G, int rt) { return (size_t)G * (rt ? rt : CSCM_WPG * 4) + (((size_t)G + 15) & ~(size_t)15); }
static inline size_t cscc_lds_bytes16(int G, int rt) { return (size_t)G * rt * 2 + 16; } // 16-bit cells (W16)

#define CSCC_MAX_BIG 8

// C16: group codes come from codes16[row] (the host's case whenever a code table exists: sparse input is limited to fewer than
// 65 536 groups); else `indices` already holds the codes (the device CSR -> CSC transposition writes them).
// WT: the sweep reads one precomputed {A, B, C} word triple per value instead of forming the terms per (group, value) cell.
// ABL: ablation bits for tools/micro/cscc_bench.hip only (timing builds with wrong results; the library instantiates 0).
// WIN: this launch holds tables for the groups [g_lo, g_lo + G) only (CscCountsParams::g_lo).
// W16: 16-bit cells for every group (RT / 2 words per group): what the route takes when more than CSCC_MAX_BIG ranked groups exceed 255
// cells -- a few hundred groups of a thousand cells each still fit LDS (300 groups x 128 bytes), and every term of the sweep is 64-bit.
template <typename InT, typename IdxT, bool OVR, int RT, bool HAS_BIG, bool MIXED, bool C16, bool WT = false, int ABL = 0, int NTT = CSCC_NT, bool LEAN = false, bool PUTB = true, bool W16 = false, bool WIN = false>
__global__ __launch_bounds__(NTT, MIXED ? (NTT == 1024 ? 8 : 4) : 2) void k_csc_counts(CscCountsParams P) { // (waves per SIMD: two workgroups per CU for the mixed form)
    static_assert(!MIXED || RT == 64, "the mixed layout holds the values 1 .. 63");
    static_assert(!W16 || (!MIXED && !HAS_BIG && !PUTB), "16-bit cells: one plain layout for every group");
    static_assert(!WIN || !PUTB, "group windows: the branching entry only");
    constexpr int NT = NTT, UL = ((HAS_BIG && MIXED) || NTT == 1024) ? CSCC_UL / 2 : CSCC_UL, WPG = MIXED ? CSCM_WPG : (W16 ? RT / 2 : RT / 4); // words per group (UL halved where 16 entries in flight would spill)
    extern __shared__ __align__(16) u32 cscc_h[];             // [WPG][G] words
    __shared__ u32 hsel[RT];   // OVO: histogram of the reference group's stored values; OVR: of the whole column
    __shared__ u32 hbig[HAS_BIG ? CSCC_MAX_BIG * RT : 1]; // 32-bit cells of the few groups with more than 255 cells
    signed char *slot = (signed char *)(cscc_h + (size_t)WPG * P.G); // [G] copy of big_slot (HAS_BIG)
    __shared__ u32 cum[RT + 1]; // cum[c] = # selected stored values < c (c >= 1)
    __shared__ uint4 wtab[WT ? RT : 1]; // per value c: {A, B, C, -}: 2 rank(c) term, 3 tS^2 - 1, 3 tS (B, C: OVO tie term)
    __shared__ u64 s_T, s_sum;
    __shared__ u32 s_nnz, s_entries, s_cells;
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63;
    const int G = P.G, ref = P.ref, g_lo = P.g_lo;
    const InT *data = (const InT *)P.data;
    const IdxT *indices = (const IdxT *)P.indices, *indptr = (const IdxT *)P.indptr;

    if (P.verdict && (double)P.verdict[0] > 0.02 * (double)P.verdict[2]) { // uniform: decided from a value sample, without the host
        for (int gene = blockIdx.x; gene < P.nb; gene += gridDim.x)
            if (tid == 0) P.fallback[gene] = 1u;
        return;
    }
    for (int gene = blockIdx.x; gene < P.nb; gene += gridDim.x) {
        const long long col = P.gene_cols ? (long long)P.gene_cols[gene] : P.col0 + gene;
        const long long k0 = (long long)indptr[col] - P.kshift, k1 = (long long)indptr[col + 1] - P.kshift;
        bool bad = false;
        u32 n_ent = 0; // MIXED: entries this thread put into the packed group tables
        // one stored entry (value v, group code cd) into the tables: ONE non-returning LDS atomic (two for the selected group)
        auto put = [&](InT v, int cd) {
            if (v != (InT)0) {
                const int c = (v > (InT)0 && v < (InT)RT) ? (int)v : 0;
                if (c == 0 || (InT)c != v) bad = true; // negative, fractional, NaN or beyond the table
                else {
                    if (ABL & 2) { n_ent += (u32)(c + cd); return; }
                    const int cl = WIN ? cd - g_lo : cd; // the group's place in this launch's window (WIN: a window of the groups)
                    if (!WIN || (unsigned)cl < (unsigned)G) {
                        const int bs = HAS_BIG ? (int)slot[cl] : -1;
                        if (bs >= 0) atomicAdd(&hbig[bs * RT + c], 1u);
                        else if (OVR || cd != ref) {
                            if (MIXED) {
                                const int wi = c < 8 ? (c >> 2) : 2 + ((c - 8) >> 3);
                                const int sh = c < 8 ? (c & 3) * 8 : ((c - 8) & 7) * 4;
                                atomicAdd(&cscc_h[wi * G + cl], 1u << sh);
                                ++n_ent;
                            } else if (W16) atomicAdd(&cscc_h[(c >> 1) * G + cl], 1u << ((c & 1) * 16)); // (a group holds fewer than 65 536 cells)
                            else atomicAdd(&cscc_h[(c >> 2) * G + cl], 1u << ((c & 3) * 8));
                        }
                    }
                    if (OVR || cd == ref) atomicAdd(&hsel[c], 1u);
                }
            }
        };
        auto code_of = [&](IdxT row) -> int {
            if (ABL & 1) return 1 + ((int)row & 1023);
            return C16 ? (int)P.codes16[(long long)row] : (int)row; // (entries past k1: row 0, value 0 -> ignored)
        };
        auto zero_tables = [&]() {
            if (!(ABL & 32)) {
                uint4 *h4 = (uint4 *)cscc_h;
                const int n4 = (G * WPG) >> 2;
                for (int i = tid; i < n4; i += NT) h4[i] = make_uint4(0u, 0u, 0u, 0u);
                for (int i = (n4 << 2) + tid; i < G * WPG; i += NT) cscc_h[i] = 0;
            }
            if (tid < RT) hsel[tid] = 0;
            if (HAS_BIG) {
                for (int i = tid; i < CSCC_MAX_BIG * RT; i += NT) hbig[i] = 0;
                for (int i = tid; i < G; i += NT) slot[i] = P.big_slot[g_lo + i];
            }
            if (tid == 0) { s_bad = 0; s_entries = 0; s_cells = 0; }
        };
        if constexpr (LEAN) {
            // The same two-stage pipeline with a straight-line body.  (i) A wavefront owns UL * 64 consecutive entries of a
            // round: every round but a column's last is FULL, and its 2 UL requests are one 32-bit lane offset + immediate
            // offsets off a scalar base -- no per-lane bound, no branch, no 64-bit address per request.  The last round clamps
            // its entry numbers to the column's last entry (requests stay unconditional) and drops the surplus lanes when it
            // counts.  (ii) An entry goes into the tables without a branch: a value that is no count in [1, RT) adds 0, a
            // reference-group entry redirects the SAME atomic to the selected histogram (OVO).
            typedef const __attribute__((address_space(1))) char *gchar_p;
            typedef const __attribute__((address_space(1))) InT *gval_p;
            typedef const __attribute__((address_space(1))) IdxT *gidx_p;
            typedef const __attribute__((address_space(1))) u16 *gu16_p;
            typedef __attribute__((address_space(3))) u32 *lds_u32_p;
            constexpr u32 span = (u32)NT * UL;
            const u32 n = (u32)(k1 - k0); // (a column holds fewer than 2^31 stored entries: at most one per row)
            const u32 R = (ABL & 16) ? 0u : (n + span - 1) / span;
            const gchar_p dbase = (gchar_p)(data + k0), ibase = (gchar_p)(indices + k0), cbase = (gchar_p)P.codes16;
            const u32 e_w = (u32)(tid >> 6) * (UL * 64) + (u32)lane;
            const u32 last = n - 1u;
            const u32 h_lds = (u32)(uintptr_t)(lds_u32_p)cscc_h, sel_lds = (u32)(uintptr_t)(lds_u32_p)hsel;
            const u32 G4 = (u32)G * 4u;
            InT vn[UL];
            IdxT in[UL];
            auto load_full = [&](u32 r) { // one lane offset, immediate offsets
                const u32 e0 = r * span + e_w;
                const gchar_p pv = dbase + (size_t)(e0 * (u32)sizeof(InT)), pi = ibase + (size_t)(e0 * (u32)sizeof(IdxT));
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    vn[u] = *(gval_p)(pv + u * 64 * (int)sizeof(InT));
                    in[u] = *(gidx_p)(pi + u * 64 * (int)sizeof(IdxT));
                }
            };
            auto load_clamped = [&](u32 r) { // the column's last, partial round
                const u32 e0 = r * span + e_w;
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const u32 e = min(e0 + u * 64u, last);
                    vn[u] = *(gval_p)(dbase + (size_t)(e * (u32)sizeof(InT)));
                    in[u] = *(gidx_p)(ibase + (size_t)(e * (u32)sizeof(IdxT)));
                }
            };
            auto put_lean = [&](InT v, u32 cd, bool live) {
                // c = the value as a table index, 0 when it is none (zero, negative, fractional, NaN, beyond the table)
                int c = (v > (InT)0 && v < (InT)RT) ? (int)v : 0;
                const bool exact = (InT)c == v; // (v == 0: c == 0, exact)
                if (!live) c = 0;
                bad |= live && !exact;
                if (ABL & 2) { n_ent += (u32)c + cd; return; }
                u32 addr, inc;
                if (MIXED) { // nibble number of the cell inside the group's words: bytes for 1 .. 7, nibbles from 8 on
                    const u32 q = (u32)c < 8u ? 2u * (u32)c : (u32)c + 8u;
                    addr = h_lds + (q >> 3) * G4 + cd * 4u;
                    inc = 1u << ((q & 7u) * 4u);
                } else {
                    addr = h_lds + ((u32)c >> 2) * G4 + cd * 4u;
                    inc = 1u << (((u32)c & 3u) * 8u);
                }
                if (c == 0) inc = 0u;
                bool packed = c != 0;
                if (HAS_BIG) { // the few groups of more than 255 cells: 32-bit cells of their own
                    const int bs = (int)slot[cd];
                    if (bs >= 0) { addr = (u32)(uintptr_t)(lds_u32_p)hbig + (u32)(bs * RT + c) * 4u; inc = c != 0 ? 1u : 0u; packed = false; }
                }
                if (!OVR) { // the reference group's entries go to the selected histogram instead
                    const bool is_ref = (int)cd == ref;
                    if (is_ref) { addr = sel_lds + (u32)c * 4u; inc = c != 0 ? 1u : 0u; packed = false; }
                }
                __hip_atomic_fetch_add((lds_u32_p)(uintptr_t)addr, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (OVR) __hip_atomic_fetch_add((lds_u32_p)(uintptr_t)(sel_lds + (u32)c * 4u), c != 0 ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (MIXED) n_ent += packed ? 1u : 0u;
            };
            // round r: its group codes are requested, then (PF) the next round's values / row indices, then it is counted.
            // Three copies of the body -- next round full / partial / none -- so that neither kind of request sits under a
            // condition inside the loop (the compiler merges two conditional request groups into one with 2 UL computed 64-bit
            // addresses).
            auto round = [&](u32 r, auto pf, auto partial) {
                InT v[UL];
                u32 cd[UL];
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    v[u] = vn[u];
                    if (ABL & 1) cd[u] = 1u + ((u32)in[u] & 1023u);
                    else cd[u] = C16 ? (u32)*(gu16_p)(cbase + (size_t)((u32)in[u] * 2u)) : (u32)in[u]; // (n_cells < 2^30: host-checked)
                }
                if constexpr (decltype(pf)::value == 1) load_full(r + 1);
                if constexpr (decltype(pf)::value == 2) load_clamped(r + 1);
                const u32 e0 = r * span + e_w;
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const bool live = !decltype(partial)::value || e0 + u * 64u <= last;
                    if constexpr (PUTB) put_lean(v[u], cd[u], live);
                    else put(live ? v[u] : (InT)0, (int)cd[u]);
                }
            };
            typedef std::integral_constant<int, 0> I0;
            typedef std::integral_constant<int, 1> I1;
            typedef std::integral_constant<int, 2> I2;
            const u32 Rf = (ABL & 16) ? 0u : n / span; // full rounds; R - Rf = 0 or 1 partial round behind them
            if (Rf > 0) load_full(0);
            else if (R > 0) load_clamped(0);
            zero_tables();
            __syncthreads();
            for (u32 r = 0; r + 1 < Rf; ++r) round(r, I1(), I0());
            if (Rf > 0) {
                if (R > Rf) round(Rf - 1, I2(), I0());
                else round(Rf - 1, I0(), I0());
            }
            if (R > Rf) round(Rf, I0(), I1());
        } else {
        // (the first round's loads are in flight while the tables are zeroed)
        InT vn[UL];
        IdxT in[UL];
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            const long long k = k0 + u * NT + tid;
            vn[u] = k < k1 ? data[k] : (InT)0;
            in[u] = k < k1 ? indices[k] : (IdxT)0;
        }
        zero_tables();
        __syncthreads();
        if ((ABL & 64) && tid >= NT / 2) __builtin_amdgcn_s_sleep(32);  // experiment: the second half of the wavefronts half a round behind
        if ((ABL & 128) && tid >= NT / 2) __builtin_amdgcn_s_sleep(96);
        // two-stage pipeline over the gene's entries: the values / row indices of round i + 1 are requested before round
        // i's group codes (a dependent gather) and LDS atomics, so one HBM round trip per round is off the critical path
        {
            for (long long kb = k0; kb < ((ABL & 16) ? k0 : k1); kb += (long long)NT * UL) {
                InT v[UL];
                int cd[UL];
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    v[u] = vn[u];
                    cd[u] = code_of(in[u]);
                }
                const long long kn = kb + (long long)NT * UL;
                if (kn < k1) { // uniform
#pragma unroll
                    for (int u = 0; u < UL; ++u) {
                        const long long k = kn + u * NT + tid;
                        vn[u] = k < k1 ? data[k] : (InT)0;
                        in[u] = k < k1 ? indices[k] : (IdxT)0;
                    }
                }
#pragma unroll
                for (int u = 0; u < UL; ++u) put(v[u], cd[u]);
            }
        }
        }
        if (MIXED) {
            n_ent = (u32)wave_sum((int)n_ent);
            if (lane == 0 && n_ent) atomicAdd(&s_entries, n_ent);
        }
        if (bad) s_bad = 1;
        __syncthreads();
        if (s_bad) { // uniform: this gene takes the general routes
            if (tid == 0) P.fallback[gene] = 1u;
            __syncthreads();
            continue;
        }
        if (WT) { // wave 0: prefix counts, tie term and value sum of the selected histogram by lane = value; the weight words
            if (tid < 64) {
                const int c = tid;
                const u32 t = (c >= 1 && c < RT) ? hsel[c] : 0u;
                u32 inc = t; // inclusive scan over the values
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const u32 o = (u32)__shfl_up((int)inc, d);
                    if (lane >= d) inc += o;
                }
                const u32 lo = inc - t, run = (u32)__shfl((int)inc, 63);
                const u64 t64 = t;
                const u64 T = wave_sum<u64>(t64 * t64 * t64 - t64), sum = wave_sum<u64>(t64 * (u64)c);
                const long long n_sel = OVR ? P.n_cells : (long long)P.counts[OVR ? 0 : ref];
                const u32 z2 = 2u * (u32)(n_sel - (long long)run); // 2 zA (OVO) or 2 n0 (OVR)
                if (c < RT) {
                    wtab[c] = make_uint4(z2 + 2u * lo + t + (OVR ? 1u : 0u), 3u * t * t - 1u, 3u * t, 0u);
                    if (HAS_BIG || W16) cum[c] = lo;
                }
                if (c == 0) {
                    s_nnz = run; s_T = T; s_sum = sum;
                    if (OVR && P.gene_total) P.gene_total[gene] = (double)sum; // integer sums: exact whatever the order of addition
                }
            }
        } else
        if (tid == 0) {
            u32 run = 0;
            u64 T = 0, sum = 0;
            cum[0] = 0; cum[1] = 0;
            for (int c = 1; c < RT; ++c) {
                const u64 t = hsel[c];
                cum[c] = run;
                run += (u32)t;
                T += t * t * t - t;
                sum += t * (u64)c;
            }
            cum[RT] = run;
            s_nnz = run;
            s_T = T;
            s_sum = sum;
            if (OVR && P.gene_total) P.gene_total[gene] = (double)sum; // integer sums: exact whatever the order of addition
        }
        __syncthreads();
        const u64 nnz_sel = s_nnz, T_sel = s_T;
        const long long n_ref = OVR ? 0 : P.counts[OVR ? 0 : ref];
        const u64 zsel = (u64)((OVR ? P.n_cells : n_ref) - (long long)nnz_sel); // zA (OVO) or n0 (OVR)
        u32 cells_seen = 0;
        auto emit = [&](int g, long long two_u, u64 tie_sum, u64 sum_g) { // (integer value sums: counts)
            const size_t o = (size_t)gene * P.G_total + g; // (g: the group's number among all)
            if (P.pack16) { // (uniform) 16 bytes per test: a third less for this kernel to write and for k_finalize to read
                P.out_2u[o] = (long long)(((u64)sum_g << 40) | ((u64)two_u & 0xFFFFFFFFFFull));
                P.out_tie[o] = tie_sum;
            } else { P.out_2u[o] = two_u; P.out_tie[o] = tie_sum; P.out_sum[o] = (double)sum_g; }
        };
        for (int g = tid; g < G; g += NT) {
            const int gg = g_lo + g; // the group's number among all
            if (!OVR && gg == ref) {
                emit(gg, -2, 0, s_sum);
                continue;
            }
            // 32-bit inner terms (host-checked: n_ref < 30000 for OVO, n_cells < 2^30), one 32 x 32 -> 64 multiply-add each;
            // a word whose cells are empty for every lane of the wavefront is skipped.  The word loop stays rolled: unrolled,
            // the compiler hoists all 2 RT table reads into registers and spills 43 of them.
            u64 acc = 0, tie = 0;
            u32 nnz_g = 0, vsum = 0;
            const int bs = HAS_BIG ? (int)slot[g] : -1;
            auto cell = [&](int c, u32 tB) {
                nnz_g += tB;
                vsum += tB * (u32)c;
                if (WT) { // A = 2 zsel + 2 cum[c] + tS (+ 1), B = 3 tS^2 - 1, C = 3 tS: tB (3 tS (tS + tB) + tB^2 - 1) = tB (B + tB (C + tB))
                    const uint4 w = wtab[c];
                    acc += (u64)tB * w.x;
                    if (!OVR) tie += (u64)tB * (u32)(w.y + tB * (w.z + tB));
                    return;
                }
                const u32 tS = hsel[c], lo = cum[c];
                if (OVR) acc += (u64)tB * (u32)(2u * (u32)zsel + 2u * lo + tS + 1u);
                else {
                    acc += (u64)tB * (u32)(2u * (u32)zsel + 2u * lo + tS);
                    tie += (u64)tB * (u32)(3u * tS * (tS + tB) + tB * tB - 1u);
                }
            };
            auto cell64 = [&](int c, u64 tB) { // a cell of a group of any size: 64-bit terms (3 tS (tS + tB) + tB^2 leaves 32 bits)
                const u64 tS = hsel[c], lo = cum[c];
                nnz_g += (u32)tB;
                vsum += (u32)tB * (u32)c;
                if (OVR) acc += tB * (2ull * zsel + 2ull * lo + tS + 1ull);
                else {
                    acc += tB * (2ull * zsel + 2ull * lo + tS);
                    tie += tB * (3ull * tS * (tS + tB) + tB * tB - 1ull);
                }
            };
            if (HAS_BIG && bs >= 0) { // 32-bit cells of their own
#pragma unroll 1
                for (int c = 1; c < RT; ++c) cell64(c, (u64)hbig[bs * RT + c]);
            }
            const bool packed = !HAS_BIG || bs < 0;
            if (ABL & 4) { acc = cscc_h[g]; tie = cscc_h[G + g]; }
            else
            if (MIXED) {
#pragma unroll
                for (int i = 0; i < 2; ++i) { // the 8-bit cells of the values 0 .. 7
                    const u32 w = packed ? cscc_h[i * G + g] : 0u;
                    if (__ballot(w != 0) == 0ull) continue;
#pragma unroll
                    for (int k = 0; k < 4; ++k) cell(i * 4 + k, (w >> (k * 8)) & 0xFFu);
                }
#pragma unroll 1
                for (int i = 2; i < WPG; ++i) { // eight 4-bit cells per word
                    const u32 w = packed ? cscc_h[i * G + g] : 0u;
                    if (__ballot(w != 0) == 0ull) continue;
#pragma unroll
                    for (int k = 0; k < 8; ++k) cell(8 + (i - 2) * 8 + k, (w >> (k * 4)) & 0xFu);
                }
                if (packed) cells_seen += nnz_g;
            } else if (W16) {
#pragma unroll 1
                for (int i = 0; i < WPG; ++i) { // two 16-bit cells per word
                    const u32 w = cscc_h[i * G + g];
                    if (__ballot(w != 0) == 0ull) continue;
                    if (w & 0xFFFFu) cell64(i * 2, (u64)(w & 0xFFFFu));
                    if (w >> 16) cell64(i * 2 + 1, (u64)(w >> 16));
                }
            } else {
#pragma unroll 1
                for (int i = 0; i < WPG; ++i) {
                    const u32 w = packed ? cscc_h[i * G + g] : 0u;
                    if (__ballot(w != 0) == 0ull) continue;
#pragma unroll
                    for (int k = 0; k < 4; ++k) cell(i * 4 + k, (w >> (k * 8)) & 0xFFu);
                }
            }
            const long long n_g = P.counts[gg];
            if ((ABL & 8) && acc != 0x123456789ull) continue;
            const u64 zB = (u64)(n_g - (long long)nnz_g);
            if (OVR) {
                acc += zB * (zsel + 1ull);
                emit(gg, 2ll * (P.n_cells - n_g) * n_g + n_g * (n_g + 1) - (long long)acc, P.tie_f64 ? tie_f64_sparse(T_sel, (long long)zsel) : T_sel + (zsel * zsel * zsel - zsel), (u64)vsum);
            } else {
                acc += zB * zsel;
                const u64 t0 = zsel + zB;
                emit(gg, 2ll * n_ref * n_g - (long long)acc, T_sel + tie + (t0 * t0 * t0 - t0), (u64)vsum);
            }
        }
        if (MIXED) {
            cells_seen = (u32)wave_sum((int)cells_seen);
            if (lane == 0 && cells_seen) atomicAdd(&s_cells, cells_seen);
            __syncthreads();
            if (tid == 0 && s_cells != s_entries) P.fallback[gene] = 2u; // a 4-bit cell overflowed: the 8-bit form redoes this gene
        }
        __syncthreads();
    }
}
