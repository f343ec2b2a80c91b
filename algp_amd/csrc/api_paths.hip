// api_paths.hip -- best_path (agent.py:358-403): the entropy gain of every enumerated path in one call.
#include "api_impl.h"

using namespace algp;

namespace algp {


// Paths of 65 .. 256 distinct sites (cpos / lpos: candidate row and train row per site, packed to the front of each path's
// maxlen entries).  Per batch of paths: the paths' rows of V^T gathered into a scratch (a site that is a train row
// already: L[lpos, :] - var_lpos * its unit row, as in the LDS kernel), the Gram matrices as ONE batched lower-tile MFMA
// product, G = C_PP + sigma_m^2 I - Gram, then the ppad x ppad blocks factored as 2 x 2 tiles of 128: diagonal-block
// kernel, L21 = G21 inv(L11)^T, G22 -= L21 L21^T, diagonal-block kernel -- every step one launch for the whole batch.
template <typename T>
int Impl<T>::score_paths_big(algp_ctx* c, const std::vector<int64_t>& cpos, const std::vector<int64_t>& lpos, int npaths, int maxlen,
                               int maxused, double mobile_std, double* dH) {
    const int64_t Npad = c->Npad;
    const int ppad = maxused <= NB ? NB : 2 * NB;
    // rows scratch: batch * ppad * Npad elements, at most ~4 GB
    const int64_t per_path = (int64_t)ppad * Npad * (int64_t)sizeof(T);
    const int bmax = (int)std::max<int64_t>(1, std::min<int64_t>(npaths, (int64_t)4e9 / per_path));
    ALGP_TRY(ensure(c, c->auxW, (size_t)bmax * per_path));
    ALGP_TRY(ensure(c, c->auxA, sizeof(T) * (size_t)bmax * ppad * ppad));
    ALGP_TRY(ensure(c, c->auxInv, sizeof(T) * (size_t)bmax * 2 * NB * NB));
    ALGP_TRY(ensure(c, c->auxD, sizeof(T) * (size_t)bmax * NB * NB));
    ALGP_TRY(ensure(c, c->auxIdx, sizeof(int64_t) * 2 * (size_t)bmax * ppad));
    ALGP_TRY(ensure(c, c->auxVar, sizeof(T) * (size_t)bmax * ppad + 256));
    ALGP_TRY(ensure(c, c->hostStage, sizeof(double) * (size_t)(npaths + bmax) + sizeof(int) * (size_t)bmax + 64));
    double* d_out = (double*)c->hostStage.p;
    double* d_ld = d_out + npaths;
    int* d_info = (int*)(d_ld + bmax);
    T* rows = p(c->auxW);
    T* G = p(c->auxA);
    T* inv = p(c->auxInv);
    T* L21 = p(c->auxD);
    int64_t* d_src = (int64_t*)c->auxIdx.p;
    int64_t* d_lrow = d_src + (size_t)bmax * ppad;
    std::vector<int64_t> src((size_t)bmax * ppad), lr((size_t)bmax * ppad);
    std::vector<T> lsc((size_t)bmax * ppad);
    for (int p0 = 0; p0 < npaths; p0 += bmax) {
        const int B = std::min(bmax, npaths - p0);
        bool second = false;
        for (int b = 0; b < B; ++b)
            for (int a = 0; a < ppad; ++a) {
                const size_t e = (size_t)b * ppad + a;
                const int64_t cp = a < maxlen ? cpos[(size_t)(p0 + b) * maxlen + a] : -1;
                const int64_t lp = a < maxlen ? lpos[(size_t)(p0 + b) * maxlen + a] : -1;
                src[e] = cp;
                lr[e] = cp >= 0 ? lp : -1;
                lsc[e] = (cp >= 0 && lp >= 0) ? (T)c->train_var_host[(size_t)lp] : (T)0;
                second |= cp >= 0 && lp >= 0;
            }
        const size_t nrow = (size_t)B * ppad;
        ALGP_HIP(hipMemcpyAsync(d_src, src.data(), sizeof(int64_t) * nrow, hipMemcpyHostToDevice, c->stream));
        ALGP_HIP(hipMemcpyAsync(d_lrow, lr.data(), sizeof(int64_t) * nrow, hipMemcpyHostToDevice, c->stream));
        ALGP_HIP(hipMemcpyAsync(c->auxVar.p, lsc.data(), sizeof(T) * nrow, hipMemcpyHostToDevice, c->stream));
        ALGP_HIP(hipMemsetAsync(d_ld, 0, sizeof(double) * B, c->stream));
        ALGP_HIP(hipMemsetAsync(d_info, 0, sizeof(int) * B, c->stream));
        ALGP_TRY(gather_rows_launch<T>(c, p(c->Vt), c->ldv, d_src, rows, Npad, (int64_t)nrow, Npad, second ? d_lrow : nullptr,
                                       second ? (const T*)c->auxVar.p : nullptr, p(c->L), c->Lld));
        ALGP_TRY(gemm_nt_launch_batched<T>(c, ALGP_PROF_GEMM_OTHER, ppad, ppad, Npad, (T)1, rows, Npad, (int64_t)ppad * Npad, rows, Npad,
                                           (int64_t)ppad * Npad, (T)0, nullptr, 0, 0, G, ppad, (int64_t)ppad * ppad, 1, B));
        ALGP_TRY(path_assemble_launch<T>(c, d_src, B, ppad, (const int64_t*)c->Cidx.p, (const T*)c->Xs.p,
                                         c->pool_is_cov ? (const T*)c->Cp.p : nullptr, c->n_pool, c->hyp.DP, c->hyp.kernel,
                                         c->hyp.outputscale, c->hyp.noise, mobile_std * mobile_std, G));
        ALGP_TRY(potrf_diag_batched_launch<T>(c, G, (int64_t)ppad * ppad, ppad, inv, 2 * NB * NB, d_ld, d_info, B));
        if (ppad > NB) {
            T* G21 = G + (int64_t)NB * ppad;
            T* G22 = G21 + NB;
            ALGP_TRY(gemm_nt_launch_batched<T>(c, ALGP_PROF_GEMM_OTHER, NB, NB, NB, (T)1, G21, ppad, (int64_t)ppad * ppad, inv, NB, 2 * NB * NB,
                                               (T)0, nullptr, 0, 0, L21, NB, NB * NB, 0, B));
            ALGP_TRY(gemm_nt_launch_batched<T>(c, ALGP_PROF_GEMM_OTHER, NB, NB, NB, (T)-1, L21, NB, NB * NB, L21, NB, NB * NB, (T)1, G22, ppad,
                                               (int64_t)ppad * ppad, G22, ppad, (int64_t)ppad * ppad, 0, B));
            ALGP_TRY(potrf_diag_batched_launch<T>(c, G22, (int64_t)ppad * ppad, ppad, inv + NB * NB, 2 * NB * NB, d_ld, d_info, B));
        }
        ALGP_TRY(path_finish_launch(c, d_src, ppad, B, d_ld, d_info, d_out + p0));
        ALGP_TRY(sync(c));                                   // the index vectors are reused by the next batch
    }
    ALGP_HIP(hipMemcpyAsync(dH, d_out, sizeof(double) * npaths, hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}


// a8 / f3: the entropy gain of every enumerated path (agent.py:374-400 computes one slogdet per path) from ONE
// resident factor and candidate solve: sites[p][a] are pool indices (-1 = none); a site that already is a train
// row receives a second (mobile) row, a new site a first one; dH[p] = H(A u path_p) - H(A)
template <typename T>
int Impl<T>::score_paths(algp_ctx* c, const int64_t* sites, int npaths, int maxlen, double mobile_std, double* dH) {
    if (!c->solved) return fail(c, ALGP_ERR_STATE, "score_paths: call algp_solve_candidates first");
    if (!c->prior_noise) return fail(c, ALGP_ERR_STATE, "score_paths: candidates were set with predictive semantics");
    if (!c->picks.empty()) return fail(c, ALGP_ERR_STATE, "score_paths: picks were committed since the candidate solve; solve again");
    const size_t tot = (size_t)npaths * maxlen;
    std::vector<int64_t> cpos(tot, -1), lpos(tot, -1);
    int maxused = 0;
    for (int pth = 0; pth < npaths; ++pth) {
        int used = 0;
        for (int a = 0; a < maxlen; ++a) {
            const int64_t j = sites[(size_t)pth * maxlen + a];
            if (j < 0) continue;
            if (j >= c->n_pool) return fail(c, ALGP_ERR_BAD_ARG, "score_paths: index outside the pool");
            const int64_t cp = c->cand_pos[j];
            if (cp < 0) return fail(c, ALGP_ERR_BAD_ARG, "score_paths: site " + std::to_string(j) + " is not a resident candidate");
            bool dup = false;
            for (int b = 0; b < used; ++b) dup |= cpos[(size_t)pth * maxlen + b] == cp;
            if (dup) continue;                                  // a site crossed twice is measured once (mobile mask)
            cpos[(size_t)pth * maxlen + used] = cp;
            lpos[(size_t)pth * maxlen + used] = c->pos_in_train[j];
            ++used;
        }
        if (used > 256) return fail(c, ALGP_ERR_BAD_ARG, "score_paths: more than 256 distinct sites in a path");
        maxused = std::max(maxused, used);
    }
    if (maxused > 64) return score_paths_big(c, cpos, lpos, npaths, maxlen, maxused, mobile_std, dH);
    ALGP_TRY(ensure(c, c->auxIdx, sizeof(int64_t) * 2 * tot));
    ALGP_TRY(ensure(c, c->hostStage, sizeof(double) * std::max<size_t>(npaths, 1)));
    int64_t* d_c = (int64_t*)c->auxIdx.p;
    int64_t* d_l = d_c + tot;
    ALGP_HIP(hipMemcpyAsync(d_c, cpos.data(), sizeof(int64_t) * tot, hipMemcpyHostToDevice, c->stream));
    ALGP_HIP(hipMemcpyAsync(d_l, lpos.data(), sizeof(int64_t) * tot, hipMemcpyHostToDevice, c->stream));
    ALGP_TRY(path_score_launch<T>(c, d_c, d_l, npaths, maxlen, (const int64_t*)c->Cidx.p, p(c->Vt), c->ldv, c->ncols, p(c->L),
                                  c->Lld, (const T*)c->varA.p, (const T*)c->Xs.p, c->pool_is_cov ? (const T*)c->Cp.p : nullptr,
                                  c->n_pool, c->hyp.DP, c->hyp.kernel, c->hyp.outputscale, c->hyp.noise,
                                  mobile_std * mobile_std, (double*)c->hostStage.p));
    ALGP_HIP(hipMemcpyAsync(dH, c->hostStage.p, sizeof(double) * npaths, hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}

template struct Impl<float>;
template struct Impl<double>;

}  // namespace algp

extern "C" {

int algp_score_paths(algp_ctx* c, const int64_t* sites, int npaths, int maxlen, double mobile_std, double* dH_out) {
    CHECK_CTX(c);
    if (npaths < 0 || maxlen < 1 || (npaths > 0 && (!sites || !dH_out))) return fail(c, ALGP_ERR_BAD_ARG, "score_paths: bad arguments");
    if (npaths == 0) return ALGP_OK;
    FINISH(c, DISPATCH(c, score_paths(c, sites, npaths, maxlen, mobile_std, dH_out)));
}

}  // extern "C"
