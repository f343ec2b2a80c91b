// api_fit.hip -- one iteration of GPR.fit (models.py:145-158): MLL and its gradient w.r.t. the log hyper-parameters.
#include "api_impl.h"

using namespace algp;

namespace algp {


// have_X: c->auxW already holds X = L^-T (it rode along with the factorisation as an identity panel); inv_enqueued: and
// S^-1 = X X^T is already running on the helper stream (event 21 marks its end)
template <typename T>
int Impl<T>::mll_grad(algp_ctx* c, double* grad_out, bool have_X, bool inv_enqueued) {
    if (!c->factored || c->train_dirty) return fail(c, ALGP_ERR_STATE, "get_mll_grad: call algp_factorize first");
    if (c->pool_is_cov) return fail(c, ALGP_ERR_BAD_ARG, "get_mll_grad needs a coordinate pool");
    const int64_t N = c->N, Npad = c->Npad;
    const int D = c->hyp.D, DP = c->hyp.DP;
    if (have_X && !c->alpha_valid) {
        // alpha = L^-T z = X z with the X the launch left in auxW: one pass over its upper triangle (0.4 GB at N = 10 000)
        // instead of the backward substitution's chain of 79 hand-offs
        ALGP_TRY(upper_gemv_launch<T>(c, p(c->auxW), Npad, Npad, (const T*)c->z.p, p(c->alpha)));
        c->alpha_valid = true;
    }
    ALGP_TRY(need_alpha(c));
    ALGP_TRY(ensure(c, c->auxA, sizeof(T) * Npad * Npad));
    // X = I L^-T = L^-T ;  S^-1 = X X^T (lower tiles)
    if (!have_X) {
        ALGP_TRY(ensure(c, c->auxW, sizeof(T) * Npad * Npad));
        ALGP_TRY(set_identity_launch<T>(c, p(c->auxW), Npad, Npad));
        ALGP_TRY(trinv_upper<T>(c, ALGP_PROF_GEMM_OTHER, p(c->auxW), Npad, Npad, p(c->L), c->Lld, p(c->invD)));
    }
    if (inv_enqueued) ALGP_HIP(hipStreamWaitEvent(c->stream, sync_event_api(c, 21), 0));
    else ALGP_TRY(syrk_upper<T>(c, ALGP_PROF_GEMM_OTHER, p(c->auxW), Npad, Npad, p(c->auxA), Npad));
    double* sc = (double*)c->scal.p + SC_GRAD;        // slots 16..27: os, trace, ls[0..8)
    ALGP_HIP(hipMemsetAsync(sc, 0, sizeof(double) * 12, c->stream));
    ALGP_TRY(mll_grad_launch<T>(c, p(c->auxA), Npad, N, (const T*)c->Xs.p, DP, (const int64_t*)c->Aidx.p,
                                (const T*)c->alpha.p, c->hyp.kernel, (T)c->hyp.outputscale, sc,
                                (double*)c->auxW.p /* X = L^-T is spent: room for the per-workgroup partials */));
    double h[12];
    ALGP_HIP(hipMemcpyAsync(h, sc, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    {
        const int rc = sync_checked(c, "get_mll_grad");
        if (rc != ALGP_OK) { c->alpha_valid = false; return rc; }
    }
    for (int d = 0; d < D; ++d) grad_out[d] = 0.5 * h[2 + d];
    grad_out[D] = 0.5 * h[0];
    grad_out[D + 1] = 0.5 * c->hyp.noise * h[1];
    return ALGP_OK;
}


// f2: the device work of ONE iteration of GPR.fit (models.py:145-158: loss = -mll(model(train_x), train_y); backward) in
// one ABI call: S, its factor AND X = L^-T out of the same task-list launch (the identity rides along as a panel
// whose zero tiles are never touched), alpha by the two one-launch substitutions, S^-1 = X X^T as one
// triangular-aware launch, the pairwise gradient reduction.  Same values as algp_factorize + algp_get_mll +
// algp_get_mll_grad (tested); N^3 flop in all (N^3/3 each for the factor, the inverse of the factor and the product).
template <typename T>
int Impl<T>::fit_step(algp_ctx* c, double* mll_out, double* grad_out) {
    if (c->pool_is_cov) return fail(c, ALGP_ERR_BAD_ARG, "fit_step needs a coordinate pool");
    const int64_t Npad = c->Npad;
    bool have_X = false, inv_enq = false;
    // Round 5: z rides along too (a dense tile row behind the identity that carries y - ybar), and alpha = X z is one pass
    // over the X the launch leaves -- no substitution chain runs beside S^-1 = X X^T any more (the two took 5.5 ms there,
    // starved by the GEMM).  (X X^T as tasks of the same launch as well was built
    // and measured in round 5 -- the launch grew by what the separate 5.1-ms GEMM launch costs, 12.9 -> 18.9 ms at N = 10 000
    // fp64: the list leaves nothing idle to fill -- and removed again: EXPERIMENTS.md.)
    const int64_t prow = grad_out ? Npad + NB : Npad;
    if (c->N > 0 && panel_fits(Npad, prow)) {
        ALGP_TRY(ensure(c, c->auxW, sizeof(T) * prow * Npad));
        ALGP_TRY(ensure(c, c->auxA, sizeof(T) * Npad * Npad));
        ALGP_TRY(set_identity_launch<T>(c, p(c->auxW), Npad, Npad));
        Panel pn{p(c->auxW), Npad, prow, 2, false};
        pn.inv_out = grad_out ? p(c->auxA) : nullptr;
        if (prow > Npad) {
            ALGP_HIP(hipMemsetAsync(p(c->auxW) + Npad * Npad, 0, sizeof(T) * NB * Npad, c->stream));
            ALGP_HIP(hipMemcpyAsync(p(c->auxW) + Npad * Npad, c->y0.p, sizeof(T) * Npad, hipMemcpyDeviceToDevice, c->stream));
            pn.z_row = Npad;
        }
        const int frc = factorize(c, 0, &pn);
        if (pn.inv_enqueued && (frc != ALGP_OK || !grad_out)) hipStreamSynchronize(c->stream2);   // nothing outlives the call
        ALGP_TRY(frc);
        have_X = pn.done;
        inv_enq = pn.inv_enqueued;
    } else {
        ALGP_TRY(factorize(c, 0));
    }
    if (mll_out) *mll_out = -0.5 * c->yalpha - 0.5 * c->logdet - 0.5 * (double)c->N * 1.8378770664093453;
    if (!grad_out) return ALGP_OK;
    const int grc = mll_grad(c, grad_out, have_X, inv_enq);
    if (grc != ALGP_OK && inv_enq) hipStreamSynchronize(c->stream2);
    return grc;
}

template struct Impl<float>;
template struct Impl<double>;

}  // namespace algp

extern "C" {

int algp_get_mll_grad(algp_ctx* c, double* grad) {
    CHECK_CTX(c);
    if (!grad) return fail(c, ALGP_ERR_BAD_ARG, "get_mll_grad: bad arguments");
    FINISH(c, DISPATCH(c, mll_grad(c, grad)));
}

int algp_fit_step(algp_ctx* c, double* mll, double* grad) {
    CHECK_CTX(c);
    NEED_HYPERS(c);
    if (c->n_pool <= 0) return fail(c, ALGP_ERR_STATE, "fit_step: set a pool first");
    if (!c->y0.p || (int64_t)c->pos_in_train.size() != c->n_pool) return fail(c, ALGP_ERR_STATE, "fit_step: call algp_set_train first");
    FINISH(c, DISPATCH(c, fit_step(c, mll, grad)));
}

}  // extern "C"
