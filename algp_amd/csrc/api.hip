// api.hip -- extern "C" entry points of libalgp_hip.so (declared in include/algp_hip.h) and the
// host-side orchestration of the device kernels.  No torch types, no CPU fallback: every matrix, vector and
// per-candidate quantity comes from the HIP kernels in this directory; what the host computes is bookkeeping
// (index maps, the mean of the targets) and O(1) scalar combinations of values the kernels returned.
// (Split by concern in round 6: see api_impl.h.)
#include "api_impl.h"

using namespace algp;

namespace algp {

int fail(algp_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

int ensure(algp_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap && b.p) return ALGP_OK;
    if (bytes == 0) bytes = 256;
    if (b.p) {
        hipStreamSynchronize(c->stream);
        hipFree(b.p);
        c->dev_bytes -= (int64_t)b.cap;
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t want = (bytes + 255) / 256 * 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        (void)hipGetLastError();
        return fail(c, ALGP_ERR_OOM, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
    }
    b.cap = want;
    c->dev_bytes += (int64_t)want;
    return ALGP_OK;
}

void release(algp_ctx* c, DevBuf& b) {
    if (b.p) {
        hipFree(b.p);
        c->dev_bytes -= (int64_t)b.cap;
    }
    b.p = nullptr;
    b.cap = 0;
}

static hipEvent_t get_event(algp_ctx* c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

// Profiling: one event pair per scope on the ctx stream; flops/bytes are the ALGORITHMIC figures
// of the launch (DESIGN.md "measurement").  Nested scopes: only the outermost records.
static thread_local int prof_depth = 0;
void prof_begin(algp_ctx* c, int klass, double flops, double bytes) {
    if (!c->prof_on) return;
    if (prof_depth++ > 0) return;
    PendingEvent pe;
    pe.a = get_event(c);
    pe.b = get_event(c);
    pe.klass = klass;
    c->prof[klass].flops += flops;
    c->prof[klass].bytes += bytes;
    c->prof[klass].launches += 1;
    hipEventRecord(pe.a, c->cur);
    c->pending.push_back(pe);
}
void prof_end(algp_ctx* c) {
    if (!c->prof_on) return;
    if (--prof_depth > 0) return;
    hipEventRecord(c->pending.back().b, c->cur);
}
// A scope that is exactly one kernel launch: the two events ride on the dispatch itself (hipExtLaunchKernelGGL: its own begin
// and end timestamps, the figures rocprofv3 reports) instead of two hipEventRecord barrier packets around it on the stream --
// with ~470 GEMM launches per candidate solve those packets cost the step 2 % (bench.py: ms_per_step vs ms_per_step_unprofiled).
// False: profiling is off or an outer scope is open -- launch plainly.
bool prof_launch_events(algp_ctx* c, int klass, double flops, double bytes, hipEvent_t* a, hipEvent_t* b) {
    if (!c->prof_on || prof_depth > 0) return false;
    PendingEvent pe;
    pe.a = get_event(c);
    pe.b = get_event(c);
    pe.klass = klass;
    c->prof[klass].flops += flops;
    c->prof[klass].bytes += bytes;
    c->prof[klass].launches += 1;
    c->pending.push_back(pe);
    *a = pe.a;
    *b = pe.b;
    return true;
}
static thread_local int span_index = -1;
void prof_span_begin(algp_ctx* c, int klass, double flops, double bytes) {
    if (!c->prof_on) return;
    PendingEvent pe;
    pe.a = get_event(c);
    pe.b = get_event(c);
    pe.klass = klass;
    c->prof[klass].flops += flops;
    c->prof[klass].bytes += bytes;
    c->prof[klass].launches += 1;
    hipEventRecord(pe.a, c->stream);
    span_index = (int)c->pending.size();
    c->pending.push_back(pe);
}
void prof_span_end_on(algp_ctx* c, hipStream_t st) {
    if (!c->prof_on || span_index < 0) return;
    hipEventRecord(c->pending[span_index].b, st);
    span_index = -1;
}
void prof_span_end(algp_ctx* c) { prof_span_end_on(c, c->stream); }
static thread_local int span2_index = -1;
void prof_span_begin2(algp_ctx* c, int klass, double flops, double bytes) {
    if (!c->prof_on) return;
    PendingEvent pe;
    pe.a = get_event(c);
    pe.b = get_event(c);
    pe.klass = klass;
    c->prof[klass].flops += flops;
    c->prof[klass].bytes += bytes;
    c->prof[klass].launches += 1;
    hipEventRecord(pe.a, c->cur);
    span2_index = (int)c->pending.size();
    c->pending.push_back(pe);
}
void prof_span_end2(algp_ctx* c) {
    if (!c->prof_on || span2_index < 0) return;
    hipEventRecord(c->pending[span2_index].b, c->stream);
    span2_index = -1;
}
void prof_collect(algp_ctx* c) {
    for (auto& pe : c->pending) {
        hipEventSynchronize(pe.b);
        float ms = 0;
        hipEventElapsedTime(&ms, pe.a, pe.b);
        c->prof[pe.klass].ms += ms;
        c->event_pool.push_back(pe.a);
        c->event_pool.push_back(pe.b);
    }
    c->pending.clear();
}


hipEvent_t sync_event_api(algp_ctx* c, size_t i) {
    while (c->sync_events.size() <= i) {
        hipEvent_t e;
        hipEventCreateWithFlags(&e, hipEventDisableTiming);
        c->sync_events.push_back(e);
    }
    return c->sync_events[i];
}


KmatSrc make_src(algp_ctx* c) {
    KmatSrc s;
    s.Xs = c->Xs.p;
    s.Cp = c->pool_is_cov ? c->Cp.p : nullptr;
    s.n_pool = c->n_pool;
    s.DP = c->hyp.DP;
    s.kernel = c->hyp.kernel;
    s.outputscale = c->hyp.outputscale;
    s.noise = c->hyp.noise;
    return s;
}

int sync(algp_ctx* c) {
    ALGP_HIP(hipStreamSynchronize(c->stream));
    c->n_syncs++;
    return ALGP_OK;
}

// A synchronisation that also reads the sticky stall word (scal[SC_STALL]): one of the one-launch kernels that hand data
// between workgroups (forward / backward substitution, the solve-only task list) ran into its spin limit and abandoned
// its work since the word was last read -- whatever it was producing is incomplete.  The word is cleared again.
int sync_checked(algp_ctx* c, const char* what) {
    int stall = 0;
    int* d = (int*)((double*)c->scal.p + SC_STALL);
    ALGP_HIP(hipMemcpyAsync(&stall, d, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    ALGP_HIP(hipStreamSynchronize(c->stream));
    c->n_syncs++;
    if (stall != 0) {
        hipMemsetAsync(d, 0, sizeof(double), c->stream);
        return fail(c, ALGP_ERR_HIP, std::string(what) + ": a one-launch kernel stalled (a hand-off between its workgroups never "
                                     "arrived within the spin limit); the result is incomplete");
    }
    return ALGP_OK;
}

}  // namespace algp

// ---------------------------------------------------------------------------------------------
// typed implementation
// ---------------------------------------------------------------------------------------------

namespace algp {


template <typename T>
int Impl<T>::rescale_pool(algp_ctx* c) {
    if (c->pool_is_cov || c->n_pool == 0 || !c->hyp.set) return ALGP_OK;
    ALGP_TRY(ensure(c, c->Xs, sizeof(T) * c->n_pool * c->hyp.DP));
    return scale_coords_launch<T>(c, (const T*)c->Xraw.p, c->n_pool, p(c->Xs));
}


template <typename T>
int Impl<T>::kernel_matrix(algp_ctx* c, const void* x1, int64_t n1, const void* x2, int64_t n2, const void* diag_add,
                             int add_lik, void* out) {
    const int D = c->hyp.D, DP = c->hyp.DP;
    const bool sym = (x2 == nullptr);
    if (sym) n2 = n1;
    if (n1 == 0 || n2 == 0) return ALGP_OK;
    const int64_t ldo = round_up(n2, 4);
    DevBuf raw1, raw2, s1, s2, dv, o;
    int rc = ALGP_OK;
    auto cleanup = [&]() { release(c, raw1); release(c, raw2); release(c, s1); release(c, s2); release(c, dv); release(c, o); };
#define KM_TRY(x) do { rc = (x); if (rc != ALGP_OK) { cleanup(); return rc; } } while (0)
#define KM_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(c, ALGP_ERR_HIP, hipGetErrorString(e_)); } } while (0)
    KM_TRY(ensure(c, raw1, sizeof(T) * n1 * D));
    KM_TRY(ensure(c, s1, sizeof(T) * n1 * DP));
    KM_HIP(hipMemcpyAsync(raw1.p, x1, sizeof(T) * n1 * D, hipMemcpyHostToDevice, c->stream));
    KM_TRY(scale_coords_launch<T>(c, (const T*)raw1.p, n1, (T*)s1.p));
    if (!sym) {
        KM_TRY(ensure(c, raw2, sizeof(T) * n2 * D));
        KM_TRY(ensure(c, s2, sizeof(T) * n2 * DP));
        KM_HIP(hipMemcpyAsync(raw2.p, x2, sizeof(T) * n2 * D, hipMemcpyHostToDevice, c->stream));
        KM_TRY(scale_coords_launch<T>(c, (const T*)raw2.p, n2, (T*)s2.p));
    }
    if (sym && diag_add) {
        KM_TRY(ensure(c, dv, sizeof(T) * n1));
        KM_HIP(hipMemcpyAsync(dv.p, diag_add, sizeof(T) * n1, hipMemcpyHostToDevice, c->stream));
    }
    KM_TRY(ensure(c, o, sizeof(T) * n1 * ldo));
    KM_TRY(kmat_xy_launch<T>(c, (const T*)s1.p, n1, sym ? nullptr : (const T*)s2.p, n2, sym ? 1 : 0,
                             (sym && diag_add) ? (const T*)dv.p : nullptr, (sym && add_lik) ? c->hyp.noise : 0.0,
                             (T*)o.p, ldo));
    KM_HIP(hipMemcpy2DAsync(out, sizeof(T) * n2, o.p, sizeof(T) * ldo, sizeof(T) * n2, n1, hipMemcpyDeviceToHost,
                            c->stream));
    KM_HIP(hipStreamSynchronize(c->stream));
    cleanup();
#undef KM_TRY
#undef KM_HIP
    return ALGP_OK;
}


template <typename T>
int Impl<T>::set_pool(algp_ctx* c, const void* x, int64_t n) {
    const int D = c->hyp.D;
    ALGP_TRY(ensure(c, c->Xraw, sizeof(T) * n * D));
    ALGP_HIP(hipMemcpyAsync(c->Xraw.p, x, sizeof(T) * n * D, hipMemcpyHostToDevice, c->stream));
    c->n_pool = n;
    c->pool_is_cov = false;
    // a fingerprint per site (FNV-1a over its coordinates' bytes): lets algp_factorize_from check that a factor
    // adopted from another context was computed for the same COORDINATES, not only the same indices
    c->site_hash.assign((size_t)n, 0);
    const unsigned char* bytes = (const unsigned char*)x;
    const size_t stride = sizeof(T) * (size_t)D;
    for (int64_t i = 0; i < n; ++i) {
        uint64_t h = 1469598103934665603ull;
        for (size_t b = 0; b < stride; ++b) h = (h ^ bytes[(size_t)i * stride + b]) * 1099511628211ull;
        c->site_hash[(size_t)i] = h;
    }
    ALGP_TRY(rescale_pool(c));
    return sync(c);
}

// fingerprint of the coordinates of a train set, in row order (0 for an explicit-covariance pool)
template <typename T>
uint64_t Impl<T>::train_sites_hash(const algp_ctx* c, const std::vector<int64_t>& idx) {
    if (c->pool_is_cov || c->site_hash.empty()) return 0;
    uint64_t h = 1469598103934665603ull;
    for (int64_t i : idx) h = (h ^ c->site_hash[(size_t)i]) * 1099511628211ull;
    return h;
}


template <typename T>
int Impl<T>::set_pool_cov(algp_ctx* c, const void* cov, int64_t n) {
    ALGP_TRY(ensure(c, c->Cp, sizeof(T) * n * n));
    ALGP_HIP(hipMemcpyAsync(c->Cp.p, cov, sizeof(T) * n * n, hipMemcpyHostToDevice, c->stream));
    c->n_pool = n;
    c->pool_is_cov = true;
    c->site_hash.clear();
    return sync(c);
}


template <typename T>
int Impl<T>::set_train(algp_ctx* c, const int64_t* idx, int64_t N, const void* y, const void* var) {
    const int64_t Npad = round_up(std::max<int64_t>(N, 1), NB);
    c->N = N;
    c->Npad = Npad;
    c->train_idx.assign(idx, idx + N);
    c->pos_in_train.assign(c->n_pool, -1);
    c->train_has_repeats = false;
    for (int64_t i = 0; i < N; ++i) {                      // a site measured more than once: its FIRST row stands for it
        if (c->pos_in_train[idx[i]] < 0) c->pos_in_train[idx[i]] = i;
        else c->train_has_repeats = true;
    }
    double ybar = 0;
    const T* yt = (const T*)y;
    for (int64_t i = 0; i < N; ++i) ybar += (double)yt[i];
    ybar = N > 0 ? ybar / (double)N : 0.0;
    if (c->mean_override) ybar = c->mean_value;             // algp_set_constant_mean
    c->ybar = ybar;
    std::vector<T> y0(Npad, (T)0), vv(Npad, (T)0);
    for (int64_t i = 0; i < N; ++i) y0[i] = (T)((double)yt[i] - ybar);
    if (var)
        for (int64_t i = 0; i < N; ++i) vv[i] = ((const T*)var)[i];
    ALGP_TRY(ensure(c, c->Aidx, sizeof(int64_t) * Npad));
    ALGP_TRY(ensure(c, c->y0, sizeof(T) * Npad));
    ALGP_TRY(ensure(c, c->varA, sizeof(T) * Npad));
    if (N > 0) ALGP_HIP(hipMemcpyAsync(c->Aidx.p, idx, sizeof(int64_t) * N, hipMemcpyHostToDevice, c->stream));
    ALGP_HIP(hipMemcpyAsync(c->y0.p, y0.data(), sizeof(T) * Npad, hipMemcpyHostToDevice, c->stream));
    std::vector<T> yr(Npad, (T)0);
    for (int64_t i = 0; i < N; ++i) yr[i] = yt[i];
    ALGP_TRY(ensure(c, c->yraw, sizeof(T) * Npad));
    ALGP_HIP(hipMemcpyAsync(c->yraw.p, yr.data(), sizeof(T) * Npad, hipMemcpyHostToDevice, c->stream));
    c->train_y_host.assign(N, 0.0);
    for (int64_t i = 0; i < N; ++i) c->train_y_host[i] = (double)yt[i];
    ALGP_HIP(hipMemcpyAsync(c->varA.p, vv.data(), sizeof(T) * Npad, hipMemcpyHostToDevice, c->stream));
    ALGP_TRY(sync(c));   // host vectors go out of scope
    c->train_var_host.assign(N, 0.0);
    if (var)
        for (int64_t i = 0; i < N; ++i) c->train_var_host[i] = (double)((const T*)var)[i];
    c->train_dirty = true;       // the resident factor (if any) still describes fact_idx / fact_var
    c->solved = false;
    if (c->comm || c->host_gather) ALGP_TRY(comm_reserve(c));   // the exchange's buffers follow the train set's size
    return ALGP_OK;
}


// ------------------------------------------------------------------ set entropies / inverse diagonals
template <typename T>
int Impl<T>::build_set_matrix(algp_ctx* c, const int64_t* idx, int64_t m, const void* var, int64_t* mpad_out, T* dst) {
    const int64_t mpad = round_up(std::max<int64_t>(m, 1), NB);
    *mpad_out = mpad;
    if (!dst) ALGP_TRY(ensure(c, c->auxA, sizeof(T) * mpad * mpad));
    ALGP_TRY(ensure(c, c->auxInv, sizeof(T) * mpad * NB));
    ALGP_TRY(ensure(c, c->auxIdx, sizeof(int64_t) * mpad));
    ALGP_TRY(ensure(c, c->auxVar, sizeof(T) * mpad));
    if (m > 0) ALGP_HIP(hipMemcpyAsync(c->auxIdx.p, idx, sizeof(int64_t) * m, hipMemcpyHostToDevice, c->stream));
    if (var && m > 0)
        ALGP_HIP(hipMemcpyAsync(c->auxVar.p, var, sizeof(T) * m, hipMemcpyHostToDevice, c->stream));
    KmatSrc s = make_src(c);
    return kmat_launch<T>(c, s, (const int64_t*)c->auxIdx.p, m, mpad, (const int64_t*)c->auxIdx.p, m, mpad,
                          var ? (const T*)c->auxVar.p : nullptr, c->pool_is_cov ? 0 : 1, nullptr, 1, dst ? dst : p(c->auxA),
                          mpad);
}


template <typename T>
int Impl<T>::set_entropy(algp_ctx* c, const int64_t* idx, int64_t m, const void* var, double* H) {
    if (m == 0) { *H = 0.0; return ALGP_OK; }
    int64_t mpad;
    ALGP_TRY(build_set_matrix(c, idx, m, var, &mpad));
    double ld = 0;
    ALGP_TRY(factor_resident(c, p(c->auxA), m, mpad, p(c->auxInv), SC_AUXLOGDET, SC_AUXINFO, &ld));
    *H = (double)m * ENT_CONST + 0.5 * ld;
    return ALGP_OK;
}


// diag(S^-1) = row sums of squares of L^-T (the triangular inverse on the MFMA GEMM)
template <typename T>
int Impl<T>::inverse_diag_resident(algp_ctx* c, int64_t m, int64_t mpad, void* diag_out) {
    ALGP_TRY(ensure(c, c->auxW, sizeof(T) * mpad * mpad));
    ALGP_TRY(ensure(c, c->auxD, sizeof(T) * mpad));
    ALGP_TRY(set_identity_launch<T>(c, p(c->auxW), mpad, mpad));
    ALGP_TRY(trinv_upper<T>(c, ALGP_PROF_GEMM_OTHER, p(c->auxW), mpad, mpad, p(c->auxA), mpad, p(c->auxInv)));
    ALGP_TRY(rows_reduce_launch<T>(c, p(c->auxW), m, mpad, mpad, (const T*)nullptr, p(c->auxD), (T*)nullptr));
    ALGP_HIP(hipMemcpyAsync(diag_out, c->auxD.p, sizeof(T) * m, hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}


template <typename T>
int Impl<T>::set_inverse_diag(algp_ctx* c, const int64_t* idx, int64_t m, const void* var, void* diag_out, double* H) {
    if (m == 0) { if (H) *H = 0.0; return ALGP_OK; }
    int64_t mpad;
    ALGP_TRY(build_set_matrix(c, idx, m, var, &mpad));
    double ld = 0;
    ALGP_TRY(factor_resident(c, p(c->auxA), m, mpad, p(c->auxInv), SC_AUXLOGDET, SC_AUXINFO, &ld));
    if (H) *H = (double)m * ENT_CONST + 0.5 * ld;
    return inverse_diag_resident(c, m, mpad, diag_out);
}


template <typename T>
int Impl<T>::upload_padded(algp_ctx* c, DevBuf& b, const void* A, int64_t rows, int64_t cols, int64_t rpad,
                             int64_t cpad) {
    ALGP_TRY(ensure(c, b, sizeof(T) * rpad * cpad));
    ALGP_HIP(hipMemsetAsync(b.p, 0, sizeof(T) * rpad * cpad, c->stream));
    if (rows > 0 && cols > 0)
        ALGP_HIP(hipMemcpy2DAsync(b.p, sizeof(T) * cpad, A, sizeof(T) * cols, sizeof(T) * cols, rows,
                                  hipMemcpyHostToDevice, c->stream));
    return ALGP_OK;
}


template <typename T>
int Impl<T>::entropy_from_cov(algp_ctx* c, const void* cov, int64_t k, double* H, void* L_out, double* logdet) {
    if (k == 0) { if (H) *H = 0.0; if (logdet) *logdet = 0.0; return ALGP_OK; }
    const int64_t kpad = round_up(k, NB);
    ALGP_TRY(upload_padded(c, c->auxA, cov, k, k, kpad, kpad));
    ALGP_TRY(pad_identity_launch<T>(c, p(c->auxA), k, kpad, kpad));
    ALGP_TRY(ensure(c, c->auxInv, sizeof(T) * kpad * NB));
    double ld = 0;
    ALGP_TRY(factor_resident(c, p(c->auxA), k, kpad, p(c->auxInv), SC_AUXLOGDET, SC_AUXINFO, &ld));
    if (H) *H = (double)k * ENT_CONST + 0.5 * ld;
    if (logdet) *logdet = ld;
    if (L_out) {
        ALGP_HIP(hipMemcpy2DAsync(L_out, sizeof(T) * k, c->auxA.p, sizeof(T) * kpad, sizeof(T) * k, k,
                                  hipMemcpyDeviceToHost, c->stream));
        ALGP_TRY(sync(c));
        T* Lh = (T*)L_out;
        for (int64_t i = 0; i < k; ++i)
            for (int64_t j = i + 1; j < k; ++j) Lh[i * k + j] = (T)0;
    }
    return ALGP_OK;
}


template <typename T>
int Impl<T>::gemm_host(algp_ctx* c, int64_t m, int64_t n, int64_t k, double alpha, const void* A, const void* B,
                         double beta, const void* C, void* D) {
    const int64_t mp = round_up(std::max<int64_t>(m, 1), NB), np = round_up(std::max<int64_t>(n, 1), NB),
                  kp = round_up(std::max<int64_t>(k, 1), NB);
    DevBuf a, b, cc;
    int rc = upload_padded(c, a, A, m, k, mp, kp);
    if (rc == ALGP_OK) rc = upload_padded(c, b, B, n, k, np, kp);
    if (rc == ALGP_OK) rc = upload_padded(c, cc, (beta != 0.0 && C) ? C : nullptr, (beta != 0.0 && C) ? m : 0, n, mp, np);
    if (rc == ALGP_OK)
        rc = gemm_nt_launch<T>(c, ALGP_PROF_GEMM_OTHER, mp, np, kp, (T)alpha, (const T*)a.p, kp, (const T*)b.p, kp,
                               (T)beta, (const T*)cc.p, np, (T*)cc.p, np, 0);
    if (rc == ALGP_OK) {
        hipError_t e = hipMemcpy2DAsync(D, sizeof(T) * n, cc.p, sizeof(T) * np, sizeof(T) * n, m,
                                        hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) rc = fail(c, ALGP_ERR_HIP, hipGetErrorString(e));
    }
    hipStreamSynchronize(c->stream);
    release(c, a); release(c, b); release(c, cc);
    return rc;
}


template <typename T>
int Impl<T>::trsm_host(algp_ctx* c, const void* L, int64_t n, const void* B, int64_t m, void* X) {
    const int64_t np = round_up(std::max<int64_t>(n, 1), NB), mp = round_up(std::max<int64_t>(m, 1), NB);
    DevBuf l, b, inv;
    int rc = upload_padded(c, l, L, n, n, np, np);
    if (rc == ALGP_OK) rc = pad_identity_launch<T>(c, (T*)l.p, n, np, np);
    if (rc == ALGP_OK) rc = upload_padded(c, b, B, m, n, mp, np);
    if (rc == ALGP_OK) rc = ensure(c, inv, sizeof(T) * np * NB);
    for (int64_t kb = 0; rc == ALGP_OK && kb < np / NB; ++kb)
        rc = trinv_diag_launch<T>(c, (const T*)l.p + kb * NB * np + kb * NB, np, (T*)inv.p + kb * NB * NB);
    if (rc == ALGP_OK) rc = trsm_blocked<T>(c, ALGP_PROF_GEMM_OTHER, (T*)b.p, mp, np, (const T*)l.p, np, np, (const T*)inv.p);
    if (rc == ALGP_OK) {
        hipError_t e = hipMemcpy2DAsync(X, sizeof(T) * n, b.p, sizeof(T) * np, sizeof(T) * n, m,
                                        hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) rc = fail(c, ALGP_ERR_HIP, hipGetErrorString(e));
    }
    hipStreamSynchronize(c->stream);
    release(c, l); release(c, b); release(c, inv);
    return rc;
}

template <typename T>
int Impl<T>::selftest(algp_ctx* c, int* mism) {
    double* sc = (double*)c->scal.p;
    int* d = (int*)(sc + SC_PROBE);
    ALGP_HIP(hipMemsetAsync(d, 0, sizeof(double), c->stream));
    ALGP_TRY(test_mfma_launch<double>(c, d));
    ALGP_TRY(test_mfma_launch<float>(c, d));
    ALGP_HIP(hipMemcpyAsync(mism, d, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}

template struct Impl<float>;
template struct Impl<double>;

}  // namespace algp

extern "C" {

int algp_version(void) { return 100; }

int algp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int algp_create(int device_id, int dtype, algp_ctx** out) {
    if (!out || (dtype != ALGP_F32 && dtype != ALGP_F64)) return ALGP_ERR_BAD_ARG;
    *out = nullptr;
    // more hardware queues than the default 4 so the helper streams do not share one with RCCL / torch
    // streams of the same process (no effect if the HIP runtime is already initialised)
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return ALGP_ERR_NO_DEVICE; }
    if (device_id < 0 || device_id >= n) return ALGP_ERR_BAD_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return ALGP_ERR_HIP;
    algp_ctx* c = new algp_ctx();
    c->device = device_id;
    c->dtype = dtype;
    c->es = dtype == ALGP_F64 ? 8 : 4;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return ALGP_ERR_HIP; }
    if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) { hipStreamDestroy(c->stream); delete c; return ALGP_ERR_HIP; }
    if (hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking) != hipSuccess) c->stream3 = nullptr;
    if (hipStreamCreateWithFlags(&c->stream4, hipStreamNonBlocking) != hipSuccess) c->stream4 = nullptr;
    c->trsm_chunks = env_int("ALGP_TRSM_CHUNKS", c->trsm_chunks);
    c->cur = c->stream;
    if (ensure(c, c->scal, sizeof(double) * SC_COUNT) != ALGP_OK) { hipStreamDestroy(c->stream); delete c; return ALGP_ERR_OOM; }
    hipMemsetAsync(c->scal.p, 0, sizeof(double) * SC_COUNT, c->stream);
    hipStreamSynchronize(c->stream);
    *out = c;
    return ALGP_OK;
}

void algp_destroy(algp_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    prof_collect(c);
    DevBuf* bufs[] = {&c->Xs, &c->Xraw, &c->Cp, &c->Aidx, &c->yA, &c->varA, &c->y0, &c->L, &c->invD, &c->z, &c->alpha,
                      &c->scal, &c->Cidx, &c->ckind, &c->cextra, &c->Vt, &c->dstat, &c->mu, &c->alive, &c->scores,
                      &c->lrow, &c->remote, &c->commbuf, &c->tvec, &c->amax, &c->prevrows, &c->fresh, &c->lazypicks, &c->yraw, &c->uvec, &c->wvec, &c->acc3, &c->rowstat, &c->inv512, &c->inv512_scr, &c->trsm_tmp, &c->splitk, &c->dag_state, &c->dag_stats, &c->trsv_ctrl, &c->miXbar, &c->miXall, &c->miDP, &c->miDQ, &c->miPos, &c->miU, &c->miW, &c->miCol, &c->miH, &c->auxA, &c->auxInv, &c->auxW, &c->auxIdx,
                      &c->auxVar, &c->auxD, &c->hostStage, &c->rowx, &c->tailE, &c->tailPart, &c->ldpart};
    for (DevBuf* b : bufs) release(c, *b);
    dag_release(c);
    comm_destroy(c);
    for (hipEvent_t e : c->event_pool) hipEventDestroy(e);
    for (hipEvent_t e : c->sync_events) hipEventDestroy(e);
    hipStreamSynchronize(c->stream2);
    hipStreamDestroy(c->stream2);
    if (c->stream3) { hipStreamSynchronize(c->stream3); hipStreamDestroy(c->stream3); }
    if (c->stream4) { hipStreamSynchronize(c->stream4); hipStreamDestroy(c->stream4); }
    hipStreamDestroy(c->stream);
    delete c;
}

const char* algp_last_error(const algp_ctx* c) { return c ? c->err.c_str() : "null context"; }

int64_t algp_last_pivot(const algp_ctx* c) { return c ? c->pivot : 0; }

double algp_last_jitter(const algp_ctx* c) { return c ? c->last_jitter : 0.0; }

int algp_dtype(const algp_ctx* c) { return c ? c->dtype : -1; }

int algp_set_hypers(algp_ctx* c, int kernel, int D, const double* log_ls, double log_os, double log_noise) {
    CHECK_CTX(c);
    if (D < 1 || D > MAXD || !log_ls) return fail(c, ALGP_ERR_BAD_ARG, "set_hypers: 1 <= D <= 8 required");
    if (kernel != ALGP_KERNEL_RBF && kernel != ALGP_KERNEL_MATERN15) return fail(c, ALGP_ERR_BAD_ARG, "set_hypers: unknown kernel");
    if (c->hyp.set && c->hyp.D != D && c->n_pool > 0 && !c->pool_is_cov) {
        // the resident coordinates have another width: drop the pool, the caller sets a new one
        c->n_pool = 0;
        c->N = 0;
        c->pos_in_train.clear();
    }
    c->hyp.kernel = kernel;
    c->hyp.D = D;
    c->hyp.DP = D <= 2 ? 2 : (D <= 4 ? 4 : 8);
    for (int d = 0; d < MAXD; ++d) c->hyp.inv_ls[d] = d < D ? exp(-log_ls[d]) : 0.0;
    c->hyp.outputscale = exp(log_os);
    c->hyp.noise = exp(log_noise);
    c->hyp.set = true;
    c->hyp_stamp++;
    c->factored = false;
    c->solved = false;
    FINISH(c, DISPATCH(c, rescale_pool(c)));
}

int algp_kernel_matrix(algp_ctx* c, const void* x1, int64_t n1, const void* x2, int64_t n2, const void* diag_add,
                       int add_lik, void* out) {
    CHECK_CTX(c);
    NEED_HYPERS(c);
    if (!x1 || n1 < 0 || n2 < 0 || !out) return fail(c, ALGP_ERR_BAD_ARG, "kernel_matrix: bad arguments");
    FINISH(c, DISPATCH(c, kernel_matrix(c, x1, n1, x2, n2, diag_add, add_lik, out)));
}

int algp_set_pool(algp_ctx* c, const void* x, int64_t n) {
    CHECK_CTX(c);
    NEED_HYPERS(c);
    if (!x || n <= 0) return fail(c, ALGP_ERR_BAD_ARG, "set_pool: bad arguments");
    c->factored = c->solved = false;
    c->hyp_stamp++;
    c->pos_in_train.clear();
    c->N = 0;
    c->site_owner.clear();                                   // the owner map belongs to the pool it was given for (re-attach it: algp_comm_set_owners)
    c->site_owner_hash = 0;
    FINISH(c, DISPATCH(c, set_pool(c, x, n)));
}

int algp_set_pool_cov(algp_ctx* c, const void* cov, int64_t n) {
    CHECK_CTX(c);
    NEED_HYPERS(c);
    if (!cov || n <= 0) return fail(c, ALGP_ERR_BAD_ARG, "set_pool_cov: bad arguments");
    c->factored = c->solved = false;
    c->hyp_stamp++;
    c->pos_in_train.clear();
    c->N = 0;
    c->site_owner.clear();                                   // the owner map belongs to the pool it was given for (re-attach it: algp_comm_set_owners)
    c->site_owner_hash = 0;
    FINISH(c, DISPATCH(c, set_pool_cov(c, cov, n)));
}

int algp_set_train(algp_ctx* c, const int64_t* idx, int64_t N, const void* y, const void* var) {
    CHECK_CTX(c);
    if (c->n_pool <= 0) return fail(c, ALGP_ERR_STATE, "set_train: set a pool first");
    if (N < 0 || (N > 0 && (!idx || !y))) return fail(c, ALGP_ERR_BAD_ARG, "set_train: bad arguments");
    for (int64_t i = 0; i < N; ++i)
        if (idx[i] < 0 || idx[i] >= c->n_pool) return fail(c, ALGP_ERR_BAD_ARG, "set_train: index outside the pool");
    FINISH(c, DISPATCH(c, set_train(c, idx, N, y, var)));
}

int algp_set_constant_mean(algp_ctx* c, int enable, double value) {
    CHECK_CTX(c);
    if (enable && !(value == value)) return fail(c, ALGP_ERR_BAD_ARG, "set_constant_mean: NaN");
    c->mean_override = enable != 0;
    c->mean_value = value;
    return ALGP_OK;
}

#if ALGP_TEST_HOOKS
int64_t algp_debug_counter(algp_ctx* c, int which) {
    if (!c) return -1;
    switch (which) {
        case 0: return c->n_syncs;
        case 1: return c->factor_rows_from_vt;
        case 2: return c->rows_from_peers;
        case 3: return c->row_exchanges;
        case 4: return c->row_fallbacks;
        default: return -1;
    }
}
#endif

#if ALGP_TEST_HOOKS
int algp_debug_set_trsm_chunks(algp_ctx* c, int chunks) {
    CHECK_CTX(c);
    if (chunks < 0 || chunks > 4) return fail(c, ALGP_ERR_BAD_ARG, "debug_set_trsm_chunks: 0 (default) .. 4");
    hipStreamSynchronize(c->stream);
    if (chunks == 0) chunks = env_int("ALGP_TRSM_CHUNKS", 3);
    c->trsm_chunks = chunks;
    return ALGP_OK;
}
#endif

#if ALGP_TEST_HOOKS
int algp_debug_trsv_stall(algp_ctx* c, int block) {
    CHECK_CTX(c);
    if (block < -1) return fail(c, ALGP_ERR_BAD_ARG, "debug_trsv_stall: a block >= 0, or -1 to disarm");
    c->debug_trsv_stall_block = block;
    return ALGP_OK;
}
#endif

#if ALGP_TEST_HOOKS
int algp_debug_dag_stall(algp_ctx* c, int ticket) {
    CHECK_CTX(c);
    if (ticket < -1) return fail(c, ALGP_ERR_BAD_ARG, "debug_dag_stall: a ticket >= 0, or -1 to disarm");
    c->debug_dag_stall_ticket = ticket;
    return ALGP_OK;
}
#endif

int algp_entropy_from_cov(algp_ctx* c, const void* cov, int64_t k, double* H) {
    CHECK_CTX(c);
    if (k < 0 || !H || (k > 0 && !cov)) return fail(c, ALGP_ERR_BAD_ARG, "entropy_from_cov: bad arguments");
    FINISH(c, DISPATCH(c, entropy_from_cov(c, cov, k, H, nullptr, nullptr)));
}

int algp_set_entropy(algp_ctx* c, const int64_t* idx, int64_t m, const void* var, double* H) {
    CHECK_CTX(c);
    NEED_HYPERS(c);
    if (m < 0 || !H || (m > 0 && !idx)) return fail(c, ALGP_ERR_BAD_ARG, "set_entropy: bad arguments");
    for (int64_t i = 0; i < m; ++i)
        if (idx[i] < 0 || idx[i] >= c->n_pool) return fail(c, ALGP_ERR_BAD_ARG, "set_entropy: index outside the pool");
    FINISH(c, DISPATCH(c, set_entropy(c, idx, m, var, H)));
}

int algp_set_inverse_diag(algp_ctx* c, const int64_t* idx, int64_t m, const void* var, void* diag_out, double* H) {
    CHECK_CTX(c);
    NEED_HYPERS(c);
    if (m < 0 || (m > 0 && (!idx || !diag_out))) return fail(c, ALGP_ERR_BAD_ARG, "set_inverse_diag: bad arguments");
    for (int64_t i = 0; i < m; ++i)
        if (idx[i] < 0 || idx[i] >= c->n_pool) return fail(c, ALGP_ERR_BAD_ARG, "set_inverse_diag: index outside the pool");
    FINISH(c, DISPATCH(c, set_inverse_diag(c, idx, m, var, diag_out, H)));
}

int algp_cholesky(algp_ctx* c, const void* A, int64_t n, void* L_out, double* logdet) {
    CHECK_CTX(c);
    if (n < 0 || (n > 0 && !A)) return fail(c, ALGP_ERR_BAD_ARG, "cholesky: bad arguments");
    FINISH(c, DISPATCH(c, entropy_from_cov(c, A, n, nullptr, L_out, logdet)));
}

int algp_gemm_nt(algp_ctx* c, int64_t m, int64_t n, int64_t k, double alpha, const void* A, const void* B, double beta,
                 const void* C, void* D) {
    CHECK_CTX(c);
    if (m <= 0 || n <= 0 || k <= 0 || !A || !B || !D) return fail(c, ALGP_ERR_BAD_ARG, "gemm_nt: bad arguments");
    FINISH(c, DISPATCH(c, gemm_host(c, m, n, k, alpha, A, B, beta, C, D)));
}

int algp_trsm_right_lt(algp_ctx* c, const void* L, int64_t n, const void* B, int64_t m, void* X) {
    CHECK_CTX(c);
    if (m <= 0 || n <= 0 || !L || !B || !X) return fail(c, ALGP_ERR_BAD_ARG, "trsm: bad arguments");
    FINISH(c, DISPATCH(c, trsm_host(c, L, n, B, m, X)));
}

int algp_selftest_mfma(algp_ctx* c, int* mismatches) {
    CHECK_CTX(c);
    if (!mismatches) return fail(c, ALGP_ERR_BAD_ARG, "selftest: bad arguments");
    return Impl<double>::selftest(c, mismatches);
}

int algp_bench_gemm(algp_ctx* c, int64_t m, int64_t n, int64_t k, int lower_only, int beta_one, int reps,
                    double* ms) {
    CHECK_CTX(c);
    if (!ms || m <= 0 || n <= 0 || k <= 0 || reps <= 0) return fail(c, ALGP_ERR_BAD_ARG, "bench_gemm: bad arguments");
    const bool was = c->prof_on;
    c->prof_on = false;
    int rc = c->dtype == ALGP_F64 ? bench_gemm<double>(c, m, n, k, lower_only, beta_one, reps, ms)
                                  : bench_gemm<float>(c, m, n, k, lower_only, beta_one, reps, ms);
    c->prof_on = was;
    return rc;
}

int algp_sync(algp_ctx* c) { CHECK_CTX(c); return sync(c); }

int64_t algp_device_bytes(const algp_ctx* c) { return c ? c->dev_bytes : 0; }

int algp_prof_enable(algp_ctx* c, int on) {
    CHECK_CTX(c);
    hipStreamSynchronize(c->stream);
    prof_collect(c);
    c->prof_on = on != 0;
    return ALGP_OK;
}

int algp_prof_reset(algp_ctx* c) {
    CHECK_CTX(c);
    hipStreamSynchronize(c->stream);
    prof_collect(c);
    for (int i = 0; i < ALGP_PROF_COUNT; ++i) c->prof[i] = ProfSlot();
    if (c->dag_stats.p) ALGP_HIP(hipMemset(c->dag_stats.p, 0, 64));
    return ALGP_OK;
}

int algp_cholesky_task_stats(algp_ctx* c, double out[4]) {
    CHECK_CTX(c);
    if (!out) return fail(c, ALGP_ERR_BAD_ARG, "cholesky_task_stats: null output");
    unsigned long long h[4] = {0, 0, 0, 0};
    if (c->dag_stats.p) {
        ALGP_HIP(hipStreamSynchronize(c->stream));
        ALGP_HIP(hipMemcpy(h, c->dag_stats.p, sizeof(h), hipMemcpyDeviceToHost));
    }
    out[0] = (double)h[0] * 0.01;                              // 100 MHz ticks -> microseconds
    out[1] = (double)h[1];
    out[2] = (double)h[2] * 0.01;
    out[3] = (double)h[3];
    return ALGP_OK;
}

int algp_prof_get(algp_ctx* c, int klass, double* ms, double* flops, double* bytes, int64_t* launches) {
    CHECK_CTX(c);
    if (klass < 0 || klass >= ALGP_PROF_COUNT) return fail(c, ALGP_ERR_BAD_ARG, "prof_get: unknown class");
    hipStreamSynchronize(c->stream);
    prof_collect(c);
    if (ms) *ms = c->prof[klass].ms;
    if (flops) *flops = c->prof[klass].flops;
    if (bytes) *bytes = c->prof[klass].bytes;
    if (launches) *launches = c->prof[klass].launches;
    return ALGP_OK;
}

}  // extern "C"

