// The host-side list schedule of the one-launch Cholesky (algp_amd/csrc/chol_dag.hip) on its own.  Needs no GPU.
//   build/dag_sched_probe 79                 critical path of the task graph, simulated makespan, utilisation and chain
//                                            progress per 500 us (build with -DALGP_DAG_DEBUG for the report)
//   build/dag_sched_probe --check LO HI [W]  for every N/128 in [LO, HI] and several worker counts: replay the ticket
//                                            list IN LIST ORDER on one imaginary bulk worker beside the chain team and check
//                                            that every task finds its inputs produced by earlier tickets or by the team,
//                                            that every tile receives its updates exactly once in ascending k, and that the
//                                            factorisation is complete at the end.  A list that passes cannot deadlock at
//                                            any residency: the lowest unfinished ticket can always run.
// hipcc --offload-arch=gfx950 -O2 -std=c++17 -w [-DALGP_DAG_DEBUG] tools/dag_sched_probe.hip -o build/dag_sched_probe
#include "../algp_amd/csrc/chol_dag.hip"
#include <string.h>
namespace algp {
int fail(algp_ctx*, int code, const std::string&) { return code; }
void prof_begin(algp_ctx*, int, double, double) {}
void prof_end(algp_ctx*) {}
int ensure(algp_ctx*, DevBuf&, size_t) { return 0; }
}
using namespace algp;

// what the kernel's waits and publishes do to the tile versions, restated independently of dag_build_schedule's graph
struct Replay {
    int nt;
    DagShape sh;
    std::vector<int> ver;
    int lead_k = 0, help_k = 0, help_phase = 0;
    // the initial versions are what dag_launch's template holds: 0, a final factor for a solve-only list, and for a
    // panel row its first column
    explicit Replay(const DagShape& s) : nt(s.nt), sh(s), ver((size_t)(s.nt + s.mt) * s.nt, 0) {
        if (s.solve_only) {
            for (int i = 0; i < nt; ++i)
                for (int j = 0; j <= i; ++j) v(i, j) = j + 1;
            lead_k = nt;
            help_k = nt;
        }
        for (int r = 0; r < s.mt; ++r)
            for (int j = s.pstart(r); j < nt; ++j) v(nt + r, j) = s.pstart(r);
    }
    int& v(int i, int j) { return ver[(size_t)i * nt + j]; }
    // the chain team advances as far as its inputs allow (leader: diagonal blocks; helpers: strips, in program order)
    void team() {
        for (bool moved = true; moved;) {
            moved = false;
            if (lead_k < nt && v(lead_k, lead_k) >= lead_k) {
                if (v(lead_k, lead_k) != lead_k) { printf("diag %d factored at version %d\n", lead_k, v(lead_k, lead_k)); exit(2); }
                v(lead_k, lead_k) = lead_k + 1;
                ++lead_k;
                moved = true;
            }
            while (help_k + 1 < nt) {
                const int k = help_k;
                bool ok = false;
                switch (help_phase) {
                    case 0: ok = v(k, k) >= k + 1 && v(k + 1, k) == k; if (ok) v(k + 1, k) = k + 1; break;           // TRSM(k+1,k)
                    case 1: ok = v(k + 1, k + 1) == k; if (ok) v(k + 1, k + 1) = k + 1; break;                       // UPD(k+1,k+1,k)
                    case 2: ok = k + 2 >= nt || v(k + 2, k) == k; if (ok && k + 2 < nt) v(k + 2, k) = k + 1; break;  // TRSM(k+2,k)
                    case 3: ok = k + 2 >= nt || v(k + 2, k + 1) == k; if (ok && k + 2 < nt) v(k + 2, k + 1) = k + 1; break;
                    case 4: ok = k + 2 >= nt || v(k + 2, k + 2) == k; if (ok && k + 2 < nt) v(k + 2, k + 2) = k + 1; break;
                }
                if (!ok) break;
                moved = true;
                if (++help_phase == 5) { help_phase = 0; ++help_k; }
            }
        }
    }
    bool run(const DagTask& t, int ticket) {
        const int i = t.i, j = t.j, k0 = t.kk >> 16, k1 = t.kk & 0xffff;
        auto bad = [&](const char* why) {
            printf("nt %d ticket %d type %d (%d,%d) k %d..%d: %s\n", nt, ticket, t.type, i, j, k0, k1, why);
            return false;
        };
        if (i >= nt + sh.mt || j >= nt) return bad("tile outside the matrix");
        if (i >= nt && (j < sh.pstart(i - nt) || k0 < sh.pstart(i - nt))) return bad("a panel row touched left of its first column");
        if (i >= nt && j >= sh.pcols(i - nt)) return bad("a short panel row touched its last column tile");
        if (t.type == DAG_TU) {
            // TRSM(i,k), then -- a wait in the middle of the task, the workgroup held -- UPD(i,k+1,k)
            const int k = j;
            if (k + 1 >= (i >= nt ? sh.pcols(i - nt) : nt)) return bad("TU in the row's last column");
            if (v(i, k) != k) return bad("tile (i,k) is not at version k");
            if (v(k, k) < k + 1) return bad("diagonal block not factored");
            v(i, k) = k + 1;
            team();
            if (v(i, k + 1) != k) return bad("second half: tile (i,k+1) is not at version k (its producer holds a LATER ticket)");
            if (v(k + 1, k) < k + 1) return bad("second half: L_(k+1)k not solved");
            v(i, k + 1) = k + 1;
        } else if (t.type == DAG_TRSM) {
            if (v(i, j) != j) return bad("tile is not at version k");
            if (v(j, j) < j + 1) return bad("diagonal block not factored");
            v(i, j) = j + 1;
        } else if (t.type == DAG_UPD) {
            if (v(i, j) != k0) return bad("tile is not at version k0");
            for (int kk = k0; kk < k1; ++kk) {
                if (v(i, kk) < kk + 1) return bad("L_i,kk not solved");
                if (v(j, kk) < kk + 1) return bad("L_j,kk not solved");
            }
            v(i, j) = k1;
        } else {
            return bad("a team task carries a ticket");
        }
        return true;
    }
};

static bool check(const DagShape& sh, int W, int workers) {
    const int nt = sh.nt;
    DagSchedule s;
    dag_build_schedule(sh, W, workers, s);
    Replay r(sh);
    r.team();
    for (size_t t = 0; t < s.tasks.size(); ++t) {
        if (!r.run(s.tasks[t], (int)t)) return false;
        r.team();
    }
    for (int i = 0; i < nt + sh.mt; ++i)
        for (int j = (i < nt ? 0 : sh.pstart(i - nt)); j <= (i < nt ? i : nt - 1); ++j)
            if (r.v(i, j) != ((i >= nt && j >= sh.pcols(i - nt)) ? 0 : j + 1)) {         // a short panel row never touches its last column tile
                printf("nt %d mt %d mode %d solve_only %d workers %d: tile (%d,%d) ends at version %d\n", nt, sh.mt, sh.mode, (int)sh.solve_only,
                       workers, i, j, r.v(i, j));
                return false;
            }
    return true;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "--check")) {
        // W = 2 is what dag_launch ships
        const int lo = argc > 2 ? atoi(argv[2]) : 8, hi = argc > 3 ? atoi(argv[3]) : 192, W = argc > 4 ? atoi(argv[4]) : 2;
        int n = 0;
        for (int nt = lo; nt <= hi; ++nt)
            for (int workers : {512, 64, 2 * DAG_TEAM + 1}) {
                DagShape sh;
                sh.nt = nt;
                if (!check(sh, W, workers)) return 1;
                ++n;
            }
        printf("CHECK OK: %d schedules (N/128 = %d..%d, W = %d)\n", n, lo, hi, W);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "--check-panel")) {
        // lists that carry a row panel: dense rows (mode 1), the identity (mode 2), each with and without the factorisation
        const int lo = argc > 2 ? atoi(argv[2]) : 8, hi = argc > 3 ? atoi(argv[3]) : 40, W = 2;
        int n = 0;
        for (int nt = lo; nt <= hi; ++nt)
            for (int mode : {1, 2})
                for (int so : {0, 1})
                    for (int mt : {1, 3, nt, nt + 1, 2 * nt + 5}) {
                        // mode 2: the identity alone, or followed by one dense tile row (the row that carries y - ybar: algp_fit_step)
                        if (mode == 2 && mt != nt && mt != nt + 1) continue;
                        if (mode == 1 && mt == nt + 1) continue;
                        for (int workers : {512, 37, 2 * DAG_TEAM + 1}) {
                            DagShape sh;
                            sh.nt = nt; sh.mt = mt; sh.mode = mode; sh.solve_only = so != 0;
                            if (!check(sh, W, workers)) return 1;
                            ++n;
                            // round 6: dense panels whose tile rows (all, or all but the one that carries z) leave the last column tile out
                            if (mode == 1)
                                for (int pshort : {mt, mt - 1}) {
                                    if (pshort < 1) continue;
                                    sh.pshort = pshort;
                                    if (!check(sh, W, workers)) return 1;
                                    ++n;
                                }
                        }
                    }
        printf("CHECK OK: %d panel schedules (N/128 = %d..%d)\n", n, lo, hi);
        return 0;
    }
    if (argc > 5 && !strcmp(argv[1], "--check-shape")) {
        // one shape, e.g. the largest the library sends: --check-shape NT MT MODE SOLVE_ONLY [PSHORT]
        DagShape sh;
        sh.nt = atoi(argv[2]); sh.mt = atoi(argv[3]); sh.mode = atoi(argv[4]); sh.solve_only = atoi(argv[5]) != 0;
        sh.pshort = argc > 6 ? atoi(argv[6]) : 0;                  // [PSHORT]: panel tile rows without their last column tile
        for (int workers : {512, 2 * DAG_TEAM + 1})
            if (!check(sh, 2, workers)) return 1;
        printf("CHECK OK: shape nt %d mt %d mode %d solve_only %d pshort %d\n", sh.nt, sh.mt, sh.mode, (int)sh.solve_only, sh.pshort);
        return 0;
    }
    // dag_sched_probe NT [MT MODE SOLVE_ONLY]
    DagShape sh;
    sh.nt = argc > 1 ? atoi(argv[1]) : 79;
    sh.mt = argc > 2 ? atoi(argv[2]) : 0;
    sh.mode = argc > 3 ? atoi(argv[3]) : (sh.mt ? 1 : 0);
    sh.solve_only = argc > 4 && atoi(argv[4]) != 0;
    DagSchedule s;
    dag_build_schedule(sh, 2, 512, s);
    printf("nt %d mt %d mode %d solve_only %d: %zu ticketed tasks, simulated makespan %.0f us\n", sh.nt, sh.mt, sh.mode, (int)sh.solve_only,
           s.tasks.size(), s.makespan);
    return 0;
}
