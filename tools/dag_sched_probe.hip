// The host-side list schedule of the one-launch Cholesky (algp_amd/csrc/chol_dag.hip) on its own: critical path of the task
// graph, simulated makespan, simulated utilisation and chain progress per 500 us.  Needs no GPU.
// hipcc --offload-arch=gfx950 -O2 -std=c++17 -w -DALGP_DAG_DEBUG tools/dag_sched_probe.hip -o build/dag_sched_probe && build/dag_sched_probe 79
#include "../algp_amd/csrc/chol_dag.hip"
namespace algp {
int fail(algp_ctx*, int code, const std::string&) { return code; }
void prof_begin(algp_ctx*, int, double, double) {}
void prof_end(algp_ctx*) {}
int ensure(algp_ctx*, DevBuf&, size_t) { return 0; }
}
int main(int argc, char** argv) {
    const int nt = argc > 1 ? atoi(argv[1]) : 79;
    algp::DagSchedule s;
    algp::dag_build_schedule(nt, 4, 512, s);
    printf("nt %d: %zu ticketed tasks\n", nt, s.tasks.size());
    return 0;
}
