# host API + kernel timeline of config 4's rank step (tools/rank_step.py): where the time between the launches goes
cd /tmp && export TMPDIR=/tmp
export REPS=3
rocprofv3 --kernel-trace --hip-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_rs_hip -o rs --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/rank_step.py > $GRAFT_REPO_ROOT/gpurun_out/prof_rs_hip.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_rs_hip.err
ls $GRAFT_REPO_ROOT/gpurun_out/prof_rs_hip
