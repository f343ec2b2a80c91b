# host API + kernel timeline of config 5's one-GPU incremental step (tools/c5_leg.py, loop only, 12 steps)
cd /tmp && export TMPDIR=/tmp
export C5_ONLY=loop C5_STEPS=12
rocprofv3 --kernel-trace --hip-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_c5_hip -o c5 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/c5_leg.py > $GRAFT_REPO_ROOT/gpurun_out/prof_c5_hip.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_c5_hip.err
ls $GRAFT_REPO_ROOT/gpurun_out/prof_c5_hip
