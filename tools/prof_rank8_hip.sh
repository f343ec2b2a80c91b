cd /tmp && export TMPDIR=/tmp
export C5_ONLY=rank8 C5_EMU_STEPS=10
rocprofv3 --kernel-trace --hip-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_rank8h -o r8 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/c5_leg.py > $GRAFT_REPO_ROOT/gpurun_out/prof_rank8h.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_rank8h.err
ls -la $GRAFT_REPO_ROOT/gpurun_out/prof_rank8h
