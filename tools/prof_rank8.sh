cd /tmp && export TMPDIR=/tmp
export C5_ONLY=rank8 C5_EMU_STEPS=10
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_rank8 -o r8 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/c5_leg.py > $GRAFT_REPO_ROOT/gpurun_out/prof_rank8.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_rank8.err
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_rank8 | head
