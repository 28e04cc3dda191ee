cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/full_gpu.txt
python __graft_entry__.py smoke >> gpurun_out/full_gpu.txt 2>&1
