cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_map_add.py tests/test_gpu_shim.py -x -q 2>&1 | tail -2
bash scripts/gpu_stream_sequence.sh 2>&1 | grep -A28 "kf_begin" | head -32
bash scripts/gpu_stream_sequence.sh 2>&1 | grep "cycle us\|submissions"
