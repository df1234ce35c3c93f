cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_configs.py -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed" | tail -6
python tools/measure/mesh_time.py 2>&1 | grep "us/step"
python tools/measure/mesh_time.py 2>&1 | grep "us/step" | grep "205, 205"
