# A/B of the packed remainder column of the tiled in-plane step (SFM_MESH_PACK=0: off)
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_maps.py tests/test_gpu_dist.py -m gpu -q -x 2>&1 | tail -4
for i in 1 2; do
for p in 1 0; do
echo "SFM_MESH_PACK=$p"
SFM_MESH_PACK=$p python tools/measure/mesh_big.py 0 2>&1 | grep us/step
SFM_MESH_PACK=$p python tools/measure/mesh_big.py 1 2>&1 | grep us/step
SFM_MESH_PACK=$p python tools/measure/montage_time.py 2>&1 | grep "tiled:"
done; done
