cd $GRAFT_REPO_ROOT
SFM_MESH_FLAGS="-DSFM_MESH_TIMING" python -c "
from sofima_amd import _build; import os; os.utime('sofima_amd/csrc/sfm_mesh.hip'); _build.build()" 2>&1 | grep error
python tools/measure/mesh_time.py 2>&1 | grep -E "MESH|SPEC|us/step" | tail -14
