# A/B of compile-time variants of the persistent mesh kernel: bash mesh_flags_ab.sh "<flags1>" "<flags2>" ...
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for f in "" "$@"; do
SFM_MESH_FLAGS="$f" python -c "
from sofima_amd import _build; _build.build()" 2>&1 | grep -i error
echo "flags='$f' $(python tools/measure/mesh_time.py 2>&1 | grep 'us/step' | grep -E '\(2, 1, 205, 205' )"
done; done
