# A/B of compile-time variants of the correlation file (SFM_MFMA_FLAGS): prep / correlation kernel times from a trace.
#   bash prep_flags_ab.sh "<flags1>" "<flags2>" ...   (the empty set runs first)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prepab; mkdir -p $O
for f in "" "$@"; do
cd $R
SFM_MFMA_FLAGS="$f" python -c "
from sofima_amd import _build; _build.build()" 2>&1 | grep -i "error"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/t
timeout 600 rocprofv3 --kernel-trace --stats -d $O/t -o t -- python $R/bench.py --no-cpu-baseline --no-legs --sustain 0 --steps ${STEPS:-4} --warmup 1 > $O/trace.log 2>&1
echo "flags='$f'"
python $R/tools/rocpd_summary.py $(find $O/t -name '*.db' | head -1) | grep -E "prep_same|xcorr_mfma" | cut -c1-150
done
find $O -name '*.db' -delete
