# masked blob case (r = 130): rounds of 1 / 2 / 4 / 8 reference batches, block-maxima skip on / off
R=$GRAFT_REPO_ROOT
for G in 1 2 4 8; do for B in 1 0; do
  echo "groups=$G blkmax=$B: $(SFM_MASKED_GROUPS=$G SFM_MASKED_BLKMAX=$B MASK_CASE='blobs r=130' python $R/tools/measure/masked_time.py 2>&1 | grep blobs)"
done; done
