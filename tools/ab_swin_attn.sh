# dev: Swin-T per sample with the fused attention half at embed dims {96,192} | {96} | none (16 samples per forward)
R=$GRAFT_REPO_ROOT
for dims in "96,192" "96" "192" ""; do
  echo "AL3D_SWIN_ATTN_DIMS=$dims"
  AL3D_SWIN_ATTN_DIMS=$dims timeout -k 10 200 python3 $R/tools/bench_swin.py 16 5 || exit 1
done
