# dev: runtime ablations of the level-0 item-stream kernel (AL3D_R16_ABL bits: 1 rows from the zero row, 2 no indices,
# 4 no products, 8 no fragment reads, 16 no stores, 32 no residual requests)
# needs the tuning build: make -C exploring-*_amd/csrc EXTRA=-DAL3D_R16_ABLATE (the shipped library ignores AL3D_R16_ABL)
cd $GRAFT_REPO_ROOT
O=gpurun_out
: > $O/l0_abl.log
for abl in 0 1 2 3 4 8 12 16 28 31 63; do
  echo "=== abl $abl" >> $O/l0_abl.log
  AL3D_R16_ABL=$abl BENCH_L0_MODES=raster16+32 timeout -k 10 200 python tools/bench_l0.py ${1:-32} 3 2>&1 | grep -E "16-> (16|32)" >> $O/l0_abl.log || exit 1
done
cat $O/l0_abl.log
