# dev: kernel form x shapes x tiles-per-wave sweep of the level-0 item-stream kernel (tools/bench_l0.py)
cd $GRAFT_REPO_ROOT
O=gpurun_out
B=${1:-32}
: > $O/l0_sweep.log
for cfg in ${CFGS:-"2,0,4" "2,0,8" "1,0,4"}; do
  IFS=, read k sh tpw <<< "$cfg"
  echo "=== kernel $k shape $sh tpw $tpw" >> $O/l0_sweep.log
  AL3D_R16_KERNEL=$k AL3D_R16_SHAPE=$sh AL3D_R16_TPW=$tpw BENCH_L0_MODES=${MODES:-raster16+32p} timeout -k 10 300 python tools/bench_l0.py $B 3 2>&1 | grep -E "16-> (16|32)|sum us|rulebook|identical" >> $O/l0_sweep.log || exit 1
done
BENCH_L0_MODES=off timeout -k 10 300 python tools/bench_l0.py $B 3 2>&1 | grep -E "16-> (16|32)|sum us|rulebook" >> $O/l0_sweep.log
cat $O/l0_sweep.log
