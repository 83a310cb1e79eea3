# Round-5 profile summaries on a GPU box (gpurun): one process per leg, collected at the round's final kernels.
#   PART=stats  kernel stats of the default (pipelined) and the serial bench, of the camera+lidar / head / lidar-only sweeps
#   PART=pmc    HBM traffic counters of the headline bench WITH the two-stream pipeline on (the round-3 attempt hung in this
#               pass and was collected serial): a stack-dump watchdog and the pass's stderr are kept either way
#   PART=sq     SQ counters of the level-0 kernels
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
PART=${PART:-stats}
HEAD="--steps 2 --warmup 1 --no-cpu-baseline --no-extra-math --no-from-files --no-bevfusion"
if [ "$PART" = stats ]; then
  rm -rf /tmp/p_*
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/p_def -o d -- python3 $R/bench.py $HEAD > $O/r05_bench_default_line.json 2> $O/r05_prof_def.err || exit 1
  python3 $R/tools/rocpd_summary.py stats $(find /tmp/p_def -name "*.db" | head -1) $O/r05_bench_default_kernel_stats.csv
  AL3D_PIPELINE=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/p_ser -o s -- python3 $R/bench.py $HEAD > $O/r05_bench_serial_line.json 2> $O/r05_prof_ser.err || exit 1
  python3 $R/tools/rocpd_summary.py stats $(find /tmp/p_ser -name "*.db" | head -1) $O/r05_bench_serial_kernel_stats.csv
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/p_cl -o c -- python3 $R/tools/bench_bevfusion_camera_lidar.py 16 3 > $O/r05_bevfusion_camera_lidar.log 2> $O/r05_prof_cl.err || exit 1
  python3 $R/tools/rocpd_summary.py stats $(find /tmp/p_cl -name "*.db" | head -1) $O/r05_bevfusion_camera_lidar_kernel_stats.csv
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/p_clh -o c -- python3 $R/tools/bench_bevfusion_camera_lidar.py 8 3 1 > $O/r05_bevfusion_camera_lidar_head.log 2> $O/r05_prof_clh.err || exit 1
  python3 $R/tools/rocpd_summary.py stats $(find /tmp/p_clh -name "*.db" | head -1) $O/r05_bevfusion_camera_lidar_head_kernel_stats.csv
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/p_li -o l -- python3 $R/tools/bench_bevfusion_lidar.py 96 32 > $O/r05_bevfusion_lidar.log 2> $O/r05_prof_li.err || exit 1
  python3 $R/tools/rocpd_summary.py stats $(find /tmp/p_li -name "*.db" | head -1) $O/r05_bevfusion_lidar_kernel_stats.csv
fi
if [ "$PART" = pmc ]; then
  rm -rf /tmp/p_fetch /tmp/p_write
  ARGS="--scenes 16 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-math --no-from-files --no-verify --no-bevfusion"
  export AL3D_STACKDUMP_AFTER=45
  echo "fetch pass (pipeline ${AL3D_PIPELINE:-ahead}) start $(date +%T)" >> $O/r05_pmc_progress.log
  timeout -k 10 170 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_fetch -o f -- python3 $R/bench.py $ARGS > $O/r05_pmc_fetch_line.json 2> $O/r05_prof_fetch.err
  rc=$?
  echo "fetch pass rc $rc $(date +%T)" >> $O/r05_pmc_progress.log
  [ $rc -eq 0 ] || exit $rc
  timeout -k 10 170 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_write -o w -- python3 $R/bench.py $ARGS > $O/r05_pmc_write_line.json 2> $O/r05_prof_write.err
  rc=$?
  echo "write pass rc $rc $(date +%T)" >> $O/r05_pmc_progress.log
  [ $rc -eq 0 ] || exit $rc
  python3 $R/tools/rocpd_summary.py hbm $(find /tmp/p_fetch -name "*.db" | head -1) $(find /tmp/p_write -name "*.db" | head -1) $O/r05_pmc_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, AL3D_PIPELINE=${AL3D_PIPELINE:-ahead} bench.py $ARGS (640 frames = 5 batches of 128, AL3D_MATH=f16x3)" $O/r05_pmc_fetch_line.json
fi
if [ "$PART" = swin ]; then
  # HBM traffic of the Swin-T token kernels (16 samples per forward), with the algorithmic bytes of the fused halves
  rm -rf /tmp/p_sf /tmp/p_sw
  timeout -k 10 170 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_sf -o f -- python3 $R/tools/bench_swin.py 16 1 > $O/r05_pmc_swin_fetch.log 2> $O/r05_prof_swin_fetch.err || exit 1
  timeout -k 10 170 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_sw -o w -- python3 $R/tools/bench_swin.py 16 1 > $O/r05_pmc_swin_write.log 2> $O/r05_prof_swin_write.err || exit 1
  python3 $R/tools/rocpd_summary.py hbm $(find /tmp/p_sf -name "*.db" | head -1) $(find /tmp/p_sw -name "*.db" | head -1) $O/r05_pmc_swin_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, tools/bench_swin.py 16 1 (96 images of 256 x 704; 3 forwards per process)" $O/r05_pmc_swin_fetch.log
fi
if [ "$PART" = sq ]; then
  # SQ counters of the block-staged sparse kernels (opt-in, AL3D_BLK_PAIRS) against the kernels they would replace
  AL3D_BLK_PAIRS=32x32,64x64,128x128 bash $R/tools/pmc_sq_blk.sh
  cp $O/pmc_blk.txt $O/r05_pmc_sq_blk.txt
fi
if [ "$PART" = calib ]; then
  bash $R/tools/calib_fetch_size.sh
  cp $O/fetch_calib.json $O/r05_fetch_size_calibration.json
fi
ls -la $O/r05_* | tail -30
