# Round-3 profile summaries on a GPU box (gpurun).  Kernel stats of the default bench (incl. the BEVFusion legs), of the
# serial bench, of the camera+lidar and lidar-only BEVFusion sweeps; HBM traffic counters (separate passes).
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_*
date +%T > $O/r03_progress.log
rocprofv3 --kernel-trace --stats -d /tmp/p_def -o d -- python3 $R/bench.py --steps 2 --warmup 1 > $O/r03_bench_default_line.json 2> $O/r03_prof_def.err
python3 $R/tools/rocpd_summary.py stats $(find /tmp/p_def -name "*.db" | head -1) $O/r03_bench_default_kernel_stats.csv
echo "default done $(date +%T)" >> $O/r03_progress.log
AL3D_PIPELINE=0 rocprofv3 --kernel-trace --stats -d /tmp/p_ser -o s -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-math --no-from-files --no-bevfusion > $O/r03_bench_serial_line.json 2> $O/r03_prof_ser.err
python3 $R/tools/rocpd_summary.py stats $(find /tmp/p_ser -name "*.db" | head -1) $O/r03_bench_serial_kernel_stats.csv
echo "serial done $(date +%T)" >> $O/r03_progress.log
rocprofv3 --kernel-trace --stats -d /tmp/p_cl -o c -- python3 $R/tools/bench_bevfusion_camera_lidar.py 16 3 > $O/r03_bevfusion_camera_lidar.log 2> $O/r03_prof_cl.err
python3 $R/tools/rocpd_summary.py stats $(find /tmp/p_cl -name "*.db" | head -1) $O/r03_bevfusion_camera_lidar_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/p_clh -o c -- python3 $R/tools/bench_bevfusion_camera_lidar.py 8 3 1 > $O/r03_bevfusion_camera_lidar_head.log 2> $O/r03_prof_clh.err
python3 $R/tools/rocpd_summary.py stats $(find /tmp/p_clh -name "*.db" | head -1) $O/r03_bevfusion_camera_lidar_head_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/p_li -o l -- python3 $R/tools/bench_bevfusion_lidar.py 96 32 > $O/r03_bevfusion_lidar.log 2> $O/r03_prof_li.err
python3 $R/tools/rocpd_summary.py stats $(find /tmp/p_li -name "*.db" | head -1) $O/r03_bevfusion_lidar_kernel_stats.csv
echo "bevfusion done $(date +%T)" >> $O/r03_progress.log
# STATS_ONLY=1: kernel stats only (the HBM counters: tools/collect_pmc_r03.sh)
if [ -n "$STATS_ONLY" ]; then ls -la $O/r03_*; exit 0; fi
ARGS="--scenes 16 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-math --no-from-files --no-verify --no-bevfusion"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_fetch -o f -- python3 $R/bench.py $ARGS > $O/r03_pmc_fetch_line.json 2> $O/r03_prof_fetch.err
echo "fetch pass rc $? $(date +%T)" >> $O/r03_progress.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_write -o w -- python3 $R/bench.py $ARGS > $O/r03_pmc_write_line.json 2> $O/r03_prof_write.err
echo "write pass rc $? $(date +%T)" >> $O/r03_progress.log
python3 $R/tools/rocpd_summary.py hbm $(find /tmp/p_fetch -name "*.db" | head -1) $(find /tmp/p_write -name "*.db" | head -1) $O/r03_pmc_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, bench.py $ARGS (640 frames = 5 batches of 128, AL3D_MATH=f16x3)" $O/r03_pmc_fetch_line.json
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_sf -o f -- python3 $R/tools/bench_swin.py 4 2 > $O/r03_pmc_swin_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_sw -o w -- python3 $R/tools/bench_swin.py 4 2 > $O/r03_pmc_swin_write.log 2>&1
python3 $R/tools/rocpd_summary.py hbm $(find /tmp/p_sf -name "*.db" | head -1) $(find /tmp/p_sw -name "*.db" | head -1) $O/r03_pmc_swin_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, tools/bench_swin.py 4 2 (Swin-T, 24 images of 256 x 704 per forward, 4 forwards)"
echo "pmc done $(date +%T)" >> $O/r03_progress.log
ls -la $O/r03_*
