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
ls -la $O/r03_*
