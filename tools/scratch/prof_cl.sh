set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_cl
rocprofv3 --kernel-trace --stats -d /tmp/p_cl -o c -- python3 $R/tools/bench_bevfusion_camera_lidar.py 16 10 > $O/cl16.log 2> $O/cl16.err
python3 $R/tools/rocpd_summary.py stats $(find /tmp/p_cl -name "*.db" | head -1) $O/cl16_kernel_stats.csv
tail -12 $O/cl16.log
