R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_cs
rocprofv3 --kernel-trace -d /tmp/p_cs -o s -- python3 $R/tools/bench_bevfusion_camera_lidar.py 16 1 > $O/cl16s.log 2> $O/cl16s.err
DB=$(find /tmp/p_cs -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py seq $DB $O/cl16_seq.csv 700
tail -2 $O/cl16s.log
