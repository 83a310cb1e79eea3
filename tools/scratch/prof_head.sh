R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_h
AL3D_PIPELINE=0 rocprofv3 --kernel-trace --stats -d /tmp/p_h -o h -- python3 $R/bench.py --scenes 16 --steps 1 --warmup 1 --no-cpu-baseline --no-extra-math --no-from-files --no-bevfusion > $O/ph_line.json 2> $O/ph.err
python3 $R/tools/rocpd_summary.py stats $(find /tmp/p_h -name "*.db" | head -1) $O/ph_stats.csv
grep -i "head_\|vox_\|sp_subm\|sp_down" $O/ph_stats.csv | cut -c1-60,100-200 | head -20
