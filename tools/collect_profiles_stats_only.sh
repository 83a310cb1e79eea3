# Collect the round's profile summaries on a GPU box (gpurun): kernel stats of the default and serial bench,
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) on a pool that is a multiple of the batch.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_def /tmp/p_ser /tmp/p_fetch /tmp/p_write
rocprofv3 --kernel-trace --stats -d /tmp/p_def -o d -- python3 $R/bench.py --steps 2 --warmup 1 > $O/r02_bench_default_line.json 2> $O/r02_prof_def.err
python3 $R/tools/rocpd_summary.py stats $(find /tmp/p_def -name "*.db" | head -1) $O/r02_bench_default_kernel_stats.csv
AL3D_PIPELINE=0 rocprofv3 --kernel-trace --stats -d /tmp/p_ser -o s -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-math --no-from-files > $O/r02_bench_serial_line.json 2> $O/r02_prof_ser.err
python3 $R/tools/rocpd_summary.py stats $(find /tmp/p_ser -name "*.db" | head -1) $O/r02_bench_serial_kernel_stats.csv
