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
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_fetch -o f -- python3 $R/bench.py --scenes 16 --steps 1 --warmup 1 --no-cpu-baseline --no-extra-math --no-from-files --no-verify > $O/r02_pmc_fetch_line.json 2> $O/r02_prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_write -o w -- python3 $R/bench.py --scenes 16 --steps 1 --warmup 1 --no-cpu-baseline --no-extra-math --no-from-files --no-verify > $O/r02_pmc_write_line.json 2> $O/r02_prof_write.err
python3 $R/tools/rocpd_summary.py hbm $(find /tmp/p_fetch -name "*.db" | head -1) $(find /tmp/p_write -name "*.db" | head -1) $O/r02_pmc_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, bench.py --scenes 16 --steps 1 --warmup 1 (640 frames = 5 batches of 128, AL3D_MATH=f16x3)"
ls -la $O/r02_*
