# HBM traffic counters (FETCH_SIZE / WRITE_SIZE, separate passes) on a small pool: 5 batches of exactly 128 frames.
# Every pass runs under its own timeout (a hung profiler pass must not take the call down) and reports progress.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_fetch /tmp/p_write
ARGS="--scenes 16 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-math --no-from-files --no-verify"
echo "fetch pass start $(date +%T)" >> $O/pmc_progress.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_fetch -o f -- python3 $R/bench.py $ARGS > $O/r02_pmc_fetch_line.json 2> $O/r02_prof_fetch.err
echo "fetch pass rc $? $(date +%T)" >> $O/pmc_progress.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_write -o w -- python3 $R/bench.py $ARGS > $O/r02_pmc_write_line.json 2> $O/r02_prof_write.err
echo "write pass rc $? $(date +%T)" >> $O/pmc_progress.log
python3 $R/tools/rocpd_summary.py hbm $(find /tmp/p_fetch -name "*.db" | head -1) $(find /tmp/p_write -name "*.db" | head -1) $O/r02_pmc_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, bench.py $ARGS (640 frames = 5 batches of 128, AL3D_MATH=f16x3)"
ls -la $O/r02_pmc_hbm_traffic.json
