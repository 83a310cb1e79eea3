# HBM traffic counters of the headline bench kernels (FETCH_SIZE / WRITE_SIZE, separate passes), serial pipeline mode (the
# per-kernel bytes do not depend on the batch pipelining; the first r03 attempt hung in the FETCH pass under the two-stream mode).
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_fetch /tmp/p_write
export AL3D_PIPELINE=0
ARGS="--scenes 16 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-math --no-from-files --no-verify --no-bevfusion"
echo "fetch pass start $(date +%T)" >> $O/r03_pmc_progress.log
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_fetch -o f -- python3 $R/bench.py $ARGS > $O/r03_pmc_fetch_line.json 2> $O/r03_prof_fetch.err
rc=$?
echo "fetch pass rc $rc $(date +%T)" >> $O/r03_pmc_progress.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_write -o w -- python3 $R/bench.py $ARGS > $O/r03_pmc_write_line.json 2> $O/r03_prof_write.err
rc=$?
echo "write pass rc $rc $(date +%T)" >> $O/r03_pmc_progress.log
[ $rc -eq 0 ] || exit $rc
python3 $R/tools/rocpd_summary.py hbm $(find /tmp/p_fetch -name "*.db" | head -1) $(find /tmp/p_write -name "*.db" | head -1) $O/r03_pmc_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, AL3D_PIPELINE=0 bench.py $ARGS (640 frames = 5 batches of 128, AL3D_MATH=f16x3)" $O/r03_pmc_fetch_line.json
ls -la $O/r03_pmc_hbm_traffic.json
