# dev: kernel stats of the serial bench on a small pool (16 scenes = 5 batches of 128), raster vs off
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="--scenes 16 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-math --no-from-files --no-verify --no-bevfusion"
for l0 in ${L0S:-raster off}; do
  rm -rf /tmp/p_q
  AL3D_L0=$l0 AL3D_R16_TPW=${TPW:-8} AL3D_PIPELINE=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/p_q -o s -- python3 $R/bench.py $ARGS > $O/q_$l0.json 2> $O/q_$l0.err || { tail -5 $O/q_$l0.err; exit 1; }
  python3 $R/tools/rocpd_summary.py stats $(find /tmp/p_q -name "*.db" | head -1) $O/q_${l0}_stats.csv
done
