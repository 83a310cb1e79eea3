# Dev tool (GPU box): FETCH_SIZE of the dense kernels under both workgroup -> tile maps (AL3D_F3_MAP=rr|band), tools/bench_conv.py 64 f16x3dma
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $O/pmc_fetch_dense.txt
for m in rr band; do
  rm -rf /tmp/p_fd
  AL3D_F3_MAP=$m timeout -k 10 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_fd -o p -- python3 $R/tools/bench_conv.py 64 f16x3dma > $O/pmc_fd.log 2> $O/pmc_fd.err || exit 1
  python3 - <<PY >> $O/pmc_fetch_dense.txt
import sqlite3, glob
db=glob.glob("/tmp/p_fd/**/*.db", recursive=True)[0]
c=sqlite3.connect(db)
print("== AL3D_F3_MAP=$m (per launch: FETCH_SIZE x 64 B as counted, MB; x 2 for whole-line loaders)")
for k,n,s,dur in c.execute("select kernel_name, count(*), sum(value), avg(duration) from counters_collection where counter_name='FETCH_SIZE' group by kernel_name order by sum(value) desc"):
    if "conv" in k: print(f"  {k[:70]:70s} n={n:4d} fetch {s/n*1024/1e6:9.1f} MB  {dur/1e3:8.1f} us")
PY
done
cat $O/pmc_fetch_dense.txt
