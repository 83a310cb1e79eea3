for v in "ABL=0 CFG=0" "ABL=1 CFG=0" "ABL=3 CFG=0" "ABL=4 CFG=0" "ABL=0 CFG=1" "ABL=0 CFG=2" "ABL=0 CFG=3" "ABL=0 CFG=4"; do
  set -- $v; a=${1#ABL=}; c=${2#CFG=}
  echo "=== $v" >> gpurun_out/r2_glds_abl.log
  AL3D_GLDS_ABL=$a AL3D_GLDS_CFG=$c timeout -k 10 120 python tools/bench_splayers.py 64 al3d_sp_conv_glds_f16x3 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10}' >> gpurun_out/r2_glds_abl.log || exit 1
done
