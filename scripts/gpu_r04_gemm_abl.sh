#!/bin/bash
# round 4: where the centred GEMM's time goes -- ablation build of q8gemm256c.hip (one ingredient removed at a time),
# and what the chip clocks / draws under the kernel
TAG=${1:-r04b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=${2:-20}
echo "== clocks and power" 
timeout 200 python tools/gemm_power.py --variants 15,20,21 --seconds 3 2>&1 | grep -v amdgpu.ids | tee $OUT/gemm_power.txt
ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>/dev/null | head -40 > $OUT/hwmon_ls.txt
echo "== ablations of variant $V"
for A in 0 1 2 3 4 8 16 24 27 32 59 31 0; do
  echo -n "ablate=$A " | tee -a $OUT/gemm_c_ablation.txt
  QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so QNNP_GFX950_ABLATE=$A timeout 120 python tools/gemm_ab.py --variants $V --rounds 3 2>&1 | grep us_median | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['us_median'], r['us_min'], r['tops_median'])" | tee -a $OUT/gemm_c_ablation.txt
done
echo "== counter list"
(cd /tmp && rocprofv3 -L > $OLDPWD/$OUT/rocprofv3_counters.txt 2>&1); grep -c . $OUT/rocprofv3_counters.txt
LDS="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"
bash scripts/gpu_pmc_cmd.sh $TAG gemm${V}_lds "python tools/gemm_ab.py --only $V" $LDS | tee $OUT/pmc_gemm${V}_lds.txt
