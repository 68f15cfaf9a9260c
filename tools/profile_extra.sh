#!/bin/bash
# The side legs' evidence, on the GPU box from the repo root (after tools/profile_round.sh <tag>):
#   bash tools/profile_extra.sh r06   -> profiles/r06_*
set -u
TAG=${1:-r06}
root=$(pwd); out=$root/gpurun_out/${TAG}x; mkdir -p "$out"
export TMPDIR=/tmp
# --- N worker processes on one GPU: the default (service), the ways the dispatcher can wait, spectra staged, few processes
timeout 600 python tools/mc3_bench.py 1,3,10 1500 > profiles/${TAG}_mc3_service.json 2> "$out/mc3.err"
{
  for sync in 0 1 2; do
    BARTRT_SVC_SYNC=$sync timeout 300 python tools/mc3_bench.py 10 1500 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); v=d['shared_10']; print('BARTRT_SVC_SYNC=$sync shared_10: %.3e spectra/s, %.1f us per call (loop), %.1f us median call, mean batch %.2f' % (v['aggregate_spectra_per_s'], v['us_per_call_median'], v['call_us_median_of_medians'], v['service']['mean_batch']))"
  done
  BARTRT_SVC_DIRECT_BYTES=0 timeout 300 python tools/mc3_bench.py 10 1500 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); v=d['shared_10']; print('spectra staged through HBM + one DMA copy (BARTRT_SVC_DIRECT_BYTES=0) shared_10: %.3e spectra/s, %.1f us per call' % (v['aggregate_spectra_per_s'], v['us_per_call_median']))"
} > profiles/${TAG}_mc3_sync_modes.txt
timeout 600 python tools/mc3_bench.py 1,2,3,4,5,6 1500 > profiles/${TAG}_mc3_few.json 2>/dev/null
# --- the whole step: per-kernel split
cd /tmp
for leg in full_step_10 wasp12b_step; do
  timeout 400 rocprofv3 --kernel-trace --stats -d "$out/stats_$leg" --output-format csv -- python3 "$root/tools/step_leg.py" $leg > "$out/stats_$leg.log" 2>&1
  f=$(find "$out/stats_$leg" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -6 "$f" > "$root/profiles/${TAG}_${leg}_kernel_stats.csv"
done
cd "$root"
timeout 300 python tools/step_leg.py wasp12b_shard8 > profiles/${TAG}_wasp12b_shard8.json 2>/dev/null
timeout 300 python tools/step_leg.py demo_1walker > profiles/${TAG}_demo_1walker.json 2>/dev/null
BARTRT_SYNC=stream timeout 300 python tools/step_leg.py demo_1walker > profiles/${TAG}_demo_1walker_streamsync.json 2>/dev/null
# --- few walkers: the launcher's choice, the adjacent-rows kernel forced, the forms it replaced
{
  for k in "" adj8 adj16 octo hexa; do BARTRT_KERNEL=$k timeout 200 python tools/ab_small.py 1 2 3 4 2>&1 | grep walkers; done
  for k in "" adj8 adj16 hexa; do AB_NWAVE=5000 BARTRT_KERNEL=$k timeout 200 python tools/ab_small.py 1 2 3 4 2>&1 | grep walkers; done
  for k in "" adj8 adj16 r32 hexa octo; do AB_CASE=demo BARTRT_KERNEL=$k timeout 200 python tools/ab_small.py 1 2 3 4 5 6 2>&1 | grep walkers; done
  echo "# two temperature planes (64 MB grid, fits the Infinity Cache):"
  for k in "" adj16; do AB_TEMPDELT=2600 BARTRT_KERNEL=$k timeout 200 python tools/ab_small.py 1 2 2>&1 | grep walkers; done
} > profiles/${TAG}_qadj_ab.txt
# --- shapes outside the ahead-of-time set: instantiated at run time against the generic kernel
{
  for shape in "--nmol 7 --cia 2" "--nmol 9 --cia 1" "--nmol 4 --cia 2 --angles 12"; do
    BARTRT_RTC_CACHE=$out/rtc_cache timeout 300 python tools/shape_bench.py $shape 1 10 64 2>/dev/null
    BARTRT_RTC=0 timeout 300 python tools/shape_bench.py $shape 1 10 64 2>/dev/null
  done
} > profiles/${TAG}_rtc.txt
# --- SQ counters of the one-walker launch
{
  for k in "" adj16; do
    BARTRT_KERNEL=$k bash tools/pmc_pass.sh a$k "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES" --walkers 1
    BARTRT_KERNEL=$k bash tools/pmc_pass.sh b$k "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" --walkers 1
  done
} > profiles/${TAG}_w1_sq.jsonl
# --- config 5: lane utilisation of the accumulation kernels (item 7's evidence)
bash tools/lbl_pmc.sh ${TAG}x 1 > /dev/null 2>&1; cp gpurun_out/${TAG}x/lbl_pmc_1.txt profiles/${TAG}_lbl_pmc_1.txt 2>/dev/null
bash tools/lbl_kernel_times.sh ${TAG}x > /dev/null 2>&1; cp gpurun_out/${TAG}x/lbl_times.txt profiles/${TAG}_lbl_times.txt 2>/dev/null
mkdir -p "$out/profiles"; cp profiles/${TAG}_* "$out/profiles/" 2>/dev/null
ls -la profiles/${TAG}_* | head -60
