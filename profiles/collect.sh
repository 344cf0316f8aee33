#!/bin/bash
# Collects the round's profiles on the GPU box (run from the repo root through gpurun):
#   bash profiles/collect.sh r01
# 1. rocprofv3 --kernel-trace --stats of the two bench workloads and of the roofline leg alone
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the roofline leg, as MI355X_MICROARCH.md prescribes
# Raw output goes under gpurun_out/<tag>_*; profiles/summarize.py turns it into the files committed in profiles/.
# Optional second argument: 1 = parts 1-3 only (kernel stats of the three workloads, roofline legs + their counter passes),
# 2 = parts 3b-4 only (byte census, whole-step counters per layer class) -- one gpurun call each (20-minute limit).
set -e
tag=${1:-r01}
part=${2:-all}
out=gpurun_out
common="--output-format csv"
if [ "$part" != "2" ]; then
rm -rf $out/${tag}_ugan $out/${tag}_unet $out/${tag}_roof $out/${tag}_pmc_fetch $out/${tag}_pmc_write
rocprofv3 --kernel-trace --stats $common -d $out/${tag}_ugan -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-unet-step --no-dist-leg > $out/${tag}_ugan.log 2>&1
tail -1 $out/${tag}_ugan.log | cut -c1-160
rocprofv3 --kernel-trace --stats $common -d $out/${tag}_unet -- python3 bench.py --workload unet --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-unet-step > $out/${tag}_unet.log 2>&1
tail -1 $out/${tag}_unet.log | cut -c1-160
rm -rf $out/${tag}_c5
rocprofv3 --kernel-trace --stats $common -d $out/${tag}_c5 -- python3 bench.py --dtype f16 --size 512 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-unet-step --no-config5 > $out/${tag}_c5.log 2>&1
tail -1 $out/${tag}_c5.log | cut -c1-160
rocprofv3 --kernel-trace --stats $common -d $out/${tag}_roof -- python3 bench.py --roofline-only > $out/${tag}_roof.log 2>&1
tail -1 $out/${tag}_roof.log | cut -c1-400
rocprofv3 --pmc FETCH_SIZE --kernel-trace $common -d $out/${tag}_pmc_fetch -- python3 bench.py --roofline-only > $out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace $common -d $out/${tag}_pmc_write -- python3 bench.py --roofline-only > $out/${tag}_pmc_write.log 2>&1
# 3. MFMA-busy of the dominant kernel (SQ block, own pass): busy cycles of the matrix pipes vs the GPU-active cycles
rm -rf $out/${tag}_pmc_sq
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace $common -d $out/${tag}_pmc_sq -- python3 bench.py --roofline-only > $out/${tag}_pmc_sq.log 2>&1
# 3c. config 5's roofline leg (fp16-operand fused-shortcut forward at 512^2): duration + the two traffic counters, own passes
rm -rf $out/${tag}_c5roof $out/${tag}_c5pmc_fetch $out/${tag}_c5pmc_write
rocprofv3 --kernel-trace --stats $common -d $out/${tag}_c5roof -- python3 bench.py --roofline-only --dtype f16 --size 512 > $out/${tag}_c5roof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace $common -d $out/${tag}_c5pmc_fetch -- python3 bench.py --roofline-only --dtype f16 --size 512 > $out/${tag}_c5pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace $common -d $out/${tag}_c5pmc_write -- python3 bench.py --roofline-only --dtype f16 --size 512 > $out/${tag}_c5pmc_write.log 2>&1
python3 profiles/summarize.py $tag || true
fi
if [ "$part" = "1" ]; then exit 0; fi
# 3b. algorithmic bytes of the step's launches (bench.py's per-shape census: every tensor of every conv / InstanceNorm / tail / pooling
#     call read once + written once), un-profiled: the floor the PMC traffic below stands against
python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-unet-step --no-dist-leg --no-config5 > $out/${tag}_bytes_ugan.log 2>/dev/null
python3 bench.py --workload unet --steps 3 --warmup 2 --no-cpu-baseline --no-unet-step > $out/${tag}_bytes_unet.log 2>/dev/null
# 4. whole-step counters per layer class (profiles/step_pmc.py: eager steps between two marker dispatches; four passes each:
#    durations without counters, FETCH_SIZE, WRITE_SIZE, SQ) -> profiles/<tag>_step_<wl>_classes.{json,md}
for wl in unet ugan; do
  rm -rf $out/${tag}_step_${wl}_time $out/${tag}_step_${wl}_fetch $out/${tag}_step_${wl}_write $out/${tag}_step_${wl}_sq
  rocprofv3 --kernel-trace $common -d $out/${tag}_step_${wl}_time -- python3 profiles/step_pmc.py $wl > $out/${tag}_step_${wl}_time.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace $common -d $out/${tag}_step_${wl}_fetch -- python3 profiles/step_pmc.py $wl > $out/${tag}_step_${wl}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace $common -d $out/${tag}_step_${wl}_write -- python3 profiles/step_pmc.py $wl > $out/${tag}_step_${wl}_write.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace $common -d $out/${tag}_step_${wl}_sq -- python3 profiles/step_pmc.py $wl > $out/${tag}_step_${wl}_sq.log 2>&1
  python3 profiles/summarize_step.py $tag $wl
done
