#!/bin/bash
# Re-collect everything under profiles/ for one round (run on the MI355X box from the repo root):
#     bash tools/collect_profiles.sh 01        -> gpurun_out/profiles_r01/*  (copy into profiles/ afterwards)
# Separate rocprofv3 passes for the kernel trace and for each PMC counter, as the MI355X guide prescribes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
RN=${1:-01}
OUT=$R/gpurun_out/profiles_r$RN
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log > $OUT/r${RN}_bench_1gpu.json
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary > $OUT/trace.log 2>&1
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/r${RN}_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f --output-format csv -- python $R/tools/perf_probe.py --only fk,ceiling,dq,o6d --sustained 20 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w --output-format csv -- python $R/tools/perf_probe.py --only fk,ceiling,dq,o6d --sustained 20 > $OUT/pmc_write.log 2>&1
python $R/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $((10#$RN)) > $OUT/r${RN}_fk_hbm_traffic.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq -o q --output-format csv -- python $R/tools/perf_probe.py --only fk,dq,mirror,o6d,ik,unroll --sustained 10 > $OUT/pmc_sq.log 2>&1
python $R/tools/sq_cycles.py $OUT/pmc_sq $((10#$RN)) > $OUT/r${RN}_sq_wave_cycles.json
python $R/tools/perf_probe.py --sustained 100 > $OUT/r${RN}_kernel_probe.txt 2>&1
PM_SWEEP_ALL=1 python $R/tools/jsweep_probe.py 4,8,16,22,23,24,28,32,40,48,52,56,64,65,72,96,128 > $OUT/r${RN}_joint_sweep.txt 2>&1
PM_SWEEP_ALL=1 PM_SWEEP_TOPO=bushy python $R/tools/jsweep_probe.py 22,40,52,64,65,72,80,96,128 >> $OUT/r${RN}_joint_sweep.txt 2>&1
DEEP_SWEEP=dq,fk,mirror python $R/tools/deep_sweep.py 56,57,60,63,64,65,66,68,72,80,88,96,97,112,127,128,129,192,250 > $OUT/r${RN}_deep_sweep.txt 2>&1
python $R/tools/dq_probe.py 22,28,40,48,52,56,64,65,96,128 > $OUT/r${RN}_to_root_dq_sweep.txt 2>&1
DQW_KINDS=body,bushy,humanoid,chain python $R/tools/dq_wide_sweep.py 16,22,32,52,64,96,128,250,512 2>&1 | grep -v amdgpu.ids > $OUT/r${RN}_dq_wide_sweep.txt
(MW_KINDS=body python $R/tools/mirror_wide_sweep.py 52; MW_KINDS=bushy,humanoid,chain python $R/tools/mirror_wide_sweep.py 22,40,48,64,96,128,250,512) 2>&1 | grep -v amdgpu.ids > $OUT/r${RN}_mirror_wide_sweep.txt
python $R/tools/ik_probe.py 4,22,28,52,96,128 2>&1 | grep -v amdgpu.ids > $OUT/r${RN}_from_root_positions_sweep.txt
python $R/tools/ik_order_probe.py 2>&1 | grep -v amdgpu.ids >> $OUT/r${RN}_from_root_positions_sweep.txt
python $R/tools/unroll_probe.py > $OUT/r${RN}_unroll_sweep.txt 2>&1
python $R/tools/small_clip_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/r${RN}_small_clips.txt
python $R/tools/prec_probe.py > $OUT/r${RN}_fk_precision_levels.txt 2>&1
python $R/tools/numpy_door_probe.py > $OUT/r${RN}_numpy_door.txt 2>&1
python $R/tools/store_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/r${RN}_store_patterns.txt
python $R/tools/fk_shape_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/r${RN}_fk_tile_shapes.txt
python $R/tools/fk_long_sweep.py 2>&1 | grep -v amdgpu.ids > $OUT/r${RN}_fk_long_sweep.txt
FKW_KINDS=humanoid,bushy,chain python $R/tools/fk_wide_sweep.py 96,100,112,128,129,130,160,200,250,256,300,400,511,512 2>&1 | grep -v amdgpu.ids > $OUT/r${RN}_fk_wide_final.txt
python $R/tools/fk_wide_variants.py 96,104,112,128,129,160,200,256,384,512 2>&1 | grep -v amdgpu.ids > $OUT/r${RN}_fk_wide_variants_final.txt
python $R/tools/fk_w4_sweep.py 24,28,32,36,40,44,48,52,56,64,72,80,92,100 2>&1 | grep -v amdgpu.ids > $OUT/r${RN}_fk_w4_final.txt
FKW_SRC=o6d FKW_KINDS=humanoid,bushy python $R/tools/fk_w4_sweep.py 24,32,40,48,52,64,80 2>&1 | grep -v amdgpu.ids >> $OUT/r${RN}_fk_w4_final.txt
python $R/tools/fmap_ab.py 22 10 14 18 26 2>&1 | grep -v amdgpu.ids > $OUT/r${RN}_fk_frame_map_ab.txt
python $R/tools/door_latency_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/r${RN}_door_latency_final.txt
python $R/bench.py --frames-per-gpu 16777216 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $OUT/r${RN}_bench_16m_frames_1gpu.json
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
ls -la $OUT
