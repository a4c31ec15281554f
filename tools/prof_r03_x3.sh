# round-3 profile batch for the three-fp16-product ("f32x3") regimes (GPU box): kernel stats + whole-step matrix-pipe PMC
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_x3 -o step -- python3 $R/bench.py --steps 10 --warmup 5 --no-extras --no-cpu-baseline --precision f32x3 > $R/gpurun_out/prof_x3.json 2> $R/gpurun_out/prof_x3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fp16x3 -o step -- python3 $R/bench.py --steps 10 --warmup 5 --no-extras --no-cpu-baseline --precision fp16 > $R/gpurun_out/prof_fp16x3.json 2> $R/gpurun_out/prof_fp16x3.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_x3 -o s -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --precision f32x3 > /dev/null 2> $R/gpurun_out/pmc_x3.err
cd $R
python tools/step_pmc.py gpurun_out/pmc_x3 > gpurun_out/r03_step_pmc_f32x3.json
rm -f gpurun_out/pmc_x3/*counter_collection.csv gpurun_out/*/*kernel_trace.csv
ls gpurun_out/prof_x3 gpurun_out/prof_fp16x3
