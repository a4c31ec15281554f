# round-3 profile batch (GPU box): kernel stats, whole-step PMC, HBM traffic (fp32 and fp16), dominant kernel stats + PMC
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_step -o step -- python3 $R/bench.py --steps 10 --warmup 5 --no-extras --no-cpu-baseline > $R/gpurun_out/prof_step.json 2> $R/gpurun_out/prof_step.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_step16 -o step -- python3 $R/bench.py --steps 10 --warmup 5 --no-extras --no-cpu-baseline --precision fp16 > $R/gpurun_out/prof_step16.json 2> $R/gpurun_out/prof_step16.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_step -o s -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/pmc_step.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o w -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/pmc_write.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch16 -o f -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --precision fp16 > /dev/null 2> $R/gpurun_out/pmc_fetch16.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write16 -o w -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --precision fp16 > /dev/null 2> $R/gpurun_out/pmc_write16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dom -o dom -- python3 $R/bench.py --dominant-kernel-only > $R/gpurun_out/prof_dom.json 2> $R/gpurun_out/prof_dom.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_dom -o d -- python3 $R/bench.py --dominant-kernel-only > /dev/null 2> $R/gpurun_out/pmc_dom.err
cd $R
python tools/step_pmc.py gpurun_out/pmc_step > gpurun_out/r03_step_pmc.json
python tools/measure_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/r03_traffic.json
python tools/measure_traffic.py gpurun_out/pmc_fetch16 gpurun_out/pmc_write16 > gpurun_out/r03_traffic_fp16.json
rm -f gpurun_out/pmc_step/*counter_collection.csv gpurun_out/pmc_fetch*/*counter_collection.csv gpurun_out/pmc_write*/*counter_collection.csv gpurun_out/*/*kernel_trace.csv
ls gpurun_out/prof_step gpurun_out/prof_dom gpurun_out/pmc_dom
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
