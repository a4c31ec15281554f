R=$GRAFT_REPO_ROOT
python -m pytest $R/tests/test_compact_gpu.py $R/tests/test_gemm_gpu.py $R/tests/test_api_corners_gpu.py -q > $R/gpurun_out/test_sel2.log 2>&1
python $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/bench_r03e.json 2>/dev/null
python $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/bench_r03e2.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_staged -o st -- python3 $R/tools/trace_staged_step.py > $R/gpurun_out/trace_staged.log 2>&1
cd $R && python tools/trace_staged_step.py analyse gpurun_out/trace_staged > gpurun_out/r03_staged_step_overlap.json 2>> gpurun_out/trace_staged.log
rm -rf gpurun_out/trace_staged
tail -n 3 gpurun_out/test_sel2.log; head -c 300 gpurun_out/bench_r03e.json; echo; head -c 300 gpurun_out/bench_r03e2.json; echo; head -c 1500 gpurun_out/r03_staged_step_overlap.json
