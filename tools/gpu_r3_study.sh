# tools/gpu_r3_study.sh -- round 3, first GPU session: parity tests on the round's first build, the default bench
# line (pipelined and in order), the formulation-study micro-benchmarks (tools/formulation_study.hip)
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log)
(timeout 400 python bench.py > gpurun_out/bench_cfg2.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg2.log)
(timeout 400 python bench.py --no-cpu-baseline --no-extra --no-pipeline > gpurun_out/bench_cfg2_nopipe.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg2_nopipe.log)
(timeout 300 ./tools/formulation_study > gpurun_out/formulation_study.txt 2>&1; echo "rc=$?" >> gpurun_out/formulation_study.txt)
tail -4 gpurun_out/pytest_gpu.log
cat gpurun_out/formulation_study.txt
python tools/summarize.py 2>/dev/null | head -20
