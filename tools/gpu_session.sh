# tools/gpu_session.sh -- the commands of the CURRENT gpurun session
# round 5, session 22: the multi-process bench path on REAL processes with a GPU under them -- two and four ranks sharing the box's one device
# (MKAMD_BENCH_SHARE_DEVICES=1: a rehearsal, not a scaling measurement): (a) without the gather legs, (b) with them -- RCCL is asked for a
# communicator of ranks that sit on one device and must be survived: rank 0's line has to print either way
mkdir -p gpurun_out
export TMPDIR=/tmp
export MKAMD_BENCH_SHARE_DEVICES=1
(timeout 400 python bench.py --gpus 2 --no-extra --no-cpu-baseline --no-single --steps 10 --warmup 3 --min-seconds 1 --no-gather > gpurun_out/rehearsal_2_nogather.log 2>gpurun_out/rehearsal_2_nogather.err; echo "rc=$?" >> gpurun_out/rehearsal_2_nogather.log)
tail -c 1500 gpurun_out/rehearsal_2_nogather.log; echo
(timeout 400 python bench.py --gpus 4 --no-extra --no-cpu-baseline --no-single --steps 10 --warmup 3 --min-seconds 1 --no-gather --batch 64 > gpurun_out/rehearsal_4_nogather.log 2>gpurun_out/rehearsal_4_nogather.err; echo "rc=$?" >> gpurun_out/rehearsal_4_nogather.log)
tail -c 700 gpurun_out/rehearsal_4_nogather.log; echo
(timeout 500 python bench.py --gpus 2 --no-extra --no-cpu-baseline --no-single --steps 10 --warmup 3 --min-seconds 1 --gather-timeout 60 > gpurun_out/rehearsal_2_gather.log 2>gpurun_out/rehearsal_2_gather.err; echo "rc=$?" >> gpurun_out/rehearsal_2_gather.log)
tail -c 1200 gpurun_out/rehearsal_2_gather.log; echo; tail -5 gpurun_out/rehearsal_2_gather.err | cut -c1-300
