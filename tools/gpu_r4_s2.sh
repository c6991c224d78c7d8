# round 4, session 2: GPU tests on the in-tree build (incl. the promised / streamed pipelining test), counters of the instruction cache,
# then the in-tree library against the rolled-channel variants
mkdir -p gpurun_out
export TMPDIR=/tmp
(rocprofv3 -L 2>/dev/null | grep -i -E "icache|inst_cache|ICACHE|SQC_" | head -60 > gpurun_out/r4_counters_icache.txt)
bash tools/gpu_r4_tile_ab.sh "$@"
