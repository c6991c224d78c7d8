# round 6, session 25: timeline of the XTC-fed leg (2 048 frames per chunk) with the copy/decode stream on a high-priority queue
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_xtc_tl
(cd /tmp && XTC_LEG_ONE=2048 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_xtc_tl -o tl --output-format csv -- python $GRAFT_REPO_ROOT/tools/xtc_leg.py 0.92 2>&1 | grep -v amdgpu | tail -25 | cut -c1-300)
python tools/xtc_timeline.py gpurun_out/prof_xtc_tl | tail -40 | tee gpurun_out/xtc_timeline.txt
rm -rf gpurun_out/prof_xtc_tl
