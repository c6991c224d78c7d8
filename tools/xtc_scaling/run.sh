# tools/xtc_scaling/run.sh -- on a many-core host: build the harness, synthesise a 2400-frame file from the 3PTB fixture, run both page modes
set -e
g++ -O3 -std=c++17 -pthread -I moleculekit_amd/csrc tools/xtc_scaling/harness.cpp -o /tmp/xtc_harness
python - <<'PY'
src = open("tests/golden/xtc/3ptb_traj_head.xtc", "rb").read()
with open("/tmp/long.xtc", "wb") as f:
    for _ in range(400): f.write(src)
PY
nproc
/tmp/xtc_harness /tmp/long.xtc
/tmp/xtc_harness /tmp/long.xtc hugepage
cat /sys/kernel/mm/transparent_hugepage/enabled
