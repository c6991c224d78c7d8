# round 6, session 53: the whole GPU tier with the few-frame contact tests
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -3)
