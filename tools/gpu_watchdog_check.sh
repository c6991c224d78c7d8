python bench.py --gpus 1 --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-single --min-seconds 0 2>/dev/null | grep -c '^{'
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-single --min-seconds 0 2>/dev/null | python -c "
import json,sys
ls=[l for l in sys.stdin if l.startswith('{')]; print('lines', len(ls)); d=json.loads(ls[0]); print(d['value'], d.get('gather_ms'), d.get('gather_error'))"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 1 --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-single --min-seconds 0 --gather-timeout 0.001 2>/dev/null | python -c "
import json,sys
ls=[l for l in sys.stdin if l.startswith('{')]; print('lines', len(ls)); d=json.loads(ls[0]); print(d['value'], d.get('gather_ms'), d.get('gather_error'))"
echo "rc=$?"
