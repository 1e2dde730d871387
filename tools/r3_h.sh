#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/r3h; rm -rf $o; mkdir -p $o
timeout 1800 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
tail -5 $o/pytest.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline > $o/bench_tr.json 2> $o/bench_tr.err; echo "torchrun bench rc=$?"; tail -2 $o/bench_tr.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3h/bench_tr.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["verified"], d["config"]["kernel_ms_over_ranks"], d["config"]["broadcast"], d["config"]["gpus_on_node"])
PY
