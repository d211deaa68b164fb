#!/bin/bash
# side timings of the bf16x3 engine after its generator / projection kernels -> gpurun_out/r02_side_x3.json
# (rank_model at_baseline is not timed here: the autoregressive model has no bf16x3 engine - cn_model_create says so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
echo '{'
echo '"esa_sample50_bf16x3": '; timeout -k 10 300 python tools/time_esa.py --reps 6 --same-seed --precision bf16x3 2>/dev/null | tail -1; echo ','
echo '"ctc_modes_bf16x3": '; timeout -k 10 300 python tools/time_ctc_modes.py --precision bf16x3 2>/dev/null | tail -1; echo ','
echo '"bf16x3_bench_steps20": '; timeout -k 10 300 python bench.py --precision bf16x3 --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 20 --warmup 5 2>/dev/null | tail -1
echo '}'
} > gpurun_out/r02_side_x3.json
python3 -c "
import json
d=json.load(open('gpurun_out/r02_side_x3.json'))
for k,v in d.items():
    print(k, {a:b for a,b in v.items() if not isinstance(b,(dict,list)) and a not in ('workload','note')})
"
