#!/bin/bash
# side timings of the round (configs 4 and 5, conformer variants, ESA, fbank) -> gpurun_out/r02n_side_timings.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
echo '{'
echo '"ast_config4": '; timeout -k 10 300 python tools/time_ast.py --streams 4 2>/dev/null | tail -1; echo ','
echo '"conformer": '; timeout -k 10 300 python tools/time_conformer.py --streams 3 2>/dev/null | tail -1; echo ','
echo '"esa_sample50": '; timeout -k 10 300 python tools/time_esa.py --reps 12 --same-seed 2>/dev/null | tail -1; echo ','
echo '"fbank": '; timeout -k 10 300 python tools/time_fbank.py 2>/dev/null | tail -1; echo ','
echo '"fp8_bench": '; timeout -k 10 300 python bench.py --precision fp8 --no-cpu-baseline --no-parity-engine --steps 100 2>/dev/null | tail -1; echo ','
echo '"bf16x3_bench_steps20": '; timeout -k 10 300 python bench.py --precision bf16x3 --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 20 --warmup 5 2>/dev/null | tail -1
echo '}'
} > gpurun_out/r02n_side_timings.json
python3 -c "
import json
d=json.load(open('gpurun_out/r02n_side_timings.json'))
for k,v in d.items():
    print(k, {a:b for a,b in v.items() if not isinstance(b,(dict,list)) and a not in ('workload','note')})
"
