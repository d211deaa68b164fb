#!/bin/bash
# BASELINE config 5's mode on the bench workload: throughput and where the encoder's time goes
timeout -k 10 300 python bench.py --no-cpu-baseline --precision fp8 --stage-profile 2> gpurun_out/fp8_stage.err | cut -c1-250
grep -E "fp8|row_chain|self_attention|conv2 " gpurun_out/fp8_stage.err | head -12
timeout -k 10 300 python bench.py --no-cpu-baseline --precision fp8 --streams 1 --steps 40 | cut -c1-200
