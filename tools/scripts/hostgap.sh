#!/bin/bash
timeout -k 10 200 python tools/host_gap.py --streams 4
timeout -k 10 200 python tools/host_gap.py --streams 6
timeout -k 10 200 python tools/host_gap.py --streams 8
GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python tools/host_gap.py --streams 8
