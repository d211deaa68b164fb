#!/bin/bash
timeout -k 10 200 python tools/host_gap.py --streams 4 --switch 0.005
timeout -k 10 200 python tools/host_gap.py --streams 4 --switch 0.0002
timeout -k 10 200 python tools/host_gap.py --streams 4 --switch 0.00002
timeout -k 10 200 python tools/host_gap.py --streams 1 --switch 0.005
