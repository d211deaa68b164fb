#!/usr/bin/env python3
"""Recogniser entry point with the reference's command line (src/bin/decode_asr.py:18-53):

    python -m cassnat_asr_public_amd.bin.decode_asr --task cassnat --test_config conf/decode.yaml \\
        --data_path data/test/feats.scp --resume_model exp/model.mdl --result_file out/token_results.txt

YAML keys become attributes of the same flat `args` namespace.  Unlike the reference it needs neither $E2EASR nor an
integer $CUDA_VISIBLE_DEVICES; under `torch.distributed.run --nproc-per-node N` it decodes on N GPUs (utterances
sharded by length, weights broadcast over RCCL, one merged result file).
"""
import json
import os
import sys

import numpy as np
import torch
import yaml

from ..tasks import ArtTask, CassNATTask
from ..utils.parser import DecodeParser


def main(argv=None):
    args = DecodeParser().get_args(argv)
    with open(args.test_config) as f:
        config = yaml.safe_load(f) or {}
    test_path = {"name": "test", "scp_path": args.data_path}
    if args.text_label:
        test_path["text_label"] = args.text_label
    config["test_paths"] = [test_path]
    for key, val in config.items():
        setattr(args, key, val)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.rank = int(os.environ.get("RANK", "0"))
    if args.rank == 0:
        merged = dict(config, **{k: v for k, v in vars(args).items() if k not in config})
        print("Experiment starts with config {}".format(json.dumps(merged, sort_keys=True, indent=4, default=str)))
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if args.hip_dist_backend == "gloo":  # rehearsal: the ranks may share a GPU (or, in the host-logic tests, have none)
            if torch.cuda.device_count() > 0:
                local = min(local, torch.cuda.device_count() - 1)
                os.environ["LOCAL_RANK"] = str(local)
                torch.cuda.set_device(local)
            torch.distributed.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    task_dict = {"cassnat": CassNATTask, "art": ArtTask}  # (the reference's ctc / lmnat* / hubert tasks are out of scope)
    if args.task not in task_dict:
        raise NotImplementedError("task '%s' is not on the accelerated path (only %s)" % (args.task, sorted(task_dict)))
    task = task_dict[args.task]("test", args)
    task.load_lm_model(args)
    task.decode(args)
    if hasattr(task, "close"):
        task.close()  # (decode pipelines: engine workspaces and worker threads)
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
