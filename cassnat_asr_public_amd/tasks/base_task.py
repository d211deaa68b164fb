"""Inference half of the reference's BaseTask (src/tasks/base_task.py:9-149): checkpoint loading by parameter
name, parameter statistics, test dataloader.  Training-side methods are out of scope."""
import torch

from ..data.speech_loader import SpeechDataLoader, SpeechDataset


class BaseTask(object):
    def __init__(self, args):
        self.use_cuda = getattr(args, "use_gpu", True)

    def set_model(self, args):
        raise NotImplementedError

    def load_test_model(self, resume_model):
        """{'model_state': state_dict} with optional 'module.' prefixes (DDP checkpoints), copied by name."""
        if not resume_model:
            return
        print("Loading model from {}".format(resume_model))
        state = torch.load(resume_model, map_location="cpu")["model_state"]
        with torch.no_grad():
            for name, param in self.model.named_parameters():
                param.copy_(state[name] if name in state else state["module." + name])

    def model_stats(self, rank, use_slurm, distributed):
        if distributed:
            raise NotImplementedError("DDP training is out of scope")
        if rank == 0:
            n = sum(p.numel() for p in self.model.parameters())
            print("Number of parameters: {}, updated params: {}".format(n, n))
            self.model_params = self.updated_params = n
        local_rank = rank % max(torch.cuda.device_count(), 1) if use_slurm else rank
        if self.use_cuda:
            torch.cuda.set_device(local_rank)
            self.model = self.model.cuda(local_rank)

    def set_test_dataloader(self, args, indices=None):
        args.use_specaug, args.specaug_conf = False, None
        if getattr(args, "dataset_type", "SpeechDataset") != "SpeechDataset":
            raise NotImplementedError("only the fbank SpeechDataset feeds the accelerated path")
        testset = SpeechDataset(self.vocab, args.test_paths, args)
        if getattr(args, "use_cmvn", False):
            testset._load_cmvn(args.global_cmvn)
        self.test_loader = SpeechDataLoader(testset, args.batch_size, args.padding_idx,
                                            num_workers=args.load_data_workers, shuffle=False, indices=indices)
        print("Finish Loading test files. Number batches: {}".format(len(self.test_loader)))
