"""Timing helpers with the interface CassNATTask.decode uses (reference: src/utils/util.py)."""


class AverageMeter(object):
    def __init__(self, name, fmt=":f"):
        self.name, self.fmt = name, fmt
        self.val = self.avg = self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / max(self.count, 1)

    def __str__(self):
        return ("{name} {val" + self.fmt + "} ({avg" + self.fmt + "})").format(name=self.name, val=self.val, avg=self.avg)


class ProgressMeter(object):
    def __init__(self, num_batches, *meters, prefix=""):
        width = len(str(num_batches))
        self._fmt = "[{:" + str(width) + "d}/" + str(num_batches) + "]"
        self.meters, self.prefix = meters, prefix

    def print(self, step):
        print("\t".join([self.prefix + self._fmt.format(step)] + [str(m) for m in self.meters]), flush=True)
