"""Static check of a gfx950 assembly listing (hipcc -S) for the hazard hipcc cannot see around inline asm: a register that an
LDS read / global load issued from an asm statement is still going to write, touched (read, overwritten, copied, spilled) by
another instruction before the s_waitcnt that covers the load.

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o /tmp/k.s file.hip && python tools/pending_reg_check.py /tmp/k.s [kernel-substring]

Linear scan: a conditional branch is followed through its fall-through only, the state is dropped at an unconditional one (a
block entered only by jumps starts clean: such paths are not checked).  LDS reads complete in order (lgkmcnt), VMEM loads in order (vmcnt).  SMEM loads also count on lgkmcnt
and may return out of order - the scan treats an s_load as one more lgkm operation, which only makes it more conservative."""
import re
import sys

REG = re.compile(r'\b([va])(?:\[(\d+):(\d+)\]|(\d+)\b)')


def regs(text):
    out = set()
    for m in REG.finditer(text):
        k = m.group(1)
        if m.group(2) is not None:
            out.update((k, i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((k, int(m.group(4))))
    return out


def scan(path, want=""):
    """-> list of findings (strings) in the kernels of the listing whose mangled name contains `want`"""
    lines = open(path).read().split("\n")
    name = None
    lgkm, vm = [], []  # queues of (line number, set of destination registers), oldest first
    found = []
    for no, raw in enumerate(lines, 1):
        l = raw.split(";")[0].strip()
        m = re.match(r'^(_Z\w+):', l)
        if m:
            name, lgkm, vm = m.group(1), [], []
            continue
        if not l or l.startswith(".") or name is None or want not in name:
            continue
        op = l.split()[0]
        args = l[len(op):]
        if op == "s_waitcnt":
            for cnt, q in (("lgkmcnt", lgkm), ("vmcnt", vm)):
                mm = re.search(cnt + r'\((\d+)\)', l)
                if mm:
                    n = int(mm.group(1))
                    del q[: max(0, len(q) - n)]
            continue
        if op in ("s_endpgm", "s_branch", "s_setpc_b64"):  # what follows is not reached from here
            lgkm, vm = [], []
            continue
        touched = regs(args)
        for q, kind in ((lgkm, "LDS read"), (vm, "load")):
            for (lno, dst) in q:
                hit = touched & dst
                if hit:
                    found.append(f"{name[:40]} line {no}: `{l[:90]}` touches {sorted(hit)[:4]} pending from the {kind} at line {lno}")
        first = args.split(",")[0]
        if op.startswith("ds_read") or op.startswith("ds_load"):
            lgkm.append((no, regs(first)))
        elif op.startswith("s_load") or op.startswith("s_buffer_load") or (op.startswith("ds_") and not op.startswith("ds_read")):
            lgkm.append((no, set()))
        elif op.startswith("global_load_lds") or op.startswith("buffer_load") and "lds" in l:
            vm.append((no, set()))
        elif op.startswith("global_load") or op.startswith("scratch_load") or op.startswith("buffer_load") or op.startswith("flat_load"):
            vm.append((no, regs(first)))
        elif op.startswith("global_store") or op.startswith("scratch_store") or op.startswith("buffer_store") or op.startswith("flat_store") or op.startswith("global_atomic"):
            vm.append((no, set()))
    return found


def main():
    found = scan(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    for f in found[:40]:
        print(f)
    print("flagged", len(found))


if __name__ == "__main__":
    main()
