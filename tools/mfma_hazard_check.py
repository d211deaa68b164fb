"""Static check of a gfx950 assembly listing for the other hazard hipcc cannot see inside inline asm: a vector / memory instruction
that reads or overwrites the destination of an MFMA before the matrix pipe has written it.  The hardware does not interlock
this case (CDNA3/4 ISA, "manually inserted wait states"): on gfx950 an XDL op (bf16 / f16 / fp8 inputs) of N passes needs
N + 4 wait states before a VALU, VMEM, LDS or export instruction touches its vDst (8 passes -> 12, 16 passes -> 20 - the
counts hipcc itself pads its builtin MFMAs with), the f32-input forms N + 2 (16 passes -> 18); every issued instruction
counts one, `s_nop k` counts k + 1.  The compiler pads its own MFMAs; MFMAs written in an asm statement are invisible to it, so a block that is
followed by compiler-scheduled code must end with the wait itself (chain.hip CH_DRAIN / CH8_DRAIN, conv2.hip).

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o /tmp/k.s file.hip && python tools/mfma_hazard_check.py /tmp/k.s

An MFMA that accumulates onto the same registers (SrcC == vDst, same shape) may follow back to back; an MFMA that reads another's
result as SrcA / SrcB is checked like a VALU read.  Linear scan; the state is dropped at unconditional branches."""
import re
import sys

# mnemonic -> wait states its result needs before a non-MFMA instruction may touch it
NEED = {"v_mfma_f32_32x32x16_bf16": 8 + 4, "v_mfma_f32_32x32x16_f16": 8 + 4, "v_mfma_f32_32x32x16_fp8_fp8": 8 + 4,
        "v_mfma_f32_16x16x32_bf16": 4 + 4, "v_mfma_scale_f32_32x32x64_f8f6f4": 16 + 4, "v_mfma_f32_32x32x64_f8f6f4": 16 + 4,
        "v_mfma_f32_32x32x2_f32": 16 + 2, "v_mfma_f32_16x16x4_f32": 8 + 2}
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
    found = []
    name = None
    recent = []  # (issue index, needed wait states, dst regs, line number)
    t = 0
    for no, raw in enumerate(open(path).read().split("\n"), 1):
        l = raw.split(";")[0].strip()
        m = re.match(r'^(_Z\w+):', l)
        if m:
            name, recent, t = m.group(1), [], 0
            continue
        if not l or l.startswith(".") or name is None or want not in name:
            continue
        op = l.split()[0]
        args = l[len(op):]
        if op in ("s_endpgm", "s_branch", "s_setpc_b64"):
            recent = []
            continue
        if op == "s_nop":
            t += int(args.strip(), 0) + 1
            continue
        parts = [a.strip() for a in args.split(",")]
        if op in NEED:
            dst = regs(parts[0])
            src_ab = regs(parts[1]) | regs(parts[2])
            for (t0, need, d, lno) in recent:
                if t - t0 < need and (src_ab & d):
                    found.append(f"{name[:40]} line {no}: `{l[:80]}` reads as A/B the result of the MFMA at line {lno} after {t - t0} of {need} wait states")
            recent = [r for r in recent if t - r[0] < r[1]]
            recent.append((t + 1, NEED[op], dst, no))
            t += 1
            continue
        if op.startswith("s_") and not op.startswith("s_waitcnt"):
            t += 1  # scalar instructions take an issue slot but touch no vector register
            continue
        if op.startswith("s_waitcnt") or op.startswith("s_barrier"):
            t += 1
            continue
        touched = regs(args)
        for (t0, need, d, lno) in recent:
            if t - t0 < need and (touched & d):
                found.append(f"{name[:40]} line {no}: `{l[:80]}` touches {sorted(touched & d)[:3]} = result of the MFMA at line {lno} after {t - t0} of {need} wait states")
        recent = [r for r in recent if t - r[0] < r[1]]
        t += 1
    return found


def main():
    found = scan(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    for f in found[:40]:
        print(f)
    print("flagged", len(found))


if __name__ == "__main__":
    main()
