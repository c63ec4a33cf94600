"""Audit of the strip kernel's ISA (dense_strip_impl.h, instantiated in dense_strip_w*.hip) (no GPU needed): the kernel keeps its bottleneck window in literal accumulator registers
a[160:255], which hipcc does not know are live.  This script fails if the compiler's own code touches them.

  python scripts/audit_strip_isa.py <dense_strip_w56 ... gfx950.s> [more .s files]   (scripts/isa_build.sh tennis_amd/csrc/dense_strip_w56.hip -fno-slp-vectorize)

For every dense_strip kernel it checks that
  1. no instruction OUTSIDE an inline-asm region (;;#ASMSTART .. ;;#ASMEND) names an accumulator register >= 160;
  2. nothing was spilled to scratch (.private_segment_fixed_size 0, .vgpr_spill_count 0);
  3. no compiler v_accvgpr_* instruction sits directly in front of an asm MFMA block whose accumulator operands it writes
     (VALU write -> MFMA operand read needs two wait states, and hipcc pads nothing for an asm consumer);
  4. no compiler instruction reads an accumulator register that an asm MFMA wrote fewer than 18 wait states earlier (hipcc
     does not know the latency of an asm MFMA; round 3: v_accvgpr_mov copies at a control-flow join, four instructions
     behind the last MFMA of the 3x3 phase).
"""
import re
import sys

WIN_LO = 160


def regs_of(tok):
    m = re.fullmatch(r"a\[(\d+):(\d+)\]", tok)
    if m:
        return range(int(m.group(1)), int(m.group(2)) + 1)
    m = re.fullmatch(r"a(\d+)", tok)
    if m:
        return range(int(m.group(1)), int(m.group(1)) + 1)
    return range(0)


def agprs(line):
    out = set()
    for tok in re.findall(r"a\[\d+:\d+\]|\ba\d+\b", line):
        out.update(regs_of(tok))
    return out


def main():
    text = "\n".join(open(p).read() for p in sys.argv[1:])
    bad = 0
    kernels = 0
    for m in re.finditer(r"^(\S*dense_strip_kernel\S*):", text, re.M):
        name = m.group(1)
        end = text.find(".end_amdhsa_kernel", m.end())
        body = text[m.end():text.rfind("s_endpgm", m.end(), end) + 8]
        kernels += 1
        in_asm = False
        prev = []          # last compiler instructions (outside asm) with the AGPRs they write
        fresh = {}         # AGPR -> wait states since an asm MFMA wrote it
        for raw in body.split("\n"):
            if "#ASMSTART" in raw:
                in_asm = True
                first_in_block = True
                continue
            if "#ASMEND" in raw:
                in_asm = False
                prev = []
                continue
            line = raw.split(";")[0].strip()
            if not line or line.startswith(".") or line.endswith(":"):
                continue
            # ---- check 4: wait states between an asm MFMA's write and a compiler read of the register
            states = 1
            mm = re.fullmatch(r"s_nop (\d+)", line)
            if mm:
                states = int(mm.group(1)) + 1
            if in_asm and line.startswith("v_mfma"):
                dst = agprs(line.split(",")[0])
                for r in fresh:
                    fresh[r] += states
                for r in dst:
                    fresh[r] = 0
            else:
                if not in_asm and not line.startswith(("s_", "v_accvgpr_write")):
                    ops = line.split(None, 1)[1] if " " in line else ""
                    srcs = agprs(",".join(ops.split(",")[1:])) if line.startswith(("v_accvgpr_read", "v_accvgpr_mov")) else agprs(ops)
                    early = [r for r in srcs if r in fresh and fresh[r] < 18]
                    if early and not line.startswith("v_mfma"):
                        print("%s: compiler '%s' reads a%d %d wait states behind the asm MFMA that wrote it" % (name, line, early[0], fresh[early[0]]))
                        bad += 1
                for r in list(fresh):
                    fresh[r] += states
                    if fresh[r] > 64:
                        del fresh[r]
            if in_asm:
                if first_in_block and line.startswith("v_mfma"):
                    used = agprs(line)
                    for pl, written in prev[-2:]:
                        if written & used:
                            print("%s: compiler '%s' directly in front of asm '%s'" % (name, pl, line))
                            bad += 1
                first_in_block = False
                continue
            touched = agprs(line)
            if any(r >= WIN_LO for r in touched):
                print("%s: compiler instruction touches the window: %s" % (name, line))
                bad += 1
            written = set()
            if line.startswith(("v_accvgpr_write", "v_accvgpr_mov")):
                written = agprs(line.split(",")[0])
            prev.append((line, written))
            if not line.startswith(("v_accvgpr", "s_nop")):
                prev = prev[-1:] if written else []
        # scratch: the kernel descriptor that follows the code (.amdhsa_private_segment_fixed_size) and hipcc's own summary
        # comment (; ScratchSize:) - both must be there and both must say 0
        desc = text[end - 6000:end]
        seen = 0
        for pat in (r"\.amdhsa_private_segment_fixed_size\s+(\d+)", r";\s*ScratchSize:\s*(\d+)"):
            found = re.findall(pat, desc if "amdhsa" in pat else text[end:end + 3000])
            if found:
                seen += 1
                if int(found[-1] if "amdhsa" in pat else found[0]) != 0:
                    print("%s: scratch in use (%s = %s)" % (name, pat.split("\\s")[0].replace("\\", ""), found[0]))
                    bad += 1
        if not seen:
            print("%s: no scratch-size record found in the listing (audit out of date with the assembler's output?)" % name)
            bad += 1
    print("audited %d kernels, %d problem(s)" % (kernels, bad))
    sys.exit(1 if bad or not kernels else 0)


if __name__ == "__main__":
    main()
