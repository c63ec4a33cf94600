"""Audit of dense_block14_kernel's / dense_block28_kernel's ISA (no GPU needed): the kernels keep their activation ring in literal registers v[192:255],
whose loads are in flight for two super-step intervals; hipcc does not know those registers are live.  Fails if

  1. any instruction OUTSIDE an inline-asm region (;;#ASMSTART .. ;;#ASMEND) names a VGPR >= 192;
  2. anything was spilled to scratch;
  3. a vector-memory LOAD appears outside an asm region inside the layer loop (the kernel's s_waitcnt vmcnt(N) constants count
     asm loads only: a compiler-issued load in the steady state would make them wrong) - loads before the first s_barrier
     (prologue) are allowed.

  python scripts/audit_block14_isa.py tennis_amd/csrc/isa/dense_block14.s
"""
import re
import sys

RING_LO = 192


def vgprs(line):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", line):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", line):
        out.add(int(a))
    return out


def main():
    text = "\n".join(open(p).read() for p in sys.argv[1:])
    bad = kernels = 0
    for m in re.finditer(r"^(\S*dense_block(?:14|28)_kernel\S*):", text, re.M):
        name = m.group(1)
        end = text.find(".end_amdhsa_kernel", m.end())
        body = text[m.end():text.rfind("s_endpgm", m.end(), end) + 8]
        kernels += 1
        in_asm = False
        seen_barrier = False
        n_asm_loads = 0
        for raw in body.split("\n"):
            if "#ASMSTART" in raw:
                in_asm = True
                continue
            if "#ASMEND" in raw:
                in_asm = False
                continue
            line = raw.split(";")[0].strip()
            if not line or line.startswith(".") or line.endswith(":"):
                continue
            if line.startswith("s_barrier"):
                seen_barrier = True
            is_load = re.match(r"(global|buffer|flat|scratch)_load", line) is not None
            if in_asm:
                n_asm_loads += is_load
                continue
            if any(r >= RING_LO for r in vgprs(line)):
                print("%s: compiler instruction touches the ring registers: %s" % (name, line))
                bad += 1
            if is_load and seen_barrier:
                print("%s: compiler-issued vector-memory load in the steady state: %s" % (name, line))
                bad += 1
        if n_asm_loads < 50:
            print("%s: only %d asm loads found (listing without ASMSTART markers?)" % (name, n_asm_loads))
            bad += 1
        seen = 0
        for pat, where in ((r"\.amdhsa_private_segment_fixed_size\s+(\d+)", text[end - 6000:end]), (r";\s*ScratchSize:\s*(\d+)", text[end:end + 3000])):
            found = re.findall(pat, where)
            if found:
                seen += 1
                if int(found[-1] if "amdhsa" in pat else found[0]) != 0:
                    print("%s: scratch in use" % name)
                    bad += 1
        if not seen:
            print("%s: no scratch-size record found in the listing" % name)
            bad += 1
    print("audited %d kernels, %d problem(s)" % (kernels, bad))
    sys.exit(1 if bad or not kernels else 0)


if __name__ == "__main__":
    main()
